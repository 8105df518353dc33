#!/usr/bin/env python3
"""Does a frame's duration depend on how long the device has been busy? Times 20
frames of the 16M-node hierarchy (after 5 untimed ones) behind an idle device,
behind 60 frames of the same work, and behind 30 ms of unrelated device work."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vulkan_renderer_amd import _native as N


def timed(hs, steps=20, warmup=5):
    for _ in range(warmup):
        hs.mark_all(N.DIRTY_WORLD)
        hs.frame(N.STAGE_ALL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hs.mark_all(N.DIRTY_WORLD)
        hs.frame(N.STAGE_ALL)
    hs.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    torch.cuda.set_device(0)
    setup = bench.Setup("deep16m", dict(bench.WORKLOADS["deep16m"]), 0, 0, 1, 1)
    hs = setup.hs
    time.sleep(0.5)
    print(f"idle 0.5 s, then 5 + 20 frames:      {timed(hs):7.1f} us/frame", flush=True)
    print(f"straight on, 5 + 20 frames:           {timed(hs):7.1f} us/frame", flush=True)
    print(f"straight on, 5 + 20 frames:           {timed(hs):7.1f} us/frame", flush=True)
    time.sleep(0.5)
    print(f"idle 0.5 s, then 5 + 20 frames:      {timed(hs):7.1f} us/frame", flush=True)
    time.sleep(0.5)
    x = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        for _ in range(10):
            x.add_(1.0)
        torch.cuda.synchronize()
    print(f"idle, 30 ms of other work, 5 + 20:    {timed(hs):7.1f} us/frame", flush=True)
    time.sleep(0.5)
    print(f"idle 0.5 s, 0 + 20 frames:            {timed(hs, 20, 0):7.1f} us/frame", flush=True)
    print(f"straight on, 0 + 200 frames:          {timed(hs, 200, 0):7.1f} us/frame", flush=True)
    setup.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
