#!/usr/bin/env python3
"""Is the calling thread the bound of the 1M-node frame loop? Enqueue time per
frame (the loop without the final wait) against total time per frame."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from vulkan_renderer_amd import _native as N
torch.cuda.set_device(0)
setup = bench.Setup("flat1m", dict(bench.WORKLOADS["flat1m"]), 0, 0, 1, 1)
hs = setup.hs
def run(k, mark=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        if mark: hs.mark_all(N.DIRTY_WORLD)
        hs.frame(N.STAGE_ALL)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6
run(500)
for _ in range(3):
    print("enqueue %.1f us/frame, total %.1f us/frame" % run(1000))
setup.close()
