#!/usr/bin/env python3
"""What plain streaming kernels reach on this device: copy, read-modify-write,
fill and a read-only sum over 1 GiB buffers (torch's own kernels and the
runtime's device-to-device copy), in TB/s of bytes moved."""
import time

import torch


def rate(fn, nbytes, reps=30):
    for _ in range(60):  # the device's clocks settle for the first 12 ms of a load
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12


def main():
    n = 1 << 28  # 1 GiB of float32
    x = torch.ones(n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    b = n * 4
    print(f"copy (y.copy_(x)), read + write      {rate(lambda: y.copy_(x), 2 * b):5.2f} TB/s")
    print(f"add out of place, 2 reads + 1 write   {rate(lambda: torch.add(x, y, out=z), 3 * b):5.2f} TB/s")
    print(f"add in place, read + write           {rate(lambda: x.add_(1.0), 2 * b):5.2f} TB/s")
    print(f"fill, write only                      {rate(lambda: y.fill_(2.0), b):5.2f} TB/s")
    print(f"sum, read only                        {rate(lambda: x.sum(), b):5.2f} TB/s")


if __name__ == "__main__":
    main()
