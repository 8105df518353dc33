#!/usr/bin/env python3
"""Two ranks on one GPU over gloo: does all_gather_into_tensor on DEVICE tensors
keep up when the host never waits for the GPU between calls?

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 tools/gloo_probe.py <mode> [iterations]

modes: plain   device tensors on the default stream, nothing else queued
       busy    a few milliseconds of kernels queued ahead of every gather
       side    like busy, on a side stream made current with torch.cuda.stream
       staged  like busy, but through pinned host buffers with a stream wait (what
               ShardExchange does over gloo)
       allreduce / gatherlist   other collectives on device tensors, as plain
Every rank prints its progress; a run that stops making progress dumps every
thread's stack after 40 s and exits with status 1."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    mode = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    words = 1 << 16
    x = torch.zeros(words, dtype=torch.int32, device="cuda")
    y = torch.zeros(world * words, dtype=torch.int32, device="cuda")
    load = torch.zeros(1 << 26, dtype=torch.float32, device="cuda")
    hx = torch.zeros(words, dtype=torch.int32).pin_memory()
    hy = torch.zeros(world * words, dtype=torch.int32).pin_memory()
    side = torch.cuda.Stream()
    scope = (lambda: torch.cuda.stream(side)) if mode == "side" else (lambda: torch.cuda.stream(None))
    t0 = time.time()
    for i in range(iters):
        faulthandler.dump_traceback_later(40, exit=True)
        with scope():
            if mode != "plain":
                for _ in range(8):
                    load.add_(1.0)
            x.fill_(i * world + rank)
            if mode == "staged":
                hx.copy_(x, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(hy, hx)
                y.copy_(hy, non_blocking=True)
            elif mode == "allreduce":
                dist.all_reduce(x)
            elif mode == "gatherlist":
                dist.all_gather(list(y.view(world, words).unbind(0)), x)
            else:
                dist.all_gather_into_tensor(y, x)
        if i % 50 == 49 or i < 3:
            print(f"[{mode}] rank {rank} iteration {i + 1} at {time.time() - t0:.2f} s", flush=True)
    torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    got = y.view(world, words)[:, 0].tolist()
    want = [(iters - 1) * world + r for r in range(world)]
    if mode == "allreduce":
        got, want = [], []
    print(f"[{mode}] rank {rank} done in {time.time() - t0:.2f} s, last gather {got} (want {want})", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if got == want else 3


if __name__ == "__main__":
    sys.exit(main())
