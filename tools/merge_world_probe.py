#!/usr/bin/env python3
"""What the merge of an 8-rank exchange costs one rank, measured on one GPU: the
frame's own packed block stands in for every rank's (eight copies), the merge
is timed on its own and beside a running frame loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vulkan_renderer_amd import _native as N
from vulkan_renderer_amd.parallel import exchange_block_words, exchange_result_words


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "deep16m"
    torch.cuda.set_device(0)
    setup = bench.Setup(workload, dict(bench.WORKLOADS[workload]), 0, 0, 1, 1)
    hs, views = setup.hs, setup.views
    for _ in range(3):
        hs.mark_all(N.DIRTY_WORLD)
        hs.frame(N.STAGE_ALL)
    info = hs.frame_info()
    vis_total = sum(st.visible_count for st in info.views)
    vis_cap = (vis_total + 1024 + 3) // 4 * 4
    key_cap = (info.transparent_draw_count + 256 + 3) // 4 * 4
    words = exchange_block_words(views, vis_cap, key_cap)
    block = torch.zeros(words, dtype=torch.int32, device="cuda")
    hs.exchange_pack(block.data_ptr(), vis_cap, key_cap)
    hs.sync()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for world in (1, 2, 4, 8):
        gathered = block.repeat(world)
        vis_merged = torch.zeros(world * vis_cap, dtype=torch.int32, device="cuda")
        key_merged = torch.zeros(world * key_cap, dtype=torch.int64, device="cuda")
        result = torch.zeros(exchange_result_words(views), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()

        def merge():
            hs.exchange_merge(gathered.data_ptr(), world, vis_cap, key_cap, vis_merged.data_ptr(),
                              key_merged.data_ptr(), result.data_ptr(), side.cuda_stream)

        for _ in range(3):
            merge()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(10):
            merge()
        b.record(side)
        torch.cuda.synchronize()
        alone = a.elapsed_time(b) * 100.0
        # beside the frame loop: frames per second with a merge per frame on the other stream
        def loop(with_merge, k=60):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                hs.mark_all(N.DIRTY_WORLD)
                hs.frame(N.STAGE_ALL)
                if with_merge:
                    merge()
            hs.sync()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k * 1e6
        loop(False, 40)
        print(f"world {world}: block {words * 4 / 1e6:.1f} MB per rank, merge alone {alone:7.1f} us, "
              f"frame loop {loop(False):7.1f} us, with a merge per frame {loop(True):7.1f} us", flush=True)
    setup.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
