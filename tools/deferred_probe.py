#!/usr/bin/env python3
"""Frames of the 16M-node hierarchy in which a tenth of the level-2 subtrees moves:
what the deferred dirty propagation on the device costs against marking every
moved node from the host (same set of updated nodes either way)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from vulkan_renderer_amd import _native as N
torch.cuda.set_device(0)
setup = bench.Setup("deep16m", dict(bench.WORKLOADS["deep16m"]), 0, 0, 1, 1)
hs, sc = setup.hs, setup.sc
lo = [int(x) for x in sc.level_offsets]
cut = 2
cut_first, cut_count = lo[cut], lo[cut + 1] - lo[cut]
run = max(1, cut_count // 10)
def ranges(start):
    a, b = cut_first + start, cut_first + start + run   # [a, b) at level cut
    out = [(a, b - a)]
    for l in range(cut + 1, len(lo) - 1):
        p = sc.parent[lo[l]:lo[l + 1]]
        idx = np.flatnonzero((p >= a) & (p < b))
        if idx.size == 0: break
        a, b = lo[l] + int(idx[0]), lo[l] + int(idx[-1]) + 1
        out.append((a, b - a))
    return out
starts = [(k * run) % max(cut_count - run + 1, 1) for k in range(10)]
table = [ranges(s) for s in starts]
def loop(mode, k=100):
    for w in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in range(k):
            rs = table[f % 10]
            if mode == "roots":
                hs.mark_nodes(rs[0][0], rs[0][1], N.DIRTY_WORLD)
            else:
                for a, n in rs: hs.mark_nodes(a, n, N.DIRTY_WORLD)
            hs.frame(N.STAGE_ALL)
        hs.sync(); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / k * 1e6
    return el, hs.frame_info().nodes_updated
print("roots only (descendants follow on the device):", loop("roots"))
print("every moved node marked:", loop("all"))
print("roots only:", loop("roots"))
setup.close()
