"""Times the device epoch hash (vkrhip_candidate_epoch) against the byte-serial
CPU oracle on the same candidate rows. Usage: python tools/epoch_bench.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkan_renderer_amd import HipScene, build_view_set, synth  # noqa: E402
from vulkan_renderer_amd import _native as N  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    sc = synth.flat_scene(n, seed=1)
    vs = synth.camera_views(cascades=0)
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame(N.STAGE_ALL | N.STAGE_CANDIDATES)
        hs.sync()
        hs.candidate_epoch()  # allocates the workspace
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            h, k = hs.candidate_epoch()
        gpu_ms = (time.perf_counter() - t0) * 1e3 / reps
        line = {"n": n, "occluders": k, "hash": hex(h), "gpu_ms": round(gpu_ms, 3)}
        if n <= (1 << 22):
            from oracle import pyoracle
            pyoracle.build()
            o = pyoracle.Oracle()
            rows = hs.read_candidates()
            t0 = time.perf_counter()
            ref = o.candidate_epoch(rows)
            line["cpu_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            line["match"] = ref == h
        print(line)


if __name__ == "__main__":
    main()
