"""Host time spent per call of frame() and of the native exchange_run() (a tiny
scene, so the GPU is never the limit). Usage: python tools/exchange_host_cost.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkan_renderer_amd import HipScene, build_view_set, synth  # noqa: E402
from vulkan_renderer_amd.parallel import RcclExchange  # noqa: E402


def main():
    import torch
    torch.cuda.set_device(0)  # torch first: it does not find the GPU once the library holds it
    torch.zeros(1, device="cuda")
    n = 4096
    sc = synth.flat_scene(n, seed=1)
    vs = synth.camera_views(cascades=0)
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        if "--comm" in sys.argv:
            hs.comm_init(hs.comm_unique_id(), 1, 0)
        ex = RcclExchange(hs, 1, 0, 1)
        for _ in range(3):
            hs.mark_all(2); hs.frame(); ex.run(); ex.finish()
        reps = 300
        tf = tx = 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            hs.mark_all(2); hs.frame()
            t1 = time.perf_counter()
            ex.run()
            t2 = time.perf_counter()
            tf += t1 - t0
            tx += t2 - t1
            if _ % 8 == 7:
                hs.sync()
        ex.finish()
        print({"frame_host_us": round(tf / reps * 1e6, 1), "exchange_run_host_us": round(tx / reps * 1e6, 1)})


if __name__ == "__main__":
    main()
