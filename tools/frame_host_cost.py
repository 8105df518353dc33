"""Host time per vkrhip_frame call (mark_all + frame through the ctypes mirror) on a
tiny scene, where the GPU is never the limit: what a frame costs the enqueuing
thread, with the tail on the context's stream and on the second stream.
Usage: python tools/frame_host_cost.py"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) > 1:
        import torch
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
        from vulkan_renderer_amd import HipScene, build_view_set, synth
        n = int(os.environ.get("HOST_COST_NODES", "4096"))
        sc = synth.flat_scene(n, seed=1)
        vs = synth.camera_views(cascades=0)
        planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
        with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
            hs.set_hierarchy(None, sc.level_offsets)
            hs.upload_trs(sc.pos, sc.quat, sc.scale)
            hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
            hs.set_views(planes, required, vs.view)
            for _ in range(5):
                hs.mark_all(2); hs.frame()
            hs.sync()
            reps, spent = 400, 0.0
            for k in range(reps):
                t0 = time.perf_counter()
                hs.mark_all(2); hs.frame()
                spent += time.perf_counter() - t0
                if k % 8 == 7:
                    hs.sync()
            # ... and the frame period when nothing but the device paces the loop
            hs.sync()
            t0 = time.perf_counter()
            inside = 0.0
            for k in range(reps):
                t1 = time.perf_counter()
                hs.mark_all(2); hs.frame()
                inside += time.perf_counter() - t1
            hs.sync()
            period = (time.perf_counter() - t0) / reps
            print(sys.argv[1], {"nodes": n, "frame_host_us": round(spent / reps * 1e6, 1),
                                "period_us": round(period * 1e6, 1),
                                "host_us_inside_calls_unsynced": round(inside / reps * 1e6, 1)})
        return
    for tail in ("0", "2"):
        env = dict(os.environ, VKRHIP_OVERLAP_TAIL=tail, VKRHIP_OVERLAP_SORT=tail)
        subprocess.run([sys.executable, __file__, "tail=" + tail], env=env, check=True)


if __name__ == "__main__":
    main()
