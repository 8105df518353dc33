#!/usr/bin/env python3
"""Benchmark of the per-frame scene-processing path on MI355X.

One "step" = one frame: every node moved (DIRTY_WORLD) -> world-transform
propagation -> multi-view sphere cull + ordered compaction -> transparent key
generation, radix sort and draw emission, over a synthetic instance hierarchy
already resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

Workloads: flat1m (1M instances, flat, one view; the default), deep16m (16M,
depth-8 hierarchy, camera + 4 cascades), flat32m (the per-GPU shard of 256M over
eight GPUs), probes64 (16M, 64 views).
With N > 1 every rank owns a contiguous shard of the same size (weak scaling)
and the per-frame exchange all-gathers the visible lists and merges the sorted
draw-key runs.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
EVENT_STRIDE = 4        # timed steps per HIP-event sample of the dominant kernel

WORKLOADS = {
    # name: (instances per GPU, hierarchy depth, cascades, extra probe views)
    "flat1m": dict(n=1 << 20, depth=1, cascades=0, probes=0,
                   desc="1M synthetic instances, flat hierarchy, 1 view"),
    "deep16m": dict(n=1 << 24, depth=8, cascades=4, probes=0,
                    desc="16M instances, depth-8 hierarchy, main view + 4 shadow cascades"),
    "flat32m": dict(n=1 << 25, depth=1, cascades=0, probes=0,
                    desc="32M instances per GPU (256M over 8), flat hierarchy, 1 view"),
    "probes64": dict(n=1 << 24, depth=1, cascades=0, probes=63,
                     desc="16M instances per GPU, flat hierarchy, 64 views"),
}

# Algorithmic HBM bytes per unit for each kernel of the frame:
#   transform: local 64 + parent index 4 (non-root) + flags 1+1 + world 64
#   classify : world 64 + sphere 16 + flags 4 in; visibility byte 1 out, plus 4 per
#              surviving (instance, view) (the staged survivor index)
#   fused    : both, minus the re-read of world (it is classified in registers)
def kernel_bytes(name: str, n: int, depth: int, survivors: int, dirty_local: bool = False,
                 submeshes: int = 1) -> float:
    transform = 64 + 64 + 2 + (4 if depth > 1 else 0)
    if dirty_local:  # TRS 48 in instead of the cached local, rebuilt local 64 out
        transform += 48

    # with several rows per transform the matrix is fetched once per node
    classify_in, classify_out = 64 / submeshes + 16 + 4 + (4 if submeshes > 1 else 0), 1
    if name == "transform":
        return (n // submeshes) * transform
    if name == "classify":
        return n * (classify_in + classify_out) + 4 * survivors
    if name == "fused":
        return n * (transform + classify_in - 64 + classify_out) + 4 * survivors
    raise KeyError(name)


KERNEL_NAMES = {"transform": "k_propagate_classify", "classify": "k_classify",
                "fused": "k_propagate_classify"}


def pmc_traffic(workload_name: str, overridden: bool, kernel: str, variant=None):
    """HBM bytes per step of the dominant kernel from the committed rocprofv3
    counter passes of this workload (FETCH_SIZE doubled, the gfx950 correction
    for wide coalesced reads, plus WRITE_SIZE; tools/pmc_passes.sh collects them
    in separate runs). None when no pass of this exact workload is committed."""
    if overridden:
        return None, None
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{workload_name}.json")))
    if not paths:
        return None, None
    with open(paths[-1]) as fh:
        data = json.load(fh)
    total = 0.0
    for name, k in data["kernels"].items():
        # the last template argument of k_propagate_classify tells the fused kernel
        # (true) from its transform-only variant (false)
        if name.startswith(kernel) and (variant is None or name.rstrip().endswith(variant + ">")):
            total += k["dispatches_per_frame"] * (k["fetch_bytes_per_dispatch_x2"] +
                                                  k["write_bytes_per_dispatch"])
    return (total or None), os.path.relpath(paths[-1], ROOT)


def build_scene(workload: dict, rank: int, seed: int, submeshes: int = 1):
    from vulkan_renderer_amd import synth
    n = workload["n"] // submeshes
    if workload["depth"] > 1:
        sc = synth.deep_scene(n, depth=workload["depth"], seed=seed + 1 + rank)
    else:
        sc = synth.flat_scene(n, seed=seed, first=rank * n)
    vs = synth.camera_views(cascades=workload["cascades"])
    extra = None
    if workload["probes"]:
        extra, _ = synth.ring_views(workload["probes"])
    if submeshes > 1:  # `submeshes` candidate rows per transform, in node order
        rows = sc.n * submeshes
        index = np.arange(rank * rows, (rank + 1) * rows, dtype=np.uint64)
        sc.ident, sc.matbits, sc.bounds = synth.make_candidates(seed, index)
        sc.node = (np.arange(rows, dtype=np.uint64) // np.uint64(submeshes)).astype(np.uint32)
    return sc, vs, extra


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="flat1m", choices=sorted(WORKLOADS))
    ap.add_argument("--instances", type=int, default=0, help="override instances per GPU")
    ap.add_argument("--depth", type=int, default=-1, help="override hierarchy depth")
    ap.add_argument("--cascades", type=int, default=-1, help="override shadow cascade count")
    ap.add_argument("--probes", type=int, default=-1, help="override extra probe views")
    ap.add_argument("--submeshes", type=int, default=1,
                    help="candidate rows per transform (>1: the stand-alone transform and classify "
                         "kernels run instead of the fused one)")
    ap.add_argument("--dirty", default="world", choices=["world", "local"],
                    help="what moves every frame: world = cached local matrices are re-propagated "
                         "(default); local = every node's TRS changed, locals are rebuilt too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend; gloo lets several ranks rehearse on one GPU")
    ap.add_argument("--exchange", action="store_true",
                    help="run the multi-GPU exchange step even with one rank (to time its launches)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    watchdog = float(os.environ.get("VKR_BENCH_WATCHDOG", "0"))
    if watchdog > 0:  # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                  "torch.distributed.run --nproc-per-node N", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    if args.backend == "gloo":  # rehearsal: ranks may share a GPU
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from vulkan_renderer_amd import HipScene, build_view_set, frustum_from_matrix
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd.parallel import RcclExchange, ShardExchange

    workload = dict(WORKLOADS[args.workload])
    if args.instances:
        workload["n"] = args.instances
    for key in ("depth", "cascades", "probes"):
        if getattr(args, key) >= 0:
            workload[key] = getattr(args, key)
    n = workload["n"]
    sc, vs, extra = build_scene(workload, rank, args.seed, args.submeshes)
    n = sc.ident.shape[0]
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    if extra is not None:
        planes = np.concatenate([planes, np.stack([frustum_from_matrix(m) for m in extra])])
        required = np.concatenate([required, np.full(len(extra), N.CAND_SHADOW_CASTER, np.uint32)])
    views = planes.shape[0]

    hs = HipScene(max_nodes=sc.n, max_candidates=n, max_views=views, device=local_rank,
                  source_index_base=rank * n)
    stream = torch.cuda.current_stream()
    hs.set_stream(stream.cuda_stream)
    hs.set_hierarchy(sc.parent if workload["depth"] > 1 else None, sc.level_offsets)
    hs.upload_trs(sc.pos, sc.quat, sc.scale)
    hs.set_candidates(sc.node if args.submeshes > 1 else None, sc.ident, sc.matbits, sc.bounds)
    hs.set_views(planes, required, vs.view)
    hs.frame(N.STAGE_TRANSFORM)  # materialise local matrices from TRS once (not timed)
    hs.sync()
    exchange = None
    if world > 1 or args.exchange:
        # the library's own RCCL exchange; the torch collective only for the gloo rehearsal
        native = not (args.backend == "gloo" and world > 1)
        if native:
            try:
                exchange = RcclExchange(hs, world, rank, views)
            except Exception as exc:  # no usable RCCL: the torch-driven exchange does the same work
                print(f"bench.py: native RCCL exchange unavailable ({exc}); using torch.distributed",
                      file=sys.stderr)
                native = False
        if world > 1:  # every rank must take the same path
            flag = torch.tensor([1 if native else 0], dtype=torch.int32,
                                device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
        if not native:
            exchange = ShardExchange(hs, world, rank, views)

    dirty_bits = N.DIRTY_LOCAL if args.dirty == "local" else N.DIRTY_WORLD

    def step():
        hs.mark_all(dirty_bits)      # every node moved this frame
        hs.frame(N.STAGE_ALL)
        if exchange is not None:
            exchange.run()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up, also used to find the dominant kernel
    hs.set_profiling(True)
    for _ in range(max(args.warmup, 1)):
        step()
        if exchange is not None:
            exchange.finish()        # sizes the exchange blocks (same decision on every rank)
    fence()
    warm = hs.timers(reset=True)
    dominant = max(("transform", "classify", "fused"), key=lambda k: warm[k][0])

    # The dominant kernel is timed with HIP events on the library's stream (an
    # event pair attached to each of its launches, i.e. the kernel's own start and
    # end) in one of every EVENT_STRIDE timed steps: the timestamped launches cost
    # a few microseconds of queue time, which is visible at this frame length.
    hs.set_profiling(False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sampled = k % EVENT_STRIDE == 0
        if sampled:
            hs.set_profiling([dominant])
        step()
        if sampled:
            hs.set_profiling(False)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timers = hs.timers(reset=True)
    info = hs.frame_info()
    if exchange is not None:
        regrown = exchange.regrown
        exchange.finish()
        if exchange.regrown != regrown:
            print("bench.py: the exchange blocks overflowed inside the timed region", file=sys.stderr)
            return 3

    ms_per_step = elapsed * 1e3 / args.steps
    total_instances = n * world
    value = total_instances * args.steps / elapsed

    dom_ms, dom_n = timers[dominant]
    dom_avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    launches = 1 if dominant == "classify" else len(sc.level_offsets) - 1
    survivors = sum(st.visible_count for st in info.views) + info.transparent_draw_count
    alg_bytes = kernel_bytes(dominant, n, workload["depth"], survivors, args.dirty == "local",
                             args.submeshes)
    achieved = alg_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    overridden = (bool(args.instances) or args.dirty != "world" or args.submeshes != 1 or
                  any(getattr(args, k) >= 0 for k in ("depth", "cascades", "probes")))
    traffic, traffic_source = pmc_traffic(args.workload, overridden, KERNEL_NAMES[dominant],
                                          {"fused": "true", "transform": "false"}.get(dominant))
    roofline = {
        "bound": "hbm", "kernel": KERNEL_NAMES[dominant],
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None if traffic is None else int(traffic), "traffic_source": traffic_source,
        "avg_launch_us": round(dom_avg_s * 1e6 / launches, 2),
        "launches_per_step": launches, "event_samples": dom_n,
        "algorithmic_bytes_per_step": int(alg_bytes),
    }
    stage_us = {k: round(v[0] * 1e3 / max(v[1], 1), 2) for k, v in warm.items() if v[1]}

    result = {
        "metric": "instances/sec (transform+cull+sort)",
        "value": value, "unit": "instances/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {workload['desc']}",
            "instances_per_gpu": n, "instances_total": total_instances, "views": views,
            "hierarchy_depth": len(sc.level_offsets) - 1, "rows_per_transform": args.submeshes,
            "visible_camera": info.views[0].visible_count,
            "transparent_draws": info.transparent_draw_count,
            "moved_per_frame": "every node: " + ("TRS (locals rebuilt)" if args.dirty == "local"
                                                 else "world matrices re-propagated from cached locals"),
            "exchange": ("one all-gather of packed blocks (visible lists + draw-key runs), "
                         "device-planned merge, overlapped with the next frame") if exchange is not None else "none",
        },
        "roofline": roofline,
        "stage_us_warmup": stage_us,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sc, planes, required, vs, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)  # out before any teardown
    if world > 1:
        dist.barrier()  # no rank tears its communicator down while another still uses it
    hs.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def cpu_baseline(sc, planes, required, vs, budget_s: float) -> dict:
    """The CPU oracle (a port of the reference's loop structure) timed on the
    host, single thread like the reference's render thread, on a bounded prefix
    of the same workload."""
    from oracle import pyoracle
    pyoracle.build()
    o = pyoracle.Oracle()
    n = sc.n
    flat = len(sc.level_offsets) == 2
    # bounded sample: a prefix of the instance range (flat) keeps the workload's
    # distribution; hierarchies are timed whole because subtrees interleave
    sample = min(n, 1 << 20) if flat else n
    local, _, _, _ = o.scene_update(sc.parent[:sample] if flat else sc.parent,
                                    pos=sc.pos[:sample], quat=sc.quat[:sample],
                                    scale=sc.scale[:sample],
                                    flags=np.full(sample, 0x01, np.uint8))
    h = o.baseline_create(sc.parent[:sample], local, None, sc.ident[:sample], sc.matbits[:sample],
                          sc.bounds[:sample], planes, required, vs.view, vs.projection)
    o.baseline_frame(h)  # warm
    frames, t0 = 0, time.perf_counter()
    while True:
        o.baseline_frame(h)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 2000:
            break
    o.baseline_destroy(h)
    out = {"value": sample * frames / dt, "unit": "instances/s", "cores": 1, "kind": "port",
           "sample": f"{frames} frames of the first {sample} instances of the workload, "
                     f"all {planes.shape[0]} views, {dt:.1f} s"}
    if flat:
        out["all_cores"] = cpu_baseline_all_cores(o, sc, sample, local, planes, required, vs,
                                                  min(10.0, budget_s))
    return out


def cpu_baseline_all_cores(o, sc, sample, local, planes, required, vs, budget_s: float) -> dict:
    """Not in the reference (its path runs on the render thread alone): the same
    single-thread loop on every host core of this GPU's share at once, each thread
    with its own copy of the sample and no merge of the results - an upper bound
    for a sharded CPU version. The foreign calls release the interpreter lock."""
    import threading
    threads = max(1, min(16, os.cpu_count() or 1))
    frames = [0] * threads
    handles = [None] * threads
    start = threading.Barrier(threads + 1)
    stop_at = [0.0]

    def work(k):
        handles[k] = o.baseline_create(sc.parent[:sample], local, None, sc.ident[:sample],
                                       sc.matbits[:sample], sc.bounds[:sample], planes, required,
                                       vs.view, vs.projection)
        o.baseline_frame(handles[k])  # warm
        start.wait()
        while time.perf_counter() < stop_at[0]:
            o.baseline_frame(handles[k])
            frames[k] += 1

    pool = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    for t in pool:
        t.start()
    stop_at[0] = float("inf")
    start.wait()
    t0 = time.perf_counter()
    stop_at[0] = t0 + budget_s
    for t in pool:
        t.join()
    dt = time.perf_counter() - t0
    for h in handles:
        o.baseline_destroy(h)
    return {"value": sample * sum(frames) / dt, "unit": "instances/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads, each looping over its own copy of the same {sample} "
                      f"instances, no merge, {dt:.1f} s; not in the reference"}


if __name__ == "__main__":
    sys.exit(main())
