#!/usr/bin/env python3
"""Benchmark of the per-frame scene-processing path on MI355X.

One "step" = one frame: every node moved (DIRTY_WORLD) -> world-transform
propagation -> multi-view sphere cull + ordered compaction -> transparent key
generation, radix sort and draw emission, over a synthetic instance hierarchy
already resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

Workloads: deep16m (16M instances, depth-8 hierarchy, camera + 4 cascades: the
default, the largest single-GPU configuration), flat1m (1M, flat, one view),
flat32m (the per-GPU shard of 256M over eight GPUs), probes64 (16M, 64 views).
With one GPU the line also carries `secondary` (the other workloads, shorter
runs, each with its own roofline) and `dropin` (the frame with the 144-byte
candidate rows, with and without the host copies a host-side consumer needs).
With N > 1 every rank owns a shard of about the same size (weak scaling: a
contiguous index range of the flat workloads, whole trees of ONE hierarchy of
N x 16M nodes for deep16m) and the per-frame exchange all-gathers the visible lists and merges the sorted
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
# untimed frames of the workload ahead of the warm-up (see measure); 0 switches them off
SETTLE_SECONDS = float(os.environ.get("VKR_BENCH_SETTLE_S", "0.03"))

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
#   classify : world 64 + sphere 16 + flags 4 in; visibility byte 1 out, plus the
#              survivor masks: one bit per (instance, view) and five more per instance
#              (four state buckets, visible blend row)
#   fused    : both, minus the re-read of world (it is classified in registers)
def kernel_bytes(name: str, n: int, depth: int, views: int, dirty_local: bool = False,
                 submeshes: int = 1, moved: int = -1) -> float:
    transform = 64 + 64 + 2 + (4 if depth > 1 else 0)
    if dirty_local:  # TRS 48 in instead of the cached local, rebuilt local 64 out
        transform += 48
    if 0 <= moved < n and name == "fused":
        # a node that did not move: its stored world matrix is read (64) instead of the
        # local one and no matrix is written; flags, parent index and the cull inputs
        # and outputs move as for any node
        clean = n - moved
        per_node_cull = 16 + 4 + 1
        return (moved * (transform + per_node_cull) + clean * (64 + 2 + (4 if depth > 1 else 0) + per_node_cull)
                + n * (views + 5) / 8)

    # with several rows per transform the matrix is fetched once per node
    classify_in, classify_out = 64 / submeshes + 16 + 4 + (4 if submeshes > 1 else 0), 1
    if name == "transform":
        return (n // submeshes) * transform
    masks = n * (views + 5) / 8
    if name == "classify":
        return n * (classify_in + classify_out) + masks
    if name == "fused":
        return n * (transform + classify_in - 64 + classify_out) + masks
    raise KeyError(name)


KERNEL_NAMES = {"transform": "k_propagate_classify", "classify": "k_classify",
                "fused": "k_propagate_classify"}

# The vector-instruction roof: a wave64 instruction occupies its 16-lane SIMD for four
# cycles, so the chip issues at most SIMDs x clock / 4 of them per second
# (256 CUs x 4 SIMDs x 2.4 GHz / 4).
VALU_PEAK_INSTS_PER_S = 256 * 4 * 2.4e9 / 4


def pmc_valu_insts(workload_name: str, overridden: bool, kernel: str, variant=None):
    """SQ_INSTS_VALU per step of the dominant kernel from the committed counter pass
    of this workload (wave instructions, summed over the kernel's dispatches of one
    frame), or None."""
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
        targs = [x.strip() for x in name[name.find("<") + 1: name.rfind(">")].split(",")] if "<" in name else []
        if (name.startswith(kernel) and "valu_insts_per_dispatch" in k
                and (variant is None or (len(targs) > 3 and targs[3] == variant))):
            total += k["dispatches_per_frame"] * k["valu_insts_per_dispatch"]
    return (total or None), os.path.relpath(paths[-1], ROOT)


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
        # the fourth template argument of k_propagate_classify tells the fused kernel
        # (true) from its transform-only variant (false)
        targs = [x.strip() for x in name[name.find("<") + 1: name.rfind(">")].split(",")] if "<" in name else []
        if name.startswith(kernel) and (variant is None or (len(targs) > 3 and targs[3] == variant)):
            total += k["dispatches_per_frame"] * (k["fetch_bytes_per_dispatch_x2"] +
                                                  k["write_bytes_per_dispatch"])
    return (total or None), os.path.relpath(paths[-1], ROOT)


def build_scene(workload: dict, rank: int, seed: int, submeshes: int = 1, world: int = 1):
    from vulkan_renderer_amd import synth
    n = workload["n"] // submeshes
    if workload["depth"] > 1 and world > 1:
        # ONE hierarchy of n * world nodes, cut into whole trees of about n nodes each
        # (vkrhip_shard_hierarchy); this rank generates and owns shard `rank`
        sc = synth.sharded_deep_scene(n, world, rank, depth=workload["depth"], seed=seed + 1)
    elif workload["depth"] > 1:
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


class Setup:
    """One workload resident in HBM: scene, views, context."""

    def __init__(self, name, workload, rank, local_rank, seed, submeshes, world=1):
        import torch
        from vulkan_renderer_amd import HipScene, build_view_set, frustum_from_matrix
        from vulkan_renderer_amd import _native as N
        self.name, self.workload = name, workload
        sc, vs, extra = build_scene(workload, rank, seed, submeshes, world)
        n = sc.ident.shape[0]
        # rows are numbered shard after shard
        self.index_base = sc.meta.get("index_base", rank * n)
        self.total = sc.meta.get("total", n * world)
        planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
        if extra is not None:
            planes = np.concatenate([planes, np.stack([frustum_from_matrix(m) for m in extra])])
            required = np.concatenate([required, np.full(len(extra), N.CAND_SHADOW_CASTER, np.uint32)])
        self.sc, self.vs, self.planes, self.required, self.n = sc, vs, planes, required, n
        self.views = planes.shape[0]
        hs = HipScene(max_nodes=sc.n, max_candidates=n, max_views=self.views, device=local_rank,
                      source_index_base=self.index_base)
        hs.set_stream(torch.cuda.current_stream().cuda_stream)
        hs.set_hierarchy(sc.parent if workload["depth"] > 1 else None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(sc.node if submeshes > 1 else None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame(N.STAGE_TRANSFORM)  # materialise local matrices from TRS once (not timed)
        hs.sync()
        self.hs = hs

    def close(self):
        self.hs.close()


EXCHANGE_TIMERS = ("exchange_pack", "exchange_gather", "exchange_merge")


def measure(setup, steps, warmup, dirty, submeshes, exchange, fence, overridden, min_seconds=0.0):
    """Timed frames of one workload. Returns (elapsed seconds, roofline dict, warm-up
    stage timings, frame info, exchange stage timings or None). min_seconds > 0
    (secondary entries only) raises `steps` so that the timed region lasts about that
    long - ten frames of 43 us end before the host has finished enqueuing them - and
    leaves the count used in setup.timed_steps.
    dirty: "world" every node moved, locals cached; "local" every node's TRS changed;
    "subtrees" a tenth of the subtrees hanging from hierarchy level 2 moved (another
    tenth every frame), everything else is clean."""
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd.parallel import RcclExchange
    hs, sc, workload, n = setup.hs, setup.sc, setup.workload, setup.n
    dirty_bits = N.DIRTY_LOCAL if dirty == "local" else N.DIRTY_WORLD
    native_exchange = isinstance(exchange, RcclExchange)
    frame_no = [0]
    if dirty == "subtrees":
        lo = sc.level_offsets
        cut = min(2, len(lo) - 2)
        cut_first, cut_count = int(lo[cut]), int(lo[cut + 1] - lo[cut])
        run = max(1, cut_count // 10)

    def step():
        if dirty == "subtrees":      # the roots of the moved subtrees; their descendants follow on the device
            start = (frame_no[0] * run) % max(cut_count - run + 1, 1)
            hs.mark_nodes(cut_first + start, run, N.DIRTY_WORLD)
            frame_no[0] += 1
        else:
            hs.mark_all(dirty_bits)  # every node moved this frame
        hs.frame(N.STAGE_ALL)
        if exchange is not None:
            exchange.run()

    # Settling frames (untimed, ahead of the warm-up steps asked for). The device's
    # power management needs about 12 ms of a new load to settle: after an idle
    # stretch (here: the scene set-up on the host) the first three frames of the
    # 16M-node hierarchy run at the steady rate, the next ten up to 12 % below it,
    # and the rate is back after about twenty (tools/clock_ramp_probe.py; 30 ms of
    # unrelated device work ahead of the frames does not change that). A renderer
    # runs frame after frame, so the steady rate is the one to report.
    probe = 3 if (SETTLE_SECONDS > 0 or min_seconds > 0) else 0
    t0 = time.perf_counter()
    for _ in range(probe):
        step()
        if exchange is not None:
            exchange.finish()
    fence()
    frame_s = max((time.perf_counter() - t0) / max(probe, 1), 1e-6)
    if min_seconds > 0:
        steps = max(steps, int(min_seconds / frame_s))
    setup.timed_steps = steps
    settle = int(min(SETTLE_SECONDS / frame_s, 2000))
    if exchange is not None and getattr(exchange, "world", 1) > 1:   # the same count on every rank
        import torch
        import torch.distributed as dist
        count = torch.tensor([settle], dtype=torch.int64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(count, op=dist.ReduceOp.MAX)
        settle = int(count.item())
    for _ in range(settle):
        step()
        if exchange is not None:
            exchange.finish()
    setup.settling_frames = probe + settle

    # warm-up, also used to find the dominant kernel
    hs.set_profiling(True)
    for _ in range(max(warmup, 1)):
        step()
        if exchange is not None:
            exchange.finish()        # sizes the exchange blocks (same decision on every rank)
    fence()
    warm = hs.timers(reset=True)
    dominant = max(("transform", "classify", "fused"), key=lambda k: warm[k][0])

    # The dominant kernel is timed with HIP events on the library's stream (an
    # event pair attached to each of its launches, i.e. the kernel's own start and
    # end) in one of every EVENT_STRIDE timed steps: the timestamped launches cost
    # a few microseconds of queue time, which is visible at short frame lengths.
    hs.set_profiling(False)
    fence()
    t0 = time.perf_counter()
    for k in range(steps):
        sampled = k % EVENT_STRIDE == 0
        if sampled:
            hs.set_profiling([dominant] + (list(EXCHANGE_TIMERS) if native_exchange else []))
        step()
        if sampled:
            hs.set_profiling(False)
    fence()
    elapsed = time.perf_counter() - t0
    timers = hs.timers(reset=True)
    info = hs.frame_info()

    dom_ms, dom_n = timers[dominant]
    dom_avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    moved = info.nodes_updated if dirty == "subtrees" else -1
    alg_bytes = kernel_bytes(dominant, n, workload["depth"], setup.views, dirty == "local", submeshes, moved)
    achieved = alg_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    traffic, traffic_source = pmc_traffic(setup.name, overridden, KERNEL_NAMES[dominant],
                                          {"fused": "true", "transform": "false"}.get(dominant))
    roofline = {
        "bound": "hbm", "kernel": KERNEL_NAMES[dominant],
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None if traffic is None else int(traffic), "traffic_source": traffic_source,
        "launch": "all launches of the kernel in one step (the merged top levels, then one per "
                  "remaining hierarchy level) taken as one",
        "avg_launch_us": round(dom_avg_s * 1e6, 2), "event_samples": dom_n,
        "algorithmic_bytes_per_step": int(alg_bytes),
    }
    # the same kernel against the vector-instruction roof (wave instructions per step
    # from the committed counter pass over the kernel's live duration): whichever
    # fraction is higher names the bound
    insts, insts_source = pmc_valu_insts(setup.name, overridden, KERNEL_NAMES[dominant],
                                         {"fused": "true", "transform": "false"}.get(dominant))
    if insts and dom_avg_s > 0:
        valu_frac = insts / dom_avg_s / VALU_PEAK_INSTS_PER_S
        roofline["roofline_valu"] = {
            "insts_per_step": int(insts), "source": insts_source,
            "peak_insts_per_s": VALU_PEAK_INSTS_PER_S,
            "peak_from": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction",
            "frac": round(valu_frac, 4)}
        if valu_frac > roofline["frac"]:
            roofline["bound"] = "valu"
    stage_us = {k: round(v[0] * 1e3 / max(v[1], 1), 2) for k, v in warm.items() if v[1]}
    exchange_us = None
    if native_exchange:
        exchange_us = {k: round(timers[k][0] * 1e3 / max(timers[k][1], 1), 2) for k in EXCHANGE_TIMERS
                       if timers[k][1]}
    return elapsed, roofline, stage_us, info, exchange_us


CANDIDATE_ROW_BYTES = 144            # VkrWorldDrawCandidate
CANDIDATE_IN_BYTES = 64 + 32 + 16 + 4  # world matrix, identity words, local sphere, candidate word


def measure_dropin(setup, steps, fence) -> dict:
    """The frame a drop-in behind application_build_world_payload runs: the timed
    stages plus the 144-byte VkrWorldDrawCandidate rows of every source row
    (STAGE_CANDIDATES), once with the outputs left in HBM (a backend that consumes
    device memory) and once with the host copies the Vulkan backend wants: frame
    counters, candidate rows, sorted draws and instances."""
    import ctypes as C
    from vulkan_renderer_amd import _native as N
    hs, n = setup.hs, setup.n
    lib = hs.lib
    stages = N.STAGE_ALL | N.STAGE_CANDIDATES

    def frame():
        hs.mark_all(N.DIRTY_WORLD)
        hs.frame(stages)

    frame()
    fence()
    hs.set_profiling(["candidates"])
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    fence()
    device_s = (time.perf_counter() - t0) / steps
    cand_ms, cand_n = hs.timers(reset=True)["candidates"]
    hs.set_profiling(False)
    cand_s = cand_ms * 1e-3 / max(cand_n, 1)

    info = hs.frame_info()
    t = info.transparent_draw_count
    rows = np.empty(n * CANDIDATE_ROW_BYTES, np.uint8)
    draws = np.empty(max(t, 1) * 48, np.uint8)
    inst = np.empty(max(t, 1) * 80, np.uint8)
    raw = N.FrameInfo()
    host_steps = max(1, min(steps, 3))
    t0 = time.perf_counter()
    for _ in range(host_steps):
        frame()
        hs._ok(lib.vkrhip_get_frame_info(hs.ctx, C.byref(raw)))   # waits for the frame
        hs._ok(lib.vkrhip_read_candidates(hs.ctx, 0, n, N.ptr(rows)))
        hs._ok(lib.vkrhip_read_transparent(hs.ctx, raw.transparent_draw_count, N.ptr(draws),
                                           N.ptr(inst), None, None))
    host_s = (time.perf_counter() - t0) / host_steps
    # the same copies into page-locked memory from the library (vkrhip_host_alloc)
    rows_p = hs.host_buffer(n * CANDIDATE_ROW_BYTES)
    draws_p, inst_p = hs.host_buffer(max(t, 1) * 48), hs.host_buffer(max(t, 1) * 80)
    t0 = time.perf_counter()
    for _ in range(host_steps):
        frame()
        hs._ok(lib.vkrhip_get_frame_info(hs.ctx, C.byref(raw)))
        hs._ok(lib.vkrhip_read_candidates(hs.ctx, 0, n, N.ptr(rows_p)))
        hs._ok(lib.vkrhip_read_transparent(hs.ctx, raw.transparent_draw_count, N.ptr(draws_p),
                                           N.ptr(inst_p), None, None))
    pinned_s = (time.perf_counter() - t0) / host_steps
    same = bool((rows_p == rows).all() and (draws_p == draws).all() and (inst_p == inst).all())
    del rows_p, draws_p, inst_p
    hs.host_buffers_free()
    emit_bytes = n * (CANDIDATE_IN_BYTES + CANDIDATE_ROW_BYTES)
    return {
        "stages": "transform + cull + sort + 144-byte candidate rows (STAGE_ALL | STAGE_CANDIDATES)",
        "ms_per_step": device_s * 1e3, "value": n / device_s,
        "ms_per_step_with_host_copies": host_s * 1e3, "value_with_host_copies": n / host_s,
        "ms_per_step_with_host_copies_pinned": pinned_s * 1e3, "value_with_host_copies_pinned": n / pinned_s,
        "pinned_copies_equal_pageable": same,
        "host_copies": f"frame counters, {n} candidate rows ({n * CANDIDATE_ROW_BYTES / 1e6:.0f} MB), "
                       f"{t} sorted draws + instances into pageable host memory, every frame; _pinned: "
                       "the same into page-locked memory from vkrhip_host_alloc "
                       "(the reference caps a frame at 262144 rows = 38 MB, see at_reference_capacity)",
        "roofline_emit_candidates": {
            "bound": "hbm", "kernel": "k_emit_candidates", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": round(emit_bytes / cand_s / 1e9, 1) if cand_s > 0 else 0.0,
            "frac": round(emit_bytes / cand_s / 1e9 / HBM_PEAK_GBS, 4) if cand_s > 0 else 0.0,
            "avg_launch_us": round(cand_s * 1e6, 2),
            "algorithmic_bytes_per_step": int(emit_bytes),
        },
    }


REFERENCE_ROW_CAPACITY = 262144  # VKR_GPU_DRAW_CANDIDATE_CAPACITY, application.h:703-711


def measure_reference_capacity(local_rank: int, seed: int, fence, cpu_seconds: float) -> dict:
    """The largest frame the reference itself accepts (262144 source rows), as a
    drop-in sees it: the whole path with candidate rows, then the reference's own
    output struct filled by vkrhip_fill_world_payload (arrays in page-locked host
    memory, the host blocked until they are there) - next to the CPU restatement
    on the same scene (one host core)."""
    import ctypes as C
    from oracle import pyoracle
    from vulkan_renderer_amd import _native as N
    workload = dict(WORKLOADS["deep16m"], n=REFERENCE_ROW_CAPACITY)
    setup = Setup("capacity", workload, 0, local_rank, seed, 1)
    hs, n, lib, sc = setup.hs, setup.n, setup.hs.lib, setup.sc
    stages = N.STAGE_ALL | N.STAGE_CANDIDATES
    payload, stats = N.WorldPayload(), N.VisibilityStats()

    def frame():  # what a call-through behind application_build_world_payload does every frame
        hs.mark_all(N.DIRTY_WORLD)
        hs.frame(stages)
        hs._ok(lib.vkrhip_fill_world_payload(hs.ctx, C.byref(payload), C.byref(stats)))

    for _ in range(3):
        frame()
    fence()
    steps = 50
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    gpu_s = (time.perf_counter() - t0) / steps
    t = payload.transparent_draw_count
    rows = C.string_at(payload.gpu_candidates, payload.gpu_candidate_count * CANDIDATE_ROW_BYTES)
    draws = C.string_at(payload.transparent_draws, t * 48) if t else b""
    inst = C.string_at(payload.instances, t * 80) if t else b""

    pyoracle.build()
    o = pyoracle.Oracle()
    local, world, _, _ = o.scene_update(sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale,
                                        flags=np.full(sc.n, N.DIRTY_LOCAL, np.uint8))
    pay = o.build_world_payload(world, None, sc.ident, sc.matbits, sc.bounds, setup.vs.view,
                                setup.vs.projection)
    # the reference's loop structure as one C call per frame, like cpu_baseline()
    h = o.baseline_create(sc.parent, local, None, sc.ident, sc.matbits, sc.bounds, setup.planes,
                          setup.required, setup.vs.view, setup.vs.projection)
    o.baseline_frame(h)
    frames, t0 = 0, time.perf_counter()
    while frames < 3 or time.perf_counter() - t0 < min(cpu_seconds, 1.0):
        o.baseline_frame(h)
        frames += 1
    cpu_s = (time.perf_counter() - t0) / frames
    o.baseline_destroy(h)
    same = bool(rows == pay["gpu_candidates"].tobytes() and draws == pay["draws"].tobytes()
                and inst == pay["instances"].tobytes()
                and stats.objects_tested == int(pay["stats"][0])
                and stats.objects_culled_camera == int(pay["stats"][1]))
    setup.close()
    return {
        "rows": n, "views": setup.views, "hierarchy_depth": len(sc.level_offsets) - 1,
        "ms_per_frame": gpu_s * 1e3, "value": n / gpu_s,
        "what": "vkrhip_frame(transform + cull + sort + candidate rows) + vkrhip_fill_world_payload: "
                f"the VkrWorldPassPayload with its candidate rows ({n * CANDIDATE_ROW_BYTES / 1e6:.0f} MB), "
                f"{t} sorted draws + instances in page-locked host memory and the visibility stats, "
                "host blocked until they are there, every frame",
        "cpu_ms_per_frame": cpu_s * 1e3, "cpu_value": n / cpu_s, "cpu_cores": 1,
        "host_arrays_equal_cpu": same,
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="deep16m", choices=sorted(WORKLOADS))
    ap.add_argument("--instances", type=int, default=0, help="override instances per GPU")
    ap.add_argument("--depth", type=int, default=-1, help="override hierarchy depth")
    ap.add_argument("--cascades", type=int, default=-1, help="override shadow cascade count")
    ap.add_argument("--probes", type=int, default=-1, help="override extra probe views")
    ap.add_argument("--submeshes", type=int, default=1,
                    help="candidate rows per transform (>1: the stand-alone transform and classify "
                         "kernels run instead of the fused one)")
    ap.add_argument("--dirty", default="world", choices=["world", "local", "subtrees"],
                    help="what moves every frame: world = cached local matrices are re-propagated "
                         "(default); local = every node's TRS changed, locals are rebuilt too; subtrees = a "
                         "tenth of the subtrees below hierarchy level 2, the rest of the scene is clean")
    ap.add_argument("--dump-merged", default="",
                    help="with an exchange: rank 0 writes the merged visible lists, ranges and draw keys of "
                         "the last frame to this .npz file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other single-GPU workloads (flat1m, flat32m, probes64)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the drop-in frame (candidate rows + host copies)")
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

    from vulkan_renderer_amd.parallel import ExchangeOverflow, RcclExchange, ShardExchange

    workload = dict(WORKLOADS[args.workload])
    if args.instances:
        workload["n"] = args.instances
    for key in ("depth", "cascades", "probes"):
        if getattr(args, key) >= 0:
            workload[key] = getattr(args, key)
    overridden = (bool(args.instances) or args.dirty != "world" or args.submeshes != 1 or
                  any(getattr(args, k) >= 0 for k in ("depth", "cascades", "probes")))
    setup = Setup(args.workload, workload, rank, local_rank, args.seed, args.submeshes, world)
    hs, sc, n, views = setup.hs, setup.sc, setup.n, setup.views
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, roofline, stage_us, info, exchange_us = measure(setup, args.steps, args.warmup, args.dirty,
                                                             args.submeshes, exchange, fence, overridden)
    rank_ms = [elapsed * 1e3 / args.steps]
    rank_frac = [roofline["frac"]]
    if world > 1:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed, roofline["frac"]], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(e[0]) * 1e3 / args.steps for e in every]
        rank_frac = [float(e[1]) for e in every]
        elapsed = max(float(e[0]) for e in every)
    if exchange is not None:
        regrown = exchange.regrown
        try:
            exchange.finish()
        except ExchangeOverflow as exc:
            print(f"bench.py: {exc}", file=sys.stderr)
            return 3
        if exchange.regrown != regrown:
            print("bench.py: the exchange blocks overflowed inside the timed region", file=sys.stderr)
            return 3
        if args.dump_merged and rank == 0:
            v = views
            np.savez(args.dump_merged,
                     vis=exchange.vis_merged.cpu().numpy().view(np.uint32)[: exchange.visible_total],
                     keys=exchange.key_merged.cpu().numpy().view(np.uint64)[: exchange.transparent_total],
                     merged_ranges=exchange.merged_ranges, view_offsets=exchange.view_offsets)

    ms_per_step = elapsed * 1e3 / args.steps
    total_instances = setup.total  # (the shards of a hierarchy differ by a few nodes)
    value = total_instances * args.steps / elapsed

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
            "settling_frames": f"{setup.settling_frames} untimed frames of the workload ahead of the {args.warmup} "
                               f"warm-up steps ({SETTLE_SECONDS * 1e3:.0f} ms: the device's power management "
                               "settles about 12 ms after a new load starts; the frames in between run up to "
                               "12 % slower, profiles/r03_frame_sequence_after_idle.txt)",
            "timed_stages": "transform + multi-view cull + compaction + transparent key sort and draw "
                            "emission, outputs left in HBM; the 144-byte candidate rows of "
                            "application_build_world_payload are NOT in it (see dropin)",
            "streams": "scenes of 16K nodes and more with up to 8 views: scan, gather, sort and emission of a "
                       "frame run on a second stream beside the next frame's transform and classify, enqueued by "
                       "the context's worker thread (more "
                       "views: sort and emission only), so the dominant kernel's event time includes that "
                       "contention; VKRHIP_OVERLAP_TAIL=0 / VKRHIP_OVERLAP_SORT=0 keep one stream"
                       + ("; the exchange's pack follows the tail on the second stream, all-gather and merge "
                          "run on a third" if exchange is not None else ""),
            "exchange": ("one all-gather of packed blocks (visible lists + draw-key runs), "
                         "device-planned merge, overlapped with the next frame") if exchange is not None else "none",
        },
        "roofline": roofline,
        "whole_frame_frac": round(roofline["algorithmic_bytes_per_step"] / (ms_per_step * 1e-3) / 1e9
                                  / HBM_PEAK_GBS, 4),
        "stage_us_warmup": stage_us,
    }
    if exchange is not None:
        from vulkan_renderer_amd.parallel import exchange_block_words
        ranks_rccl = None
        if getattr(exchange, "scene", None) is not None and world > 1:
            try:
                ranks_rccl = hs.comm_count()
            except Exception:
                ranks_rccl = None
        result["config"]["ranks_seen"] = {
            "torch_distributed": dist.get_world_size() if world > 1 else 1,
            "rccl_communicator": ranks_rccl,
            "transport": "rccl (library)" if ranks_rccl else
                         ("none (one rank)" if world == 1 else f"torch.distributed/{args.backend}")}
        result["ms_per_step_ranks"] = {"min": min(rank_ms), "max": max(rank_ms)}
        result["roofline_frac_ranks"] = {"min": min(rank_frac), "max": max(rank_frac)}
        result["exchange"] = {
            "bytes_per_rank": 4 * exchange_block_words(views, exchange.vis_cap, exchange.key_cap),
            "us_pack": (exchange_us or {}).get("exchange_pack"),
            "us_allgather": (exchange_us or {}).get("exchange_gather"),
            "us_merge": (exchange_us or {}).get("exchange_merge"),
            "timed": "HIP events around the stages on the streams they run on (rank 0), in one of every "
                     f"{EVENT_STRIDE} timed steps; null: not timed (torch-driven exchange) or absent (one rank "
                     "has no all-gather)"}

    single = rank == 0 and world == 1
    want_capacity = False
    if single and not args.no_dropin and args.submeshes == 1 and exchange is None:
        result["dropin"] = measure_dropin(setup, min(args.steps, 10), fence)
        want_capacity = not args.no_cpu_baseline and not overridden
    if single and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sc, setup.planes, setup.required, setup.vs, args.cpu_seconds)
    if world > 1:
        dist.barrier()  # no rank tears its communicator down while another still uses it
    setup.close()
    del setup, hs, sc

    if want_capacity:
        result["dropin"]["at_reference_capacity"] = measure_reference_capacity(local_rank, args.seed, fence,
                                                                               args.cpu_seconds)

    if single and not args.no_secondary and not overridden and exchange is None:
        # the other single-GPU configurations, each with its own roofline (shorter runs)
        secondary = []
        moved_text = {"world": "every node: world matrices re-propagated from cached locals",
                      "local": "every node's TRS changed: local matrices rebuilt as well (48 B/node more in, "
                               "64 B/node more out)",
                      "subtrees": "a tenth of the subtrees below hierarchy level 2 (another tenth every frame); "
                                  "clean nodes read their stored world matrix and write none"}
        runs = [(name, "world") for name in ("flat1m", "flat32m", "probes64", "deep16m") if name != args.workload]
        runs += [(args.workload, "local")]
        if WORKLOADS[args.workload]["depth"] > 2:
            runs += [(args.workload, "subtrees")]
        for name, dirty in runs:
            other = Setup(name, dict(WORKLOADS[name]), 0, local_rank, args.seed, 1)
            steps = max(10, args.steps // 2)
            el, roof, _, inf, _ = measure(other, steps, max(3, args.warmup // 2), dirty, 1, None, fence,
                                          dirty != "world", min_seconds=0.02)
            steps = other.timed_steps
            secondary.append({
                "workload": f"{name}: {WORKLOADS[name]['desc']}", "instances_per_gpu": other.n,
                "views": other.views, "steps": steps, "settling_frames": other.settling_frames,
                "ms_per_step": el * 1e3 / steps,
                "moved_per_frame": moved_text[dirty], "nodes_updated": inf.nodes_updated,
                "value": other.n * steps / el, "roofline": roof,
                "whole_frame_frac": round(roof["algorithmic_bytes_per_step"] / (el / steps) / 1e9
                                          / HBM_PEAK_GBS, 4),
                "visible_camera": inf.views[0].visible_count,
                "transparent_draws": inf.transparent_draw_count})
            other.close()
            del other
        result["secondary"] = secondary
    if rank == 0:
        print(json.dumps(result), flush=True)
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
           "host": host_description(),
           "sample": f"{frames} frames of " + ("the first " if flat else "all ") +
                     f"{sample} instances of the workload, all {planes.shape[0]} views, {dt:.1f} s; "
                     "includes the 144-byte candidate emission of application_build_world_payload "
                     "(the GPU figure beside it: dropin)"}
    if flat:
        out["all_cores"] = cpu_baseline_all_cores(o, sc, sample, local, planes, required, vs,
                                                  min(10.0, budget_s))
    else:
        out["all_cores"] = cpu_baseline_all_cores_sharded(o, sc, local, planes, required, vs,
                                                          min(10.0, budget_s))
    return out


def host_description() -> dict:
    """The box the CPU figures were taken on: CPU model and logical CPUs as the
    kernel reports them, and the share this process may use."""
    model, logical = None, 0
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    logical += 1
                    if model is None:
                        model = line.split(":", 1)[1].strip()
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"model": model, "nproc": logical or (os.cpu_count() or 1), "usable_by_this_process": usable,
            "threads_env": os.environ.get("OMP_NUM_THREADS")}


def cpu_baseline_all_cores_sharded(o, sc, local, planes, required, vs, budget_s: float) -> dict:
    """Not in the reference (its path runs on the render thread alone): the hierarchy
    cut into one shard of whole subtrees per host thread (vkrhip_shard_hierarchy, the
    cut the multi-GPU path uses; levels above the cut are replicated), every thread
    looping over its own shard with the single-thread loop, no merge of the results."""
    import threading
    from vulkan_renderer_amd.scene import shard_hierarchy, shard_local_hierarchy
    threads = max(1, min(16, os.cpu_count() or 1))
    levels = len(sc.level_offsets) - 1
    # the first level that holds at least four subtrees per thread
    cut = next((l for l in range(levels) if sc.level_offsets[l + 1] - sc.level_offsets[l] >= 4 * threads),
               levels - 1)
    ranges = shard_hierarchy(sc.parent, sc.level_offsets, threads, replicated_levels=cut)
    frames = [0] * threads
    handles = [None] * threads
    sizes = [0] * threads
    start = threading.Barrier(threads + 1)
    stop_at = [0.0]

    def work(k):
        nodes, parent_k, _ = shard_local_hierarchy(sc.parent, ranges[k])
        idx = nodes.astype(np.int64)
        sizes[k] = int(idx.shape[0])
        handles[k] = o.baseline_create(parent_k, local[idx], None, sc.ident[idx], sc.matbits[idx],
                                       sc.bounds[idx], planes, required, vs.view, vs.projection)
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
    done = sum(f * n for f, n in zip(frames, sizes))
    return {"value": done / dt, "unit": "instances/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads, one shard of whole subtrees each (cut at hierarchy level {cut}, "
                      f"{min(sizes)}..{max(sizes)} nodes per shard, {sum(frames)} shard frames), no merge, "
                      f"{dt:.1f} s; not in the reference"}


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
