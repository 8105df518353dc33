"""The library driven by a plain C program (examples/scene_frame.c, built with
gcc against include/vkr_hip_scene.h): scene in, frame results out through files,
compared bit for bit with the CPU oracle. No Python or C++ on the host side of
the library in this test."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "vulkan_renderer_amd", "scene_frame")


@pytest.mark.parametrize("identity", [True, False])
def test_c_host_frame(oracle, tmp_path, identity):
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd import synth
    assert os.path.exists(BINARY), "build the C host first (vulkan_renderer_amd.build.build_native)"
    sc = synth.deep_scene(5000, depth=5, seed=12, fanout=5, roots=3, blend_fraction=0.25,
                          transmissive_fraction=0.1, no_bounds_fraction=0.05)
    if not identity:
        rng = np.random.default_rng(2)
        rows = 7001
        sc.node = rng.integers(0, sc.n, rows).astype(np.uint32)
        sc.ident, sc.matbits, sc.bounds = synth.make_candidates(5, np.arange(rows, dtype=np.uint64),
                                                                blend_fraction=0.25,
                                                                transmissive_fraction=0.1,
                                                                no_bounds_fraction=0.05)
    node = sc.node if sc.node is not None else np.arange(sc.n, dtype=np.uint32)
    rows = node.shape[0]
    vs = synth.camera_views(cascades=2, far=900.0)
    planes, required = view_set(vs)
    views = planes.shape[0]
    scene_path, result_path = tmp_path / "scene.bin", tmp_path / "result.bin"
    payload_path = tmp_path / "payload.bin"
    with open(scene_path, "wb") as f:
        np.array([sc.n, sc.level_offsets.shape[0] - 1, rows, views], np.uint64).tofile(f)
        for arr, dt in ((sc.parent, np.uint32), (sc.level_offsets, np.uint64), (sc.pos, np.float32),
                        (sc.quat, np.float32), (sc.scale, np.float32), (node, np.uint32),
                        (sc.ident, np.uint32), (sc.matbits, np.uint32), (sc.bounds, np.float32),
                        (planes, np.float32), (required, np.uint32), (vs.view, np.float32)):
            np.ascontiguousarray(arr, dtype=dt).tofile(f)
    run = subprocess.run([BINARY, str(scene_path), str(result_path), str(payload_path)],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)

    import ctypes as C
    raw = open(result_path, "rb").read()
    info = N.FrameInfo.from_buffer_copy(raw[: C.sizeof(N.FrameInfo)])
    pos = C.sizeof(N.FrameInfo)
    world = np.frombuffer(raw, np.float32, sc.n * 16, pos).reshape(sc.n, 16)
    pos += sc.n * 64
    assert world.tobytes() == ref["world"].tobytes()
    assert info.nodes_updated == sc.n
    for v in range(views):
        st = ref["states"][v]
        count = info.views[v].visible_count
        assert count == int(st["visible_count"])
        got = np.frombuffer(raw, np.uint32, count, pos)
        pos += count * 4
        assert (got == ref["visible"][v, :count]).all(), f"view {v}"
    draws = info.transparent_draw_count
    pay = ref["payload"]
    assert draws == pay["sort_keys"].shape[0] > 10
    keys = np.frombuffer(raw, np.uint64, draws, pos)
    pos += draws * 8
    assert (keys == pay["sort_keys"]).all()
    assert raw[pos: pos + draws * 48] == pay["draws"].tobytes()
    assert pos + draws * 48 == len(raw)

    # the reference's VkrWorldPassPayload as vkrhip_fill_world_payload hands it back
    blob = open(payload_path, "rb").read()
    head = N.WorldPayload.from_buffer_copy(blob[:88])
    stats = np.frombuffer(blob, np.uint32, 3, 88)
    assert (stats == pay["stats"]).all()
    assert head.gpu_candidate_count == rows == head.gpu_shadow_candidate_count
    assert head.gpu_camera_opaque_candidate_count == int(pay["counts"][1])
    nt = int(pay["counts"][2])
    assert head.transmission_gpu_candidate_count == nt > 100
    assert head.transparent_draw_count == head.instance_count == draws
    assert head.text_draw_count == 0 and not head.text_draws
    at = 100
    assert blob[at: at + rows * 144] == pay["gpu_candidates"].tobytes()
    at += rows * 144
    assert blob[at: at + nt * 144] == pay["transmission_candidates"].tobytes()
    at += nt * 144
    assert blob[at: at + draws * 48] == pay["draws"].tobytes()
    at += draws * 48
    assert blob[at: at + draws * 80] == pay["instances"].tobytes()
    assert at + draws * 80 == len(blob)


def test_payload_overflow_convention():
    """More source rows than the reference's candidate capacity: an empty payload
    with gpu_candidate_count = capacity + 1 (application.h:703-711), no error."""
    import ctypes as C
    from vulkan_renderer_amd import HipScene, _native as N, synth
    n = 262144 + 5
    sc = synth.flat_scene(n, seed=2)
    with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        pay, stats = N.WorldPayload(), N.VisibilityStats()
        assert hs.lib.vkrhip_fill_world_payload(hs.ctx, C.byref(pay), C.byref(stats)) == N.OK
        assert pay.gpu_candidate_count == 262145 and stats.objects_tested == 262145
        assert not pay.gpu_candidates and pay.transparent_draw_count == 0
