"""The C-ABI library on a machine without a GPU: it loads, exports every entry
point include/vkr_hip_scene.h declares, agrees with the header on struct
layouts, its host-side helpers match the oracle bit for bit, and creating a
context fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, _gpu_available, assert_bits_equal

HEADER = os.path.join(ROOT, "include", "vkr_hip_scene.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vkrhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vulkan_renderer_amd import _native as N
    names = declared_functions()
    assert len(names) >= 30
    lib = C.CDLL(N.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert sorted(N.EXPORTED_SYMBOLS) == names, "binding list and header disagree"


def test_struct_layouts_match_header(tmp_path):
    """Compile the header as C and compare sizeof/offsetof with the ctypes mirrors."""
    from vulkan_renderer_amd import _native as N
    probe = tmp_path / "probe.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vkr_hip_scene.h"\n'
        "int main(void){\n"
        ' printf("%zu %zu %zu %zu %zu\\n", sizeof(vkrhip_config), sizeof(vkrhip_frame_info),\n'
        "   sizeof(vkrhip_device_outputs), sizeof(vkrhip_compaction_state),\n"
        "   offsetof(vkrhip_config, source_index_base));\n return 0;}\n")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(probe), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes[0] == C.sizeof(N.Config)
    assert sizes[1] == C.sizeof(N.FrameInfo)
    assert sizes[2] == C.sizeof(N.DeviceOutputs)
    assert sizes[3] == C.sizeof(N.CompactionState) == 80
    assert sizes[4] == N.Config.source_index_base.offset


def test_host_frustum_extraction_matches_oracle(oracle):
    from vulkan_renderer_amd import build_view_set, frustum_from_matrix, mat4_mul, synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "frustum.npz"))
    for k in range(g["view"].shape[0]):
        vp = mat4_mul(g["projection"][k], g["view"][k])
        assert_bits_equal(vp.reshape(-1), g["view_projection"][k], "mat4_mul")
        assert_bits_equal(frustum_from_matrix(vp).reshape(-1), g["planes_matrix"][k], "planes")
    vs = synth.camera_views(cascades=4)
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    assert required.tolist() == [2, 4, 4, 4, 4]
    assert_bits_equal(planes[0].reshape(-1),
                      oracle.frustum_from_view_projection(vs.view, vs.projection).reshape(-1), "camera")
    for c in range(4):
        assert_bits_equal(planes[1 + c].reshape(-1),
                          oracle.frustum_from_matrix(vs.light_view_proj[c]).reshape(-1), "cascade")


def test_levelize_matches_oracle_topo_order(oracle):
    from vulkan_renderer_amd import levelize, synth
    sizes = synth.level_sizes(5000, 6, fanout=4, roots=3)
    parent, offsets = synth.build_parents(sizes)
    n = parent.shape[0]
    # scramble the node numbering, then ask both sides for the BFS order
    rng = np.random.default_rng(3)
    perm = rng.permutation(n).astype(np.uint32)      # new id of old node
    scrambled = np.full(n, 0xFFFFFFFF, np.uint32)
    has_parent = parent != 0xFFFFFFFF
    scrambled[perm[has_parent]] = perm[parent[has_parent]]
    order, new_parent, level_offsets = levelize(scrambled)
    topo, level, levels = oracle.build_topo_order(scrambled)
    assert levels == len(sizes) == level_offsets.shape[0] - 1
    assert (np.diff(level_offsets.astype(np.int64)) == np.array(sizes)).all()
    # same level partition; inside a level both orders list parents before children
    lvl_of = np.empty(n, np.int64)
    for l in range(levels):
        lvl_of[order[level_offsets[l]:level_offsets[l + 1]]] = l
    assert (lvl_of == level.astype(np.int64)).all()
    assert sorted(order.tolist()) == list(range(n))
    # the relabelled parent array describes the same tree in level order
    for new_id in range(0, n, 17):
        old = order[new_id]
        if scrambled[old] == 0xFFFFFFFF:
            assert new_parent[new_id] == 0xFFFFFFFF
        else:
            assert order[new_parent[new_id]] == scrambled[old]
            assert new_parent[new_id] < new_id


@pytest.mark.skipif(_gpu_available(), reason="only meaningful without a GPU")
def test_create_fails_loudly_without_gpu():
    from vulkan_renderer_amd import HipScene
    with pytest.raises(RuntimeError) as e:
        HipScene(max_nodes=16, max_candidates=16, max_views=1)
    assert "hip" in str(e.value).lower() or "device" in str(e.value).lower()
