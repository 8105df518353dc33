import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own code compiled into oracle/_ref (absent off this box)."""
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/libvkr_ref.so not built (no reference checkout here)")
    return pyoracle.Reference()


def bits(a):
    """Bit view for exact float comparison."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ne = a.view(np.uint32) != b.view(np.uint32)
    if ne.any():
        idx = np.argwhere(ne)[:5]
        raise AssertionError(f"{what}: {int(ne.sum())} of {ne.size} floats differ bitwise, first at "
                             f"{idx.tolist()}: {a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")


@pytest.fixture(params=["default", "staged"])
def staged_mode(request, monkeypatch):
    """Runs a test twice: as the library decides by itself (small frames leave the
    expansion of the survivor masks to the gather kernel) and with
    VKRHIP_STAGED=2, where the classify blocks of every frame the variants exist
    for expand their own survivors and k_place moves the staged runs (what large
    frames do by default). The switch is read when a context is created."""
    if request.param == "staged":
        monkeypatch.setenv("VKRHIP_STAGED", "2")
    return request.param


@pytest.fixture(params=["default", "tail", "tail_one_thread", "tail_events"])
def tail_mode(request, monkeypatch):
    """Runs a test several times: as the library decides by itself (small scenes keep
    the whole frame on one stream) and with VKRHIP_OVERLAP_TAIL=2, where every frame
    the fused kernel serves puts scan, gather, sort and emission on the second stream
    and the next frame starts right behind the classify (what scenes of 16K nodes and
    more do by default) - with the tail enqueued by the context's worker thread and
    the streams ordered by signal and gate kernels (the default), with the calling
    thread enqueuing everything (VKRHIP_TAIL_WORKER=0), and with events between the
    streams (VKRHIP_SIDE_GATE=0). The switches are read when a context is created."""
    if request.param != "default":
        monkeypatch.setenv("VKRHIP_OVERLAP_TAIL", "2")
    if request.param == "tail_one_thread":
        monkeypatch.setenv("VKRHIP_TAIL_WORKER", "0")
    if request.param == "tail_events":
        monkeypatch.setenv("VKRHIP_SIDE_GATE", "0")
    return request.param
