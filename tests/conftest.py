import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    from slideo_amd import _capi
    return _capi


@pytest.fixture(scope="session")
def synth():
    from slideo_amd import synth
    return synth


def small_cfg(mod, **over):
    """BASELINE config 0 shape: ORB-500; thresholds scaled for 640x360 (SURVEY §7 hard parts)."""
    kw = dict(nfeatures=500, min_rating=12.0)
    kw.update(over)
    return mod.default_config(**kw)


@pytest.fixture(scope="session")
def cfg0_data(synth):
    """8 synthetic 640x360 frames vs 4 pages of 800x450 (BASELINE configs[0])."""
    pages = synth.pages(4, 800, 450)
    frames, truth, tm = synth.frames(pages, 8, 640, 360)
    return pages, frames, truth, tm
