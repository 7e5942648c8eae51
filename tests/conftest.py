import os
import sys

import pytest

# torch first: it brings its own copy of the HIP runtime, and a process that has already initialised the system's copy
# (through librt_amd.so) cannot initialise torch's afterwards ("No HIP GPUs are available").  Loaded in this order the
# library binds to the copy torch loaded, and the tests that hold accumulators in torch tensors work in any selection.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import ctypes
        from raytracing_amd import _ffi
        n = ctypes.c_int(0)
        return _ffi.lib().rt_device_count(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never fall back to anything: without a device they are skipped, with one they must pass
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


SCENES = ["cornell_box", "cornell_smoke_box", "cover1", "cu_cover2", "cover2", "teapot"]


@pytest.fixture(scope="session")
def scene_cache():
    """Scene name -> (world, kwargs); worlds are deterministic (seeded) so they can be shared."""
    import raytracing_amd as rt
    cache = {}

    def get(name, spp=8):
        key = (name, spp)
        if key not in cache:
            cache[key] = rt.get_scene(name, spp=spp)
        return cache[key]
    return get
