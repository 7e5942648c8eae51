"""rt_render_accumulate_static (the entry point that allocates nothing: its working buffers are a fixed array inside the
library) against rt_render_accumulate, through the C ABI: the un-normalised sums must be identical bit for bit, on two
calls in a row, the second with a smaller input, whatever the first left behind in the array."""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd.render import camera_record

pytestmark = pytest.mark.gpu
F = np.float32


def _sums(dev, fn, cam, pat, bg, W, H, first=0, stride=1, watch=None):
    """watch: a list that receives the device memory the call itself took (free before - free after); the accumulator is
    torch's (its allocator takes memory in 2 MiB pieces of its own) and is made before the first reading"""
    import torch
    acc = torch.full((3 * W * H,), float("nan"), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    fn(cam, pat, bg, W, H, acc.data_ptr(), sample_first=first, sample_stride=stride, stream=stream)
    torch.cuda.synchronize()
    if watch is not None:
        watch.append(free_before - torch.cuda.mem_get_info()[0])
    return acc.cpu().numpy()


def test_static_entry_point_equals_the_allocating_one():
    import torch
    world, kw = rt.get_scene("cu_cover2", spp=8)          # spheres, boxes, transforms, two constant media
    cam, bg = kw["camera"], kw["background"]
    dev = rt.DeviceScene(world)
    try:
        # first call: more paths than the static pool has slots (672 x 480 x 6 = 1.9 M paths, 256 Ki slots), three shards
        W, H, pat = 672, 480, rt.Halton_sequence32(18)
        want = _sums(dev, dev.accumulate, cam, pat, bg, W, H, 1, 3)
        took = []
        got = _sums(dev, dev.accumulate_static, cam, pat, bg, W, H, 1, 3, watch=took)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.isfinite(got).all() and got.max() > 0
        assert took[0] <= 0, "rt_render_accumulate_static took %d bytes of device memory (first, large call)" % took[0]
        # second call in a row, smaller input (fewer paths than slots): what the first call left in the array must not show,
        # and the call must not take device memory
        W2, H2, pat2 = 96, 64, rt.Halton_sequence32(5)
        took = []
        got2 = _sums(dev, dev.accumulate_static, cam, pat2, bg, W2, H2, watch=took)
        want2 = _sums(dev, dev.accumulate, cam, pat2, bg, W2, H2)
        np.testing.assert_array_equal(got2.view(np.uint32), want2.view(np.uint32))
        assert took[0] <= 0, "rt_render_accumulate_static took %d bytes of device memory" % took[0]
        # a third, again larger, with several column batches (1 Mi pixels: two columns per batch of the 2 Mi entry buffer)
        W3, H3, pat3 = 1024, 1024, rt.Halton_sequence32(5)
        took = []
        got3 = _sums(dev, dev.accumulate_static, cam, pat3, bg, W3, H3, watch=took)
        want3 = _sums(dev, dev.accumulate, cam, pat3, bg, W3, H3)
        np.testing.assert_array_equal(got3.view(np.uint32), want3.view(np.uint32))
        assert took[0] <= 0, "rt_render_accumulate_static took %d bytes of device memory (several column batches)" % took[0]
        # a camera far outside the scene: the call rebuilds the node boxes for that distance before it renders (the one
        # route on which the static entry point reaches the tree builder); that, too, must take no device memory -- the
        # builder works in memory the scene handle got when it was created
        from raytracing_amd.hittables import Camera, Vec3
        far = Camera(Vec3(478.0, 278.0, -60000.0), Vec3(0.0, 0.0, -1.0), Vec3(0, 1, 0), 2.0, 1.0, 0.0, 10.0, 0.0, 1.0)
        took = []
        got4 = _sums(dev, dev.accumulate_static, far, pat2, bg, 160, 160, watch=took)
        want4 = _sums(dev, dev.accumulate, far, pat2, bg, 160, 160)
        np.testing.assert_array_equal(got4.view(np.uint32), want4.view(np.uint32))
        assert took[0] <= 0, "rt_render_accumulate_static took %d bytes of device memory (rebuild for a far camera)" % took[0]
        assert got4.max() > 0       # (the scene is in the picture)
        # ... and back: the next call from the scene's own camera gets the tight boxes again, still without memory
        took = []
        got5 = _sums(dev, dev.accumulate_static, cam, pat2, bg, W2, H2, watch=took)
        np.testing.assert_array_equal(got5.view(np.uint32), want2.view(np.uint32))
        assert took[0] <= 0, "rt_render_accumulate_static took %d bytes of device memory (rebuild on the way back)" % took[0]
        # an empty shard is a zero sum
        zero = _sums(dev, dev.accumulate_static, cam, pat2, bg, W2, H2, 7, 8)
        assert not zero.any()
        # beyond the limits: a clear error, nothing rendered
        L = ffi.lib()
        c = camera_record(cam)
        acc = torch.zeros(3 * 64 * 64, dtype=torch.float32, device="cuda")
        big = np.zeros(2 * (ffi.RT_STATIC_MAX_COLUMNS + 1), F)
        b = np.asarray(bg, F)
        assert L.rt_render_accumulate_static(dev.handle, C.byref(c), ffi.fptr(big), ffi.RT_STATIC_MAX_COLUMNS + 1, 0, 1, ffi.fptr(b),
                                             64, 64, 50, 0, C.c_void_p(acc.data_ptr()), None) == -1
        assert b"pattern columns" in L.rt_last_error()
        assert L.rt_render_accumulate_static(dev.handle, C.byref(c), ffi.fptr(big), 4, 0, 1, ffi.fptr(b), 2048, 1025, 50, 0,
                                             C.c_void_p(acc.data_ptr()), None) == -1
        assert b"pixels" in L.rt_last_error()
    finally:
        dev.close()
