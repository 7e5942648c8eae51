"""Parity of the HIP path (through the C ABI, raytracing_amd.DeviceScene) with the CPU oracle and the committed golden
vectors.  Bar (integer / index work and, because both sides evaluate the same IEEE operations in the same order, floats too):
bit-exact primitive ids, hit distances, normals, per-path radiance and images."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from conftest import SCENES, blocks_showing_the_tall_box_front, cornell_box_of_meshes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
F = np.float32


def bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint32)


@pytest.fixture(scope="module")
def scenes(scene_cache):
    cache = {}

    def get(name):
        if name not in cache:
            world, kw = scene_cache(name, 8)
            cache[name] = (world, kw, rt.DeviceScene(world), O.OracleScene(world))
        return cache[name]
    yield get
    for _, _, dev, _ in cache.values():
        dev.close()


@pytest.mark.parametrize("name", SCENES)
def test_golden_vectors(name, scenes):
    """The committed vectors (closest hits of 1024 fixed rays, 256 path radiances, a 24x16 image) reproduced bit for bit."""
    from make_golden import golden_inputs, W, H
    g = np.load(os.path.join(HERE, "golden", f"golden_{name}.npz"))
    world, cam, pat, bg, rays, ids = golden_inputs(name)
    dev = rt.DeviceScene(world)
    try:
        hits = dev.trace_rays(g["rays"])
        np.testing.assert_array_equal(hits["prim_id"], g["hit_prim_id"])          # check (1): ids bit-exact
        np.testing.assert_array_equal(hits["sub_id"], g["hit_sub_id"])
        np.testing.assert_array_equal(bits(hits["t"]), bits(g["hit_t"]))           # check (2): exceeds the 1e-5 bar
        np.testing.assert_array_equal(bits(hits["normal"]), bits(g["hit_normal"]))
        np.testing.assert_array_equal(bits(hits["uv"]), bits(g["hit_uv"]))
        np.testing.assert_array_equal(hits["front"], g["hit_front"])
        np.testing.assert_array_equal(bits(dev.trace_paths(cam, pat, bg, W, H, g["ids"])), bits(g["radiance"]))
        img, _ = dev.render(cam, pat, bg, W, H)
        np.testing.assert_array_equal(bits(img), bits(g["image"]))
    finally:
        dev.close()


@pytest.mark.parametrize("name", SCENES)
def test_primary_and_secondary_rays_vs_oracle(name, scenes):
    """Fixed ray sets: a dense primary grid, and incoherent rays started inside the scene in random directions."""
    world, kw, dev, ref = scenes(name)
    cam = kw["camera"]
    rays = O.primary_rays(cam, kw["width"], kw["height"], 160, 120, time=0.25)
    rng = np.random.default_rng(11)
    a = ref.trace_rays(rays)
    pts = rays[:, :3] + rays[:, 3:6] * np.where(np.isfinite(a["t"]), a["t"] * 0.9, 1.0)[:, None]
    d = rng.normal(size=(len(rays), 3)).astype(F)
    sec = np.concatenate([pts, d, rng.uniform(0, 1, (len(rays), 1))], axis=1).astype(F)
    for batch in (rays, sec):
        a = ref.trace_rays(batch)
        b = dev.trace_rays(batch)
        np.testing.assert_array_equal(a["prim_id"], b["prim_id"])
        np.testing.assert_array_equal(a["sub_id"], b["sub_id"])
        np.testing.assert_array_equal(bits(a["t"]), bits(b["t"]))
        np.testing.assert_array_equal(bits(a["normal"]), bits(b["normal"]))
        np.testing.assert_array_equal(bits(a["uv"]), bits(b["uv"]))
        np.testing.assert_array_equal(a["front"], b["front"])
        assert (a["prim_id"] >= 0).sum() > 1000


@pytest.mark.parametrize("name", SCENES)
def test_paths_and_image_vs_oracle(name, scenes):
    world, kw, dev, ref = scenes(name)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, H = 72, 40                       # not multiples of 8 or 64: ragged tiles
    ids = np.array([(i, j, n) for j in range(0, H, 3) for i in range(0, W, 2) for n in range(0, 8, 3)], np.int32)
    np.testing.assert_array_equal(bits(ref.trace_paths(cam, pat, bg, W, H, ids)), bits(dev.trace_paths(cam, pat, bg, W, H, ids)))
    img, st = dev.render(cam, pat, bg, W, H)
    want = ref.render(cam, pat, bg, W, H)
    # check (3): the image is not merely within an RMSE, it is identical (same paths, same summation order)
    np.testing.assert_array_equal(bits(img), bits(want))
    assert st.paths == W * H * 8


def test_cornell_box_of_meshes_vs_oracle():
    """one mesh of 36 triangles with four materials picked per triangle (the teapot has one): ids, t, normals, uv and
    the image bit for bit"""
    world, kw = cornell_box_of_meshes()
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    dev, ref = rt.DeviceScene(world), O.OracleScene(world)
    try:
        rays = O.primary_rays(cam, kw["width"], kw["height"], 96, 96, time=0.5)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        for k in ("prim_id", "sub_id", "front"):
            np.testing.assert_array_equal(a[k], b[k])
        for k in ("t", "normal", "uv"):
            np.testing.assert_array_equal(bits(a[k]), bits(b[k]))
        assert len(set(a["sub_id"][a["prim_id"] >= 0])) >= 20          # (sub_id = the triangle: most of the 36 are seen)
        img, _ = dev.render(cam, pat, bg, 80, 80)
        np.testing.assert_array_equal(bits(img), bits(ref.render(cam, pat, bg, 80, 80)))
    finally:
        dev.close()


def test_edge_cases(scenes, monkeypatch):
    world, kw, dev, ref = scenes("cornell_smoke_box")
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(5)
    # smallest legal image, one pattern column, MSAA table as pattern, depth limits
    for W, H, p, depth in ((2, 2, pat[:, :1], 50), (9, 3, rt.MSAA_S16.astype(F), 50), (16, 16, pat, 1), (16, 16, pat, 2), (8, 8, pat, 0)):
        img, _ = dev.render(cam, p, bg, W, H, max_depth=depth)
        np.testing.assert_array_equal(bits(img), bits(ref.render(cam, p, bg, W, H, max_depth=depth)))
    # few slots (every slot is reused many times, the path dispensers hand out ragged last units) and a tiny radiance
    # budget (several column batches): same image
    want = ref.render(cam, pat, bg, 40, 24)
    big = ref.render(cam, pat, bg, 150, 110)
    for slots, mib in (("1024", "1"), ("16384", "1"), ("49152", "12288")):
        monkeypatch.setenv("RT_AMD_SLOTS", slots)
        monkeypatch.setenv("RT_AMD_RADIANCE_MIB", mib)
        img, _ = dev.render(cam, pat, bg, 40, 24)
        np.testing.assert_array_equal(bits(img), bits(want))
        img, st = dev.render(cam, pat, bg, 150, 110)
        np.testing.assert_array_equal(bits(img), bits(big))
        assert st.paths == 150 * 110 * 5
    monkeypatch.delenv("RT_AMD_SLOTS")
    monkeypatch.delenv("RT_AMD_RADIANCE_MIB")
    monkeypatch.setenv("RT_AMD_NO_COMPACT", "1")     # moving the last live slots together only changes where a path is kept
    img, _ = dev.render(cam, pat, bg, 150, 110)
    np.testing.assert_array_equal(bits(img), bits(big))
    monkeypatch.delenv("RT_AMD_NO_COMPACT")
    monkeypatch.setenv("RT_AMD_NO_REGROUP", "1")     # material regrouping only changes which lane shades which slot
    img, _ = dev.render(cam, pat, bg, 40, 24)
    np.testing.assert_array_equal(bits(img), bits(want))
    monkeypatch.delenv("RT_AMD_NO_REGROUP")
    # a different seed gives a different (but again reproducible) image
    a, _ = dev.render(cam, pat, bg, 40, 24, seed=7)
    b, _ = dev.render(cam, pat, bg, 40, 24, seed=7)
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, img)
    np.testing.assert_array_equal(bits(a), bits(ref.render(cam, pat, bg, 40, 24, seed=7)))
    # counters variant renders the same image and reports plausible counts
    c, st = dev.render(cam, pat, bg, 40, 24, flags=ffi.RT_FLAG_COUNTERS)
    np.testing.assert_array_equal(c, img)
    assert st.rays >= st.paths > 0 and st.node_visits > st.rays


def test_the_size_of_the_slot_pool_changes_no_picture(monkeypatch):
    """A frame of 128 Mi paths or more over a tree at least eight levels deep is given 32 Mi slots instead of 4 Mi (fewer,
    longer launches: fewer ends of launches with the waves running empty), looked at every 8 bounces instead of every
    16 and compacted as often once its unstarted paths have run out; a scene of a dozen primitives keeps 4 Mi whatever its
    length.  Which slot a path lives in changes nothing: pictures are equal bit for bit."""
    def frame(dev, kw, n, W, H, slots):
        if slots:
            monkeypatch.setenv("RT_AMD_SLOTS", str(slots))
        else:
            monkeypatch.delenv("RT_AMD_SLOTS", raising=False)
        img, st = dev.render(kw["camera"], rt.Halton_sequence32(n), kw["background"], W, H, flags=ffi.RT_FLAG_TIMING)
        return img, st.extend_launches
    world, kw = rt.get_scene("cu_cover2", spp=1)
    dev = rt.DeviceScene(world)
    try:
        # a pool above 8 Mi slots with fewer slots than paths: the large pool's batches, its dispensers and its compaction
        a, la = frame(dev, kw, 80, 512, 512, 9 << 20)
        b, lb = frame(dev, kw, 80, 512, 512, 0)
        np.testing.assert_array_equal(bits(a), bits(b))
        # (the batch in which the unstarted paths run out is cut short, and the draining pool is compacted every 4 bounces:
        # without either the same picture from fewer looks)
        monkeypatch.setenv("RT_AMD_NO_TAIL_PREDICTION", "1")
        b2, lb2 = frame(dev, kw, 80, 512, 512, 0)
        monkeypatch.delenv("RT_AMD_NO_TAIL_PREDICTION")
        np.testing.assert_array_equal(bits(b), bits(b2))
        assert lb2 % 16 == 0 and lb != lb2, (lb, lb2)
        # 210 columns of 800 x 800 are 128.2 Mi paths: the library takes 32 Mi slots by itself
        c, lc = frame(dev, kw, 210, 800, 800, 0)
        d, ld = frame(dev, kw, 210, 800, 800, 4 << 20)
        np.testing.assert_array_equal(bits(c), bits(d))
        assert lc * 3 < ld, (lc, ld)
    finally:
        dev.close()
    world, kw = rt.get_scene("cornell_box", spp=1)
    dev = rt.DeviceScene(world)
    try:
        e, le = frame(dev, kw, 840, 400, 400, 0)           # as many paths, a tree of three levels: 4 Mi slots
        f, lf = frame(dev, kw, 840, 400, 400, 4 << 20)
        g, lg = frame(dev, kw, 840, 400, 400, 32 << 20)
        np.testing.assert_array_equal(bits(e), bits(f))
        np.testing.assert_array_equal(bits(e), bits(g))
        assert le == lf and lg * 3 < lf, (le, lf, lg)
    finally:
        dev.close()


@pytest.mark.parametrize("width,height", [(1024, 64), (512, 128)])
def test_uneven_columns_drain_every_path_dispenser(width, height, monkeypatch):
    """An image whose width is a multiple of 256 maps every path dispenser to a fixed band of columns; here the left
    half of the picture is the sky (one ray per path) and the right half looks into a deep white cavity (14 rays per
    path on average), so the dispensers run dry at very different times, with few slots and the end-of-frame compaction
    at work.  Every path must still be traced: the image is the oracle's bit for bit, and the picture at full size
    (1024 x 1024) does not depend on the number of slots."""
    from raytracing_amd.hittables import (Camera, Dielectric, Lambertian, Material, Metal, Sphere, Vec3, xy_rect, xz_rect,
                                          yz_rect)
    white = Material(Lambertian, Vec3(0.97, 0.97, 0.97))
    world = [yz_rect(-100, 100, 0, 400, -200, white), yz_rect(-100, 100, 0, 400, -5, white),
             xz_rect(-200, -5, 0, 400, -100, white), xz_rect(-200, -5, 0, 400, 100, white),
             xy_rect(-200, -5, -100, 100, 400, white), Sphere(Vec3(-100, 0, 200), 60, Material(Metal, Vec3(0.9, 0.9, 0.9), 0.0)),
             Sphere(Vec3(-60, 0, 60), 30, Material(Dielectric, Vec3(1, 1, 1), 0.0, 1.5))]
    origin = Vec3(0, 0, -260)
    cam = Camera(origin, origin - Vec3(0, 0, 50), Vec3(0, 1, 0), 60.0, width / height, 0.0, 10.0, 0.0, 1.0)
    bg = Vec3(0.7, 0.8, 1.0)
    pat = rt.Halton_sequence32(48)
    ref, dev = O.OracleScene(world), rt.DeviceScene(world)
    try:
        monkeypatch.setenv("RT_AMD_SLOTS", "65536")   # 64 path dispensers, and a pool that shrinks far below 64 workgroups
        img, st = dev.render(cam, pat, bg, width, height)
        want = ref.render(cam, pat, bg, width, height)
        np.testing.assert_array_equal(bits(img), bits(want))
        assert st.paths == width * height * 48
        left, right = img[:, :width // 4].mean(), img[:, -width // 4:].mean()
        assert abs(left - np.sqrt(np.array([0.7, 0.8, 1.0])).mean()) < 1e-3 and right < 0.8  # sky on the left only
        if width == 1024:
            cam2 = Camera(origin, origin - Vec3(0, 0, 50), Vec3(0, 1, 0), 60.0, 1.0, 0.0, 10.0, 0.0, 1.0)
            pat2 = rt.Halton_sequence32(12)
            a, _ = dev.render(cam2, pat2, bg, 1024, 1024)
            monkeypatch.setenv("RT_AMD_SLOTS", "16384")
            b, _ = dev.render(cam2, pat2, bg, 1024, 1024)
            monkeypatch.delenv("RT_AMD_SLOTS")
            c, _ = dev.render(cam2, pat2, bg, 1024, 1024)
            np.testing.assert_array_equal(bits(a), bits(c))
            np.testing.assert_array_equal(bits(b), bits(c))
    finally:
        dev.close()


@pytest.mark.parametrize("name", SCENES)
def test_walk_statistics_stay_small(name, scenes):
    """The device tree's quality, counted, not timed: node tests and primitive tests per ray of the counting build.  A
    tree whose leaves overlap everything still gives the right pictures, only ten times slower (sphere leaves grown for
    rays from the far end of the huge ground sphere of cover1 once did that: 309 node tests and 122 primitive tests per
    ray).  Measured: 9 .. 24 node tests and 1.3 .. 1.8 primitive tests per ray."""
    world, kw, dev, ref = scenes(name)
    W, H = kw["width"] // 4, kw["height"] // 4
    _, st = dev.render(kw["camera"], kw["pattern"][:, :4], kw["background"], W, H, flags=ffi.RT_FLAG_COUNTERS)
    assert st.rays >= st.paths == W * H * 4
    assert st.node_visits / st.rays < 40 and st.prim_tests / st.rays < 4, (st.node_visits / st.rays, st.prim_tests / st.rays)


def test_errors_on_device(scenes):
    world, kw, dev, ref = scenes("cornell_box")
    L = ffi.lib()
    out = np.zeros(12, F)
    pat = np.zeros(2, F)
    bg = np.zeros(3, F)
    cam = rt.render.__globals__["camera_record"](kw["camera"])
    assert L.rt_render(dev.handle, C.byref(cam), ffi.fptr(pat), 1, ffi.fptr(bg), 1, 2, 50, 0, 0, ffi.fptr(out), None) == -1
    assert b"at least 2" in L.rt_last_error()
    assert L.rt_render(dev.handle, C.byref(cam), ffi.fptr(pat), 0, ffi.fptr(bg), 2, 2, 50, 0, 0, ffi.fptr(out), None) == -1
    assert L.rt_render(dev.handle, C.byref(cam), ffi.fptr(pat), 1, ffi.fptr(bg), 2, 2, 99, 0, 0, ffi.fptr(out), None) == -1
    assert dev.trace_rays(np.zeros((0, 7), F))["t"].shape == (0,)


def test_sample_sharding_sums_to_the_full_frame(scenes):
    """rt_render_accumulate over disjoint column shards + host-side sum + rt_finalize == one full render up to FP32
    summation order (what the multi-GPU all-reduce does)."""
    import torch
    world, kw, dev, ref = scenes("cu_cover2")
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(12)
    W, H = 64, 48
    full, _ = dev.render(cam, pat, bg, W, H)
    total = torch.zeros(3 * W * H, dtype=torch.float32, device="cuda")
    for g in (3, 16):                  # 16 shards of 12 columns: the last four are empty and must contribute zero
        total.zero_()
        for r in range(g):
            acc = torch.full((3 * W * H,), 7.0, dtype=torch.float32, device="cuda")
            dev.accumulate(cam, pat, bg, W, H, acc.data_ptr(), sample_first=r, sample_stride=g,
                           stream=torch.cuda.current_stream().cuda_stream)
            total += acc
        out = torch.zeros_like(total)
        dev.finalize(total.data_ptr(), 12, W, H, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        img = out.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2)
        np.testing.assert_allclose(img, full, rtol=2e-6, atol=1e-7)


def test_group_of_devices_in_one_process(scenes, monkeypatch):
    """rt_group_*: one host process, several members.  With one GPU to test on: a group of one device is rt_render bit for
    bit; the same with the sum forced through RCCL (a communicator of one: librccl is loaded, initialised and called);
    three members on the one GPU (three host threads, three shards, peer copies and adds in member order) give the sum of
    the three shards in that order bit for bit, i.e. the full frame up to the order of the FP32 sums."""
    import torch
    world, kw, dev, ref = scenes("cu_cover2")
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(12)
    W, H = 64, 48
    full, _ = dev.render(cam, pat, bg, W, H)
    one = rt.DeviceGroup(world, devices=[0])
    try:
        assert one.n_devices == 1
        img, st = one.render(cam, pat, bg, W, H)
        np.testing.assert_array_equal(bits(img), bits(full))
        assert st.paths == W * H * 12 and st.seconds > 0
        monkeypatch.setenv("RT_AMD_FORCE_RCCL", "1")
        img, _ = one.render(cam, pat, bg, W, H)
        monkeypatch.delenv("RT_AMD_FORCE_RCCL")
        np.testing.assert_array_equal(bits(img), bits(full))
    finally:
        one.close()
    three = rt.DeviceGroup(world, devices=[0, 0, 0])
    try:
        assert three.n_devices == 3
        img, _ = three.render(cam, pat, bg, W, H)
        # the members' host threads and the group's events outlive a frame: a larger frame (the accumulators grow),
        # then the first one again
        larger, _ = three.render(cam, pat, bg, 96, 80)
        again, _ = three.render(cam, pat, bg, W, H)
    finally:
        three.close()
    np.testing.assert_allclose(larger, dev.render(cam, pat, bg, 96, 80)[0], rtol=2e-6, atol=1e-7)
    np.testing.assert_array_equal(bits(img), bits(again))
    np.testing.assert_allclose(img, full, rtol=2e-6, atol=1e-7)
    stream = torch.cuda.current_stream().cuda_stream
    total = torch.zeros(3 * W * H, dtype=torch.float32, device="cuda")
    for r in range(3):
        acc = torch.zeros(3 * W * H, dtype=torch.float32, device="cuda")
        dev.accumulate(cam, pat, bg, W, H, acc.data_ptr(), sample_first=r, sample_stride=3, stream=stream)
        torch.cuda.synchronize()
        total = acc if r == 0 else total + acc
    out = torch.zeros_like(total)
    dev.finalize(total.data_ptr(), 12, W, H, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(bits(img), bits(out.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2)))
    # all visible devices: here one
    every = rt.DeviceGroup(world)
    try:
        assert every.n_devices == torch.cuda.device_count()
    finally:
        every.close()
    with pytest.raises(ffi.RtError):
        rt.DeviceGroup(world, devices=[torch.cuda.device_count()])


def test_group_render_without_rccl_is_rt_erccl_and_the_peer_sum_still_works():
    """rt_group_render on a host whose RCCL library cannot be loaded (RT_AMD_RCCL_LIB names a file that is not there):
    the frame that needs the collective fails with RT_ERCCL and the loader's message, nothing crashes, and the same
    group renders through peer copies (RT_AMD_REDUCE=peer).  In a child process: the library asks the loader once."""
    import subprocess
    code = """
import os, numpy as np
import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
world, kw = rt.get_scene("cornell_box", spp=4)
g = rt.DeviceGroup(world, devices=[0])
try:
    g.render(kw["camera"], kw["pattern"], kw["background"], 32, 32)
    print("no error")
except ffi.RtError as e:
    print("error:", e)
os.environ["RT_AMD_REDUCE"] = "peer"
os.environ.pop("RT_AMD_FORCE_RCCL")
img, _ = g.render(kw["camera"], kw["pattern"], kw["background"], 32, 32)
one = rt.DeviceScene(world)
want, _ = one.render(kw["camera"], kw["pattern"], kw["background"], 32, 32)
print("peer ok" if np.array_equal(img, want) else "peer differs")
g.close(); one.close()
"""
    env = dict(os.environ, RT_AMD_FORCE_RCCL="1", RT_AMD_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("error: librt_amd: status -6") and "/nonexistent/librccl.so.1 could not be loaded" in lines[0], lines
    assert "unknown error" not in lines[0] and lines[1] == "peer ok", lines


@pytest.mark.parametrize("name", ["cornell_smoke_box", "cu_cover2", "teapot"])
def test_degenerate_rays_terminate_and_match(name, scenes):
    """zero, axis-parallel, huge, infinite and NaN ray components: the stackless walk must end (node indices only grow)
    and the closest hit must still be the oracle's"""
    world, kw, dev, ref = scenes(name)
    cam = kw["camera"]
    base = O.primary_rays(cam, 8, 8, 8, 8)
    o = base[0, :3].copy()
    rays = []
    for d in ([0, 0, 0], [1, 0, 0], [0, -1, 0], [0, 0, 1], [1e30, 1e30, 1e30], [1e-30, -1e-30, 1e-30], [-0.0, 0.0, -1.0],
              [np.inf, 0, 0], [np.nan, 1, 0], [np.nan, np.nan, np.nan], [1, np.inf, -np.inf]):
        rays.append(np.concatenate([o, np.asarray(d, F), [0.5]]))
    for oo in ([np.nan, 0, 0], [np.inf, 0, 0], [1e30, -1e30, 0], [0, 0, 0]):
        rays.append(np.concatenate([np.asarray(oo, F), base[5, 3:6], [0.5]]))
    # nearly axis-parallel rays (huge reciprocals: the rounding slack of the box test is largest here) from the camera
    # and from points inside the scene, and rays that start on a coordinate plane (origin component exactly 0)
    rng = np.random.default_rng(11)
    centre = base[:, :3].mean(axis=0) + base[:, 3:6].mean(axis=0)
    for k in range(400):
        src = base[rng.integers(len(base))]
        oo = src[:3].copy() if k % 2 else (src[:3] + src[3:6] * F(rng.uniform(0.2, 1.5))).astype(F)
        d = (centre - oo + rng.normal(0, 1, 3) * np.abs(centre - oo).max()).astype(F) if k % 2 == 0 else src[3:6].copy()
        for axis in rng.choice(3, size=rng.integers(1, 3), replace=False):
            d[axis] *= F(10.0) ** F(-rng.integers(4, 40))
        if k % 7 == 0:
            oo[rng.integers(3)] = 0.0
        rays.append(np.concatenate([oo, d, [rng.uniform()]]))
    rays = np.asarray(rays, F)
    a, b = ref.trace_rays(rays), dev.trace_rays(rays)
    np.testing.assert_array_equal(a["prim_id"], b["prim_id"])
    np.testing.assert_array_equal(bits(a["t"]), bits(b["t"]))
    assert (a["prim_id"] >= 0).sum() > 40


def test_boxes_the_reference_culls():
    """The reference tests the box of a leaf before the primitive in it (src/bounding_box.jl:20-67, `t_max <= t_min`
    rejects).  A rectangle's box is 0.0002 thick: seen from 60000 away its two plane distances round to the same float
    and the reference never reaches the rectangle.  The Cornell box from that distance: the back wall (z = 555) is gone
    for every camera ray, the sphere behind it shows, and inside the box the paths go on as usual.  Same rays, same
    image as the oracle, although every primary hit on the back wall has to be walked again on the device."""
    from raytracing_amd.hittables import Camera, Emissive, Lambertian, Material, Sphere, Vec3
    world, kw = rt.get_scene("cornell_box", spp=1)
    bg = kw["background"]
    world = list(world) + [Sphere(Vec3(278, 278, 1500), 700, Material(Emissive, Vec3(0.5, 0.7, 0.9))),
                           Sphere(Vec3(278, 278, -20000), 60, Material(Lambertian, Vec3(0.3, 0.3, 0.8)))]
    origin = Vec3(278, 278, -60000)
    cam = Camera(origin, origin - Vec3(278, 278, 0), Vec3(0, 1, 0), 0.57, 1.0, 0.0, 10.0, 0.0, 1.0)
    ref, dev = O.OracleScene(world), rt.DeviceScene(world)
    try:
        rays = O.primary_rays(cam, 96, 96, 96, 96)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        np.testing.assert_array_equal(a["prim_id"], b["prim_id"])
        np.testing.assert_array_equal(bits(a["t"]), bits(b["t"]))
        assert (a["prim_id"] == 5).sum() == 0 and (a["prim_id"] == 8).sum() > 1000   # no back wall; the sphere behind it
        assert (a["prim_id"] == 3).sum() > 20                                        # the floor is still there
        pat = rt.Halton_sequence32(6)
        img, st = dev.render(cam, pat, bg, 96, 96)
        np.testing.assert_array_equal(bits(img), bits(ref.render(cam, pat, bg, 96, 96)))
        assert st.paths == 96 * 96 * 6 and img.max() > 0.5
    finally:
        dev.close()


@pytest.mark.parametrize("inline", ["1", "0"])
def test_boxes_the_reference_culls_around_a_nested_tree(inline, monkeypatch):
    """The same from far away for a Transform{BVHTree}: the reference tests the box of the whole tree with the ray as
    cast and the boxes inside with the ray in the tree's frame (src/transform.jl:104-116 around src/BVH.jl:22-53).  A tree
    of thin rectangles and small spheres, seen from 40000 away, both ways the device can hold such a tree; few slots, so
    that slots on hold are met by several launches."""
    import math
    from raytracing_amd import build_bvh
    from raytracing_amd.hittables import (Camera, Emissive, Lambertian, Material, RotY, Sphere, Transform, Vec3, affine,
                                          xy_rect, xz_rect)
    rng = np.random.default_rng(5)
    grey, light = Material(Lambertian, Vec3(0.6, 0.6, 0.6)), Material(Emissive, Vec3(2, 2, 2))
    members = []
    for k in range(40):
        c = rng.uniform(-200, 200, 3)
        members.append(Sphere(Vec3(*c), float(rng.uniform(5, 25)), grey) if k % 2 else
                       xy_rect(c[0] - 30, c[0] + 30, c[1] - 30, c[1] + 30, float(c[2]), light if k % 4 == 0 else grey))
    world = [Transform(build_bvh(members), affine(RotY(math.radians(20)), [10, 20, 30])),
             xz_rect(-400, 400, -400, 400, -250, grey), Sphere(Vec3(0, 0, 900), 500, light)]
    origin = Vec3(0, 100, -40000)
    cam = Camera(origin, origin - Vec3(0, 0, 0), Vec3(0, 1, 0), 1.2, 1.0, 0.0, 10.0, 0.0, 1.0)
    monkeypatch.setenv("RT_AMD_INLINE_GROUPS", inline)
    monkeypatch.setenv("RT_AMD_SLOTS", "16384")
    ref, dev = O.OracleScene(world), rt.DeviceScene(world)
    try:
        rays = O.primary_rays(cam, 80, 80, 80, 80)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        np.testing.assert_array_equal(a["prim_id"], b["prim_id"])
        np.testing.assert_array_equal(a["sub_id"], b["sub_id"])
        np.testing.assert_array_equal(bits(a["t"]), bits(b["t"]))
        assert (a["prim_id"] == 0).sum() > 100
        pat = rt.Halton_sequence32(8)
        img, st = dev.render(cam, pat, Vec3(0.1, 0.1, 0.1), 80, 80)
        np.testing.assert_array_equal(bits(img), bits(ref.render(cam, pat, Vec3(0.1, 0.1, 0.1), 80, 80)))
    finally:
        dev.close()


def test_nested_tree_kept_as_instance(scenes, monkeypatch):
    """By default the objects of a Transform{BVHTree} are taken into the enclosing device tree; with
    RT_AMD_INLINE_GROUPS=0 the tree stays a nested instance (entered and left by the extend kernel).  Same image."""
    world, kw, dev, ref = scenes("cover2")
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, H = 64, 48
    want = ref.render(cam, pat, bg, W, H)
    monkeypatch.setenv("RT_AMD_INLINE_GROUPS", "0")
    nested = rt.DeviceScene(world)
    try:
        img, _ = nested.render(cam, pat, bg, W, H)
        rays = O.primary_rays(cam, W, H, 64, 64)
        a, b = ref.trace_rays(rays), nested.trace_rays(rays)
    finally:
        nested.close()
    np.testing.assert_array_equal(bits(img), bits(want))
    np.testing.assert_array_equal(a["prim_id"], b["prim_id"])
    np.testing.assert_array_equal(a["sub_id"], b["sub_id"])
    np.testing.assert_array_equal(bits(a["t"]), bits(b["t"]))


def test_progressive_rendering_and_resume(scenes, tmp_path):
    """render_progressive: interleaved shards of the pattern, an image after each; stopped after two shards, stored with
    np.savez, resumed from the file: the final image is the one-shot image up to float32 summation order, and the same
    whether or not the run was interrupted."""
    world, kw, dev, ref = scenes("cornell_smoke_box")
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(20)
    W, H = 48, 40
    full, _ = dev.render(cam, pat, bg, W, H)
    steps = list(dev.render_progressive(cam, pat, bg, W, H, passes=5))
    assert [s["done"] for s, _ in steps] == [4, 8, 12, 16, 20]
    np.testing.assert_allclose(steps[-1][1], full, rtol=2e-6, atol=1e-7)
    # the first shard alone is the picture of pattern columns 0, 5, 10, 15
    sub, _ = dev.render(cam, pat[:, 0::5], bg, W, H)
    assert np.abs(steps[0][1] - sub).max() > 0            # other sample indices -> other random streams ...
    lin = lambda im: float((im.astype(np.float64) ** 2).mean())           # the gamma is a square root
    assert abs(lin(steps[0][1]) - lin(full)) < 0.25 * lin(full)            # ... but the same picture
    gen = dev.render_progressive(cam, pat, bg, W, H, passes=5)
    next(gen)
    state, _ = next(gen)
    gen.close()
    np.savez(tmp_path / "state.npz", **state)
    stored = dict(np.load(tmp_path / "state.npz"))
    rest = list(dev.render_progressive(cam, pat, bg, W, H, passes=5, state=stored))
    assert [s["done"] for s, _ in rest] == [12, 16, 20]
    np.testing.assert_array_equal(bits(rest[-1][1]), bits(steps[-1][1]))
    with pytest.raises(ValueError):
        next(dev.render_progressive(cam, pat, bg, W, H, passes=4, state=stored))


@pytest.mark.parametrize("name", ["cornell_smoke_box", "cover2"])
def test_scene_from_a_file_renders_the_same_picture(name, scenes, tmp_path):
    """save_scene + DeviceScene.from_file (rt_scene_desc_save / rt_scene_desc_load): the picture of the scene that came
    back from the file is, bit for bit, the picture of the scene it was written from"""
    world, kw, dev, ref = scenes(name)
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(6)
    W, H = 56, 40
    want, _ = dev.render(cam, pat, bg, W, H)
    path = tmp_path / "scene.rtscene"
    rt.save_scene(path, world)
    loaded = rt.DeviceScene.from_file(path)
    try:
        got, _ = loaded.render(cam, pat, bg, W, H)
    finally:
        loaded.close()
    np.testing.assert_array_equal(bits(got), bits(want))


def test_a_hit_in_front_of_its_box():
    """The two far spheres of the oracle's known-answer test: whether the smaller root wins depends on the type groups
    the reference sorts them into; the device gives the reference's answer in each case (hit_stands, surface_again)"""
    from test_oracle_kat import _two_far_spheres
    ray, worlds = _two_far_spheres()
    for name, world in worlds.items():
        want = O.OracleScene(world).trace_rays(ray)
        dev = rt.DeviceScene(world)
        try:
            got = dev.trace_rays(ray)
        finally:
            dev.close()
        assert got["prim_id"][0] == want["prim_id"][0], name
        np.testing.assert_array_equal(bits(got["t"]), bits(want["t"]))
        np.testing.assert_array_equal(bits(got["normal"]), bits(want["normal"]))


def test_device_output_stage_matches_host_quantisation(scenes):
    """rt_finalize_rgb8 (clamp01 + N0f8 on the device) == rt_finalize followed by the host quantisation"""
    import torch
    world, kw, dev, ref = scenes("cornell_box")                       # its light (15) is far outside [0, 1]
    cam, bg = kw["camera"], kw["background"]
    pat = rt.Halton_sequence32(6)
    W, H = 50, 38
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(3 * W * H, dtype=torch.float32, device="cuda")
    dev.accumulate(cam, pat, bg, W, H, acc.data_ptr(), stream=stream)
    out = torch.zeros_like(acc)
    rgb = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    dev.finalize(acc.data_ptr(), 6, W, H, out.data_ptr(), stream=stream)
    dev.finalize_rgb8(acc.data_ptr(), 6, W, H, rgb.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    img = out.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2)
    got = rgb.cpu().numpy().reshape(H, W, 3)
    np.testing.assert_array_equal(got, rt.to_rgb8(img))
    assert got.max() == 255 and img.max() > 1.0


def test_full_size_cu_cover2_against_oracle_and_properties(scenes):
    """The headline configuration at full resolution (800x800): a few pattern columns against the oracle bit for bit, and
    size-independent properties at more columns: run-to-run determinism and column-additivity."""
    world, kw, dev, ref = scenes("cu_cover2")
    cam, bg = kw["camera"], kw["background"]
    W, H = kw["width"], kw["height"]
    pat = rt.Halton_sequence32(2)
    img, _ = dev.render(cam, pat, bg, W, H)
    np.testing.assert_array_equal(bits(img), bits(ref.render(cam, pat, bg, W, H)))
    pat = rt.Halton_sequence32(64)
    a, st = dev.render(cam, pat, bg, W, H)
    b, _ = dev.render(cam, pat, bg, W, H)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all() and st.paths == W * H * 64
    # gross appearance of the reference's own cu_cover2 picture (its scene is unseeded, so only the level is comparable):
    # linear-space mean RGB (0.106, 0.164, 0.143) in outputs/cu_cover2_1000_samples.png
    lin = np.clip(a, 0, 1) ** 2
    assert np.all(np.abs(lin.mean(axis=(0, 1)) - np.array([0.1063, 0.1642, 0.1427])) < 0.03)


@pytest.mark.parametrize("name,key,spp,tol", [("cornell_box", "cornell_box_10000", 2000, 0.012),
                                              ("cornell_box_of_meshes", "cornell_box_10000", 2000, 0.012),
                                              ("cornell_smoke_box", "cornell_smoke_box_10000", 1000, 0.012)])
def test_converged_image_against_reference_gallery(name, key, spp, tol, scene_cache):
    """The two deterministic scenes at full resolution against statistics of the reference's own rendered PNGs
    (tests/golden/gallery_stats.json): per-channel RMSE of block means in gamma space, mean level, black frame.
    Stated RMSE bound: 0.012 on 16x16 (24x24) block means, about twice the 1000-vs-10000 spp noise floor (0.0053)
    measured between the reference's own two Cornell images."""
    stats = json.load(open(os.path.join(HERE, "golden", "gallery_stats.json")))[key]
    if name == "cornell_box_of_meshes":
        # the Cornell box built from triangles: the same picture, so the reference's own picture pins the triangle test
        # (src/triangle.jl:73-121) and the mesh tree (src/mesh.jl:115-163), which no picture of the gallery shows as such
        world, _ = cornell_box_of_meshes()
        _, kw = rt.get_scene("cornell_box", spp=spp)
    else:
        world, kw = rt.get_scene(name, spp=spp)
    dev = rt.DeviceScene(world)
    try:
        img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
    finally:
        dev.close()
    assert img.shape == (stats["height"], stats["width"], 3)
    g = np.clip(img, 0, 1).astype(np.float64)       # the PNG is clamp01 of the same sqrt-gamma values
    g = np.round(g * 255) / 255
    blk = stats["block"]
    gh, gw = g.shape[0] // blk, g.shape[1] // blk
    mine = g[:gh * blk, :gw * blk].reshape(gh, blk, gw, blk, 3).mean(axis=(1, 3))
    want = np.array(stats["block_means_gamma"])
    keep = np.ones((gh, gw), bool)
    if name == "cornell_box_of_meshes":
        # the one face on which triangles and rectangles differ in the reference's own arithmetic is left out
        probe = rt.DeviceScene(world)
        try:
            keep = ~blocks_showing_the_tall_box_front(probe.trace_rays, kw, blk)
        finally:
            probe.close()
        assert 0.02 < (~keep).mean() < 0.15, (~keep).mean()      # (the face is a tenth of the picture)
    rmse = np.sqrt(((mine - want)[keep] ** 2).mean(axis=0))
    assert (rmse < tol).all(), rmse
    level = (mine ** 2)[keep].mean(axis=0)          # (mean of squared block means: the same statistic on both sides)
    level_want = (want ** 2)[keep].mean(axis=0)
    np.testing.assert_allclose(level, level_want, atol=0.004)
    if name != "cornell_box_of_meshes":
        np.testing.assert_allclose((g ** 2).mean(axis=(0, 1)), stats["linear_mean"], atol=0.004)
    print(name, "rmse", rmse.round(4), "linear mean", np.round(level, 4), "reference", np.round(level_want, 4))
    row = g[g.shape[0] // 2].sum(axis=1)
    assert abs(int(np.argmax(row > 0.02)) - stats["black_frame_px"]) <= 1


GALLERY = {   # scene -> (key in gallery_blocks.npz, spp of that picture, spp of the three renders that decide the mask,
              #           spread limit, RMSE bound, bias bound, least mask share)
    "cover1": ("cover1_500", 500, 500, 0.01, 0.004, 0.002, 0.25),
    "cu_cover2": ("cu_cover2_1000", 1000, 1000, 0.02, 0.012, 0.004, 0.50),
    "cover2": ("cover2_100", 100, 100, 0.03, 0.030, 0.004, 0.40),
}


def _erode(m):
    e = m.copy()
    e[1:, :] &= m[:-1, :]; e[:-1, :] &= m[1:, :]; e[:, 1:] &= m[:, :-1]; e[:, :-1] &= m[:, 1:]
    return e


@pytest.mark.parametrize("name", list(GALLERY))
def test_deterministic_regions_against_reference_gallery(name):
    """The reference's pictures of the scenes it builds with the unseeded rand() (outputs/cover1_500_samples.png,
    cu_cover2_1000_samples.png, cover2_100_samples.png; 8x8 block means in tests/golden/gallery_blocks.npz) pin whatever in
    them does not depend on the random part of the scene: the sky, the three large spheres of cover1 (Dielectric, Lambertian,
    Metal with fuzz 0); the light, the fog, the moving sphere, the glass, metal and blue-medium spheres of cu_cover2; the
    earth sphere of cover2 (ImageTexture + get_sphere_uv).  Which blocks those are is decided by rendering the scene with
    three scene seeds: a block takes part if the three renders agree within the spread limit there and in its four
    neighbours.  On those blocks the mean of the three renders, made at the picture's own sample count (equal counts: the
    sqrt of a noisy mean is biased and bright pixels are clamped, equally on both sides), must equal the reference's picture:
    per-channel RMSE and bias below the stated bounds (gamma space), the worst 1 % of blocks left out (blocks where the
    reference's own random spheres show and none of ours do).  Stated bounds: cover1 0.004 (measured 0.001), cu_cover2
    0.012 (0.008: the noise of a 1000-spp picture), cover2 at 100 spp 0.03 (0.02).
    Two things in the reference's gallery do not fit the rest of it and are left out.  outputs/cover2_10000_samples.png is
    brighter than outputs/cover2_100_samples.png by 14 % (fog) to 42 % (moving sphere) in linear radiance, which no sample
    count explains (means of linear radiance do not depend on it; this library gives the same means at 100, 1000 and 10 000
    spp, and they are those of the 100-sample picture and of the CUDA picture cu_cover2_1000): it must come from another
    revision of the scene or the code.  And the fuzzy metal sphere of cover2_100 is evenly grey where the reference's two
    other pictures of that sphere (and this library) show a highlight and a dark side, so that one object is not compared
    in cover2 (it is in cu_cover2)."""
    from raytracing_amd.hittables import ConstantMedium
    key, spp, mask_spp, spread_limit, tol, bias_tol, least = GALLERY[name]
    G = np.load(os.path.join(HERE, "golden", "gallery_blocks.npz"))
    want, B = G[key].astype(np.float64), int(G["block"])

    def blocks_of(seed, n):
        world, kw = rt.get_scene(name, spp=n, seed=seed)
        dev = rt.DeviceScene(world)
        try:
            img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
        finally:
            dev.close()
        g = np.round(np.clip(np.nan_to_num(img), 0, 1).astype(np.float64) * 255) / 255
        h, w, _ = g.shape
        return g.reshape(h // B, B, w // B, B, 3).mean(axis=(1, 3)), world, kw

    renders = []
    for seed in (1, 2, 3):
        blocks, world, kw = blocks_of(seed, mask_spp)
        renders.append(blocks)
        if seed == 1:
            # which object a block shows (primary ray through its centre, constant media left out: their hits are random)
            solid = [o for o in world if not isinstance(o, ConstantMedium)]
            probe = rt.DeviceScene(solid)
            try:
                h, w = kw["height"], kw["width"]
                hit = probe.trace_rays(O.primary_rays(kw["camera"], w, h, w // B, h // B, time=0.5))["prim_id"].reshape(h // B, w // B)[::-1]
            finally:                              # (rows of rays count from the bottom)
                probe.close()
    assert renders[0].shape == want.shape
    spread = np.max([np.abs(renders[i] - renders[j]).max(axis=2) for i in range(3) for j in range(i)], axis=0)
    mask = _erode(spread < spread_limit)
    assert mask.mean() > least, mask.mean()
    mean = sum(renders) / 3 if spp == mask_spp else blocks_of(1, spp)[0]   # (every entry of GALLERY has equal counts today)
    d = (mean - want)[mask]
    worst = np.abs(d).max(axis=1)
    d = d[worst <= np.quantile(worst, 0.99)]
    rmse, bias = np.sqrt((d ** 2).mean(axis=0)), d.mean(axis=0)
    # the named objects one by one: enough of each must be among the blocks compared, and agree
    if name == "cover1":
        named = {"sky": -1, "glass sphere": len(solid) - 3, "lambertian sphere": len(solid) - 2, "metal sphere": len(solid) - 1}
    else:
        named = {"light": 0, "moving sphere": 1, "glass sphere": 2, "metal sphere": 3, "sphere of the blue medium": 4, "fog": -1}
        named["earth sphere" if name == "cover2" else "placeholder sphere"] = 5
        if name == "cover2":
            del named["metal sphere"]
    report = {}
    for label, pid in named.items():
        sel = mask & (hit == pid)
        dd = (mean - want)[sel]
        if len(dd) > 20:
            w_ = np.abs(dd).max(axis=1)
            dd = dd[w_ <= np.quantile(w_, 0.98)]
        report[label] = (int(sel.sum()), float(np.sqrt((dd ** 2).mean())) if len(dd) else float("nan"))
        if os.environ.get("RT_AMD_GALLERY_VERBOSE") and len(dd):
            print("   ", label, "blocks", int(sel.sum()), "bias", dd.mean(axis=0).round(4), "level", want[sel].mean(axis=0).round(3))
    print(name, "rmse", rmse.round(4), "bias", bias.round(4), "mask", round(float(mask.mean()), 3), report)
    assert (rmse < tol).all() and (np.abs(bias) < bias_tol).all(), (rmse, bias)
    for label, (count, r) in report.items():
        assert count >= 12, (label, count)
        assert r < 2.0 * tol, (label, r)
