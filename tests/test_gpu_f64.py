"""The Float64 mode (render(Float64, ...), main.jl:21,46) against the oracle built for Float64 (oracle/rt_ref.cpp with
-DREF_F64): closest hits of fixed rays, per-path radiance and whole images must be identical in every bit of every
double -- the device walks the reference's own trees with the reference's own box test, so nothing is tolerated.
Through the C ABI (rt_scene_create_f64, rt_render_f64, rt_render_accumulate_f64, rt_finalize_f64, rt_trace_rays_f64,
rt_trace_paths_f64)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd import hittables as H

pytestmark = pytest.mark.gpu
F = np.float64
SCENES = ["cornell_box", "cornell_smoke_box", "cover1", "cu_cover2", "cover2", "teapot"]


def bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint64)


def same_hits(a, b):
    same = (a["prim_id"] == b["prim_id"]) & (a["sub_id"] == b["sub_id"]) & (bits(a["t"]) == bits(b["t"]))
    assert same.all(), np.flatnonzero(~same)[:10]
    for field in ("normal", "uv"):
        np.testing.assert_array_equal(bits(a[field]), bits(b[field]))
    np.testing.assert_array_equal(a["front"], b["front"])


@pytest.mark.parametrize("key", SCENES)
def test_sample_scene_hits_paths_and_image(key):
    world, kw = rt.get_scene(key, spp=5, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    assert pat.dtype == F and cam.origin.dtype == F
    W, Hh = 56, 40
    ref = O.OracleScene(world, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        rays = O.primary_rays(cam, W, Hh, 64, 48, time=0.37, F=F)
        a = ref.trace_rays(rays)
        same_hits(a, dev.trace_rays(rays))
        assert (a["prim_id"] >= 0).any()
        # secondary rays from just in front of the first hits, in random directions
        g = np.random.default_rng(7)
        pts = rays[:, :3] + rays[:, 3:6] * np.where(np.isfinite(a["t"]), a["t"] * 0.95, 1.0)[:, None]
        sec = np.concatenate([pts, g.normal(size=(len(rays), 3)), g.uniform(0, 1, (len(rays), 1))], axis=1).astype(F)
        same_hits(ref.trace_rays(sec), dev.trace_rays(sec))
        ids = np.stack([g.integers(0, W, 600), g.integers(0, Hh, 600), g.integers(0, 5, 600)], axis=1).astype(np.int32)
        np.testing.assert_array_equal(bits(ref.trace_paths(cam, pat, bg, W, Hh, ids)), bits(dev.trace_paths(cam, pat, bg, W, Hh, ids)))
        img, st = dev.render(cam, pat, bg, W, Hh)
        assert img.dtype == F and st.paths == W * Hh * 5
        want = ref.render(cam, pat, bg, W, Hh)
        assert (bits(img) != bits(want)).sum() == 0
        assert np.isfinite(img[np.isfinite(want)]).all()
    finally:
        dev.close()


@pytest.mark.parametrize("key,size,cols", [("cu_cover2", (200, 200), 6), ("cover2", (160, 160), 4), ("teapot", (320, 180), 4)])
def test_larger_frames(key, size, cols):
    """Frames large enough for several generations of paths per slot (path regeneration, the end of a batch) and, with
    the small pool, for slots that are handed a new path many times."""
    world, kw = rt.get_scene(key, spp=cols, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = size
    want = O.OracleScene(world, F=F).render(cam, pat, bg, W, Hh)
    dev = rt.DeviceScene(world, F=F)
    try:
        img, st = dev.render(cam, pat, bg, W, Hh)
        assert (bits(img) != bits(want)).sum() == 0 and st.paths == W * Hh * cols
        os.environ["RT_AMD_F64_SLOTS"] = "4096"
        try:
            img, _ = dev.render(cam, pat, bg, W, Hh)
        finally:
            del os.environ["RT_AMD_F64_SLOTS"]
        assert (bits(img) != bits(want)).sum() == 0
    finally:
        dev.close()


def test_float64_differs_from_float32_only_by_rounding_at_the_first_hit():
    """The two element types see the same scene: first hits agree in id, and in t to Float32 accuracy."""
    w32, kw32 = rt.get_scene("cornell_box", spp=1)
    w64, kw64 = rt.get_scene("cornell_box", spp=1, F=F)
    d32, d64 = rt.DeviceScene(w32), rt.DeviceScene(w64, F=F)
    try:
        r64 = O.primary_rays(kw64["camera"], 400, 400, 40, 40, F=F)
        a, b = d32.trace_rays(r64.astype(np.float32)), d64.trace_rays(r64)
        agree = a["prim_id"] == b["prim_id"]
        assert agree.mean() > 0.99
        hit = agree & (b["prim_id"] >= 0)
        np.testing.assert_allclose(a["t"][hit], b["t"][hit], rtol=2e-5)
    finally:
        d32.close()
        d64.close()


def test_batches_and_shards_sum_like_one_frame(monkeypatch):
    """Columns traced in several batches (radiance buffer smaller than the frame) give the same picture bit for bit;
    two interleaved shards (rt_render_accumulate_f64 twice, summed, rt_finalize_f64) give it up to the order of the
    sums."""
    import torch
    world, kw = rt.get_scene("cornell_smoke_box", spp=7, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 48, 36
    dev = rt.DeviceScene(world, F=F)
    try:
        whole, _ = dev.render(cam, pat, bg, W, Hh)
        monkeypatch.setenv("RT_AMD_F64_BATCH_COLUMNS", "3")
        again, _ = dev.render(cam, pat, bg, W, Hh)
        monkeypatch.delenv("RT_AMD_F64_BATCH_COLUMNS")
        assert (bits(whole) != bits(again)).sum() == 0
        parts = [torch.empty(3 * W * Hh, dtype=torch.float64, device="cuda") for _ in range(3)]
        stream = torch.cuda.current_stream().cuda_stream
        for c in range(3):  # the third shard is empty for c >= n: here n = 7, shards 0, 1, 2 of 3
            dev.accumulate(cam, pat, bg, W, Hh, parts[c].data_ptr(), sample_first=c, sample_stride=3, stream=stream)
        total = parts[0] + parts[1] + parts[2]
        out = torch.empty_like(total)
        dev.finalize(total.data_ptr(), 7, W, Hh, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        img = out.cpu().numpy().reshape(W, Hh, 3).transpose(1, 0, 2)
        np.testing.assert_allclose(img, whole, rtol=1e-13, atol=1e-300)
        empty = torch.full((3 * W * Hh,), 5.0, dtype=torch.float64, device="cuda")
        dev.accumulate(cam, pat, bg, W, Hh, empty.data_ptr(), sample_first=9, sample_stride=3, stream=stream)
        torch.cuda.synchronize()
        assert float(empty.abs().max()) == 0.0
    finally:
        dev.close()


def test_edge_cases():
    """smallest legal image, one pattern column, the MSAA table as pattern, depth limits 0, 1, 2, another seed"""
    world, kw = rt.get_scene("cornell_smoke_box", spp=5, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    ref = O.OracleScene(world, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        with H.element_type(F):
            msaa = H.MSAA_S16.astype(F)
        for W, Hh, p, depth in ((2, 2, pat[:, :1], 50), (9, 3, msaa, 50), (16, 16, pat, 1), (16, 16, pat, 2), (8, 8, pat, 0)):
            img, _ = dev.render(cam, p, bg, W, Hh, max_depth=depth)
            np.testing.assert_array_equal(bits(img), bits(ref.render(cam, p, bg, W, Hh, max_depth=depth)))
        a, _ = dev.render(cam, pat, bg, 24, 16, seed=7)
        b, _ = dev.render(cam, pat, bg, 24, 16, seed=7)
        c, _ = dev.render(cam, pat, bg, 24, 16)
        np.testing.assert_array_equal(bits(a), bits(b))
        assert not np.array_equal(a, c)
        np.testing.assert_array_equal(bits(a), bits(ref.render(cam, pat, bg, 24, 16, seed=7)))
    finally:
        dev.close()


@pytest.mark.parametrize("key", ["cornell_box", "cu_cover2", "cover2", "teapot"])
def test_degenerate_rays_terminate_and_match(key):
    """zero, axis-parallel, huge, infinite and NaN ray components (the box test by reciprocals hands those to the
    divisions), rays that start on coordinate planes, nearly axis-parallel rays: every walk ends and gives the
    oracle's hit"""
    world, kw = rt.get_scene(key, spp=1, F=F)
    cam = kw["camera"]
    ref = O.OracleScene(world, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        base = O.primary_rays(cam, 8, 8, 8, 8, F=F)
        o = base[0, :3].copy()
        rays = []
        for d in ([0, 0, 0], [1, 0, 0], [0, -1, 0], [0, 0, 1], [1e300, 1e300, 1e300], [1e-300, -1e-300, 1e-300], [-0.0, 0.0, -1.0],
                  [np.inf, 0, 0], [np.nan, 1, 0], [np.nan, np.nan, np.nan], [1, np.inf, -np.inf]):
            rays.append(np.concatenate([o, np.asarray(d, F), [0.5]]))
        for oo in ([np.nan, 0, 0], [np.inf, 0, 0], [1e300, -1e300, 0], [0, 0, 0]):
            rays.append(np.concatenate([np.asarray(oo, F), base[5, 3:6], [0.5]]))
        rng = np.random.default_rng(11)
        centre = base[:, :3].mean(axis=0) + base[:, 3:6].mean(axis=0)
        for k in range(400):
            src = base[rng.integers(len(base))]
            oo = src[:3].copy() if k % 2 else src[:3] + src[3:6] * rng.uniform(0.2, 1.5)
            d = centre - oo + rng.normal(0, 1, 3) * np.abs(centre - oo).max() if k % 2 == 0 else src[3:6].copy()
            for axis in rng.choice(3, size=rng.integers(1, 3), replace=False):
                d[axis] *= 10.0 ** -float(rng.integers(4, 300))
            if k % 7 == 0:
                oo[rng.integers(3)] = 0.0
            rays.append(np.concatenate([oo, d, [rng.uniform()]]))
        rays = np.array(rays, dtype=F)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        same = (a["prim_id"] == b["prim_id"]) & (a["sub_id"] == b["sub_id"]) & (bits(a["t"]) == bits(b["t"]))
        assert same.all(), np.flatnonzero(~same)[:10]
        hit = a["prim_id"] >= 0
        np.testing.assert_array_equal(bits(a["normal"])[hit], bits(b["normal"])[hit])
    finally:
        dev.close()


def test_planes_outside_the_float_range_and_rays_that_graze():
    """The walk reads the nodes' boxes narrowed to float and takes the double box wherever the narrowing could decide
    the reference's test: spheres whose boxes lie beyond the float range, closer to zero than the smallest float, and at
    coordinates where a float resolves less than their size; rays aimed at the boxes' faces, edges and corners"""
    mat = rt.Material(rt.Lambertian, rt.Vec3(0.5, 0.5, 0.5))
    rng = np.random.default_rng(5)
    with H.element_type(F):
        world = [rt.Sphere(rt.Vec3(1e39, 0, 0), 1e38, mat), rt.Sphere(rt.Vec3(-3e38, 1e39, 0), 2e38, mat),
                 rt.Sphere(rt.Vec3(1e-40, 2e-40, -1e-40), 5e-41, mat), rt.Sphere(rt.Vec3(-3e-39, 0, 1e-39), 1e-39, mat)]
        for k in range(60):   # float resolves 2^-24 * 1e6 = 0.06 here: boxes 0.02 to 2 across
            c = rng.uniform(-1, 1, 3) * 50 + np.array([1e6, -1e6, 1e6])
            world.append(rt.Sphere(rt.Vec3(*c), float(rng.uniform(0.01, 1.0)), mat))
        for k in range(40):
            c = rng.uniform(-40, 40, 3)
            world.append(rt.Box(rt.Vec3(*c), rt.Vec3(*(c + rng.uniform(0.5, 6, 3))), mat))
    ref = O.OracleScene(world, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        rays = []
        targets = [np.array([1e39, 0, 0]), np.array([-3e38, 1e39, 0]), np.array([1e-40, 2e-40, -1e-40]), np.array([-3e-39, 0, 1e-39])]
        for tgt in targets:
            scale = max(np.abs(tgt).max(), 1e-300)
            for _ in range(40):
                o = tgt + rng.normal(0, 3, 3) * scale
                rays.append(np.concatenate([o, (tgt + rng.normal(0, 0.12, 3) * scale) - o, [0.5]]))
        for obj in world[4:]:
            box6 = np.asarray(H.bounding_box(obj).as6(), dtype=F)
            lo, hi = box6[:3], box6[3:]
            for _ in range(12):
                # a point on a face, an edge or a corner of the object's box, moved by nothing or by an ulp or two
                p = np.where(rng.random(3) < 0.5, lo, hi).astype(F)
                free = rng.random(3) < 0.4
                p = np.where(free, lo + (hi - lo) * rng.random(3), p)
                for _ in range(int(rng.integers(0, 3))):
                    p = np.nextafter(p, p + rng.choice([-1.0, 1.0], 3))
                o = p + rng.normal(0, 1, 3) * (np.abs(hi - lo).max() * rng.choice([2.0, 50.0, 1e4]))
                rays.append(np.concatenate([o, p - o, [0.5]]))
        rays = np.array(rays, dtype=F)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        same = (a["prim_id"] == b["prim_id"]) & (a["sub_id"] == b["sub_id"]) & (bits(a["t"]) == bits(b["t"]))
        assert same.all(), np.flatnonzero(~same)[:10]
        assert (a["prim_id"] >= 0).sum() > len(rays) // 10
    finally:
        dev.close()


def test_progressive_rendering_and_resume_in_float64(tmp_path):
    """render_progressive on a Float64 scene: the accumulator, the state that np.savez stores and the pictures are
    double; the last picture is the one-shot picture up to the order of the sums, interrupted or not."""
    world, kw = rt.get_scene("cornell_box", spp=12, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 40, 32
    dev = rt.DeviceScene(world, F=F)
    try:
        full, _ = dev.render(cam, pat, bg, W, Hh)
        steps = list(dev.render_progressive(cam, pat, bg, W, Hh, passes=3))
        assert [s["done"] for s, _ in steps] == [4, 8, 12] and steps[-1][1].dtype == F and steps[-1][0]["accum"].dtype == F
        np.testing.assert_allclose(steps[-1][1], full, rtol=1e-13, atol=1e-300)
        gen = dev.render_progressive(cam, pat, bg, W, Hh, passes=3)
        state, _ = next(gen)
        gen.close()
        np.savez(tmp_path / "state.npz", **state)
        rest = list(dev.render_progressive(cam, pat, bg, W, Hh, passes=3, state=dict(np.load(tmp_path / "state.npz"))))
        assert [s["done"] for s, _ in rest] == [8, 12]
        np.testing.assert_array_equal(bits(rest[-1][1]), bits(steps[-1][1]))
    finally:
        dev.close()


def test_compacted_pool_gives_the_same_picture(monkeypatch):
    """Once a batch has handed out its last path the pools are compacted as they empty (rt64_compact_kernel); a path's
    arithmetic does not depend on its slot, so the picture is that of the pools left as they are -- and the oracle's"""
    world, kw = rt.get_scene("cornell_smoke_box", spp=12, F=F)   # long paths inside the media: a long tail of few paths
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 96, 64
    dev = rt.DeviceScene(world, F=F)
    try:
        got, _ = dev.render(cam, pat, bg, W, Hh)
        monkeypatch.setenv("RT_AMD_F64_NO_COMPACT", "1")
        plain, _ = dev.render(cam, pat, bg, W, Hh)
        monkeypatch.delenv("RT_AMD_F64_NO_COMPACT")
        monkeypatch.setenv("RT_AMD_F64_COMPACT_AT", "16")          # at every look
        monkeypatch.setenv("RT_AMD_F64_SLOTS", "8192")             # and with paths still to hand out for most of the frame
        eager, _ = dev.render(cam, pat, bg, W, Hh)
    finally:
        dev.close()
    np.testing.assert_array_equal(bits(got), bits(plain))
    np.testing.assert_array_equal(bits(eager), bits(plain))
    want = O.OracleScene(world, F=F).render(cam, pat, bg, W, Hh)
    np.testing.assert_array_equal(bits(got), bits(want))


def test_device_output_stage_matches_host_quantisation():
    """rt_finalize_rgb8_f64 (clamp01 + N0f8 in double on the device) == rt_finalize_f64 followed by the host's"""
    import torch
    world, kw = rt.get_scene("cornell_box", spp=6, F=F)                # its light (15) is far outside [0, 1]
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 50, 38
    dev = rt.DeviceScene(world, F=F)
    try:
        stream = torch.cuda.current_stream().cuda_stream
        acc = torch.zeros(3 * W * Hh, dtype=torch.float64, device="cuda")
        dev.accumulate(cam, pat, bg, W, Hh, acc.data_ptr(), stream=stream)
        out = torch.zeros_like(acc)
        rgb = torch.zeros(Hh * W * 3, dtype=torch.uint8, device="cuda")
        dev.finalize(acc.data_ptr(), 6, W, Hh, out.data_ptr(), stream=stream)
        dev.finalize_rgb8(acc.data_ptr(), 6, W, Hh, rgb.data_ptr(), stream=stream)
        torch.cuda.synchronize()
    finally:
        dev.close()
    img = out.cpu().numpy().reshape(W, Hh, 3).transpose(1, 0, 2)
    got = rgb.cpu().numpy().reshape(Hh, W, 3)
    assert img.dtype == F
    np.testing.assert_array_equal(got, rt.to_rgb8(img))
    assert got.max() == 255 and img.max() > 1.0


@pytest.mark.parametrize("name", ["cornell_smoke_box", "cover2"])
def test_scene_from_a_file_renders_the_same_picture(name, tmp_path):
    """save_scene(F=float64) + DeviceScene.from_file(F=float64): the picture of the scene that came back from the file
    is, bit for bit, the picture of the scene it was written from"""
    world, kw = rt.get_scene(name, spp=4, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 40, 28
    dev = rt.DeviceScene(world, F=F)
    try:
        want, _ = dev.render(cam, pat, bg, W, Hh)
    finally:
        dev.close()
    path = tmp_path / "scene.rtscene64"
    rt.save_scene(path, world, F=F)
    loaded = rt.DeviceScene.from_file(path, F=F)
    try:
        got, _ = loaded.render(cam, pat, bg, W, Hh)
    finally:
        loaded.close()
    assert got.dtype == F
    np.testing.assert_array_equal(bits(got), bits(want))
    with pytest.raises(ffi.RtError):
        rt.DeviceScene.from_file(path)


def test_group_of_float64_scenes(monkeypatch):
    """rt_group_*_f64: one member renders rt_render_f64's picture bit for bit; two members on the one GPU (peer copies
    and adds) and one member through the RCCL binding give it up to the order of the sums."""
    world, kw = rt.get_scene("cornell_smoke_box", spp=6, F=F)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, Hh = 40, 32
    dev = rt.DeviceScene(world, F=F)
    want, _ = dev.render(cam, pat, bg, W, Hh)
    dev.close()
    one = rt.DeviceGroup(world, devices=[0], F=F)
    try:
        assert one.n_devices == 1
        img, st = one.render(cam, pat, bg, W, Hh)
        assert img.dtype == F and st.paths == W * Hh * 6
        np.testing.assert_array_equal(bits(img), bits(want))
        monkeypatch.setenv("RT_AMD_FORCE_RCCL", "1")
        img, _ = one.render(cam, pat, bg, W, Hh)
        monkeypatch.delenv("RT_AMD_FORCE_RCCL")
        np.testing.assert_array_equal(bits(img), bits(want))
    finally:
        one.close()
    two = rt.DeviceGroup(world, devices=[0, 0], F=F)
    try:
        assert two.n_devices == 2
        img, _ = two.render(cam, pat, bg, W, Hh)
        np.testing.assert_allclose(img, want, rtol=1e-13, atol=1e-300)
        assert np.array_equal(rt.render("cornell_smoke_box", gpu=True, spp=6, width=W, height=Hh, devices=[0, 0], F=F), img)
    finally:
        two.close()
    # more members than pattern columns: the member without a column adds a zero sum
    three = rt.DeviceGroup(world, devices=[0, 0, 0], F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        img, st = three.render(cam, pat[:, :2], bg, W, Hh)
        want2, _ = dev.render(cam, pat[:, :2], bg, W, Hh)
        assert st.paths == W * Hh * 2
        np.testing.assert_allclose(img, want2, rtol=1e-13, atol=1e-300)
    finally:
        three.close()
        dev.close()


def test_bad_arguments_are_errors():
    from raytracing_amd import _ffi as ffi
    world, kw = rt.get_scene("cornell_box", spp=2, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        with pytest.raises(ffi.RtError):
            dev.render(kw["camera"], kw["pattern"], kw["background"], 1, 1)
        with pytest.raises(ffi.RtError):
            dev.trace_paths(kw["camera"], kw["pattern"], kw["background"], 8, 8, [[9, 0, 0]])
        with pytest.raises(ffi.RtError):
            dev.layout()
    finally:
        dev.close()
    with pytest.raises(ffi.RtError):  # Transform{Transform}: the nestings the Float32 path turns down
        inner = rt.Transform(rt.Sphere(rt.Vec3(0), 1.0, rt.Material(rt.Lambertian, rt.Vec3(0.5))), np.eye(4))
        rt.DeviceScene([rt.Transform(inner, np.eye(4))], F=F)


# RT_AMD_FUZZ_SCENES_F64=N changes the size of the sweep
@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("RT_AMD_FUZZ_SCENES_F64", "400"))))
def test_random_scene_f64(seed):
    from test_gpu_fuzz import random_scene
    with H.element_type(F):
        world, cam, bg = random_scene(seed)
        pat = rt.Halton_sequence32(4)
    W, Hh = 32, 24
    ref = O.OracleScene(world, F=F)
    dev = rt.DeviceScene(world, F=F)
    try:
        rays = O.primary_rays(cam, W, Hh, 40, 32, time=0.37, F=F)
        same_hits(ref.trace_rays(rays), dev.trace_rays(rays))
        img, _ = dev.render(cam, pat, bg, W, Hh)
        assert (bits(img) != bits(ref.render(cam, pat, bg, W, Hh))).sum() == 0
    finally:
        dev.close()
