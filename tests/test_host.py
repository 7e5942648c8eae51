"""Host-side logic (no GPU): the C ABI surface, SceneManager grouping, the restated BVH builders, scene definitions,
the OBJ loader, and argument validation of the library."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd.scene_manager import flatten_scene
from conftest import SCENES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(rt_\w+)\s*\(", header, flags=re.M))
    assert declared == set(ffi.SYMBOLS), declared ^ set(ffi.SYMBOLS)
    lib = C.CDLL(ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert ffi.lib().rt_abi_version() == 1


def test_scenes_list_and_shapes():
    assert set(rt.scenes_list()) == {"cornell_box", "cornell_smoke_box", "cover1", "cover2", "cu_cover2"}
    sizes = {"cornell_box": (400, 400, 8), "cornell_smoke_box": (600, 600, 8), "cover1": (1200, 800, None),
             "cu_cover2": (800, 800, 1409), "cover2": (800, 800, 410)}
    for name, (w, h, n) in sizes.items():
        world, kw = rt.get_scene(name, spp=4)
        assert (kw["width"], kw["height"]) == (w, h)
        assert kw["pattern"].shape == (2, 4)
        if n is not None:
            assert len(world) == n
    # seeded scenes are reproducible
    a, _ = rt.get_scene("cover1", spp=1)
    b, _ = rt.get_scene("cover1", spp=1)
    assert len(a) == len(b) and all(np.array_equal(x.center, y.center) for x, y in zip(a, b))


def test_scene_manager_groups_by_concrete_type():
    world, _ = rt.get_scene("cu_cover2", spp=1)
    sm = rt.SceneManager(world)
    # samples/cu_cover_the_next_week.jl:81-86: xz_rect, MovingSphere, Sphere x5, ConstantMedium x2, Box x400, Transform{Sphere} x1000
    assert [len(g) for g in sm] == [1, 1, 5, 2, 400, 1000]
    world, _ = rt.get_scene("cover2", spp=1)
    keys = [rt.type_key(g[0]) for g in rt.SceneManager(world)]
    # spheres with image / noise textures are different concrete types, and the nested tree is its own type
    assert keys.count("Sphere<SolidColor>") == 1 and "Sphere<ImageTexture>" in keys and "Sphere<NoiseTexture<Vec3>>" in keys
    assert keys[-1] == "Transform<BVHTree<Sphere<SolidColor>>>"


@pytest.mark.parametrize("name", SCENES)
def test_library_bvh_builder_equals_oracle_builder(name):
    """rt_bvh_build / rt_bvh_build_meshes against the oracle's restatement of src/BVH.jl:65-109, src/mesh.jl:217-305."""
    world, _ = rt.get_scene(name, spp=1)
    ref = O.OracleScene(world).buckets()
    bvh = rt.build_bvh_scene(rt.SceneManager(world))
    pos = {id(o): i for i, o in enumerate(world)}
    assert len(ref) == len(bvh)
    for o, t in zip(ref, bvh):
        np.testing.assert_array_equal([pos[id(x)] for x in t.objects], o["objects"])
        v = o["valid"].astype(bool) if o["is_mesh"] else np.ones(len(o["leaf"]), bool)
        if o["is_mesh"]:
            np.testing.assert_array_equal(v, t.node_leaf != 255)
        for a, b in ((o["bounds"], t.node_bounds), (o["left"], t.node_left), (o["right"], t.node_right), (o["leaf"], t.node_leaf)):
            np.testing.assert_array_equal(a[v], b[v])


def test_bvh_builder_structure_random():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 64, 257):
        lo = rng.uniform(-10, 10, (n, 3)).astype(F)
        bounds = np.concatenate([lo, lo + rng.uniform(0.1, 2, (n, 3)).astype(F)], axis=1)
        order = np.zeros(n, np.int32)
        nb = np.zeros((2 * n - 1, 6), F)
        nl = np.zeros(2 * n - 1, np.int32)
        nr = np.zeros(2 * n - 1, np.int32)
        lf = np.zeros(2 * n - 1, np.uint8)
        ffi.check(ffi.lib().rt_bvh_build(n, ffi.fptr(bounds), ffi.iptr(order), ffi.fptr(nb), ffi.iptr(nl), ffi.iptr(nr), ffi.bptr(lf)))
        assert sorted(order.tolist()) == list(range(n))
        seen = []

        def walk(i, depth=0):   # 1-based ids, pre-order layout: left child = id + 1
            if lf[i - 1]:
                assert nl[i - 1] == nr[i - 1]
                seen.append(nl[i - 1])
                np.testing.assert_array_equal(nb[i - 1], bounds[order[nl[i - 1] - 1]])
                return
            assert nl[i - 1] == i + 1
            for c in (nl[i - 1], nr[i - 1]):
                assert (nb[c - 1][:3] >= nb[i - 1][:3]).all() and (nb[c - 1][3:] <= nb[i - 1][3:]).all()
                walk(c, depth + 1)
        walk(1)
        assert seen == list(range(1, n + 1))


def test_obj_loader_teapot():
    from raytracing_amd.obj_loader import load_obj
    v, idx = load_obj(os.path.join(ROOT, "raytracing_amd", "assets", "teapot.obj"))
    assert idx.shape == (6320, 4)          # 2856 quads + 608 triangles
    assert v.shape[1] == 8 and idx[:, :3].max() == len(v) - 1
    np.testing.assert_allclose(v[:, :3].min(axis=0), [-2, 0, -3.434], atol=1e-3)
    np.testing.assert_allclose(v[:, :3].max(axis=0), [2, 3.15, 3], atol=1e-3)
    assert (np.abs(v[:, 3:6]).sum(axis=1) > 0).all()   # every vertex carries a normal


def test_transform_bbox_of_eight_corners():
    b = rt.Box(rt.Vec3(0), rt.Vec3(165, 330, 165), rt.Material(rt.Lambertian, rt.Vec3(0.7)))
    t = rt.Transform(b, rt.affine(rt.RotY(np.radians(15)), [265, 0, 295]))
    c, s = np.cos(np.radians(15)), np.sin(np.radians(15))
    xs = [265 + c * x + s * z for x in (0, 165) for z in (0, 165)]
    zs = [295 - s * x + c * z for x in (0, 165) for z in (0, 165)]
    np.testing.assert_allclose(t.bbox.min, [min(xs), 0, min(zs)], rtol=1e-6)
    np.testing.assert_allclose(t.bbox.max, [max(xs), 330, max(zs)], rtol=1e-6)
    # xform is stored column-major 3x4 (src/transform.jl:65): element 9..11 is the translation
    np.testing.assert_allclose(t.xform[9:12], [265, 0, 295])
    m = np.eye(4)
    m[:3, :4] = t.xform.reshape(4, 3).T
    mi = np.eye(4)
    mi[:3, :4] = t.inv_xform.reshape(4, 3).T
    np.testing.assert_allclose(m @ mi, np.eye(4), atol=1e-4)


def test_render_refuses_cpu_path_and_missing_library(monkeypatch):
    with pytest.raises(NotImplementedError):
        rt.render("cornell_box", spp=1, gpu=False)
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", "/nonexistent/librt_amd.so")
    with pytest.raises(ffi.RtError):
        ffi.lib()


def test_flattened_description_and_validation():
    """rt_scene_create validates before it touches the GPU: malformed descriptions give RT_EINVAL / RT_EUNSUPPORTED and a
    message; a valid one passes validation and then fails with RT_ENODEV in this GPU-less container."""
    L = ffi.lib()
    world, _ = rt.get_scene("cornell_smoke_box", spp=1)
    bvh = rt.build_bvh_scene(rt.SceneManager(world))
    desc, keep = flatten_scene(world, bvh)
    assert desc.n_top == 8 and desc.n_buckets == 4 and desc.n_transforms == 2
    handle = C.c_void_p()
    n = C.c_int(0)
    assert L.rt_device_count(C.byref(n)) == 0
    st = L.rt_scene_create(C.byref(desc), 0, C.byref(handle))
    if n.value == 0:
        assert st == -2 and b"device" in L.rt_last_error().lower()      # RT_ENODEV
    else:
        assert st == 0
        L.rt_scene_destroy(handle)
    # corrupt a material index
    keep.objects[0].material = 999
    desc2, keep2 = flatten_scene(world, bvh)
    arr = (ffi.rtObject * desc2.n_objects)(*[desc2.objects[i] for i in range(desc2.n_objects)])
    arr[0].material = 999
    desc2.objects = arr
    assert L.rt_scene_create(C.byref(desc2), 0, C.byref(handle)) == -1 and b"material" in L.rt_last_error()
    # a cyclic tree
    desc3, keep3 = flatten_scene(world, bvh)
    b0 = desc3.buckets[1]
    left = np.array([b0.node_left[i] for i in range(b0.n_nodes)], np.int32)
    left[0] = 1
    leaf = np.zeros(b0.n_nodes, np.uint8)
    b0.node_left, b0.node_leaf = ffi.iptr(left), ffi.bptr(leaf)
    bl = (ffi.rtBvh * desc3.n_buckets)(*[desc3.buckets[i] for i in range(desc3.n_buckets)])
    bl[1] = b0
    desc3.buckets = bl
    assert L.rt_scene_create(C.byref(desc3), 0, C.byref(handle)) == -1
    assert L.rt_scene_create(None, 0, C.byref(handle)) == -1
    # unsupported nesting is reported, not mis-rendered
    s = rt.Sphere(rt.Vec3(0), 1, rt.Material(rt.Lambertian, rt.Vec3(0.5)))
    nested = [rt.Transform(rt.Transform(s, rt.affine(rt.RotY(0), [0, 0, 1])), rt.affine(rt.RotY(0), [0, 0, 1]))]
    d4, k4 = flatten_scene(nested, rt.build_bvh_scene(rt.SceneManager(nested)))
    assert L.rt_scene_create(C.byref(d4), 0, C.byref(handle)) == -5     # RT_EUNSUPPORTED
    # argument checks of the other entry points
    assert L.rt_bvh_build(0, None, None, None, None, None, None) == -1
    assert L.rt_render(None, None, None, 0, None, 0, 0, 0, 0, 0, None, None) == -1
    assert L.rt_trace_rays(None, None, 0, 0.0, 1.0, 0, None, None, None, None, None, None) == -1
    info = ffi.rtLayout()
    assert L.rt_scene_nodes(None, C.byref(info), None, 0, None, 0, None) == -1
    assert b"rt_scene_nodes" in L.rt_last_error()


def test_group_entry_points_without_a_device():
    """rt_group_*: argument checks, and without a GPU a clear RT_ENODEV instead of a crash"""
    L = ffi.lib()
    handle = C.c_void_p()
    assert L.rt_group_create(None, None, 0, C.byref(handle)) == -1
    n = C.c_int(0)
    assert L.rt_group_size(None, C.byref(n)) == -1
    assert L.rt_group_destroy(None) == 0
    world, _ = rt.get_scene("cornell_box", spp=1)
    desc, keep = flatten_scene(world, rt.build_bvh_scene(rt.SceneManager(world)))
    count = C.c_int(0)
    assert L.rt_device_count(C.byref(count)) == 0
    bad = np.array([0, 1], np.int32)
    assert L.rt_group_create(C.byref(desc), ffi.iptr(bad), -1, C.byref(handle)) == -1
    if count.value == 0:
        assert L.rt_group_create(C.byref(desc), None, 0, C.byref(handle)) == -2       # RT_ENODEV
        assert b"no HIP device" in L.rt_last_error() and not handle
    del keep


def _child(code, **env):
    """runs `code` in a fresh interpreter (the library caches what it learns about RCCL for the life of a process)"""
    import subprocess
    import sys
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-c", code], env=e, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                          capture_output=True, text=True, timeout=300)


def test_a_missing_rccl_library_is_rt_erccl_with_the_loaders_message():
    """A group of several devices sums through librccl, loaded at run time; a host without it must get RT_ERCCL and the
    loader's reason (not a crash: dlerror() hands its message out once).  RT_AMD_RCCL_LIB names the file to load, so a
    name that does not exist is a host without the library.  No GPU is needed to ask (rt_rccl_available)."""
    code = ("from raytracing_amd import _ffi as f\nL = f.lib()\nst = L.rt_rccl_available()\n"
            "print(st, L.rt_last_error().decode())\nst2 = L.rt_rccl_available()\nprint(st2, L.rt_last_error().decode())")
    r = _child(code, RT_AMD_RCCL_LIB="/nonexistent/librccl.so.1")
    assert r.returncode == 0, r.stderr
    first, second = r.stdout.strip().splitlines()
    assert first.startswith("-6 ") and "/nonexistent/librccl.so.1 could not be loaded: " in first, first
    assert "No such file" in first or "cannot open" in first, first        # the loader's own words, not "unknown error"
    assert second == first                                                 # the answer is kept, and asking again does not crash
    # with the library where the image has it the answer is RT_OK (skipped where there is no ROCm installation)
    if os.path.exists("/opt/rocm/lib/librccl.so.1"):
        r = _child(code)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip().splitlines()[0].startswith("0 "), r.stdout


def test_a_group_of_more_devices_than_the_host_has_fails_cleanly():
    """a duplicate-free list that names a device the host does not have: RT_EINVAL with the range (RT_ENODEV without any
    GPU), no handle, and the library goes on working"""
    L = ffi.lib()
    world, _ = rt.get_scene("cornell_box", spp=1)
    desc, keep = flatten_scene(world, rt.build_bvh_scene(rt.SceneManager(world)))
    count = C.c_int(0)
    assert L.rt_device_count(C.byref(count)) == 0
    want = np.arange(count.value + 1, dtype=np.int32)                      # 0 .. visible: one too many, none twice
    handle = C.c_void_p()
    st = L.rt_group_create(C.byref(desc), ffi.iptr(want), len(want), C.byref(handle))
    if count.value == 0:
        assert st == -2 and not handle
    else:
        assert st == -1 and b"out of range" in L.rt_last_error() and not handle
    assert L.rt_abi_version() == 1
    del keep
