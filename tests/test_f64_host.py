"""The Float64 mode without a GPU: the oracle built for Float64 (oracle/rt_ref.cpp with -DREF_F64), its deterministic
routines against libm, the host-side mirror with element type float64, the library's Float64 tree builders against the
oracle's, and the checks rt_scene_create_f64 makes before it looks for a device."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd import hittables as H
from conftest import SCENES

F = np.float64


def test_deterministic_routines_at_double_accuracy():
    L = O.lib(F)
    g = np.random.default_rng(5)
    for x in g.random(4000):
        assert abs(L.ref_logf(float(x)) - math.log(x)) <= 4e-16 * abs(math.log(x)) + 1e-300
    assert L.ref_logf(0.0) == -math.inf
    for x in (g.random(4000) - 0.5) * 4e4:
        assert abs(L.ref_sinf(float(x)) - math.sin(x)) <= 3e-16
    for x in (g.random(4000) - 0.5) * 2:
        assert abs(L.ref_acosf(float(x)) - math.acos(x)) <= 2e-15
    for y, x in (g.random((4000, 2)) - 0.5) * 2:
        assert abs(L.ref_atan2f(float(y), float(x)) - math.atan2(y, x)) <= 2e-15
    assert L.ref_acosf(1.0) == 0.0 and abs(L.ref_acosf(-1.0) - math.pi) <= 1e-15
    assert abs(L.ref_atan2f(0.0, -1.0) - math.pi) <= 1e-15


def test_random_numbers_have_53_bits():
    L = O.lib(F)
    xs = np.array([L.ref_rand01(p, 3, 7, 0x5EED, lane) for p in range(2000) for lane in (0, 1)])
    assert ((xs >= 0) & (xs < 1)).all()
    assert abs(xs.mean() - 0.5) < 0.02
    scaled = xs * 2.0 ** 53
    assert (scaled == np.floor(scaled)).all()            # multiples of 2^-53 ...
    assert (np.floor(scaled) % 2 ** 29 != 0).mean() > 0.99  # ... that use the low bits a 24-bit number would not have
    # the two numbers of one hash come from different words
    assert L.ref_rand01(1, 2, 3, 4, 0) != L.ref_rand01(1, 2, 3, 4, 1)


def test_halton_in_double():
    out = np.zeros(2 * 64, F)
    O.lib(F).ref_halton_sequence32(64, out.ctypes.data_as(ffi.c_f64p))
    with H.element_type(F):
        mine = rt.Halton_sequence32(64)
    assert mine.dtype == F
    np.testing.assert_array_equal(out.reshape(64, 2).T, mine)
    assert mine[1, 0] == 1.0 / 3.0 - 0.5                  # a third in double, not the Float32 third widened
    assert rt.Halton_sequence32(4).dtype == np.float32    # outside the block the element type is the default again


def test_scenes_are_built_in_the_element_type():
    world, kw = rt.get_scene("cornell_box", spp=3, F=F)
    assert kw["camera"].origin.dtype == F and kw["pattern"].dtype == F and kw["background"].dtype == F
    assert world[0].k.dtype == F
    t = world[-1]
    assert t.xform.dtype == F and t.inv_xform.dtype == F and t.bbox.min.dtype == F
    # cos(-18 degrees) as Float64, where the Float32 scene holds it rounded (samples/cornell_box.jl:33: F[RotY(...) ...])
    assert t.xform[0] == math.cos(math.radians(-18)) and np.float32(t.xform[0]) != t.xform[0]
    w32, _ = rt.get_scene("cornell_box", spp=3)
    assert w32[-1].xform.dtype == np.float32


@pytest.mark.parametrize("name", SCENES)
def test_library_f64_builders_equal_the_oracle_builder(name):
    """rt_bvh_build_f64 / rt_bvh_build_meshes_f64 against the Float64 oracle's restatement of src/BVH.jl:65-109,
    src/mesh.jl:217-305"""
    world, _ = rt.get_scene(name, spp=1, F=F)
    ref = O.OracleScene(world, F=F).buckets()
    with H.element_type(F):
        bvh = rt.build_bvh_scene(rt.SceneManager(world))
    pos = {id(o): i for i, o in enumerate(world)}
    assert len(ref) == len(bvh)
    for o, t in zip(ref, bvh):
        assert t.node_bounds.dtype == F
        np.testing.assert_array_equal([pos[id(x)] for x in t.objects], o["objects"])
        v = o["valid"].astype(bool) if o["is_mesh"] else np.ones(len(o["leaf"]), bool)
        for a, b in ((o["bounds"], t.node_bounds), (o["left"], t.node_left), (o["right"], t.node_right), (o["leaf"], t.node_leaf)):
            np.testing.assert_array_equal(a[v], b[v])


def test_float64_oracle_agrees_with_the_float32_oracle_at_the_first_hit():
    """cornell_box has no random part: both element types describe the same scene"""
    w32, kw32 = rt.get_scene("cornell_box", spp=1)
    w64, kw64 = rt.get_scene("cornell_box", spp=1, F=F)
    r64 = O.primary_rays(kw64["camera"], 400, 400, 48, 48, F=F)
    a = O.OracleScene(w32).trace_rays(r64.astype(np.float32))
    b = O.OracleScene(w64, F=F).trace_rays(r64)
    agree = a["prim_id"] == b["prim_id"]
    assert agree.mean() > 0.99
    hit = agree & (b["prim_id"] >= 0)
    assert hit.sum() > 1500
    np.testing.assert_allclose(a["t"][hit], b["t"][hit], rtol=3e-5)
    np.testing.assert_allclose(a["normal"][hit], b["normal"][hit], atol=1e-5)


def test_description_is_checked_before_the_device_is_looked_for():
    """On a host without a GPU a good scene ends in RT_ENODEV (-2), a bad one in the error that names what is wrong."""
    def status_of(world):
        try:
            rt.DeviceScene(world, F=F).close()
        except ffi.RtError as e:
            return str(e)
        return "ok"
    mat = rt.Material(rt.Lambertian, rt.Vec3(0.5))
    with H.element_type(F):
        good = [rt.Sphere(rt.Vec3(0), 1.0, mat), rt.Transform(rt.Box(rt.Vec3(0), rt.Vec3(1), mat), np.eye(4))]
        nested = [rt.Transform(rt.Transform(rt.Sphere(rt.Vec3(0), 1.0, mat), np.eye(4)), np.eye(4))]
        medium = [rt.ConstantMedium(rt.xy_rect(0, 1, 0, 1, 0, mat), 0.1, rt.Vec3(1))]
    msg = status_of(good)
    assert msg == "ok" or "status -2" in msg
    assert "status -5" in status_of(nested) and "nesting" in status_of(nested)
    assert "status -5" in status_of(medium) and "boundary" in status_of(medium)
    # a tree whose links form a cycle
    world, _ = rt.get_scene("cornell_box", spp=1, F=F)
    with H.element_type(F):
        from raytracing_amd.scene_manager import SceneManager, build_bvh_scene, flatten_scene
        bvh = build_bvh_scene(SceneManager(world))
        bvh[1].node_left[0] = 1
        desc, keep = flatten_scene(world, bvh)
    handle = C.c_void_p()
    st = ffi.lib().rt_scene_create_f64(C.byref(desc), 0, C.byref(handle))
    assert st == -1 and b"cycle" in ffi.lib().rt_last_error()
    assert ffi.lib().rt_scene_create_f64(None, 0, C.byref(handle)) == -1
