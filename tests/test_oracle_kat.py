"""Known-answer tests that pin the CPU oracle (oracle/rt_ref.cpp) to hand-derived results of the reference's formulas.
The reference has no tests of its own; every expected value below is derived from the cited source lines."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd import (Box, Camera, CheckerTexture, ConstantMedium, Dielectric, Emissive, ImageTexture, Lambertian,
                            Material, Mesh, Metal, MovingSphere, Sphere, Transform, Vec3, affine, xy_rect, xz_rect, yz_rect)

F = np.float32
INF = np.float32(np.inf)
white = Material(Lambertian, Vec3(0.73))


def ray(o, d, time=0.0):
    return np.array([*o, *d, time], dtype=F)


def hit1(world, o, d, time=0.0, t_min=0.001, t_max=np.inf):
    h = O.OracleScene(world).trace_rays(ray(o, d, time)[None], t_min, t_max)
    return {k: v[0] for k, v in h.items()}


# ------------------------------------------------------------------ src/MSAA_patterns.jl:25-58
def test_halton_prefix():
    out = np.zeros(16, dtype=F)
    O.lib().ref_halton_sequence32(8, out.ctypes.data_as(C.POINTER(C.c_float)))
    base2 = [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]
    base3 = [1 / 3, 2 / 3, 1 / 9, 4 / 9, 7 / 9, 2 / 9, 5 / 9, 8 / 9]
    np.testing.assert_array_equal(out[0::2], np.array(base2, F) - F(0.5))
    np.testing.assert_array_equal(out[1::2], np.array([F(a) / F(b) for a, b in
                                                       [(1, 3), (2, 3), (1, 9), (4, 9), (7, 9), (2, 9), (5, 9), (8, 9)]],
                                                      F) - F(0.5))
    assert np.allclose(out[1::2] + 0.5, base3, atol=1e-7)
    # the host mirror computes the same numbers
    np.testing.assert_array_equal(rt.Halton_sequence32(8).T.reshape(-1), out)


# ------------------------------------------------------------------ the shared counter-based generator
def _pcg4d(v):
    M = 0xFFFFFFFF
    v = [(x * 1664525 + 1013904223) & M for x in v]
    for rnd in range(2):
        v[0] = (v[0] + v[1] * v[3]) & M
        v[1] = (v[1] + v[2] * v[0]) & M
        v[2] = (v[2] + v[0] * v[1]) & M
        v[3] = (v[3] + v[1] * v[2]) & M
        if rnd == 0:
            v = [x ^ (x >> 16) for x in v]
    return v


def test_pcg4d_matches_published_algorithm():
    for v in ([0, 0, 0, 0], [1, 2, 3, 4], [0xFFFFFFFF, 0x12345678, 0x9E3779B9, 7], [639999, 999, (49 << 26) | (2 << 24) | 5, 0x5EED]):
        arr = (C.c_uint32 * 4)(*v)
        O.lib().ref_pcg4d(arr)
        assert list(arr) == _pcg4d(v)
    # rand01 is the top 24 bits: uniform on [0, 1), never 1
    r = O.lib().ref_rand01(5, 6, 7, 8, 2)
    assert r == F(_pcg4d([5, 6, 7, 8])[2] >> 8) * F(2.0 ** -24) and 0.0 <= r < 1.0


def test_deterministic_math_close_to_libm():
    L = O.lib()
    xs = (np.arange(1, 4000, dtype=np.float64) * 4099 % (1 << 24)).astype(F) * F(2.0 ** -24)
    got = np.array([L.ref_logf(float(x)) for x in xs], F)
    want = np.log(xs.astype(np.float64))
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)) < 3e-7
    assert L.ref_logf(0.0) == -np.inf
    ang = np.linspace(-6000.0, 6000.0, 4001).astype(F)
    got = np.array([L.ref_sinf(float(a)) for a in ang], F)
    assert np.max(np.abs(got - np.sin(ang.astype(np.float64)))) < 3e-7
    cs = np.linspace(-1, 1, 501).astype(F)
    got = np.array([L.ref_acosf(float(c)) for c in cs], F)
    assert np.max(np.abs(got - np.arccos(cs.astype(np.float64)))) < 4e-7
    ys, xs2 = np.meshgrid(np.linspace(-2, 2, 41), np.linspace(-2, 2, 41))
    got = np.array([L.ref_atan2f(float(y), float(x)) for y, x in zip(ys.ravel(), xs2.ravel())], F)
    assert np.max(np.abs(got - np.arctan2(ys.ravel(), xs2.ravel()))) < 4e-7
    assert L.ref_atan2f(-0.0, -1.0) == F(-math.pi) and L.ref_atan2f(0.0, -1.0) == F(math.pi)


# ------------------------------------------------------------------ src/sphere.jl:45-72,104-108
def test_sphere_hit_front_and_inside():
    s = Sphere(Vec3(0, 0, 5), 1, white)
    h = hit1([s], (0, 0, 0), (0, 0, 1))
    assert h["prim_id"] == 0 and h["t"] == F(4) and h["front"] == 1
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    # uv from the outward normal (0,0,-1): theta = acos(0) -> v = 1/2, phi = atan(1, 0) + pi -> u = 3/4
    np.testing.assert_allclose(h["uv"], [0.75, 0.5], atol=1e-7)
    # t is measured in units of |direction| (the reference never normalises directions)
    assert hit1([s], (0, 0, 0), (0, 0, 2))["t"] == F(2)
    # from the centre: near root is behind t_min, far root is used; the stored normal faces the ray
    h = hit1([s], (0, 0, 5), (0, 0, 1))
    assert h["t"] == F(1) and h["front"] == 0
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    # grazing miss and the t_max window (root test is done scaled by a = d.d)
    assert hit1([s], (0, 1.5, 0), (0, 0, 1))["prim_id"] == -1
    assert hit1([s], (0, 0, 0), (0, 0, 1), t_max=3.9)["prim_id"] == -1
    # t == t_max passes the sphere's own window (src/sphere.jl:58) but the BVH's slab test rejects a box whose
    # entry distance equals t_max (t_max <= t_min, src/bounding_box.jl:62), so through the scene it is a miss
    assert hit1([s], (0, 0, 0), (0, 0, 1), t_max=4.0)["prim_id"] == -1
    assert hit1([s], (0, 0, 0), (0, 0, 1), t_max=4.5)["t"] == F(4)


def test_moving_sphere_centre_at_time():
    m = MovingSphere(Vec3(0, 0, 5), Vec3(2, 0, 5), 0, 1, 1, white)  # src/sphere.jl:24-26
    assert hit1([m], (1, 0, 0), (0, 0, 1), time=0.5)["t"] == F(4)   # centre (1,0,5)
    assert hit1([m], (1, 0, 0), (0, 0, 1), time=0.0)["t"] == F(5)   # centre (0,0,5): tangent at x = 1
    assert hit1([m], (3.5, 0, 0), (0, 0, 1), time=0.0)["prim_id"] == -1


# ------------------------------------------------------------------ src/aarect.jl:49-119
def test_rect_hits_and_closed_bounds():
    r = xy_rect(0, 2, 0, 2, 3, white)
    h = hit1([r], (1, 0.5, 0), (0, 0, 1))
    assert h["t"] == F(3) and h["front"] == 0
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])  # (-1 + 2*front) * (0,0,1): always faces the ray
    np.testing.assert_array_equal(h["uv"], [0.5, 0.25])
    h = hit1([r], (1, 0.5, 6), (0, 0, -1))
    assert h["front"] == 1 and h["normal"][2] == 1
    assert hit1([r], (2, 2, 0), (0, 0, 1))["prim_id"] == 0          # x == x1 and y == y1 are inside (closed)
    assert hit1([r], (2.0001, 1, 0), (0, 0, 1))["prim_id"] == -1
    assert hit1([r], (1, 1, 0), (1, 0, 0))["prim_id"] == -1         # parallel: t = +-Inf / NaN -> miss
    assert hit1([xz_rect(0, 2, 0, 2, 3, white)], (1, 0, 0.5), (0, 1, 0))["t"] == F(3)
    h = hit1([yz_rect(0, 2, 0, 4, 3, white)], (0, 1, 1), (1, 0, 0))
    assert h["t"] == F(3)
    np.testing.assert_array_equal(h["uv"], [0.5, 0.25])


# ------------------------------------------------------------------ src/bounding_box.jl:20-67 in front of src/aarect.jl:49-95
def test_leaf_box_is_tested_before_the_primitive():
    """The tree tests the box of a leaf first and turns it down when `t_max <= t_min` after any axis.  A rectangle's
    box is its plane +- 0.0001: from far enough away the two plane distances round to one float, and the rectangle is
    never reached although its own hit function would accept the ray."""
    r = xy_rect(-10, 10, -10, 10, 3, white)
    assert hit1([r], (0, 0, -1000), (0, 0, 1))["t"] == F(1003)          # ulp(1003) = 6.1e-5 < 0.0002: two distances
    assert hit1([r], (0, 0, -5000), (0, 0, 1))["prim_id"] == -1         # ulp(5003) = 4.9e-4: both planes at one distance
    # a ray inside the plane of a box face: 0 * Inf = NaN on that axis is ignored, the other axes decide
    b = Box(Vec3(0, 0, 0), Vec3(2, 2, 2), white)
    assert hit1([b], (-1, 1, 2), (1, 0, 0))["t"] == F(1)
    # ... also when the primitive only touches that plane (tangent ray along the top of a sphere's box)
    assert hit1([Sphere(Vec3(0, 0, 0), 1, white)], (-5, 1, 0), (1, 0, 0))["t"] == F(5)


# ------------------------------------------------------------------ src/box.jl:9-24,43-83
def test_box_sides_in_reference_order():
    b = Box(Vec3(0, 0, 0), Vec3(1, 2, 3), white)
    cases = {0: ((0.5, 1, 9), (0, 0, -1)), 1: ((0.5, 1, -9), (0, 0, 1)), 2: ((0.5, 9, 1), (0, -1, 0)),
             3: ((0.5, -9, 1), (0, 1, 0)), 4: ((9, 1, 1), (-1, 0, 0)), 5: ((-9, 1, 1), (1, 0, 0))}
    for side, (o, d) in cases.items():
        h = hit1([b], o, d)
        assert h["prim_id"] == 0 and h["sub_id"] == side, side
        np.testing.assert_array_equal(h["normal"], [-x for x in d])
    # from inside the box the exit face is hit
    assert hit1([b], (0.5, 1, 1), (0, 0, 1))["sub_id"] == 0


# ------------------------------------------------------------------ src/transform.jl:104-116
def test_transform_translation_and_rotation():
    s = Sphere(Vec3(0, 0, 0), 1, white)
    t = Transform(s, affine(rt.RotY(0.0), [0, 0, 5]))
    h = hit1([t], (0, 0, 0), (0, 0, 1))
    assert h["t"] == F(4)
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    # a box rotated by 90 degrees about y: its +x face (side 4) now faces -z
    b = Transform(Box(Vec3(-1, -1, -1), Vec3(1, 1, 1), white), affine(rt.RotY(math.pi / 2), [0, 0, 5]))
    h = hit1([b], (0, 0, 0), (0, 0, 1))
    assert abs(h["t"] - 4) < 1e-5 and h["sub_id"] == 4
    np.testing.assert_allclose(h["normal"], [0, 0, -1], atol=1e-6)


# ------------------------------------------------------------------ src/triangle.jl:73-121, src/mesh.jl
def _tri_mesh(normals=True):
    v = np.zeros((3, 8), F)
    v[:, :3] = [[0, 0, 5], [2, 0, 5], [0, 2, 5]]
    v[:, 3:6] = [0, 0, -1] if normals else [0, 0, 0]
    v[:, 6:8] = [[0.1, 0.2], [0.5, 0.2], [0.1, 0.9]]
    return Mesh(v, [white], np.array([[0, 1, 2, 0]], np.int32))


def test_triangle_hit_and_inverted_normal_flag():
    h = hit1([_tri_mesh()], (0.5, 0.5, 0), (0, 0, 1))
    assert h["prim_id"] == 0 and h["sub_id"] == 0 and h["t"] == F(5)
    # the mesh carries vertex normals, so (inverted flag, src/triangle.jl:110-113) the GEOMETRIC normal is used:
    # cross(e1, e2) = (0,0,4) -> +z, the ray travels +z -> back face, stored normal faces the ray
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    assert h["front"] == 0
    # interpolated vertex uv (both >= 0): w=0.5, u=0.25, v=0.25
    np.testing.assert_allclose(h["uv"], [0.5 * 0.1 + 0.25 * 0.5 + 0.25 * 0.1, 0.5 * 0.2 + 0.25 * 0.2 + 0.25 * 0.9], atol=1e-6)
    assert hit1([_tri_mesh()], (1.5, 1.5, 0), (0, 0, 1))["prim_id"] == -1   # u + v > 1
    # a mesh WITHOUT normals gets the (zero) vertex normal -> normalize(0) = NaN
    h = hit1([_tri_mesh(normals=False)], (0.5, 0.5, 0), (0, 0, 1))
    assert h["t"] == F(5) and np.isnan(h["normal"]).all()


# ------------------------------------------------------------------ tie-breaks, src/scene_manager.jl:42-49, src/BVH.jl:41-49
def test_tie_breaks():
    a, b = xy_rect(0, 2, 0, 2, 3, white), xy_rect(0, 2, 0, 2, 3, white)
    # same type group: the stack pops the right child first, so the LATER of two coincident objects is found first
    # and a strict < keeps it
    assert hit1([a, b], (1, 1, 0), (0, 0, 1))["prim_id"] == 1
    # different type groups (different texture type): <= keeps the EARLIER group
    c = xy_rect(0, 2, 0, 2, 3, Material(Lambertian, CheckerTexture(Vec3(0), Vec3(1))))
    assert hit1([a, c], (1, 1, 0), (0, 0, 1))["prim_id"] == 0
    assert hit1([c, a], (1, 1, 0), (0, 0, 1))["prim_id"] == 0
    # groups are formed by first appearance: [a, c, b] -> groups (a, b), (c); inside the first group b wins
    assert hit1([a, c, b], (1, 1, 0), (0, 0, 1))["prim_id"] == 2


def test_closest_of_many_and_type_groups():
    world = [Sphere(Vec3(0, 0, 10), 1, white), xy_rect(-5, 5, -5, 5, 20, white), Sphere(Vec3(0, 0, 6), 1, white),
             Box(Vec3(-1, -1, 2), Vec3(1, 1, 3), white)]
    sm = rt.SceneManager(world)
    assert [len(g) for g in sm] == [2, 1, 1] and [type(g[0]).__name__ for g in sm] == ["Sphere", "xy_rect", "Box"]
    h = hit1(world, (0, 0, 0), (0, 0, 1))
    assert h["prim_id"] == 3 and h["sub_id"] == 1 and h["t"] == F(2)
    assert hit1(world, (0, 3, 0), (0, 0, 1))["prim_id"] == 1
    assert hit1(world, (0, 30, 0), (0, 0, 1))["prim_id"] == -1


# ------------------------------------------------------------------ src/constant_medium.jl:29-65
def test_constant_medium():
    boundary = Sphere(Vec3(0, 0, 5), 1, Material(Dielectric, 1.5))
    dense = ConstantMedium(boundary, 1e6, Vec3(1))
    h = hit1([dense], (0, 0, 0), (0, 0, 1))
    assert h["prim_id"] == 0 and 4.0 <= h["t"] < 4.001 and h["front"] == 0
    np.testing.assert_array_equal(h["normal"], [1, 0, 0])
    np.testing.assert_array_equal(h["uv"], [0, 0])
    thin = ConstantMedium(boundary, 1e-9, Vec3(1))
    assert hit1([thin], (0, 0, 0), (0, 0, 1))["prim_id"] == -1
    # deterministic: same seed, same distance; the distance obeys t = t1 - log(u)/density with u from the shared generator
    med = ConstantMedium(boundary, 2.0, Vec3(1))
    sc = O.OracleScene([med])
    r = ray((0, 0, 0), (0, 0, 1))[None]
    t_a, t_b = sc.trace_rays(r, seed=1)["t"][0], sc.trace_rays(r, seed=1)["t"][0]
    assert t_a == t_b
    # object index of the medium in the flattened table is 1 (the boundary sphere is registered first)
    u = O.lib().ref_rand01(0, 0, (0 << 26) | (2 << 24) | 1, 1, 0)
    want = 4.0 - math.log(u) / 2.0
    if want <= 6.0:
        assert abs(t_a - want) < 1e-5
    else:
        assert t_a == INF
    # the far window: starting inside the boundary the first probe is behind the origin and t1 is clamped to t_min, then 0
    h = hit1([dense], (0, 0, 5), (0, 0, 1))
    assert 0.001 <= h["t"] < 0.002


# ------------------------------------------------------------------ src/Ray_Tracing.jl:11-29 through the camera
def _camera():
    return Camera(Vec3(0, 0, 0), Vec3(0, 0, 1), Vec3(0, 1, 0), 90.0, 1.0, 0.0, 1.0)   # looks down -z (direction = origin - target)


def _radiance(world, background, ids, pattern=None, max_depth=50, W=4, H=4):
    pattern = np.zeros((2, 1), F) if pattern is None else pattern
    return O.OracleScene(world).trace_paths(_camera(), pattern, background, W, H, ids, max_depth=max_depth)


def test_trace_emission_background_and_depth():
    light = Material(Emissive, Vec3(7, 5, 3))
    wall = xy_rect(-100, 100, -100, 100, -10, light)
    ids = np.array([[1, 2, 0]], np.int32)
    # a ray that reaches an Emissive surface returns acc * emitted and stops (zero attenuation), both faces emit
    np.testing.assert_array_equal(_radiance([wall], Vec3(0.2), ids)[0], [7, 5, 3])
    # a miss returns acc * background
    np.testing.assert_array_equal(_radiance([xy_rect(-1, 1, -1, 1, 50, light)], Vec3(0.25, 0.5, 0.75), ids)[0], [0.25, 0.5, 0.75])
    # two facing perfect mirrors: the path runs out of depth and returns the emission collected so far (zero), NOT the background
    mirror = Material(Metal, Vec3(1, 1, 1), 0.0)
    mirrors = [xy_rect(-1e4, 1e4, -1e4, 1e4, -10, mirror), xy_rect(-1e4, 1e4, -1e4, 1e4, 10, mirror)]
    np.testing.assert_array_equal(_radiance(mirrors, Vec3(9, 9, 9), ids)[0], [0, 0, 0])
    # depth 1: one bounce off a 0.5-grey mirror, then the loop ends without looking at the background
    np.testing.assert_array_equal(_radiance([xy_rect(-1e4, 1e4, -1e4, 1e4, -10, mirror)], Vec3(1), ids, max_depth=1)[0], [0, 0, 0])
    # depth 2: mirror -> miss -> albedo * background
    grey = Material(Metal, Vec3(0.5, 0.25, 1.0), 0.0)
    np.testing.assert_array_equal(_radiance([xy_rect(-1e4, 1e4, -1e4, 1e4, -10, grey)], Vec3(1), ids, max_depth=2)[0], [0.5, 0.25, 1.0])


def test_path_rays_lists_the_rays_of_trace():
    """path_rays: the camera ray and every scattered ray of one path, in order (src/Ray_Tracing.jl:11-29)"""
    mirror = Material(Metal, Vec3(1, 1, 1), 0.0)
    mirrors = [xy_rect(-1e4, 1e4, -1e4, 1e4, -10, mirror), xy_rect(-1e4, 1e4, -1e4, 1e4, 10, mirror)]
    S = O.OracleScene(mirrors)
    cam, pat = _camera(), np.zeros((2, 1), F)
    rays = S.path_rays(cam, pat, Vec3(0), 4, 4, 1, 2, 0, max_depth=7)
    assert rays.shape == (7, 7)                                            # depth 7: seven rays, all of them hit
    np.testing.assert_array_equal(rays[0, :3], np.asarray(cam.origin, F))
    h = S.trace_rays(rays)
    assert (h["prim_id"] >= 0).all()
    for k in range(6):  # each ray starts where the one before it ended, reflected about z
        p = rays[k, :3] + h["t"][k] * rays[k, 3:6]
        np.testing.assert_allclose(rays[k + 1, :3], p, rtol=1e-6, atol=1e-6)
        assert np.sign(rays[k + 1, 5]) == -np.sign(rays[k, 5])
    assert S.path_rays(cam, pat, Vec3(0), 4, 4, 1, 2, 0, max_depth=0).shape == (0, 7)


def test_image_texture_lookup_quirks():
    # src/texture.jl:55-76: v is flipped, indices are trunc(v*H), trunc(u*W) with 0 -> 1 (1-based): row/column 1 serve two
    # cells, the last row/column only the exact border
    img = np.zeros((4, 4, 3), np.uint8)
    for r in range(4):
        for c in range(4):
            img[r, c] = (10 * (r + 1), 10 * (c + 1), 255)
    tex = Material(Emissive, ImageTexture(img))
    wall = xy_rect(-8, 8, -8, 8, -8, tex)      # fills the 90-degree view at z = -8
    cam = _camera()
    W = H = 8
    ids = np.array([(i, j, 0) for j in range(H) for i in range(W)], np.int32)
    rad = O.OracleScene([wall]).trace_paths(cam, np.zeros((2, 1), F), Vec3(0), W, H, ids)
    for (i, j, _), c in zip(ids, rad):
        # pixel (i, j): s = i/(W-1), t = j/(H-1); the wall spans the viewport exactly, so u = s, v = t
        u, v = F(i) / F(W - 1), F(j) / F(H - 1)
        row = max(int(np.trunc((F(1) - v) * F(4))), 1)
        col = max(int(np.trunc(u * F(4))), 1)
        want = img[row - 1, col - 1].astype(F) / F(255)
        np.testing.assert_allclose(c, want, atol=1e-6, err_msg=f"pixel {i},{j}")


def test_camera_frame():
    # src/camera.jl:16-30: fov is HORIZONTAL, direction = origin - target, llc = o - h/2 - v/2 - fd*w
    cam = Camera(Vec3(0, 0, 0), Vec3(0, 0, 1), Vec3(0, 1, 0), 90.0, 2.0, 0.0, 3.0)
    np.testing.assert_allclose(cam.horizontal, [6, 0, 0], atol=1e-6)      # 2*tan(45 deg)*fd
    np.testing.assert_allclose(cam.vertical, [0, 3, 0], atol=1e-6)        # width / aspect
    np.testing.assert_allclose(cam.lower_left_corner, [-3, -1.5, -3], atol=1e-6)
    assert cam.lens_radius == 0 and cam.t0 == 0 and cam.t1 == 0


def test_render_layout_and_gamma():
    # draw!: img[height-j, i+1] = sqrt(mean) (src/Ray_Tracing.jl:90-92): row 0 is the TOP of the picture, values unclamped
    light = Material(Emissive, Vec3(4, 4, 4))
    top_half = xy_rect(-100, 100, 0, 100, -10, light)
    img = O.OracleScene([top_half]).render(_camera(), np.zeros((2, 1), F), Vec3(0), 6, 4)
    assert img.shape == (4, 6, 3)
    np.testing.assert_array_equal(img[0], np.full((6, 3), 2.0, F))   # top row sees the light, sqrt(4) = 2 (not clamped)
    np.testing.assert_array_equal(img[3], np.zeros((6, 3), F))


def _two_far_spheres():
    """Two spheres seen from 9000 away, where the sphere test (src/sphere.jl:45-65) has lost its digits: B's root
    (0.8658094) lies in front of B's own box (entered at 0.8659409), A's root (0.8658121) in between."""
    ray = np.array([[-103.47892761230469, 8866.81640625, -2951.947509765625,
                     124.83503723144531, -10233.1455078125, 3403.513916015625, 0.1916230320930481]], np.float32)
    a = lambda tex: rt.Sphere(rt.Vec3(4.6211205, 6.9095955, -4.923138), 0.9200988, rt.Material(rt.Lambertian, tex))
    b = lambda tex: rt.Sphere(rt.Vec3(3.8248086, 3.4844747, -5.6529064), 2.0317829, rt.Material(rt.Lambertian, tex))
    solid = lambda: rt.SolidColor(rt.Vec3(0.5))
    check = lambda: rt.CheckerTexture(rt.Vec3(0.1), rt.Vec3(0.9))      # another texture type: another type group
    worlds = {"A|B": [a(solid()), b(check())], "B|A": [b(check()), a(solid())],
              "A,B": [a(solid()), b(solid())], "B,A": [b(solid()), a(solid())]}
    return ray, worlds


def test_a_hit_in_front_of_its_box_and_the_bound_between_type_groups():
    """In one tree A is met first and its distance culls B's box although B's root is the smaller one
    (src/BVH.jl:36-47 with src/bounding_box.jl:20-67).  In two type groups B is tested with the ray's own t_max, whichever
    comes first: hit(::SceneManager) only hands on `closest_t` when a later group's hit replaces the one held, so the first
    group's hit never becomes a bound (src/scene_manager.jl:40-52), and B's smaller root wins the reduction."""
    ray, worlds = _two_far_spheres()
    t_a, t_b = np.float32(0.8658121228218079), np.float32(0.865809440612793)
    want = {"A|B": (1, t_b), "B|A": (0, t_b), "A,B": (0, t_a), "B,A": (1, t_a)}
    for name, world in worlds.items():
        hit = O.OracleScene(world).trace_rays(ray)
        assert (int(hit["prim_id"][0]), hit["t"][0]) == want[name], name
    # B's box for this ray, as the reference computes it
    o, d = ray[0, :3], ray[0, 3:6]
    c, r = np.array([3.8248086, 3.4844747, -5.6529064], np.float32), np.float32(2.0317829)
    t0, t1 = ((c - r) - o) / d, ((c + r) - o) / d
    entry = np.minimum(t0, t1).max()
    assert t_b < t_a < entry < np.maximum(t0, t1).min()
