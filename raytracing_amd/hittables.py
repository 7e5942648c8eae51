"""Host-side mirror of the reference's scene vocabulary (Vec3, Material, textures, the
Hittable types, Camera, sub-pixel patterns).  Names, argument order and arithmetic
follow the Julia sources so that scene scripts and tests read like the reference's.
The element type F of render(F, ...) (main.jl:21) is the module variable `F`: numpy.float32, the reference's
default, unless code runs inside `with element_type(numpy.float64):` -- every constructor and helper below reads it
when it is called, as the reference's take Vec3{F}.
"""
import contextlib
import math

import numpy as np

F = np.float32


@contextlib.contextmanager
def element_type(T):
    """Scenes, cameras and patterns made inside the block have element type T (numpy.float32 or numpy.float64):
    the F of render(F, ...), get_scene(key; F=F) and Vec3{F} (main.jl:21,90-94)."""
    global F
    T = np.dtype(T).type
    if T not in (np.float32, np.float64):
        raise TypeError("element type must be float32 or float64")
    old, F = F, T
    try:
        yield
    finally:
        F = old

# material types, src/material.jl:3-9
Lambertian, Metal, Dielectric, Emissive, Isotropic = 1, 2, 3, 4, 5


def Vec3(x, y=None, z=None):
    """src/vec3.jl:1-9"""
    if y is None:
        if np.ndim(x) == 0:
            return np.array([x, x, x], dtype=F)
        return np.asarray(x, dtype=F).copy()
    return np.array([x, y, z], dtype=F)


def length(a):
    """src/vec3.jl:52: sqrt((x*x + y*y) + z*z) in Float32"""
    a = np.asarray(a, dtype=F)
    return np.sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])


def normalize(a):
    """src/vec3.jl:57 with src/vec3.jl:31 (a * F(1/len))"""
    return np.asarray(a, dtype=F) * (F(1) / length(a))


def cross(a, b):
    """src/vec3.jl:49"""
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=F)


# ---------------------------------------------------------------- textures (src/texture.jl)
class Texture:
    pass


class SolidColor(Texture):
    def __init__(self, color=(0, 0, 0)):
        self.color = Vec3(color)


class CheckerTexture(Texture):
    def __init__(self, odd, even):
        self.odd = odd if isinstance(odd, Texture) else SolidColor(odd)
        self.even = even if isinstance(even, Texture) else SolidColor(even)


class ImageTexture(Texture):
    """src/texture.jl:23-28.  `image` is an (H, W, 3) uint8 array (row 0 = top) or the path of a raw
    RGB8 fixture written by tools/decode_assets.py ('<name>.rgb8' with a '<name>.json' holding its shape)."""

    def __init__(self, image=None):
        if image is None:
            image = np.zeros((0, 0, 3), dtype=np.uint8)
        elif isinstance(image, str):
            import json
            with open(image.rsplit(".", 1)[0] + ".json") as f:
                meta = json.load(f)
            image = np.fromfile(image, dtype=np.uint8).reshape(meta["height"], meta["width"], 3)
        self.image = np.ascontiguousarray(image, dtype=np.uint8)


class Perlin3:
    """src/perlin.jl:5-33: 256 normalised random vectors (or scalars) and three permutations of 0..255, drawn by the
    library (rt_perlin_generate) from the position `rng` (a scenes.SceneRng) stands at; `rng` moves past the draws."""

    def __init__(self, rng, vector=True, point_count=256):
        import ctypes as C
        from . import _ffi as ffi
        assert point_count == 256
        self.vector = vector
        table, ctr = ffi.rtPerlin(), C.c_uint64(rng.ctr)
        ffi.check(ffi.lib().rt_perlin_generate(rng.seed, C.byref(ctr), int(bool(vector)), C.byref(table)))
        rng.ctr = ctr.value
        self.random = np.array(table.ranvec, dtype=F).reshape(point_count, 3)
        self.perm_x, self.perm_y, self.perm_z = list(table.perm_x), list(table.perm_y), list(table.perm_z)


class NoiseTexture(Texture):
    """src/texture.jl:30-37: NoiseTexture{Vec3{F}}(scale) is the marble texture, NoiseTexture{F} the scalar one."""

    def __init__(self, scale, rng, vector=True):
        self.noise = Perlin3(rng, vector=vector)
        self.scale = float(scale)
        self.vector = vector


class Material:
    """src/material.jl:11-21.  Material(type, albedo[, fuzz]) or Material(Dielectric, eta)."""

    def __init__(self, type, albedo=None, fuzz=0.0, eta=0.0):
        self.type = int(type)
        if albedo is not None and np.ndim(albedo) == 0 and not isinstance(albedo, Texture):
            eta = albedo  # Material(type, eta::F), src/material.jl:21
            albedo = SolidColor(Vec3(0))
        elif albedo is None:
            albedo = SolidColor(Vec3(0))
        elif not isinstance(albedo, Texture):
            albedo = SolidColor(Vec3(albedo))
        self.albedo = albedo
        self.fuzz = F(fuzz)
        self.eta = F(eta)


# ---------------------------------------------------------------- hittables
class AABB:
    def __init__(self, mn, mx):
        self.min = Vec3(mn)
        self.max = Vec3(mx)

    def as6(self):
        return np.concatenate([self.min, self.max])  # in the element type the box was made in


class Hittable:
    pass


class Sphere(Hittable):
    def __init__(self, center, radius, material):
        self.center, self.radius, self.material = Vec3(center), F(radius), material


class MovingSphere(Hittable):
    """MovingSphere(center0, center1, t0, t1, radius, material), src/sphere.jl:21"""

    def __init__(self, center0, center1, t0, t1, radius, material):
        self.center0, self.center1 = Vec3(center0), Vec3(center1)
        self.t0, self.t1, self.radius, self.material = F(t0), F(t1), F(radius), material


class _Rect(Hittable):
    def __init__(self, a0, a1, b0, b1, k, material):
        self.a0, self.a1, self.b0, self.b1, self.k, self.material = F(a0), F(a1), F(b0), F(b1), F(k), material


class xy_rect(_Rect):
    pass


class xz_rect(_Rect):
    pass


class yz_rect(_Rect):
    pass


class Box(Hittable):
    def __init__(self, p0, p1, material):
        self.box_min, self.box_max, self.material = Vec3(p0), Vec3(p1), material


def RotY(theta):
    """Rotations.RotY as a Float64 3x3 matrix."""
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float64)


def affine(rot, t):
    """F[rot t; 0 0 0 1] as the sample scenes write it (samples/cornell_box.jl:32)."""
    m = np.zeros((4, 4), dtype=np.float64)
    m[:3, :3] = rot
    m[:3, 3] = t
    m[3, 3] = 1.0
    return m.astype(F)


class Transform(Hittable):
    """src/transform.jl:44-66: stores xform and inv(xform) as 3x4 column-major 12-tuples and the AABB of the
    eight transformed corners of the child's box."""

    def __init__(self, obj, xform):
        xform = np.asarray(xform, dtype=F)
        inv_xform = np.linalg.inv(xform).astype(F)
        bbox = bounding_box(obj)
        mn = [math.inf] * 3
        mx = [-math.inf] * 3
        for k in (0, 1):
            for j in (0, 1):
                for i in (0, 1):
                    v = Vec3(F(i) * bbox.max[0] + F(1 - i) * bbox.min[0],
                             F(j) * bbox.max[1] + F(1 - j) * bbox.min[1],
                             F(k) * bbox.max[2] + F(1 - k) * bbox.min[2])
                    tv = [((xform[r, 0] * v[0] + xform[r, 1] * v[1]) + xform[r, 2] * v[2]) + xform[r, 3]
                          for r in range(3)]
                    for a in range(3):
                        mn[a] = min(float(tv[a]), mn[a])
                        mx[a] = max(float(tv[a]), mx[a])
        self.object = obj
        self.xform = xform[:3, :4].T.reshape(-1).astype(F).copy()          # column-major 3x4
        self.inv_xform = inv_xform[:3, :4].T.reshape(-1).astype(F).copy()
        self.bbox = AABB(mn, mx)


class ConstantMedium(Hittable):
    """ConstantMedium(boundary, density, color), src/constant_medium.jl:7,21-23"""

    def __init__(self, boundary, density, color):
        self.boundary = boundary
        self.phase_function = Material(Isotropic, Vec3(color))
        self.neg_inv_density = F(-1) / F(density)


class Mesh(Hittable):
    """src/mesh.jl:5-9.  vertices: (nv, 8) float32 rows of position, normal, u, v; indices: (ntri, 4) int32
    rows = the columns of the reference's 4 x N matrix, 0-based here."""

    def __init__(self, vertices, materials, indices):
        self.vertices = np.ascontiguousarray(vertices, dtype=F)
        self.materials = list(materials)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        assert self.vertices.ndim == 2 and self.vertices.shape[1] == 8
        assert self.indices.ndim == 2 and self.indices.shape[1] == 4


def bounding_box(obj):
    """bounding_box(...) of every hittable, in the reference's Float32 arithmetic."""
    if isinstance(obj, Sphere):  # src/sphere.jl:32-34
        r = np.abs(obj.radius)
        return AABB(obj.center - r, obj.center + r)
    if isinstance(obj, MovingSphere):  # src/sphere.jl:36-38
        r = np.abs(obj.radius)
        return AABB(np.minimum(obj.center0 - r, obj.center1 - r), np.maximum(obj.center0 + r, obj.center1 + r))
    e = F(0.0001)
    if isinstance(obj, xy_rect):  # src/aarect.jl:37-39
        return AABB([obj.a0, obj.b0, obj.k - e], [obj.a1, obj.b1, obj.k + e])
    if isinstance(obj, xz_rect):  # src/aarect.jl:41-43
        return AABB([obj.a0, obj.k - e, obj.b0], [obj.a1, obj.k + e, obj.b1])
    if isinstance(obj, yz_rect):  # src/aarect.jl:45-47
        return AABB([obj.k - e, obj.a0, obj.b0], [obj.k + e, obj.a1, obj.b1])
    if isinstance(obj, Box):  # src/box.jl:85-87
        return AABB(obj.box_min, obj.box_max)
    if isinstance(obj, Transform):  # src/transform.jl:100-102
        return obj.bbox
    if isinstance(obj, ConstantMedium):  # src/constant_medium.jl:25-27
        return bounding_box(obj.boundary)
    if isinstance(obj, Mesh):  # src/mesh.jl:45-49 (Float64 pad, narrowed)
        pos = obj.vertices[:, :3]
        mn = (pos.min(axis=0).astype(np.float64) - 0.0001).astype(F)
        mx = (pos.max(axis=0).astype(np.float64) + 0.0001).astype(F)
        return AABB(mn, mx)
    from .scene_manager import BVHTree
    if isinstance(obj, BVHTree):  # src/BVH.jl:61-63
        return AABB(obj.node_bounds[0, :3], obj.node_bounds[0, 3:])
    raise TypeError(f"bounding_box: unsupported hittable {type(obj).__name__}")


# ---------------------------------------------------------------- camera (src/camera.jl)
class Camera:
    """Camera(origin, direction, up, fov, aspect_ratio, aperture, focus_dist[, t0, t1]); fov is HORIZONTAL."""

    def __init__(self, origin=None, direction=None, up=None, fov=45.0, aspect_ratio=16 / 9, aperture=0.0,
                 focus_dist=None, t0=0.0, t1=0.0):
        origin = Vec3(0, 0, -1) if origin is None else Vec3(origin)
        direction = origin.copy() if direction is None else Vec3(direction)
        up = Vec3(0, 1, 0) if up is None else Vec3(up)
        fov, aspect_ratio, aperture = F(fov), F(aspect_ratio), F(aperture)
        focus_dist = length(direction) if focus_dist is None else F(focus_dist)
        theta = fov * (F(np.pi) / F(180))  # deg2rad(::Float32)
        l = np.tan(theta / F(2))
        viewport_width = F(2) * l
        viewport_height = viewport_width / aspect_ratio
        w = normalize(direction)
        u = normalize(cross(up, w))
        v = cross(w, u)
        horizontal = (focus_dist * viewport_width) * u
        vertical = (focus_dist * viewport_height) * v
        self.origin = origin
        self.lower_left_corner = ((origin - horizontal * F(0.5)) - vertical * F(0.5)) - focus_dist * w
        self.horizontal, self.vertical = horizontal, vertical
        self.u, self.v, self.w = u, v, w
        self.lens_radius = aperture * F(0.5)
        self.t0, self.t1 = F(t0), F(t1)


def hfov_from_vfov(vfov, aspect_ratio):
    """rad2deg(atan(tan(deg2rad(vfov)/2)*aspect_ratio)*2) in Float32 (samples/cornell_box.jl:13-15)."""
    vfov, aspect_ratio = F(vfov), F(aspect_ratio)
    d2r = F(np.pi) / F(180)
    half = np.tan((vfov * d2r) / F(2))
    return (np.arctan(half * aspect_ratio) * F(2)) * (F(180) / F(np.pi))


# ---------------------------------------------------------------- sub-pixel patterns (src/MSAA_patterns.jl)
MSAA_S1 = np.array([[0.5, 0.5]]).T
MSAA_S2 = np.array([[0.25, 0.25], [0.75, 0.75]]).T
MSAA_S4 = np.array([[0.375, 0.125], [0.875, 0.375], [0.625, 0.875], [0.125, 0.625]]).T
MSAA_S8 = np.array([[0.5625, 0.6875], [0.4375, 0.3125], [0.8125, 0.4375], [0.3125, 0.8125],
                    [0.1875, 0.1875], [0.0625, 0.5625], [0.6875, 0.0625], [0.9375, 0.9375]]).T
MSAA_S16 = np.array([[0.5625, 0.4375], [0.4375, 0.6875], [0.3125, 0.3750], [0.7500, 0.5625],
                     [0.1875, 0.6250], [0.6250, 0.1875], [0.1875, 0.3125], [0.6875, 0.8125],
                     [0.3750, 0.1250], [0.5000, 0.9375], [0.2500, 0.8750], [0.1250, 0.2500],
                     [0.0000, 0.5000], [0.9375, 0.7500], [0.8750, 0.0625], [0.0625, 0.0000]]).T


def generate_Halton_sequence(N, b):
    """src/MSAA_patterns.jl:25-43 (integer n/d recurrence, T(n)/d in Float32)."""
    seq = np.zeros(N, dtype=F)
    n, d = 0, 1
    for i in range(N):
        x = d - n
        if x == 1:
            n = 1
            d *= b
        else:
            y = d // b
            while x <= y:
                y //= b
            n = (b + 1) * y - x
        seq[i] = F(n) / F(d)
    return seq


def Halton_sequence32(N):
    """src/MSAA_patterns.jl:50-58: 2 x N, row 0 = Halton base 2, row 1 = base 3, each minus 0.5."""
    out = np.zeros((2, N), dtype=F)
    out[0] = generate_Halton_sequence(N, 2) - F(0.5)
    out[1] = generate_Halton_sequence(N, 3) - F(0.5)
    return out
