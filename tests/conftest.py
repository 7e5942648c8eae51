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


def cornell_box_of_meshes():
    """The Cornell box of samples/cornell_box.jl:21-34 as ONE Mesh of 36 triangles and four materials: every rectangle
    and every side of the two boxes as two triangles (the boxes' corners taken through their Transform matrices in
    Float32), the material chosen per triangle by the fourth row of the index matrix (src/mesh.jl:53-59).  Lambertian and
    Emissive shading read only the hit point and the face-flipped geometric normal, and every vertex carries a non-zero
    normal so that the reference's triangle test uses the geometric one (src/triangle.jl:110-113): the picture of this
    world is the picture of the rectangle world, which the reference's gallery holds (outputs/cornell_box_*.png).  That
    puts src/triangle.jl:73-121 and the mesh tree (src/mesh.jl:115-163, 217-305) under a reference-held pin.
    One mesh, because that is what the reference can build: its tree over several meshes numbers the nodes heap-style
    (node * 2, src/mesh.jl:241-242, 255, 276-277) into 4 * triangles - 1 slots (src/mesh.jl:300), and with eight small
    meshes -- or any split of these 36 triangles into two or more -- the ids run past the end of that vector (a BoundsError
    in the reference; rt_bvh_build_meshes answers RT_EUNSUPPORTED).  Returns (world, kwargs) like get_scene."""
    import math

    import numpy as np

    import raytracing_amd as rt
    from raytracing_amd.hittables import Lambertian, Emissive, Material, Mesh, RotY, Vec3, affine
    F = np.float32
    _, kw = rt.get_scene("cornell_box", spp=8)
    red, white = Material(Lambertian, Vec3(0.65, 0.05, 0.05)), Material(Lambertian, Vec3(0.73, 0.73, 0.73))
    green, light = Material(Lambertian, Vec3(0.12, 0.45, 0.15)), Material(Emissive, Vec3(15, 15, 15))
    materials = [green, red, light, white]
    vertices, indices = [], []

    def quad(corners, material, matrix=None):
        c = np.asarray(corners, dtype=F)
        if matrix is not None:
            c = (np.concatenate([c, np.ones((4, 1), F)], axis=1) @ np.asarray(matrix, F).T)[:, :3].astype(F)
        n = np.cross(c[1] - c[0], c[2] - c[0]).astype(F)
        n /= np.linalg.norm(n)
        v = np.zeros((4, 8), F)
        v[:, :3], v[:, 3:6], v[:, 6:] = c, n, [[0, 0], [1, 0], [1, 1], [0, 1]]
        base, m = 4 * len(vertices), materials.index(material)
        vertices.append(v)
        indices.append(np.array([[base, base + 1, base + 2, m], [base, base + 2, base + 3, m]], np.int32))

    def yz(y0, y1, z0, z1, k, m): quad([[k, y0, z0], [k, y1, z0], [k, y1, z1], [k, y0, z1]], m)
    def xz(x0, x1, z0, z1, k, m): quad([[x0, k, z0], [x1, k, z0], [x1, k, z1], [x0, k, z1]], m)
    def xy(x0, x1, y0, y1, k, m): quad([[x0, y0, k], [x1, y0, k], [x1, y1, k], [x0, y1, k]], m)

    def box(p1, matrix):
        x, y, z = p1
        for s in ([[0, 0, z], [x, 0, z], [x, y, z], [0, y, z]], [[0, 0, 0], [x, 0, 0], [x, y, 0], [0, y, 0]],
                  [[0, y, 0], [x, y, 0], [x, y, z], [0, y, z]], [[0, 0, 0], [x, 0, 0], [x, 0, z], [0, 0, z]],
                  [[x, 0, 0], [x, y, 0], [x, y, z], [x, 0, z]], [[0, 0, 0], [0, y, 0], [0, y, z], [0, 0, z]]):
            quad(s, white, matrix)

    yz(0, 555, 0, 555, 555, green)
    yz(0, 555, 0, 555, 0, red)
    xz(213, 343, 227, 332, 554, light)
    xz(0, 555, 0, 555, 0, white)
    xz(0, 555, 0, 555, 555, white)
    xy(0, 555, 0, 555, 555, white)
    box((165, 330, 165), affine(RotY(math.radians(0)), [265, 0, 295]))
    box((165, 165, 165), affine(RotY(math.radians(-18)), [130, 0, 65]))
    return [Mesh(np.concatenate(vertices), materials, np.concatenate(indices))], kw


def blocks_showing_the_tall_box_front(trace_rays, kw, block):
    """Which block x block pixel blocks of the Cornell picture show the front of the tall box (triangles 14 and 15 of
    cornell_box_of_meshes, reported as sub_id), grown by one block: (rows, columns) boolean, row 0 = top.  That face is
    the one place where the box of triangles does NOT look like the box of rectangles in the reference's own arithmetic:
    it is lit by the part of the light that lies in front of its plane, at grazing angles; the triangle test
    (src/triangle.jl:78-84: three rounded products and a reciprocal) puts the camera rays' hit points one ulp of
    |t * d| ~ 1100 (1.2e-4) behind the face where the rectangle test (one division, src/aarect.jl:50) puts them on it,
    and a scattered ray that leaves such a point with |d.z| < 0.12 -- most rays towards the light -- meets the same
    triangle again beyond t_min = 0.001 and ends inside the box.  Measured on the GPU at 4000 samples: that face
    0.49 .. 0.79 of the rectangle world's radiance, every other block of the picture within 0.2 %."""
    import numpy as np
    import oracle_ffi as O
    h, w = kw["height"], kw["width"]
    hits = trace_rays(O.primary_rays(kw["camera"], w, h, w // block, h // block, time=0.5))
    front = np.isin(hits["sub_id"], (14, 15)).reshape(h // block, w // block)[::-1]      # (rows of rays count from the bottom)
    grown = front.copy()
    grown[1:, :] |= front[:-1, :]; grown[:-1, :] |= front[1:, :]; grown[:, 1:] |= front[:, :-1]; grown[:, :-1] |= front[:, 1:]
    return grown
