"""The trees the library builds on the device (accel_device.hip: surface-area heuristic top down, level by level, then a
layout kernel), read back through rt_scene_nodes: they are the host builder's trees (rt_scene_layout) node for node,
in every layout of every node set; they are held to the same invariants (tests/test_accel_layout.py: every primitive once
per layout, links in range, pre-order consistency, enclosure, conservative culling); and the pictures do not depend on
who built the trees."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (first: its HIP runtime has to be the first in the process)

import raytracing_amd as rt
import test_accel_layout as TL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device_layouts():
    cache = {}

    def get(name):
        if name not in cache:
            world, kw = rt.get_scene(name, spp=1)
            dev = rt.DeviceScene(world)
            try:
                info, nodes, prim_top, build = dev.layout()
            finally:
                dev.close()
            assert build.on_device == 1 and build.n_trees >= 1 and build.max_depth >= 1, (build.on_device, build.n_trees)
            print(name, "device build: trees", build.n_trees, "depth", build.max_depth, "ms", round(build.build_ms, 3))
            cache[name] = (world, kw, info, nodes, prim_top)
        return cache[name]
    return get


@pytest.mark.parametrize("name", TL.SCENES)
def test_device_built_trees_structure_and_coverage(name, device_layouts):
    TL.test_structure_and_coverage(name, device_layouts)


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke_box", "cu_cover2", "cover1"])
def test_device_built_trees_never_cull_what_the_reference_visits(name, device_layouts):
    TL.test_walk_never_culls_what_the_reference_visits(name, device_layouts)


@pytest.mark.parametrize("name", TL.SCENES)
def test_device_built_trees_are_the_host_builders(name, device_layouts):
    world, kw, info, nodes, prim_top = device_layouts(name)
    h_info, h_nodes, h_prim_top = TL.scene_layout(world)
    assert (info.nodes_per_layout, info.n_sets, info.surface_root, info.surface_size) == \
        (h_info.nodes_per_layout, h_info.n_sets, h_info.surface_root, h_info.surface_size)
    assert np.array_equal(prim_top, h_prim_top)
    assert np.array_equal(nodes, h_nodes), np.argwhere(nodes != h_nodes)[:8]


@pytest.mark.parametrize("name", TL.SCENES)
def test_device_built_pair_nodes_are_the_host_builders(name):
    """the second form of the tree of surfaces (pair nodes, walked with a stack) as the device writes it and as the host
    builder does"""
    world, kw = rt.get_scene(name, spp=1)
    dev = rt.DeviceScene(world)
    try:
        info = dev.layout()[0]
        pairs, used = dev.pairs()
    finally:
        dev.close()
    h_pairs, depth = TL.scene_layout_pairs(world, info.n_sets)
    assert pairs.shape == h_pairs.shape and np.array_equal(pairs, h_pairs)
    assert used == (pairs.size > 0 and depth <= 20)


@pytest.mark.parametrize("name", ["cu_cover2", "cover1", "teapot"])
def test_pictures_do_not_depend_on_the_walk(name, monkeypatch):
    """the walk over pair nodes with a stack and the stackless walk over single nodes find the same closest hits"""
    world, kw = rt.get_scene(name, spp=4)
    w, h = (kw["width"] // 4, kw["height"] // 4)

    def picture():
        dev = rt.DeviceScene(world)
        try:
            return dev.render(kw["camera"], kw["pattern"], kw["background"], w, h)[0], dev.pairs()[1]
        finally:
            dev.close()
    a, used_a = picture()
    monkeypatch.setenv("RT_AMD_NO_PAIRS", "1")
    b, used_b = picture()
    assert used_a and not used_b
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_far_camera_rebuild_on_the_device():
    """a camera far outside the scene makes the library rebuild the node sets for its distance (boxes of objects taken out
    of nested trees are padded in proportion): with the device builder that is a kernel launch, and it gives what the
    host builder gives for the same reach"""
    world, kw = rt.get_scene("cover2", spp=1)
    origin, target = np.array([30000.0, 20000.0, -40000.0]), np.array([278.0, 278.0, 0.0])
    cam = rt.Camera(origin, origin - target, (0, 1, 0), 1.0, 1.0)
    dev = rt.DeviceScene(world)
    try:
        before = dev.layout()[1]
        img = dev.render(cam, kw["pattern"], kw["background"], 64, 64)[0]
        after = dev.layout()[1]
    finally:
        dev.close()
    assert not np.array_equal(before, after) and np.isfinite(img).all()
    os.environ["RT_AMD_HOST_BUILD"] = "1"
    try:
        host = rt.DeviceScene(world)
        try:
            img_h = host.render(cam, kw["pattern"], kw["background"], 64, 64)[0]
            after_h = host.layout()[1]
        finally:
            host.close()
    finally:
        del os.environ["RT_AMD_HOST_BUILD"]
    assert np.array_equal(after, after_h) and np.array_equal(img.view(np.uint32), img_h.view(np.uint32))


def grid_mesh(n):
    """a height field of 2 n^2 triangles"""
    xs = np.linspace(-1, 1, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    V = np.zeros(((n + 1) ** 2, 8), np.float32)
    V[:, 0], V[:, 1], V[:, 2] = X.ravel(), (0.1 * np.sin(5 * X) * np.cos(7 * Z)).ravel(), Z.ravel()
    V[:, 4] = 1.0
    V[:, 6:] = -1
    idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    z = np.zeros_like(a)
    T = np.concatenate([np.stack([a, b, c, z], 1), np.stack([a, c, d, z], 1)]).astype(np.int32)
    return rt.Mesh(V, [rt.Material(rt.Lambertian, rt.Vec3(0.7, 0.7, 0.7))], T)


@pytest.mark.parametrize("n", [40, 181])
def test_large_mesh_device_build_is_the_host_builders(n):
    """3200 and 65 522 triangles: ranges that the whole workgroup splits over several levels, many centroids on common planes
    (ties), and a picture through each tree"""
    world = [grid_mesh(n)]
    dev = rt.DeviceScene(world)
    try:
        info, nodes, prim_top, build = dev.layout()
        cam = rt.Camera((0.0, 1.5, -3.0), (0.0, 1.5, -3.0), (0, 1, 0), 45.0, 1.0)
        img = dev.render(cam, rt.Halton_sequence32(2), (0.7, 0.8, 1.0), 96, 96)[0]
    finally:
        dev.close()
    assert build.on_device == 1 and info.nodes_per_layout == 4 * n * n - 1
    print(2 * n * n, "triangles: device build", round(build.build_ms, 3), "ms, depth", build.max_depth)
    h_info, h_nodes, _ = TL.scene_layout(world)
    assert np.array_equal(nodes, h_nodes), np.argwhere(nodes != h_nodes)[:8]
    os.environ["RT_AMD_HOST_BUILD"] = "1"
    try:
        host = rt.DeviceScene(world)
        try:
            img_h = host.render(cam, rt.Halton_sequence32(2), (0.7, 0.8, 1.0), 96, 96)[0]
        finally:
            host.close()
    finally:
        del os.environ["RT_AMD_HOST_BUILD"]
    assert np.array_equal(img.view(np.uint32), img_h.view(np.uint32)) and img.std() > 0.01


def test_device_build_is_deterministic():
    world, kw = rt.get_scene("cu_cover2", spp=1)
    a, b = rt.DeviceScene(world), rt.DeviceScene(world)
    try:
        na, nb = a.layout()[1], b.layout()[1]
    finally:
        a.close(); b.close()
    assert np.array_equal(na, nb)


@pytest.mark.parametrize("name", ["cornell_box", "cu_cover2", "cover2", "teapot"])
def test_pictures_do_not_depend_on_the_builder(name, monkeypatch):
    world, kw = rt.get_scene(name, spp=4)
    w, h = (kw["width"] // 4, kw["height"] // 4)

    def picture():
        dev = rt.DeviceScene(world)
        try:
            return dev.render(kw["camera"], kw["pattern"], kw["background"], w, h)[0], dev.layout()[3].on_device
        finally:
            dev.close()
    img_dev, on_dev = picture()
    monkeypatch.setenv("RT_AMD_HOST_BUILD", "1")
    img_host, on_dev2 = picture()
    assert on_dev == 1 and on_dev2 == 0
    assert np.array_equal(img_dev.view(np.uint32), img_host.view(np.uint32))


@pytest.mark.parametrize("ratio,count,walk", [(8.0, 28, "pairs"), (4.0, 40, "pairs"), (4.0, 48, "stackless")])
def test_deep_trees_walk_the_pair_nodes_with_the_deep_stack(ratio, count, walk):
    """Spheres whose sizes and distances grow geometrically make the surface-area heuristic split one off at a time: a
    tree of depth 23 / 29 / beyond 32 from a few dozen primitives.  Up to 20 levels the walk over pair nodes keeps its
    stack in 20 KB of LDS a workgroup, up to 32 in 32 KB (four workgroups per CU instead of six), beyond that the
    stackless walk takes over; whichever walks, hits and pictures are the oracle's."""
    import oracle_ffi as O
    m = rt.Material(rt.Lambertian, rt.Vec3(0.6, 0.6, 0.6))
    world = [rt.Sphere(rt.Vec3(ratio ** k, 0.0, 0.0), 0.2 * ratio ** k, m) for k in range(count)]
    dev, ref = rt.DeviceScene(world), O.OracleScene(world)
    try:
        build = dev.layout()[3]
        in_use = dev.pairs()[1]
        assert (build.max_depth > 20) and in_use == (walk == "pairs") and (build.max_depth <= 32) == (walk == "pairs"), (build.max_depth, in_use)
        rng = np.random.default_rng(5)
        k = rng.integers(0, count, 4096)                # every ray aims near one of the spheres, from a few radii away
        centre = np.stack([ratio ** k.astype(np.float64), np.zeros(4096), np.zeros(4096)], axis=1)
        radius = (0.2 * ratio ** k.astype(np.float64))[:, None]
        away = rng.normal(size=(4096, 3))
        away /= np.linalg.norm(away, axis=1, keepdims=True)
        o = centre + away * radius * rng.uniform(1.5, 30.0, (4096, 1))
        d = centre + rng.normal(size=(4096, 3)) * radius * 0.6 - o
        rays = np.concatenate([o, d, np.full((4096, 1), 0.5)], axis=1).astype(np.float32)
        a, b = ref.trace_rays(rays), dev.trace_rays(rays)
        for k in ("prim_id", "sub_id", "front"):
            np.testing.assert_array_equal(a[k], b[k])
        for k in ("t", "normal", "uv"):
            np.testing.assert_array_equal(np.ascontiguousarray(a[k]).view(np.uint32), np.ascontiguousarray(b[k]).view(np.uint32))
        assert (a["prim_id"] >= 0).sum() > 500 and len(set(a["prim_id"])) > 12
        origin = rt.Vec3(-3.0, 1.0, -6.0)
        cam = rt.Camera(origin, origin - rt.Vec3(20.0, 0.0, 0.0), rt.Vec3(0, 1, 0), 70.0, 1.0, 0.0, 10.0, 0.0, 1.0)
        pat, bg = rt.Halton_sequence32(6), rt.Vec3(0.7, 0.8, 1.0)
        img, _ = dev.render(cam, pat, bg, 96, 96)            # (through rt_extend_kernel: the variant with the deep stack)
        np.testing.assert_array_equal(img.view(np.uint32), ref.render(cam, pat, bg, 96, 96).view(np.uint32))
    finally:
        dev.close()
