"""The device acceleration structure (own SAH trees, eight ray-octant layouts per node set, 16-byte half-precision
nodes; one set of tight boxes for rays that start inside the scene's ball, one of the reference's boxes for the others)
checked on the host through rt_scene_layout: structure, coverage of the primitives, box enclosure, and that the stackless walk
visits every primitive whose reference bounding box (src/*.jl bounding_box) the ray passes -- the walk may only cull
what the reference's exact slab test (src/bounding_box.jl:20-67) culls too.  The same for the second form of the tree of
surfaces, the pair nodes that rt_extend_kernel walks with a stack (rt_scene_layout_pairs).  No GPU needed."""
import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import hittables as H
from raytracing_amd.render import scene_layout, scene_layout_pairs

LEAF = 0x80000000
PRIM_MASK = 0x0FFFFFFF
SCENES = ["cornell_box", "cornell_smoke_box", "cover1", "cu_cover2", "cover2", "teapot"]


def boxes_of(nodes, octant=0):
    """(..., 6) float32 {min.xyz, max.xyz} of the stored half-precision boxes of the layout of `octant`: word a holds
    the two planes of axis a, the one a ray of that octant meets first in the low half"""
    h = nodes[..., :3].copy().view(np.float16).astype(np.float32)      # (.., 6): x0 x1 y0 y1 z0 z1 in stored order
    first, second = h[..., 0::2], h[..., 1::2]
    neg = np.array([(octant >> a) & 1 for a in range(3)], bool)
    lo = np.where(neg, second, first)
    hi = np.where(neg, first, second)
    return np.concatenate([lo, hi], axis=-1)


def subtree_end(link, i):
    return i + 1 if link[i] & LEAF else int(link[i])


@pytest.fixture(scope="module")
def layouts():
    cache = {}

    def get(name):
        if name not in cache:
            world, kw = rt.get_scene(name, spp=1)
            cache[name] = (world, kw) + scene_layout(world)
        return cache[name]
    return get


@pytest.mark.parametrize("name", SCENES)
def test_structure_and_coverage(name, layouts):
    world, kw, info, nodes, prim_top = layouts(name)
    N = info.nodes_per_layout
    assert info.n_sets in (1, 2) and nodes.shape == (8 * info.n_sets, N, 4) and N > 0 and info.n_prims == len(prim_top)
    assert (info.n_sets == 2) == bool(np.isfinite(info.radius)) and info.radius > 0
    scale = info.node_scale
    assert scale > 0 and np.log2(scale) == np.round(np.log2(scale))          # a power of two
    ref = None
    leaf_boxes = {}
    for layout in range(8 * info.n_sets):
        o = layout % 8
        if o == 0:
            ref = None
        link = nodes[layout, :, 3].astype(np.int64)
        box = boxes_of(nodes[layout], o)
        base = layout * N
        # every surface primitive is the leaf of exactly one node of the layout (the constant media are kept as a list:
        # the kernels try every medium for every ray)
        leaves = link[(link & LEAF) != 0]
        prims = (leaves & PRIM_MASK) - 1
        surfaces = np.array([k for k in range(info.n_prims) if not isinstance(world[prim_top[k]], H.ConstantMedium)])
        assert np.array_equal(np.sort(prims), surfaces) and info.medium_size == 0
        # skip links stay inside the layout and point forward
        inter = np.nonzero((link & LEAF) == 0)[0]
        assert np.all(link[inter] > inter + base) and np.all(link[inter] <= base + N)
        # pre-order consistency and enclosure of the two main trees: an interior node is followed by its first child,
        # whose subtree ends where the second child starts, whose subtree ends at the node's skip link
        rel = np.where((link & LEAF) != 0, link, link - base)
        for root, size in ((info.surface_root, info.surface_size), (info.medium_root, info.medium_size)):
            if size == 0:
                continue
            stack = [root]
            seen = 0
            while stack:
                i = stack.pop()
                seen += 1
                if rel[i] & LEAF:
                    continue
                c1 = i + 1
                c2 = subtree_end(rel, c1)
                assert subtree_end(rel, c2) == rel[i]
                for c in (c1, c2):
                    assert np.all(box[c, :3] >= box[i, :3]) and np.all(box[c, 3:] <= box[i, 3:])
                stack += [c1, c2]
            assert seen == size and subtree_end(rel, root) == root + size
        if o == 0:
            is_leaf = (link & LEAF) != 0
            leaf_boxes[layout // 8] = box[is_leaf][np.argsort((link[is_leaf] & PRIM_MASK) - 1)]
        # the eight layouts of a set hold the same nodes (boxes, leaf words), only in different orders
        key = np.concatenate([box.view(np.uint32), np.where((link & LEAF) != 0, link, 0).astype(np.uint32)[:, None]], axis=1)
        key = key[np.lexsort(key.T[::-1])]
        if ref is None:
            ref = key
        else:
            assert np.array_equal(ref, key)
    # the tight boxes of set 0 lie inside the boxes of set 1 (up to the outward rounding of both)
    if info.n_sets == 2:
        a, b = leaf_boxes[0], leaf_boxes[1]
        assert np.all(a[:, :3] >= b[:, :3]) and np.all(a[:, 3:] <= b[:, 3:]) and np.any(a != b)
    # layouts of opposite octants visit the children of the root in opposite order
    if info.surface_size > 1:
        r = info.surface_root
        rel0 = nodes[0, :, 3].astype(np.int64)
        rel7 = np.where((nodes[7, :, 3] & LEAF) != 0, nodes[7, :, 3].astype(np.int64), nodes[7, :, 3].astype(np.int64) - 7 * N)
        a, b = boxes_of(nodes[0], 0), boxes_of(nodes[7], 7)
        assert np.array_equal(a[r + 1], b[subtree_end(rel7, r + 1)])
        assert np.array_equal(a[subtree_end(rel0, r + 1)], b[r + 1])


def walk(nodes, info, octant, root, size, o, d, t_min=0.001):
    """the stackless walk of rt_kernels.hip for one ray, in float32 like the kernel (exact division instead of the
    hardware reciprocal): the layout of the ray's octant stores every axis as (near plane, far plane), so the entry
    distance is the largest of the first halves and the exit distance the smallest of the second halves; a plane
    distance is plane * (1/d) - origin * (1/d) fused, with the rounding slack of the second product moved outward.  Returns the
    list of leaf words visited with the box passed."""
    N = info.nodes_per_layout
    F = np.float32
    scale = F(info.node_scale)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1) / d).astype(F) * (F(1) / scale)
        oi = ((o * scale).astype(F) * inv).astype(F)
        slack = (np.abs(oi) * F(2.0 ** -22)).astype(F)
        cn, cf = (-oi - slack).astype(F), (-oi + slack).astype(F)
    # node set: the tight one for rays that start inside the ball, else the last one (node_set() of rt_kernels.hip)
    with np.errstate(invalid="ignore", over="ignore"):
        rel = (o - np.asarray(info.centre, F)).astype(F)
        inside = F(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) < F(info.radius) * F(info.radius)
    layout = (0 if inside else info.n_sets - 1) * 8 + octant
    link = nodes[layout, :, 3]
    planes = nodes[layout, :, :3].copy().view(np.float16).astype(np.float64)     # x near, x far, y near, y far, z near, z far
    node = layout * N + root
    end = node + size
    out = []
    while node != end:
        i = node - layout * N
        with np.errstate(invalid="ignore", over="ignore"):
            # plane * inv + c, fused (one rounding): the product of a half and a float is exact in float64
            t0 = (planes[i, 0::2] * inv.astype(np.float64) + cn.astype(np.float64)).astype(F)
            t1 = (planes[i, 1::2] * inv.astype(np.float64) + cf.astype(np.float64)).astype(F)
        tnear = max(np.fmax.reduce(t0), F(t_min))
        tfar = np.fmin.reduce(t1)
        with np.errstate(invalid="ignore"):
            tfar = tfar + abs(tfar) * F(4.8e-7)
        hit = tnear <= tfar
        w = int(link[i])
        if w & LEAF:
            if hit:
                out.append(w & ~LEAF)
            node += 1
        else:
            node = node + 1 if hit else w
    return out


def walk_pairs(pairs, info, octant, o, d, t_min=0.001, stack_limit=20):
    """the walk of rt_extend_kernel over the pair nodes, with its stack: the same box test as walk(), on the two child boxes
    of a record; a hit leaf is reported, a hit interior child is entered (the near one first, the far one waits on the
    stack).  Returns (leaf words visited, deepest stack)."""
    F = np.float32
    scale = F(info.node_scale)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1) / d).astype(F) * (F(1) / scale)
        oi = ((o * scale).astype(F) * inv).astype(F)
        slack = (np.abs(oi) * F(2.0 ** -22)).astype(F)
        cn, cf = (-oi - slack).astype(F), (-oi + slack).astype(F)
        rel = (o - np.asarray(info.centre, F)).astype(F)
        inside = F(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) < F(info.radius) * F(info.radius)
    layout = (0 if inside else info.n_sets - 1) * 8 + octant
    P = pairs.shape[1]
    rec = pairs[layout]

    def hit(words):
        planes = words.copy().view(np.float16).astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            t0 = (planes[0::2] * inv.astype(np.float64) + cn.astype(np.float64)).astype(F)
            t1 = (planes[1::2] * inv.astype(np.float64) + cf.astype(np.float64)).astype(F)
            tnear = max(np.fmax.reduce(t0), F(t_min))
            tfar = np.fmin.reduce(t1)
            tfar = tfar + abs(tfar) * F(4.8e-7)
        return tnear <= tfar
    out, stack, deepest = [], [], 0
    cur = layout * P
    while cur is not None:
        r = rec[cur - layout * P]
        go = []
        for words, w in ((r[0:3], int(r[3])), (r[4:7], int(r[7]))):
            if hit(words):
                if w & LEAF:
                    out.append(w & ~LEAF)
                else:
                    assert layout * P <= w < (layout + 1) * P
                    go.append(w)
        if len(go) == 2:
            stack.append(go[1])
            deepest = max(deepest, len(stack))
        cur = go[0] if go else (stack.pop() if stack else None)
    assert deepest <= stack_limit
    return out, deepest


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke_box", "cu_cover2", "cover1", "teapot", "cover2"])
def test_pair_nodes_structure(name, layouts):
    """every record is reached exactly once from the first, every surface primitive is the leaf of exactly one child, and
    the boxes are those of the 16-byte nodes (the same trees written down twice)"""
    world, kw, info, nodes, prim_top = layouts(name)
    pairs, depth = scene_layout_pairs(world, info.n_sets)
    if info.nested:
        assert pairs.size == 0
        return
    n_surfaces = sum(not isinstance(world[prim_top[k]], H.ConstantMedium) for k in range(info.n_prims))
    P = max(n_surfaces - 1, 1)
    assert pairs.shape == (8 * info.n_sets, P, 8) and 1 <= depth <= 40
    for layout in range(8 * info.n_sets):
        rec = pairs[layout]
        seen = np.zeros(P, bool)
        leaves = []
        stack = [layout * P]
        while stack:
            i = stack.pop() - layout * P
            assert 0 <= i < P and not seen[i]
            seen[i] = True
            for w in (int(rec[i, 3]), int(rec[i, 7])):
                if w & LEAF:
                    leaves.append((w & PRIM_MASK) - 1)
                else:
                    stack.append(w)
        assert seen.all()
        if n_surfaces > 1:
            assert sorted(leaves) == sorted(k for k in range(info.n_prims) if not isinstance(world[prim_top[k]], H.ConstantMedium))
        # the child boxes are node boxes of the same layout
        node_boxes = {tuple(b) for b in nodes[layout, :, :3].tolist()}
        assert all(tuple(b) in node_boxes for b in rec[:, 0:3].tolist()) and all(tuple(b) in node_boxes for b in rec[:, 4:7].tolist())


def slab_exact(bmin, bmax, o, d, t_min=0.001):
    """src/bounding_box.jl:20-67 in float64"""
    tmin, tmax = float(t_min), np.inf
    for a in range(3):
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.float64(1.0) / np.float64(d[a])
            t0 = (np.float64(bmin[a]) - np.float64(o[a])) * inv
            t1 = (np.float64(bmax[a]) - np.float64(o[a])) * inv
        if inv < 0.0:
            t0, t1 = t1, t0
        tmin = t0 if t0 > tmin else tmin
        tmax = t1 if t1 < tmax else tmax
        if tmax <= tmin:
            return False
    return True


def must_visit(obj, o, d, tight, t_min=0.001):
    """True if the reference can report a hit of `obj` for the ray: the ray passes the reference's bounding box of the
    object (exact slab test) and, for a ray that walks the set of tight boxes, also meets the shape itself (in float64,
    with the ray taken into the object's frame like src/transform.jl:104-105)."""
    b = H.bounding_box(obj)
    if not slab_exact(b.min, b.max, o, d, t_min):
        return False
    if not tight:
        return True
    shape, oo, dd = obj, o.astype(np.float64), d.astype(np.float64)
    if isinstance(shape, H.Transform):
        T = np.asarray(shape.inv_xform, np.float64).reshape(4, 3).T       # column-major 3x4
        oo, dd = T[:, :3] @ oo + T[:, 3], T[:, :3] @ dd
        shape = shape.object
    if isinstance(shape, H.Sphere):
        oc = oo - np.asarray(shape.center, np.float64)
        a, hb, c = dd @ dd, oc @ dd, oc @ oc - float(shape.radius) ** 2
        disc = hb * hb - a * c
        if disc < 0:
            return False
        return any(t_min <= t for t in ((-hb - np.sqrt(disc)) / a, (-hb + np.sqrt(disc)) / a))
    if isinstance(shape, H.Box):
        return slab_exact(np.asarray(shape.box_min), np.asarray(shape.box_max), oo, dd, t_min)
    return True


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke_box", "cu_cover2", "cover1"])
def test_walk_never_culls_what_the_reference_visits(name, layouts):
    world, kw, info, nodes, prim_top = layouts(name)
    pairs = layouts.pairs(name) if hasattr(layouts, "pairs") else scene_layout_pairs(world, info.n_sets)[0]
    rng = np.random.default_rng(7)
    cam = kw["camera"]
    boxes = [H.bounding_box(obj) for obj in world]
    lo = np.array([b.min for b in boxes], np.float32)
    hi = np.array([b.max for b in boxes], np.float32)
    finite = np.isfinite(lo).all(axis=1) & np.isfinite(hi).all(axis=1) & (np.abs(hi - lo).max(axis=1) < 3000)
    centre = 0.5 * (lo[finite].min(axis=0) + hi[finite].max(axis=0))
    extent = (hi[finite].max(axis=0) - lo[finite].min(axis=0)).max()
    n_rays = 64
    n_must = n_far = 0
    # rays that run inside a face of an object's reference box (origin on the plane, direction parallel to it): the
    # reference's slab test ignores the 0 * Inf = NaN of that axis and goes on to the primitive
    special = []
    for idx in rng.choice(np.nonzero(finite)[0], size=min(12, int(finite.sum())), replace=False):
        c = 0.5 * (lo[idx] + hi[idx])
        for axis in range(3):
            for plane in (lo[idx], hi[idx]):
                o = c.copy()
                o[axis] = plane[axis]
                o[(axis + 1) % 3] = lo[idx][(axis + 1) % 3] - np.float32(3.0)
                d = np.zeros(3, np.float32)
                d[(axis + 1) % 3] = 1.0
                special.append((o.astype(np.float32), d))
    for k in range(n_rays + len(special)):
        if k >= n_rays:
            o, d = special[k - n_rays]
            target = None
        elif k % 2 == 0:   # from the camera towards the scene
            o = np.asarray(cam.origin, np.float32)
            target = centre + rng.uniform(-0.5, 0.5, 3) * extent
        elif k % 16 == 5:  # from far outside (beyond the ball of the tight set, if there is one) towards the scene
            o = (centre + rng.choice([-1.0, 1.0], 3) * rng.uniform(4.0, 9.0, 3) * extent).astype(np.float32)
            target = centre + rng.uniform(-0.3, 0.3, 3) * extent
        else:            # between two points inside the scene bounds (secondary rays), incl. axis-parallel ones
            o = (centre + rng.uniform(-0.5, 0.5, 3) * extent).astype(np.float32)
            target = centre + rng.uniform(-0.5, 0.5, 3) * extent
            if k % 8 == 1:
                target[rng.integers(3)] = o[rng.integers(3)]
        if target is not None:
            d = (target - o).astype(np.float32)
            if k % 16 == 3:
                d[rng.integers(3)] = 0.0
        octant = int(np.signbit(d[0])) | (int(np.signbit(d[1])) << 1) | (int(np.signbit(d[2])) << 2)
        visited = set()
        for root, size in ((info.surface_root, info.surface_size), (info.medium_root, info.medium_size)):
            if size:
                visited |= {prim_top[(w & PRIM_MASK) - 1] for w in walk(nodes, info, octant, root, size, o, d)}
        if pairs is not None and pairs.size:
            by_pairs = {prim_top[(w & PRIM_MASK) - 1] for w in walk_pairs(pairs, info, octant, o, d)[0]}
            assert by_pairs == visited, (name, k)     # the same boxes, tested at the parent instead of at the node
        rel = o.astype(np.float64) - np.asarray(info.centre, np.float64)
        tight = info.n_sets == 2 and rel @ rel < 0.99 * float(info.radius) ** 2
        if info.n_sets == 2 and not tight and rel @ rel < 1.01 * float(info.radius) ** 2:
            continue  # too close to the ball's surface to say which set the kernel's float32 test picks
        n_far += 0 if tight else 1
        for idx in range(len(world)):
            if isinstance(world[idx], H.ConstantMedium):
                continue
            if must_visit(world[idx], o, d, tight):
                n_must += 1
                assert idx in visited, (name, k, idx, type(world[idx]).__name__)
    assert n_must > n_rays  # the rays do meet the scene
    assert info.n_sets == 1 or 0 < n_far < n_rays  # ... and both node sets were walked


def test_nested_tree_inlined_or_kept(monkeypatch):
    world, kw = rt.get_scene("cover2", spp=1)
    info, nodes, prim_top = scene_layout(world)
    assert info.nested == 0                              # the 1000 spheres of the Transform{BVHTree} sit in the main tree
    monkeypatch.setenv("RT_AMD_INLINE_GROUPS", "0")
    info2, nodes2, prim_top2 = scene_layout(world)
    assert info2.nested == 1 and info2.n_prims == info.n_prims + 1      # + the instance primitive
    assert info2.surface_size < info.surface_size
    assert sorted(set(prim_top)) == sorted(set(prim_top2))
    # rotated spheres: tight boxes in the first set, valid inside a ball around them; the nested tree itself is walked in
    # its own frame, where a sphere's box is the reference's and its place in the world is only known to lie within the
    # scene: no ball in which its small spheres' leaves could stay small, one set
    assert info.n_sets == 2 and 0 < info.radius < 1e5 and info2.n_sets == 1
