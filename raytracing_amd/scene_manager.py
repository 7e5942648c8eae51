"""Host-side mirror of src/scene_manager.jl, src/BVH.jl (tree containers) and of the
flattening step that src/cuscene_manager.jl / src/cuBVH.jl perform with Adapt/cu():
objects are grouped by concrete type, one BVH is built per group, and the hittable
tree is flattened into the plain records of include/rt_amd.h.
"""
import ctypes as C

import numpy as np

from . import _ffi as ffi
from . import hittables as H



# --------------------------------------------------------------------- type keys
def _tex_key(tex):
    if isinstance(tex, H.SolidColor):
        return "SolidColor"
    if isinstance(tex, H.CheckerTexture):
        return f"CheckerTexture<{_tex_key(tex.odd)},{_tex_key(tex.even)}>"
    if isinstance(tex, H.ImageTexture):
        return "ImageTexture"
    if isinstance(tex, H.NoiseTexture):
        return "NoiseTexture<Vec3>" if tex.vector else "NoiseTexture<F>"
    raise TypeError(type(tex).__name__)


def type_key(obj):
    """The concrete Julia type of an object, i.e. what unique!(typeof.(objects)) distinguishes
    (src/scene_manager.jl:10): shape, material texture type and wrapped types."""
    if isinstance(obj, H.Transform):
        return f"Transform<{type_key(obj.object)}>"
    if isinstance(obj, H.ConstantMedium):
        return f"ConstantMedium<{_tex_key(obj.phase_function.albedo)},{type_key(obj.boundary)}>"
    if isinstance(obj, BVHTree):
        keys = {type_key(o) for o in obj.objects}
        return f"BVHTree<{keys.pop() if len(keys) == 1 else 'Hittable'}>"
    if isinstance(obj, H.Mesh):
        return f"Mesh<{_tex_key(obj.materials[0].albedo)}>"
    return f"{type(obj).__name__}<{_tex_key(obj.material.albedo)}>"


def SceneManager(objects):
    """src/scene_manager.jl:9-19: a tuple of homogeneous vectors, one per concrete type, in order of
    first appearance."""
    fields, order = {}, []
    for obj in objects:
        k = type_key(obj)
        if k not in fields:
            fields[k] = []
            order.append(k)
        fields[k].append(obj)
    return tuple(fields[k] for k in order)


# --------------------------------------------------------------------- trees
class BVHTree(H.Hittable):
    """src/BVH.jl:12-16.  kind 0: generic tree, kind 1: tree over meshes (src/mesh.jl).  Node ids are 1-based
    as in the reference; row i-1 of the arrays describes node i."""

    def __init__(self, kind, objects, original, node_bounds, node_left, node_right, node_leaf, meshes=None):
        self.kind = kind
        self.objects = objects          # tree order (the build permutes tree.objects)
        self.original = original        # insertion order
        self.node_bounds = node_bounds  # (slots, 6)
        self.node_left, self.node_right, self.node_leaf = node_left, node_right, node_leaf
        self.mesh_data = meshes or {}   # id(mesh) -> (permuted indices, tri_ids)


def build_bvh(objects):
    """build_bvh(objects::Vector{<:Hittable}) (src/BVH.jl:98-109) and build_bvh(meshes::Vector{<:Mesh})
    (src/mesh.jl:294-305), executed by the library's restated builders."""
    objects = list(objects)
    rec = ffi.records(H.F)  # the element type picks the builder: rt_bvh_build or rt_bvh_build_f64
    F = rec.np_type
    n = len(objects)
    if n == 0:
        raise ValueError("build_bvh: empty object list")
    if all(isinstance(o, H.Mesh) for o in objects):
        total = sum(o.indices.shape[0] for o in objects)
        slots = 4 * total - 1
        nv = np.array([o.vertices.shape[0] for o in objects], dtype=np.int32)
        nt = np.array([o.indices.shape[0] for o in objects], dtype=np.int32)
        idx = [o.indices.copy() for o in objects]
        tri = [np.zeros(o.indices.shape[0], dtype=np.int32) for o in objects]
        verts = [np.ascontiguousarray(o.vertices, dtype=F) for o in objects]
        vptr = (rec.real_p * n)(*[rec.ptr(v) for v in verts])
        iptr = (ffi.c_i32p * n)(*[ffi.iptr(a) for a in idx])
        tptr = (ffi.c_i32p * n)(*[ffi.iptr(a) for a in tri])
        order = np.zeros(n, dtype=np.int32)
        nb = np.zeros((slots, 6), dtype=F)
        nl = np.zeros(slots, dtype=np.int32)
        nr = np.zeros(slots, dtype=np.int32)
        lf = np.zeros(slots, dtype=np.uint8)
        ffi.check(rec.fn("rt_bvh_build_meshes")(n, ffi.iptr(nv), vptr, ffi.iptr(nt), iptr, tptr, ffi.iptr(order),
                                                rec.ptr(nb), ffi.iptr(nl), ffi.iptr(nr), ffi.bptr(lf)))
        md = {id(o): (idx[i], tri[i]) for i, o in enumerate(objects)}
        return BVHTree(1, [objects[i] for i in order], objects, nb, nl, nr, lf, md)
    bounds = np.stack([H.bounding_box(o).as6() for o in objects]).astype(F)
    order = np.zeros(n, dtype=np.int32)
    slots = 2 * n - 1
    nb = np.zeros((slots, 6), dtype=F)
    nl = np.zeros(slots, dtype=np.int32)
    nr = np.zeros(slots, dtype=np.int32)
    lf = np.zeros(slots, dtype=np.uint8)
    ffi.check(rec.fn("rt_bvh_build")(n, rec.ptr(bounds), ffi.iptr(order), rec.ptr(nb), ffi.iptr(nl), ffi.iptr(nr),
                                     ffi.bptr(lf)))
    return BVHTree(0, [objects[i] for i in order], objects, nb, nl, nr, lf)


def build_bvh_scene(scene_manager):
    """build_bvh(scene_manager::SceneManager) = map(build_bvh_scene, ...) (src/scene_manager.jl:25-38)"""
    return tuple(build_bvh(bucket) for bucket in scene_manager)


# --------------------------------------------------------------------- flattening
class Flattener:
    """Collects the record tables of include/rt_amd.h from a hittable tree.  Used for the device scene
    description (trees included) and, by the tests, for the plain world list given to the checker."""

    def __init__(self, rec=None):
        self.rec = rec or ffi.records(H.F)  # record types of the element type (float or double fields)
        self.objects, self.obj_index = [], {}
        self.transforms, self.materials, self.mat_index = [], [], {}
        self.textures, self.tex_index = [], {}
        self.images, self.perlins, self.perlin_index = [], [], {}
        self.meshes, self.mesh_index = [], {}
        self.groups, self.group_index = [], {}
        self.keep = []

    # textures / materials are deduplicated by identity
    def texture(self, tex):
        if id(tex) in self.tex_index:
            return self.tex_index[id(tex)]
        t = self.rec.rtTexture()
        if isinstance(tex, H.SolidColor):
            t.kind = ffi.RT_TEX_SOLID
            t.color[:] = [float(x) for x in tex.color]
        elif isinstance(tex, H.CheckerTexture):
            t.kind = ffi.RT_TEX_CHECKER
            t.a, t.b = self.texture(tex.odd), self.texture(tex.even)
        elif isinstance(tex, H.ImageTexture):
            t.kind = ffi.RT_TEX_IMAGE
            im = ffi.rtImage()
            im.height, im.width = tex.image.shape[0], tex.image.shape[1]
            im.rgb = ffi.bptr(tex.image)
            self.keep.append(tex.image)
            self.images.append(im)
            t.a = len(self.images) - 1
        elif isinstance(tex, H.NoiseTexture):
            t.kind = ffi.RT_TEX_NOISE_MARBLE if tex.vector else ffi.RT_TEX_NOISE_SCALAR
            t.scale = tex.scale
            if id(tex.noise) not in self.perlin_index:
                p = self.rec.rtPerlin()
                p.ranvec[:] = [float(x) for x in tex.noise.random.reshape(-1)]
                p.perm_x[:] = tex.noise.perm_x
                p.perm_y[:] = tex.noise.perm_y
                p.perm_z[:] = tex.noise.perm_z
                self.perlins.append(p)
                self.perlin_index[id(tex.noise)] = len(self.perlins) - 1
            t.a = self.perlin_index[id(tex.noise)]
        else:
            raise TypeError(type(tex).__name__)
        self.textures.append(t)
        self.tex_index[id(tex)] = len(self.textures) - 1
        self.keep.append(tex)
        return self.tex_index[id(tex)]

    def material(self, mat):
        if id(mat) in self.mat_index:
            return self.mat_index[id(mat)]
        m = self.rec.rtMaterial(mat.type, self.texture(mat.albedo), float(mat.fuzz), float(mat.eta))
        self.materials.append(m)
        self.mat_index[id(mat)] = len(self.materials) - 1
        self.keep.append(mat)
        return self.mat_index[id(mat)]

    def obj(self, o, with_trees):
        if id(o) in self.obj_index:
            return self.obj_index[id(o)]
        r = self.rec.rtObject()
        r.material, r.child, r.aux = -1, -1, -1
        if isinstance(o, H.Sphere):
            r.kind, r.material = ffi.RT_SPHERE, self.material(o.material)
            r.p[0:4] = [*map(float, o.center), float(o.radius)]
        elif isinstance(o, H.MovingSphere):
            r.kind, r.material = ffi.RT_MOVING_SPHERE, self.material(o.material)
            r.p[0:9] = [*map(float, o.center0), float(o.t0), *map(float, o.center1), float(o.t1), float(o.radius)]
        elif isinstance(o, H._Rect):
            r.kind = {H.xy_rect: ffi.RT_XY_RECT, H.xz_rect: ffi.RT_XZ_RECT, H.yz_rect: ffi.RT_YZ_RECT}[type(o)]
            r.material = self.material(o.material)
            r.p[0:5] = [float(o.a0), float(o.a1), float(o.b0), float(o.b1), float(o.k)]
        elif isinstance(o, H.Box):
            r.kind, r.material = ffi.RT_BOX, self.material(o.material)
            r.p[0:6] = [*map(float, o.box_min), *map(float, o.box_max)]
        elif isinstance(o, H.Transform):
            r.kind = ffi.RT_TRANSFORM
            r.child = self.obj(o.object, with_trees)
            t = self.rec.rtTransform()
            t.xform[:] = [float(x) for x in o.xform]
            t.inv_xform[:] = [float(x) for x in o.inv_xform]
            t.bbox[:] = [float(x) for x in o.bbox.as6()]
            self.transforms.append(t)
            r.aux = len(self.transforms) - 1
        elif isinstance(o, H.ConstantMedium):
            r.kind = ffi.RT_CONSTANT_MEDIUM
            r.child = self.obj(o.boundary, with_trees)
            r.material = self.material(o.phase_function)
            r.p[0] = float(o.neg_inv_density)
        elif isinstance(o, BVHTree):
            r.kind = ffi.RT_BVH
            # children are registered in insertion order so that object indices do not depend on the build
            for c in o.original:
                self.obj(c, with_trees)
            self.groups.append(o)
            r.aux = len(self.groups) - 1
        elif isinstance(o, H.Mesh):
            r.kind = ffi.RT_MESH
            self.meshes.append(o)
            self.mesh_index[id(o)] = len(self.meshes) - 1
            r.aux = len(self.meshes) - 1
            for m in o.materials:
                self.material(m)
        else:
            raise TypeError(f"unsupported hittable {type(o).__name__}")
        self.objects.append(r)
        self.obj_index[id(o)] = len(self.objects) - 1
        self.keep.append(o)
        return self.obj_index[id(o)]

    # ---- table export
    def _arr(self, ctype, items):
        a = (ctype * max(1, len(items)))(*items)
        self.keep.append(a)
        return a

    def _bvh_record(self, tree):
        b = self.rec.rtBvh()
        b.kind = tree.kind
        objs = np.array([self.obj_index[id(o)] for o in tree.objects], dtype=np.int32)
        self.keep += [objs, tree]
        b.n_objects, b.objects = len(objs), ffi.iptr(objs)
        b.n_nodes = tree.node_bounds.shape[0]
        bounds = np.ascontiguousarray(tree.node_bounds, dtype=self.rec.np_type)
        self.keep.append(bounds)
        b.node_bounds = self.rec.ptr(bounds)
        b.node_left, b.node_right = ffi.iptr(tree.node_left), ffi.iptr(tree.node_right)
        b.node_leaf = ffi.bptr(tree.node_leaf)
        return b

    def _mesh_record(self, mesh, sorted_data):
        m = self.rec.rtMesh()
        m.n_vertices, m.n_triangles, m.n_materials = mesh.vertices.shape[0], mesh.indices.shape[0], len(mesh.materials)
        mats = np.array([self.material(x) for x in mesh.materials], dtype=np.int32)
        idx, tri = sorted_data if sorted_data is not None else (mesh.indices, None)
        verts = np.ascontiguousarray(mesh.vertices, dtype=self.rec.np_type)
        self.keep += [mats, idx, tri, verts]
        m.vertices, m.indices, m.materials = self.rec.ptr(verts), ffi.iptr(idx), ffi.iptr(mats)
        m.tri_ids = ffi.iptr(tri) if tri is not None else None
        return m

    def common_tables(self, desc, mesh_sorted=None):
        mesh_sorted = mesh_sorted or {}
        desc.n_objects, desc.objects = len(self.objects), self._arr(self.rec.rtObject, self.objects)
        desc.n_transforms, desc.transforms = len(self.transforms), self._arr(self.rec.rtTransform, self.transforms)
        mesh_records = [self._mesh_record(m, mesh_sorted.get(id(m))) for m in self.meshes]
        desc.n_meshes, desc.meshes = len(mesh_records), self._arr(self.rec.rtMesh, mesh_records)
        desc.n_materials, desc.materials = len(self.materials), self._arr(self.rec.rtMaterial, self.materials)
        desc.n_textures, desc.textures = len(self.textures), self._arr(self.rec.rtTexture, self.textures)
        desc.n_images, desc.images = len(self.images), self._arr(ffi.rtImage, self.images)
        desc.n_perlins, desc.perlins = len(self.perlins), self._arr(self.rec.rtPerlin, self.perlins)


def flatten_scene(world, bvh):
    """The device scene description: `world` is the render() vector (defines prim ids), `bvh` the tuple of
    per-type trees built from SceneManager(world)."""
    fl = Flattener()
    top = np.array([fl.obj(o, True) for o in world], dtype=np.int32)
    desc = fl.rec.rtSceneDesc()
    mesh_sorted = {}
    for tree in bvh:
        mesh_sorted.update(tree.mesh_data)
    fl.common_tables(desc, mesh_sorted)
    desc.n_top, desc.top = len(top), ffi.iptr(top)
    groups = [fl._bvh_record(g) for g in fl.groups]
    buckets = [fl._bvh_record(t) for t in bvh]
    desc.n_groups, desc.groups = len(groups), fl._arr(fl.rec.rtBvh, groups)
    desc.n_buckets, desc.buckets = len(buckets), fl._arr(fl.rec.rtBvh, buckets)
    fl.keep.append(top)
    return desc, fl
