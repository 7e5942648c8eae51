"""Wavefront OBJ -> Mesh, the host-side step of render(filename, material) (main.jl:53-81).

The reference delegates loading to MeshIO/GeometryBasics (unpinned versions); what it keeps is one Vertex per
distinct (position, texcoord, normal) combination and one index column per triangle.  This loader defines:
vertices = distinct v/vt/vn triples in order of first appearance, polygons fan-triangulated (1,2,3),(1,3,4),...
"""
import numpy as np

from .hittables import Mesh

F = np.float32


def load_obj(filename):
    """(vertices (nv, 8) float32, indices (ntri, 4) int32) through the library's loader (rt_obj_load)"""
    import ctypes as C
    from . import _ffi as ffi
    mesh = ffi.rtObjMesh()
    L = ffi.lib()
    ffi.check(L.rt_obj_load(str(filename).encode(), C.byref(mesh)))
    try:
        vertices = np.ctypeslib.as_array(mesh.vertices, shape=(max(1, mesh.n_vertices), 8))[:mesh.n_vertices].copy()
        indices = np.ctypeslib.as_array(mesh.indices, shape=(max(1, mesh.n_triangles), 4))[:mesh.n_triangles].copy()
    finally:
        L.rt_obj_free(C.byref(mesh))
    return vertices.astype(F), indices.astype(np.int32)


def load_obj_py(filename):
    """the same loader in plain Python (kept to cross-check the native one)"""
    pos, tex, nrm = [], [], []
    verts, vert_index, tris = [], {}, []
    with open(filename) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                pos.append([float(x) for x in parts[1:4]])
            elif parts[0] == "vt":
                tex.append([float(x) for x in parts[1:3]])
            elif parts[0] == "vn":
                nrm.append([float(x) for x in parts[1:4]])
            elif parts[0] == "f":
                ids = []
                for tok in parts[1:]:
                    fields = (tok.split("/") + ["", ""])[:3]
                    key = tuple(int(x) if x else 0 for x in fields)
                    key = tuple(k + (len(src) + 1 if k < 0 else 0) for k, src in zip(key, (pos, tex, nrm)))
                    if key not in vert_index:
                        vert_index[key] = len(verts)
                        verts.append(key)
                    ids.append(vert_index[key])
                for k in range(1, len(ids) - 1):
                    tris.append([ids[0], ids[k], ids[k + 1], 0])
    vertices = np.zeros((len(verts), 8), dtype=F)
    for i, (vi, ti, ni) in enumerate(verts):
        vertices[i, 0:3] = pos[vi - 1]
        # main.jl:62-67: missing normals become zero vectors, missing texcoords (-1, -1)
        vertices[i, 3:6] = nrm[ni - 1] if ni else (0.0, 0.0, 0.0)
        vertices[i, 6:8] = tex[ti - 1] if ti else (-1.0, -1.0)
    return vertices, np.array(tris, dtype=np.int32).reshape(-1, 4)


def load_obj_mesh(filename, material):
    vertices, indices = load_obj(filename)
    return Mesh(vertices, [material], indices)
