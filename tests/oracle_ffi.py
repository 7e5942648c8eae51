"""ctypes driver for the CPU oracle (oracle/rt_ref.cpp).  Test infrastructure only: imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py, never by the raytracing_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import _ffi as ffi
from raytracing_amd.render import camera_record, _pattern
from raytracing_amd.scene_manager import BVHTree, Flattener

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "librt_ref.so")
F = np.float32


class refGroup(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32)]


class refWorld(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(ffi.rtObject)),
                ("n_top", C.c_int32), ("top", ffi.c_i32p),
                ("n_transforms", C.c_int32), ("transforms", C.POINTER(ffi.rtTransform)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(ffi.rtMaterial)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(ffi.rtTexture)),
                ("n_images", C.c_int32), ("images", C.POINTER(ffi.rtImage)),
                ("n_perlins", C.c_int32), ("perlins", C.POINTER(ffi.rtPerlin)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(ffi.rtMesh)),
                ("n_groups", C.c_int32), ("groups", C.POINTER(refGroup)),
                ("group_items", ffi.c_i32p)]


class refStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64 * 12), ("scatters", C.c_uint64 * 6)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.ref_last_error.restype = C.c_char_p
        L.ref_scene_create.argtypes = [C.POINTER(refWorld), C.POINTER(C.c_void_p)]
        L.ref_scene_destroy.argtypes = [C.c_void_p]
        L.ref_render.argtypes = [C.c_void_p, C.POINTER(ffi.rtCamera), ffi.c_f32p, C.c_int, ffi.c_f32p, C.c_int,
                                 C.c_int, C.c_int, C.c_uint64, C.c_int, ffi.c_f32p, C.POINTER(refStats)]
        L.ref_trace_paths.argtypes = [C.c_void_p, C.POINTER(ffi.rtCamera), ffi.c_f32p, C.c_int, ffi.c_f32p, C.c_int,
                                      C.c_int, C.c_int, C.c_uint64, ffi.c_i32p, C.c_int, ffi.c_f32p]
        L.ref_path_rays.argtypes = [C.c_void_p, C.POINTER(ffi.rtCamera), ffi.c_f32p, C.c_int, ffi.c_f32p, C.c_int,
                                    C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, ffi.c_f32p, C.c_int]
        L.ref_trace_rays.argtypes = [C.c_void_p, ffi.c_f32p, C.c_int, C.c_float, C.c_float, C.c_uint64, ffi.c_i32p,
                                     ffi.c_i32p, ffi.c_f32p, ffi.c_f32p, ffi.c_f32p, ffi.c_u8p]
        L.ref_bucket_count.argtypes = [C.c_void_p]
        L.ref_bucket_info.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p, ffi.c_i32p, ffi.c_i32p]
        L.ref_bucket_bvh.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p, ffi.c_f32p, ffi.c_i32p, ffi.c_i32p, ffi.c_u8p,
                                     ffi.c_u8p]
        L.ref_mesh_order.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p]
        L.ref_halton_sequence32.argtypes = [C.c_int, ffi.c_f32p]
        L.ref_halton_sequence32.restype = None
        L.ref_pcg4d.argtypes = [C.POINTER(C.c_uint32)]
        L.ref_pcg4d.restype = None
        L.ref_rand01.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.ref_rand01.restype = C.c_float
        for name in ("ref_logf", "ref_sinf", "ref_acosf"):
            getattr(L, name).argtypes = [C.c_float]
            getattr(L, name).restype = C.c_float
        L.ref_atan2f.argtypes = [C.c_float, C.c_float]
        L.ref_atan2f.restype = C.c_float
        _lib = L
    return _lib


def _check(st):
    if st != 0:
        raise RuntimeError("oracle: " + lib().ref_last_error().decode())


class OracleScene:
    """The reference's CPU pipeline on a plain world list: SceneManager(world) -> build_bvh -> hit/trace/draw!."""

    def __init__(self, world):
        fl = Flattener()
        top = np.array([fl.obj(o, False) for o in world], dtype=np.int32)
        w = refWorld()
        fl.common_tables(w)
        w.n_top, w.top = len(top), ffi.iptr(top)
        items, groups = [], []
        for g in fl.groups:  # nested BVHTree objects: hand over the unsorted members, the oracle builds the tree itself
            assert isinstance(g, BVHTree)
            groups.append(refGroup(len(items), len(g.original)))
            items += [fl.obj_index[id(o)] for o in g.original]
        items = np.array(items or [0], dtype=np.int32)
        garr = (refGroup * max(1, len(groups)))(*groups)
        w.n_groups, w.groups, w.group_items = len(groups), garr, ffi.iptr(items)
        self._keep = (fl, top, items, garr)
        self.flattener = fl
        self.handle = C.c_void_p()
        _check(lib().ref_scene_create(C.byref(w), C.byref(self.handle)))

    def __del__(self):
        try:
            if self.handle:
                lib().ref_scene_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def render(self, camera, pattern, background, width, height, max_depth=50, seed=0x5EED, threads=0, stats=False):
        pat, n = _pattern(pattern)
        bg = np.asarray(background, dtype=F)
        out = np.zeros(3 * width * height, dtype=F)
        st = refStats()
        cam = camera_record(camera)
        _check(lib().ref_render(self.handle, C.byref(cam), ffi.fptr(pat), n, ffi.fptr(bg), width, height, max_depth,
                                seed, threads, ffi.fptr(out), C.byref(st) if stats else None))
        img = out.reshape(width, height, 3).transpose(1, 0, 2).copy()
        return (img, st) if stats else img

    def trace_paths(self, camera, pattern, background, width, height, ids, max_depth=50, seed=0x5EED):
        pat, n = _pattern(pattern)
        bg = np.asarray(background, dtype=F)
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 3)
        rad = np.zeros((ids.shape[0], 3), dtype=F)
        cam = camera_record(camera)
        _check(lib().ref_trace_paths(self.handle, C.byref(cam), ffi.fptr(pat), n, ffi.fptr(bg), width, height,
                                     max_depth, seed, ffi.iptr(ids), ids.shape[0], ffi.fptr(rad)))
        return rad

    def path_rays(self, camera, pattern, background, width, height, i, j, s, max_depth=50, seed=0x5EED):
        """The rays path (i, j, s) casts, one row of 7 floats per bounce."""
        pat, n = _pattern(pattern)
        bg = np.asarray(background, dtype=F)
        rays = np.zeros((max_depth, 7), dtype=F)
        cam = camera_record(camera)
        m = lib().ref_path_rays(self.handle, C.byref(cam), ffi.fptr(pat), n, ffi.fptr(bg), width, height, max_depth,
                                seed, i, j, s, ffi.fptr(rays), max_depth)
        if m < 0:
            _check(m)
        return rays[:m].copy()

    def trace_rays(self, rays, t_min=0.001, t_max=np.inf, seed=0x5EED):
        rays = np.ascontiguousarray(rays, dtype=F).reshape(-1, 7)
        n = rays.shape[0]
        out = dict(prim_id=np.zeros(n, np.int32), sub_id=np.zeros(n, np.int32), t=np.zeros(n, F),
                   normal=np.zeros((n, 3), F), uv=np.zeros((n, 2), F), front=np.zeros(n, np.uint8))
        _check(lib().ref_trace_rays(self.handle, ffi.fptr(rays), n, float(t_min), float(t_max), seed,
                                    ffi.iptr(out["prim_id"]), ffi.iptr(out["sub_id"]), ffi.fptr(out["t"]),
                                    ffi.fptr(out["normal"]), ffi.fptr(out["uv"]), ffi.bptr(out["front"])))
        return out

    def buckets(self):
        """Per type group: sorted top-level positions and the node arrays of its tree."""
        res = []
        for b in range(lib().ref_bucket_count(self.handle)):
            n_obj, n_slots, is_mesh = C.c_int32(), C.c_int32(), C.c_int32()
            lib().ref_bucket_info(self.handle, b, C.byref(n_obj), C.byref(n_slots), C.byref(is_mesh))
            so = np.zeros(n_obj.value, np.int32)
            bounds = np.zeros((n_slots.value, 6), F)
            left = np.zeros(n_slots.value, np.int32)
            right = np.zeros(n_slots.value, np.int32)
            leaf = np.zeros(n_slots.value, np.uint8)
            valid = np.zeros(n_slots.value, np.uint8)
            lib().ref_bucket_bvh(self.handle, b, ffi.iptr(so), ffi.fptr(bounds), ffi.iptr(left), ffi.iptr(right),
                                 ffi.bptr(leaf), ffi.bptr(valid))
            res.append(dict(objects=so, bounds=bounds, left=left, right=right, leaf=leaf, valid=valid,
                            is_mesh=bool(is_mesh.value)))
        return res

    def mesh_order(self, mesh, ntri):
        out = np.zeros(ntri, np.int32)
        lib().ref_mesh_order(self.handle, mesh, ffi.iptr(out))
        return out


def primary_rays(camera, width, height, nx, ny, time=0.5):
    """A fixed nx x ny grid of pinhole rays through pixel centres (no lens offset), as 7-float records."""
    rays = np.zeros((ny, nx, 7), dtype=F)
    o = camera.origin.astype(F)
    for b in range(ny):
        for a in range(nx):
            s = F((a + 0.5) / nx)
            t = F((b + 0.5) / ny)
            d = ((camera.lower_left_corner + s * camera.horizontal) + t * camera.vertical) - o
            rays[b, a, :3] = o
            rays[b, a, 3:6] = d
            rays[b, a, 6] = time
    return rays.reshape(-1, 7)
