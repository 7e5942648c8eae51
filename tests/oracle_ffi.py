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
LIB64 = os.path.join(ORACLE_DIR, "_build", "librt_ref64.so")
F = np.float32


class refGroup(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32)]


def _world_type(rec):
    class refWorld(C.Structure):
        _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(rec.rtObject)),
                    ("n_top", C.c_int32), ("top", ffi.c_i32p),
                    ("n_transforms", C.c_int32), ("transforms", C.POINTER(rec.rtTransform)),
                    ("n_materials", C.c_int32), ("materials", C.POINTER(rec.rtMaterial)),
                    ("n_textures", C.c_int32), ("textures", C.POINTER(rec.rtTexture)),
                    ("n_images", C.c_int32), ("images", C.POINTER(ffi.rtImage)),
                    ("n_perlins", C.c_int32), ("perlins", C.POINTER(rec.rtPerlin)),
                    ("n_meshes", C.c_int32), ("meshes", C.POINTER(rec.rtMesh)),
                    ("n_groups", C.c_int32), ("groups", C.POINTER(refGroup)),
                    ("group_items", ffi.c_i32p)]
    return refWorld


refWorld = _world_type(ffi.R32)
refWorld64 = _world_type(ffi.R64)


class refStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64 * 12), ("scatters", C.c_uint64 * 6)]


_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib(F=np.float32):
    """The oracle built for element type F: librt_ref.so (Float32) or librt_ref64.so (the same source with
    -DREF_F64: every real a double, same function names)."""
    rec = ffi.records(F)
    if rec not in _libs:
        path = LIB if rec is ffi.R32 else LIB64
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        rp, real, world = rec.real_p, rec.real, (refWorld if rec is ffi.R32 else refWorld64)
        L.ref_last_error.restype = C.c_char_p
        L.ref_scene_create.argtypes = [C.POINTER(world), C.POINTER(C.c_void_p)]
        L.ref_scene_destroy.argtypes = [C.c_void_p]
        L.ref_render.argtypes = [C.c_void_p, C.POINTER(rec.rtCamera), rp, C.c_int, rp, C.c_int,
                                 C.c_int, C.c_int, C.c_uint64, C.c_int, rp, C.POINTER(refStats)]
        L.ref_trace_paths.argtypes = [C.c_void_p, C.POINTER(rec.rtCamera), rp, C.c_int, rp, C.c_int,
                                      C.c_int, C.c_int, C.c_uint64, ffi.c_i32p, C.c_int, rp]
        L.ref_path_rays.argtypes = [C.c_void_p, C.POINTER(rec.rtCamera), rp, C.c_int, rp, C.c_int,
                                    C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, rp, C.c_int]
        L.ref_trace_rays.argtypes = [C.c_void_p, rp, C.c_int, real, real, C.c_uint64, ffi.c_i32p,
                                     ffi.c_i32p, rp, rp, rp, ffi.c_u8p]
        L.ref_bucket_count.argtypes = [C.c_void_p]
        L.ref_bucket_info.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p, ffi.c_i32p, ffi.c_i32p]
        L.ref_bucket_bvh.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p, rp, ffi.c_i32p, ffi.c_i32p, ffi.c_u8p,
                                     ffi.c_u8p]
        L.ref_mesh_order.argtypes = [C.c_void_p, C.c_int, ffi.c_i32p]
        L.ref_halton_sequence32.argtypes = [C.c_int, rp]
        L.ref_halton_sequence32.restype = None
        L.ref_pcg4d.argtypes = [C.POINTER(C.c_uint32)]
        L.ref_pcg4d.restype = None
        L.ref_rand01.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.ref_rand01.restype = real
        for name in ("ref_logf", "ref_sinf", "ref_acosf"):
            getattr(L, name).argtypes = [real]
            getattr(L, name).restype = real
        L.ref_atan2f.argtypes = [real, real]
        L.ref_atan2f.restype = real
        _libs[rec] = L
    return _libs[rec]


def _check(st, L=None):
    if st != 0:
        raise RuntimeError("oracle: " + (L or lib()).ref_last_error().decode())


class OracleScene:
    """The reference's CPU pipeline on a plain world list: SceneManager(world) -> build_bvh -> hit/trace/draw!."""

    def __init__(self, world, F=np.float32):
        self.rec = rec = ffi.records(F)
        self.F = rec.np_type
        self.lib = lib(self.F)
        from raytracing_amd import hittables as H
        with H.element_type(self.F):
            fl = Flattener(rec)
            top = np.array([fl.obj(o, False) for o in world], dtype=np.int32)
        w = (refWorld if rec is ffi.R32 else refWorld64)()
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
        _check(self.lib.ref_scene_create(C.byref(w), C.byref(self.handle)), self.lib)

    def __del__(self):
        try:
            if self.handle:
                self.lib.ref_scene_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def render(self, camera, pattern, background, width, height, max_depth=50, seed=0x5EED, threads=0, stats=False):
        pat, n = _pattern(pattern, self.F)
        bg = np.asarray(background, dtype=self.F)
        out = np.zeros(3 * width * height, dtype=self.F)
        st = refStats()
        cam = camera_record(camera, self.rec)
        _check(self.lib.ref_render(self.handle, C.byref(cam), self.rec.ptr(pat), n, self.rec.ptr(bg), width, height, max_depth,
                                seed, threads, self.rec.ptr(out), C.byref(st) if stats else None))
        img = out.reshape(width, height, 3).transpose(1, 0, 2).copy()
        return (img, st) if stats else img

    def trace_paths(self, camera, pattern, background, width, height, ids, max_depth=50, seed=0x5EED):
        pat, n = _pattern(pattern, self.F)
        bg = np.asarray(background, dtype=self.F)
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 3)
        rad = np.zeros((ids.shape[0], 3), dtype=self.F)
        cam = camera_record(camera, self.rec)
        _check(self.lib.ref_trace_paths(self.handle, C.byref(cam), self.rec.ptr(pat), n, self.rec.ptr(bg), width, height,
                                     max_depth, seed, ffi.iptr(ids), ids.shape[0], self.rec.ptr(rad)))
        return rad

    def path_rays(self, camera, pattern, background, width, height, i, j, s, max_depth=50, seed=0x5EED):
        """The rays path (i, j, s) casts, one row of 7 floats per bounce."""
        pat, n = _pattern(pattern, self.F)
        bg = np.asarray(background, dtype=self.F)
        rays = np.zeros((max_depth, 7), dtype=self.F)
        cam = camera_record(camera, self.rec)
        m = self.lib.ref_path_rays(self.handle, C.byref(cam), self.rec.ptr(pat), n, self.rec.ptr(bg), width, height, max_depth,
                                seed, i, j, s, self.rec.ptr(rays), max_depth)
        if m < 0:
            _check(m)
        return rays[:m].copy()

    def trace_rays(self, rays, t_min=0.001, t_max=np.inf, seed=0x5EED):
        rays = np.ascontiguousarray(rays, dtype=self.F).reshape(-1, 7)
        n = rays.shape[0]
        out = dict(prim_id=np.zeros(n, np.int32), sub_id=np.zeros(n, np.int32), t=np.zeros(n, self.F),
                   normal=np.zeros((n, 3), self.F), uv=np.zeros((n, 2), self.F), front=np.zeros(n, np.uint8))
        _check(self.lib.ref_trace_rays(self.handle, self.rec.ptr(rays), n, float(self.F(t_min)), float(self.F(t_max)), seed,
                                    ffi.iptr(out["prim_id"]), ffi.iptr(out["sub_id"]), self.rec.ptr(out["t"]),
                                    self.rec.ptr(out["normal"]), self.rec.ptr(out["uv"]), ffi.bptr(out["front"])))
        return out

    def buckets(self):
        """Per type group: sorted top-level positions and the node arrays of its tree."""
        res = []
        for b in range(self.lib.ref_bucket_count(self.handle)):
            n_obj, n_slots, is_mesh = C.c_int32(), C.c_int32(), C.c_int32()
            self.lib.ref_bucket_info(self.handle, b, C.byref(n_obj), C.byref(n_slots), C.byref(is_mesh))
            so = np.zeros(n_obj.value, np.int32)
            bounds = np.zeros((n_slots.value, 6), self.F)
            left = np.zeros(n_slots.value, np.int32)
            right = np.zeros(n_slots.value, np.int32)
            leaf = np.zeros(n_slots.value, np.uint8)
            valid = np.zeros(n_slots.value, np.uint8)
            self.lib.ref_bucket_bvh(self.handle, b, ffi.iptr(so), self.rec.ptr(bounds), ffi.iptr(left), ffi.iptr(right),
                                 ffi.bptr(leaf), ffi.bptr(valid))
            res.append(dict(objects=so, bounds=bounds, left=left, right=right, leaf=leaf, valid=valid,
                            is_mesh=bool(is_mesh.value)))
        return res

    def mesh_order(self, mesh, ntri):
        out = np.zeros(ntri, np.int32)
        self.lib.ref_mesh_order(self.handle, mesh, ffi.iptr(out))
        return out


def primary_rays(camera, width, height, nx, ny, time=0.5, F=np.float32):
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
