"""ctypes view of the C ABI declared in include/rt_amd.h.

The record layouts (object, transform, material, texture, image, perlin, mesh, camera)
are shared with the test oracle's header so that one flattener feeds both; nothing in
this module loads or references the oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_amd.so")

# object kinds (include/rt_amd.h)
RT_SPHERE, RT_MOVING_SPHERE, RT_XY_RECT, RT_XZ_RECT, RT_YZ_RECT = 1, 2, 3, 4, 5
RT_BOX, RT_TRANSFORM, RT_CONSTANT_MEDIUM, RT_BVH, RT_MESH = 6, 7, 8, 9, 10
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE_MARBLE, RT_TEX_NOISE_SCALAR = 0, 1, 2, 3, 4
RT_FLAG_COUNTERS = 1
RT_FLAG_TIMING = 2
RT_STATIC_SLOTS, RT_STATIC_MAX_COLUMNS, RT_STATIC_MAX_PIXELS = 256 * 1024, 16384, 2 * 1024 * 1024

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class rtObject(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("child", C.c_int32), ("aux", C.c_int32),
                ("p", C.c_float * 12)]


class rtTransform(C.Structure):
    _fields_ = [("xform", C.c_float * 12), ("inv_xform", C.c_float * 12), ("bbox", C.c_float * 6)]


class rtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("fuzz", C.c_float), ("eta", C.c_float)]


class rtTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("pad", C.c_int32),
                ("color", C.c_float * 3), ("scale", C.c_float)]


class rtImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", c_u8p)]


class rtPerlin(C.Structure):
    _fields_ = [("ranvec", C.c_float * 768), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class rtMesh(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_triangles", C.c_int32), ("n_materials", C.c_int32),
                ("pad", C.c_int32), ("vertices", c_f32p), ("indices", c_i32p), ("materials", c_i32p),
                ("tri_ids", c_i32p)]


class rtBvh(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_objects", C.c_int32), ("objects", c_i32p), ("n_nodes", C.c_int32),
                ("pad", C.c_int32), ("node_bounds", c_f32p), ("node_left", c_i32p), ("node_right", c_i32p),
                ("node_leaf", c_u8p)]


class rtSceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(rtObject)),
                ("n_top", C.c_int32), ("top", c_i32p),
                ("n_transforms", C.c_int32), ("transforms", C.POINTER(rtTransform)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(rtMaterial)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(rtTexture)),
                ("n_images", C.c_int32), ("images", C.POINTER(rtImage)),
                ("n_perlins", C.c_int32), ("perlins", C.POINTER(rtPerlin)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(rtMesh)),
                ("n_groups", C.c_int32), ("groups", C.POINTER(rtBvh)),
                ("n_buckets", C.c_int32), ("buckets", C.POINTER(rtBvh))]


class rtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float), ("t0", C.c_float), ("t1", C.c_float)]


class rtStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("paths", C.c_uint64), ("rays", C.c_uint64),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("node_wave_steps", C.c_uint64),
                ("prim_wave_steps", C.c_uint64), ("bounce_wave_steps", C.c_uint64),
                ("extend_seconds", C.c_double), ("shade_seconds", C.c_double),
                ("extend_launches", C.c_uint64), ("shade_launches", C.c_uint64)]


# ---- F = Float64 (render(Float64, ...), main.jl:21,46): the same records with double fields
c_f64p = C.POINTER(C.c_double)


class rtObject64(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("child", C.c_int32), ("aux", C.c_int32),
                ("p", C.c_double * 12)]


class rtTransform64(C.Structure):
    _fields_ = [("xform", C.c_double * 12), ("inv_xform", C.c_double * 12), ("bbox", C.c_double * 6)]


class rtMaterial64(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("fuzz", C.c_double), ("eta", C.c_double)]


class rtTexture64(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("pad", C.c_int32),
                ("color", C.c_double * 3), ("scale", C.c_double)]


class rtPerlin64(C.Structure):
    _fields_ = [("ranvec", C.c_double * 768), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class rtMesh64(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_triangles", C.c_int32), ("n_materials", C.c_int32),
                ("pad", C.c_int32), ("vertices", c_f64p), ("indices", c_i32p), ("materials", c_i32p),
                ("tri_ids", c_i32p)]


class rtBvh64(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_objects", C.c_int32), ("objects", c_i32p), ("n_nodes", C.c_int32),
                ("pad", C.c_int32), ("node_bounds", c_f64p), ("node_left", c_i32p), ("node_right", c_i32p),
                ("node_leaf", c_u8p)]


class rtSceneDesc64(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(rtObject64)),
                ("n_top", C.c_int32), ("top", c_i32p),
                ("n_transforms", C.c_int32), ("transforms", C.POINTER(rtTransform64)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(rtMaterial64)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(rtTexture64)),
                ("n_images", C.c_int32), ("images", C.POINTER(rtImage)),
                ("n_perlins", C.c_int32), ("perlins", C.POINTER(rtPerlin64)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(rtMesh64)),
                ("n_groups", C.c_int32), ("groups", C.POINTER(rtBvh64)),
                ("n_buckets", C.c_int32), ("buckets", C.POINTER(rtBvh64))]


class rtCamera64(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("lens_radius", C.c_double), ("t0", C.c_double), ("t1", C.c_double)]


class Records:
    """The record types, pointer type and entry-point suffix of one element type."""

    def __init__(self, np_type, suffix, **types):
        self.np_type, self.suffix = np_type, suffix
        self.__dict__.update(types)

    def ptr(self, a):
        return a.ctypes.data_as(self.real_p)

    def fn(self, name):
        """the entry point `name` of this element type (rt_render -> rt_render_f64)"""
        return getattr(lib(), name + self.suffix)


def records(np_type):
    """Records of an element type: numpy.float32 (the reference's default F) or numpy.float64."""
    import numpy as np
    return R64 if np.dtype(np_type) == np.float64 else R32


# every symbol include/rt_amd.h declares: name -> (restype, argtypes)
class rtLayout(C.Structure):
    _fields_ = [("nodes_per_layout", C.c_uint32), ("n_prims", C.c_uint32), ("surface_root", C.c_uint32),
                ("surface_size", C.c_uint32), ("medium_root", C.c_uint32), ("medium_size", C.c_uint32),
                ("nested", C.c_uint32), ("node_scale", C.c_float), ("n_sets", C.c_uint32), ("centre", C.c_float * 3),
                ("radius", C.c_float)]


class rtBuildInfo(C.Structure):
    _fields_ = [("on_device", C.c_uint32), ("n_trees", C.c_uint32), ("max_depth", C.c_uint32), ("build_ms", C.c_float)]


class rtObjMesh(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_triangles", C.c_int32), ("vertices", c_f32p), ("indices", c_i32p)]


SYMBOLS = {
    "rt_abi_version": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_scene_create": (C.c_int, [C.POINTER(rtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (C.c_int, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rtCamera), c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int,
                            C.c_uint64, C.c_uint32, c_f32p, C.POINTER(rtStats)]),
    "rt_render_accumulate": (C.c_int, [C.c_void_p, C.POINTER(rtCamera), c_f32p, C.c_int, C.c_int, C.c_int, c_f32p,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_render_accumulate_static": (C.c_int, [C.c_void_p, C.POINTER(rtCamera), c_f32p, C.c_int, C.c_int, C.c_int, c_f32p,
                                              C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rt_group_create": (C.c_int, [C.POINTER(rtSceneDesc), c_i32p, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_group_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rt_group_render": (C.c_int, [C.c_void_p, C.POINTER(rtCamera), c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int,
                                  C.c_uint64, C.c_uint32, c_f32p, C.POINTER(rtStats)]),
    "rt_group_destroy": (C.c_int, [C.c_void_p]),
    "rt_rccl_available": (C.c_int, []),
    "rt_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_get_counters": (C.c_int, [C.c_void_p, C.POINTER(rtStats)]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.c_float, C.c_float, C.c_uint64, c_i32p, c_i32p,
                                c_f32p, c_f32p, c_f32p, c_u8p]),
    "rt_trace_paths": (C.c_int, [C.c_void_p, C.POINTER(rtCamera), c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int,
                                 C.c_uint64, c_i32p, C.c_int, c_f32p]),
    "rt_finalize_rgb8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_image_to_rgb8": (C.c_int, [c_f32p, C.c_int, C.c_int, c_u8p]),
    "rt_finalize_rgb8_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_image_to_rgb8_f64": (C.c_int, [c_f64p, C.c_int, C.c_int, c_u8p]),
    "rt_write_png": (C.c_int, [C.c_char_p, c_u8p, C.c_int, C.c_int]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, c_u8p, C.c_int, C.c_int]),
    "rt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(rtObjMesh)]),
    "rt_obj_free": (None, [C.POINTER(rtObjMesh)]),
    "rt_scene_desc_save": (C.c_int, [C.POINTER(rtSceneDesc), C.c_char_p]),
    "rt_scene_desc_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(rtSceneDesc))]),
    "rt_scene_desc_free": (None, [C.POINTER(rtSceneDesc)]),
    "rt_scene_desc_save_f64": (C.c_int, [C.POINTER(rtSceneDesc64), C.c_char_p]),
    "rt_scene_desc_load_f64": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(rtSceneDesc64))]),
    "rt_scene_desc_free_f64": (None, [C.POINTER(rtSceneDesc64)]),
    "rt_perlin_generate": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(rtPerlin)]),
    "rt_scene_nodes": (C.c_int, [C.c_void_p, C.POINTER(rtLayout), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_int32),
                                 C.c_size_t, C.POINTER(rtBuildInfo)]),
    "rt_scene_layout_pairs": (C.c_int, [C.POINTER(rtSceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.c_size_t]),
    "rt_scene_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]),
    "rt_scene_layout": (C.c_int, [C.POINTER(rtSceneDesc), C.POINTER(rtLayout), C.POINTER(C.c_uint32), C.c_size_t,
                                  c_i32p, C.c_size_t]),
    "rt_bvh_build": (C.c_int, [C.c_int, c_f32p, c_i32p, c_f32p, c_i32p, c_i32p, c_u8p]),
    "rt_bvh_build_meshes": (C.c_int, [C.c_int, c_i32p, C.POINTER(c_f32p), c_i32p, C.POINTER(c_i32p),
                                      C.POINTER(c_i32p), c_i32p, c_f32p, c_i32p, c_i32p, c_u8p]),
    "rt_scene_create_f64": (C.c_int, [C.POINTER(rtSceneDesc64), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy_f64": (C.c_int, [C.c_void_p]),
    "rt_render_f64": (C.c_int, [C.c_void_p, C.POINTER(rtCamera64), c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, C.c_int,
                                C.c_uint64, C.c_uint32, c_f64p, C.POINTER(rtStats)]),
    "rt_render_accumulate_f64": (C.c_int, [C.c_void_p, C.POINTER(rtCamera64), c_f64p, C.c_int, C.c_int, C.c_int, c_f64p,
                                           C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rt_finalize_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_trace_rays_f64": (C.c_int, [C.c_void_p, c_f64p, C.c_int, C.c_double, C.c_double, C.c_uint64, c_i32p, c_i32p,
                                    c_f64p, c_f64p, c_f64p, c_u8p]),
    "rt_trace_paths_f64": (C.c_int, [C.c_void_p, C.POINTER(rtCamera64), c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, C.c_int,
                                     C.c_uint64, c_i32p, C.c_int, c_f64p]),
    "rt_group_create_f64": (C.c_int, [C.POINTER(rtSceneDesc64), c_i32p, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_group_size_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rt_group_render_f64": (C.c_int, [C.c_void_p, C.POINTER(rtCamera64), c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, C.c_int,
                                      C.c_uint64, C.c_uint32, c_f64p, C.POINTER(rtStats)]),
    "rt_group_destroy_f64": (C.c_int, [C.c_void_p]),
    "rt_bvh_build_f64": (C.c_int, [C.c_int, c_f64p, c_i32p, c_f64p, c_i32p, c_i32p, c_u8p]),
    "rt_bvh_build_meshes_f64": (C.c_int, [C.c_int, c_i32p, C.POINTER(c_f64p), c_i32p, C.POINTER(c_i32p),
                                          C.POINTER(c_i32p), c_i32p, c_f64p, c_i32p, c_i32p, c_u8p]),
}

_lib = None


class RtError(RuntimeError):
    pass


def lib():
    """Load librt_amd.so (built in-tree by __graft_entry__.build()).  There is no fallback:
    without the HIP library nothing in this package can render."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                          f"(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.rt_abi_version() != 1:
            raise RtError("librt_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != 0:
        msg = lib().rt_last_error()
        raise RtError(f"librt_amd: status {status}: {msg.decode() if msg else ''}")


def fptr(a):
    return a.ctypes.data_as(c_f32p)


def iptr(a):
    return a.ctypes.data_as(c_i32p)


def bptr(a):
    return a.ctypes.data_as(c_u8p)


def dptr(a):
    return a.ctypes.data_as(c_f64p)


R32 = Records(__import__("numpy").float32, "", real=C.c_float, real_p=c_f32p, rtObject=rtObject, rtTransform=rtTransform,
              rtMaterial=rtMaterial, rtTexture=rtTexture, rtPerlin=rtPerlin, rtMesh=rtMesh, rtBvh=rtBvh,
              rtSceneDesc=rtSceneDesc, rtCamera=rtCamera)
R64 = Records(__import__("numpy").float64, "_f64", real=C.c_double, real_p=c_f64p, rtObject=rtObject64,
              rtTransform=rtTransform64, rtMaterial=rtMaterial64, rtTexture=rtTexture64, rtPerlin=rtPerlin64,
              rtMesh=rtMesh64, rtBvh=rtBvh64, rtSceneDesc=rtSceneDesc64, rtCamera=rtCamera64)
