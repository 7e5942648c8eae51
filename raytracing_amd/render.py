"""Host-side mirror of the reference's public API for the accelerated path:
render(...) (main.jl:21-96) with gpu=True, and the device half of draw! (src/Ray_Tracing.jl:141-157).
The CPU branch of the reference (gpu=false) is the reference's own Julia code and is not
reimplemented here: this package renders on the MI355X or not at all.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi as ffi
from . import hittables as H
from .scene_manager import SceneManager, build_bvh_scene, flatten_scene

F = np.float32
DEFAULT_SEED = 0x5EED


def camera_record(cam, rec=ffi.R32):
    c = rec.rtCamera()
    for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        getattr(c, name)[:] = [float(x) for x in getattr(cam, name)]
    c.lens_radius, c.t0, c.t1 = float(cam.lens_radius), float(cam.t0), float(cam.t1)
    return c


def _pattern(pattern, dtype=F):
    """pattern is [x_0 ... x_n; y_0 ... y_n] (2 x N); the ABI wants it column-major like Julia stores it."""
    p = np.asarray(pattern, dtype=dtype)
    assert p.ndim == 2 and p.shape[0] == 2, "pattern must be 2 x N"
    return np.ascontiguousarray(p.T).reshape(-1), p.shape[1]


def save_scene(path, world, F=np.float32):
    """Writes the scene description of `world` (its tables and the trees build_bvh makes of it: what rt_scene_create
    takes) to one binary file (rt_scene_desc_save; F=numpy.float64: the Float64 description, rt_scene_desc_save_f64);
    DeviceScene.from_file(path) renders it without the world list."""
    rec = ffi.records(F)
    world = list(world)
    with H.element_type(rec.np_type):
        bvh = build_bvh_scene(SceneManager(world))
        desc, keep = flatten_scene(world, bvh)
    ffi.check(rec.fn("rt_scene_desc_save")(C.byref(desc), os.fsencode(path)))
    del keep


def scene_layout(world):
    """The acceleration structure the library builds for `world` (rt_scene_layout; computed on the host, no GPU needed).
    Returns (info, nodes, prim_top): nodes is a (8 * n_sets, nodes_per_layout, 4) uint32 array (per node set one copy per
    ray octant; words 0-2 hold the box as six IEEE halves, word 3 the link; set 0: tight boxes for rays that start inside
    the ball (info.centre, info.radius), set 1 if present: the reference's boxes), prim_top[k] the position of primitive
    k in `world`."""
    world = list(world)
    bvh = build_bvh_scene(SceneManager(world))
    desc, keep = flatten_scene(world, bvh)
    info = ffi.rtLayout()
    L = ffi.lib()
    ffi.check(L.rt_scene_layout(C.byref(desc), C.byref(info), None, 0, None, 0))
    nodes = np.zeros((8 * info.n_sets, info.nodes_per_layout, 4), dtype=np.uint32)
    prim_top = np.zeros(info.n_prims, dtype=np.int32)
    ffi.check(L.rt_scene_layout(C.byref(desc), C.byref(info), nodes.ctypes.data_as(C.POINTER(C.c_uint32)), nodes.size // 4,
                                ffi.iptr(prim_top), prim_top.size))
    del keep
    return info, nodes, prim_top


def scene_layout_pairs(world, n_sets):
    """The tree of surfaces as pair nodes, as the host builder makes them (rt_scene_layout_pairs): (pairs, depth) with
    pairs a (8 * n_sets, pairs_per_layout, 8) uint32 array -- words 0-2 the near child's box, 3 its word, 4-6 the far
    child's box, 7 its word -- or an empty array if the scene has none (no surfaces, or nested trees)."""
    world = list(world)
    desc, keep = flatten_scene(world, build_bvh_scene(SceneManager(world)))
    L = ffi.lib()
    n, depth = C.c_uint32(0), C.c_uint32(0)
    ffi.check(L.rt_scene_layout_pairs(C.byref(desc), C.byref(n), C.byref(depth), None, 0))
    pairs = np.zeros((8 * n_sets, n.value, 8), dtype=np.uint32)
    if pairs.size:
        ffi.check(L.rt_scene_layout_pairs(C.byref(desc), C.byref(n), C.byref(depth), pairs.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          pairs.size // 8))
    del keep
    return pairs, depth.value


class DeviceScene:
    """What draw!(::CuArray, ...) builds before its kernel launch (src/Ray_Tracing.jl:144-145):
    bvh = build_bvh(world); cu_bvh = CuSceneManager(bvh) -- here a scene resident in one GPU's HBM."""

    def __init__(self, world, device=0, F=np.float32):
        """F: the element type of render(F, ...) (main.jl:21): numpy.float32, or numpy.float64 for the Float64 entry
        points (rt_*_f64: render, accumulate, finalize, the 8-bit output stage and the two parity hooks; layout() and
        pairs() show the Float32 path's acceleration structure and exist for it only)."""
        self.rec = ffi.records(F)
        self.F = self.rec.np_type
        self.world = list(world)
        with H.element_type(self.F):
            self.scene_manager = SceneManager(self.world)
            self.bvh = build_bvh_scene(self.scene_manager)
            desc, keep = flatten_scene(self.world, self.bvh)
        self._keep = keep
        self.handle = C.c_void_p()
        ffi.check(self.rec.fn("rt_scene_create")(C.byref(desc), int(device), C.byref(self.handle)))

    def _f32_only(self, what):
        if self.rec is not ffi.R32:
            raise ffi.RtError(f"{what} exists for Float32 scenes only")

    def layout(self):
        """The node sets as they are on the device (rt_scene_nodes): (info, nodes, prim_top, build) with info, nodes and
        prim_top as scene_layout() returns them and build = rtBuildInfo (who built the trees, and how)."""
        self._f32_only("layout()")
        info, build = ffi.rtLayout(), ffi.rtBuildInfo()
        L = ffi.lib()
        ffi.check(L.rt_scene_nodes(self.handle, C.byref(info), None, 0, None, 0, None))
        nodes = np.zeros((8 * info.n_sets, info.nodes_per_layout, 4), dtype=np.uint32)
        prim_top = np.zeros(info.n_prims, dtype=np.int32)
        ffi.check(L.rt_scene_nodes(self.handle, C.byref(info), nodes.ctypes.data_as(C.POINTER(C.c_uint32)), nodes.size // 4,
                                   ffi.iptr(prim_top), prim_top.size, C.byref(build)))
        return info, nodes, prim_top, build

    def pairs(self):
        """The pair nodes as they are on the device (rt_scene_pairs): (pairs, in_use), pairs as scene_layout_pairs()"""
        self._f32_only("pairs()")
        info = self.layout()[0]
        L = ffi.lib()
        n, used = C.c_uint32(0), C.c_uint32(0)
        ffi.check(L.rt_scene_pairs(self.handle, C.byref(n), C.byref(used), None, 0))
        pairs = np.zeros((8 * info.n_sets, n.value, 8), dtype=np.uint32)
        if pairs.size:
            ffi.check(L.rt_scene_pairs(self.handle, C.byref(n), C.byref(used), pairs.ctypes.data_as(C.POINTER(C.c_uint32)), pairs.size // 8))
        return pairs, bool(used.value)

    @classmethod
    def from_file(cls, path, device=0, F=np.float32):
        """A scene written by save_scene (rt_scene_desc_load + rt_scene_create, or their _f64 forms for a file written
        with F=numpy.float64): no world list, no tree build.  A file of the other element type is an error."""
        self = cls.__new__(cls)
        self.world = self.scene_manager = self.bvh = self._keep = None
        self.rec = rec = ffi.records(F)
        self.F = rec.np_type
        self.handle = C.c_void_p()
        desc = C.POINTER(rec.rtSceneDesc)()
        ffi.check(rec.fn("rt_scene_desc_load")(os.fsencode(path), C.byref(desc)))
        try:
            ffi.check(rec.fn("rt_scene_create")(desc, int(device), C.byref(self.handle)))
        finally:
            rec.fn("rt_scene_desc_free")(desc)
        return self

    def close(self):
        if self.handle:
            self.rec.fn("rt_scene_destroy")(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, pattern, background, width, height, max_depth=50, seed=DEFAULT_SEED, flags=0):
        """Returns (img, stats): img is (height, width, 3) of the scene's element type, row 0 = top, sqrt-gamma,
        unclamped."""
        rec, F = self.rec, self.F
        pat, n = _pattern(pattern, F)
        bg = np.asarray(background, dtype=F)
        out = np.zeros(3 * width * height, dtype=F)
        st = ffi.rtStats()
        cam = camera_record(camera, rec)
        ffi.check(rec.fn("rt_render")(self.handle, C.byref(cam), rec.ptr(pat), n, rec.ptr(bg), width, height,
                                      max_depth, seed, flags, rec.ptr(out), C.byref(st)))
        # Julia Array{RGB}(H, W) memory order is [col][row][rgb]
        return out.reshape(width, height, 3).transpose(1, 0, 2).copy(), st

    def accumulate(self, camera, pattern, background, width, height, d_accum, sample_first=0, sample_stride=1,
                   max_depth=50, seed=DEFAULT_SEED, flags=0, stream=None):
        """Un-normalised sums of this shard's samples into device memory (d_accum = device pointer)."""
        rec, F = self.rec, self.F
        pat, n = _pattern(pattern, F)
        bg = np.asarray(background, dtype=F)
        cam = camera_record(camera, rec)
        args = [self.handle, C.byref(cam), rec.ptr(pat), n, sample_first, sample_stride, rec.ptr(bg), width, height,
                max_depth, seed] + ([flags] if rec is ffi.R32 else []) + [C.c_void_p(d_accum), C.c_void_p(stream or 0)]
        ffi.check(rec.fn("rt_render_accumulate")(*args))

    def accumulate_static(self, camera, pattern, background, width, height, d_accum, sample_first=0, sample_stride=1,
                          max_depth=50, seed=DEFAULT_SEED, stream=None):
        """accumulate() without any device allocation (rt_render_accumulate_static): the working buffers are a fixed
        array inside the library; at most RT_STATIC_MAX_COLUMNS pattern columns and RT_STATIC_MAX_PIXELS pixels."""
        self._f32_only("accumulate_static()")
        pat, n = _pattern(pattern)
        bg = np.asarray(background, dtype=F)
        cam = camera_record(camera)
        ffi.check(ffi.lib().rt_render_accumulate_static(self.handle, C.byref(cam), ffi.fptr(pat), n, sample_first,
                                                        sample_stride, ffi.fptr(bg), width, height, max_depth, seed,
                                                        C.c_void_p(d_accum), C.c_void_p(stream or 0)))

    def finalize(self, d_accum, n_total, width, height, d_out, stream=None):
        ffi.check(self.rec.fn("rt_finalize")(self.handle, C.c_void_p(d_accum), n_total, width, height,
                                        C.c_void_p(d_out), C.c_void_p(stream or 0)))

    def finalize_rgb8(self, d_accum, n_total, width, height, d_rgb8, stream=None):
        """rt_finalize + clamp01 + 8-bit quantisation on the device; d_rgb8 receives height x width x 3 bytes."""
        ffi.check(self.rec.fn("rt_finalize_rgb8")(self.handle, C.c_void_p(d_accum), n_total, width, height,
                                             C.c_void_p(d_rgb8), C.c_void_p(stream or 0)))

    def render_progressive(self, camera, pattern, background, width, height, passes=8, state=None, max_depth=50,
                           seed=DEFAULT_SEED):
        """Progressive refinement with resume.  The pattern columns are dealt to `passes` interleaved shards (shard c =
        columns c, c + passes, ...: the sample sharding of the multi-GPU path, in time instead of across GPUs); after
        every shard the generator yields (state, img): img is the picture of the columns traced so far (same layout as
        render()), state a dict of host arrays {"accum", "next", "passes", "n", "done"} that np.savez can store and a
        later call takes back through `state=` to go on where it stopped.  After the last shard the image equals
        render()'s up to the order of the float32 sums."""
        import torch
        if not torch.cuda.is_available():
            raise ffi.RtError("render_progressive keeps its accumulators in torch tensors and torch sees no GPU; if the "
                              "library was loaded first, import torch before raytracing_amd (torch ships its own HIP "
                              "runtime, which cannot start after the system's)")
        F = self.F
        pat, n = _pattern(pattern, F)
        passes = max(1, min(int(passes), n))
        dev = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream().cuda_stream
        total = torch.zeros(3 * width * height, dtype=torch.float64 if F is np.float64 else torch.float32, device=dev)
        first, done = 0, 0
        if state is not None:
            if int(state["passes"]) != passes or int(state["n"]) != n or np.asarray(state["accum"]).size != total.numel():
                raise ValueError("render_progressive: the state belongs to a different pattern, pass count or image size")
            total.copy_(torch.from_numpy(np.ascontiguousarray(state["accum"], dtype=F)))
            first, done = int(state["next"]), int(state["done"])
        part = torch.empty_like(total)
        out = torch.empty_like(total)
        for c in range(first, passes):
            self.accumulate(camera, pattern, background, width, height, part.data_ptr(), sample_first=c,
                            sample_stride=passes, max_depth=max_depth, seed=seed, stream=stream)
            total += part
            done += len(range(c, n, passes))
            self.finalize(total.data_ptr(), done, width, height, out.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            img = out.cpu().numpy().reshape(width, height, 3).transpose(1, 0, 2).copy()
            yield dict(accum=total.cpu().numpy(), next=c + 1, passes=passes, n=n, done=done), img

    def counters(self):
        self._f32_only("counters()")
        st = ffi.rtStats()
        ffi.check(ffi.lib().rt_get_counters(self.handle, C.byref(st)))
        return st

    def trace_rays(self, rays, t_min=0.001, t_max=np.inf, seed=DEFAULT_SEED):
        rec, F = self.rec, self.F
        rays = np.ascontiguousarray(rays, dtype=F).reshape(-1, 7)
        n = rays.shape[0]
        out = dict(prim_id=np.zeros(n, np.int32), sub_id=np.zeros(n, np.int32), t=np.zeros(n, F),
                   normal=np.zeros((n, 3), F), uv=np.zeros((n, 2), F), front=np.zeros(n, np.uint8))
        ffi.check(rec.fn("rt_trace_rays")(self.handle, rec.ptr(rays), n, float(F(t_min)), float(F(t_max)), seed,
                                          ffi.iptr(out["prim_id"]), ffi.iptr(out["sub_id"]), rec.ptr(out["t"]),
                                          rec.ptr(out["normal"]), rec.ptr(out["uv"]), ffi.bptr(out["front"])))
        return out

    def trace_paths(self, camera, pattern, background, width, height, ids, max_depth=50, seed=DEFAULT_SEED):
        rec, F = self.rec, self.F
        pat, n = _pattern(pattern, F)
        bg = np.asarray(background, dtype=F)
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 3)
        rad = np.zeros((ids.shape[0], 3), dtype=F)
        cam = camera_record(camera, rec)
        ffi.check(rec.fn("rt_trace_paths")(self.handle, C.byref(cam), rec.ptr(pat), n, rec.ptr(bg), width, height,
                                           max_depth, seed, ffi.iptr(ids), ids.shape[0], rec.ptr(rad)))
        return rad


class DeviceGroup:
    """The scene on several GPUs of one host process (rt_group_create): render() deals the pattern columns to them,
    sums their accumulators on the first device (RCCL, or peer copies with RT_AMD_REDUCE=peer) and returns the image.
    devices: a list of device numbers (a number may appear twice: several shards on one GPU), None = every visible GPU.
    F: the element type, numpy.float32 or numpy.float64 (rt_group_*_f64)."""

    def __init__(self, world, devices=None, F=np.float32):
        self.rec = ffi.records(F)
        self.F = self.rec.np_type
        self.world = list(world)
        with H.element_type(self.F):
            bvh = build_bvh_scene(SceneManager(self.world))
            desc, keep = flatten_scene(self.world, bvh)
        self.handle = C.c_void_p()
        dev = np.ascontiguousarray(devices if devices is not None else [], dtype=np.int32)
        ffi.check(self.rec.fn("rt_group_create")(C.byref(desc), ffi.iptr(dev) if len(dev) else None, len(dev), C.byref(self.handle)))
        del keep
        n = C.c_int(0)
        ffi.check(self.rec.fn("rt_group_size")(self.handle, C.byref(n)))
        self.n_devices = n.value

    def close(self):
        if self.handle:
            self.rec.fn("rt_group_destroy")(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, pattern, background, width, height, max_depth=50, seed=DEFAULT_SEED, flags=0):
        """Returns (img, stats) like DeviceScene.render."""
        rec, F = self.rec, self.F
        pat, n = _pattern(pattern, F)
        bg = np.asarray(background, dtype=F)
        out = np.zeros(3 * width * height, dtype=F)
        st = ffi.rtStats()
        cam = camera_record(camera, rec)
        ffi.check(rec.fn("rt_group_render")(self.handle, C.byref(cam), rec.ptr(pat), n, rec.ptr(bg), width, height,
                                            max_depth, seed, flags, rec.ptr(out), C.byref(st)))
        return out.reshape(width, height, 3).transpose(1, 0, 2).copy(), st


def render(world, material=None, *, width=None, height=None, camera=None, background=None, pattern=None,
           gpu=False, spp=None, seed=DEFAULT_SEED, device=0, devices=None, F=np.float32, **kwargs):
    """render(world::Vector{<:Hittable}; ...), render(filename, material; ...) and render(key::Symbol; ...)
    (main.jl:21-96).  Defaults as main.jl:22-27: 1280x720, Camera(), sky-blue background, MSAA_S16.
    F is the reference's leading type argument (render(Float64, ...), main.jl:21,46): numpy.float32 or numpy.float64.
    Returns an (height, width, 3) image of that type, row 0 = top, sqrt-gamma, unclamped."""
    F = ffi.records(F).np_type
    with H.element_type(F):
        return _render(world, material, width, height, camera, background, pattern, gpu, spp, seed, device, devices)


def _render(world, material, width, height, camera, background, pattern, gpu, spp, seed, device, devices):
    F = H.F
    given = {k: v for k, v in dict(width=width, height=height, camera=camera, background=background,
                                   pattern=pattern).items() if v is not None}
    opts = dict(width=1280, height=720, camera=None, background=(0.70, 0.80, 1.00), pattern=None)
    if isinstance(world, str) and material is not None:
        # render(filename::String, material; kwargs...) (main.jl:53-81)
        from .obj_loader import load_obj_mesh
        world = [load_obj_mesh(world, material)]
    elif isinstance(world, str):
        # render(key::Symbol; gpu, kwargs...) (main.jl:90-94): scene values, overridden by the caller's kwargs
        from .scenes import get_scene
        world, scene_kwargs = get_scene(world, F=F, **({} if spp is None else {"spp": spp}))
        opts.update(scene_kwargs)
    opts.update(given)
    if opts["camera"] is None:
        opts["camera"] = H.Camera()
    if opts["pattern"] is None:
        opts["pattern"] = H.MSAA_S16.astype(F)
    if not gpu:
        raise NotImplementedError("gpu=False is the reference's own Julia CPU path; this package only provides "
                                  "the MI355X path (gpu=True)")
    # devices: a list of GPUs to spread the samples over (rt_group_*), "all" = every visible GPU; default: the one `device`
    if devices is not None:
        scene = DeviceGroup(world, devices=None if isinstance(devices, str) and devices == "all" else devices, F=F)
    else:
        scene = DeviceScene(world, device=device, F=F)
    try:
        img, _ = scene.render(opts["camera"], opts["pattern"], opts["background"], opts["width"], opts["height"],
                              seed=seed)
    finally:
        scene.close()
    return img


def clamp01(img):
    """clamp01.(img) with NaN -> 0 (clamp01nan), as the reference's README asks before saving (README.md:56-63)"""
    T = np.float64 if np.asarray(img).dtype == np.float64 else F
    a = np.asarray(img, dtype=T)
    return np.clip(np.where(np.isnan(a), T(0), a), T(0), T(1))


def to_rgb8(img):
    """(height, width, 3) float image -> uint8 the way Images.jl stores N0f8: round-to-nearest-even of clamp01(x) * 255"""
    rec = ffi.records(np.float64 if np.asarray(img).dtype == np.float64 else F)  # an RGB{Float64} image is quantised in double
    a = np.ascontiguousarray(np.asarray(img, dtype=rec.np_type).transpose(1, 0, 2))  # rt_render's memory order [col][row][rgb]
    h, w = a.shape[1], a.shape[0]
    out = np.zeros((h, w, 3), dtype=np.uint8)
    ffi.check(rec.fn("rt_image_to_rgb8")(rec.ptr(a), w, h, ffi.bptr(out)))
    return out


def save(path, img):
    """save(path, clamp01.(img)) (README.md:50-67): PNG (or binary PPM for a .ppm path), 8 bits per channel"""
    rgb = to_rgb8(img)
    h, w = rgb.shape[0], rgb.shape[1]
    fn = ffi.lib().rt_write_ppm if str(path).lower().endswith(".ppm") else ffi.lib().rt_write_png
    ffi.check(fn(str(path).encode(), ffi.bptr(rgb), w, h))
    return path
