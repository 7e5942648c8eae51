// rt_api.cpp -- the extern "C" entry points of librt_amd.so (include/rt_amd.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/rt_amd.h"
#include "rt_internal.h"
#include "rt_kernels.h"
#include "rt_scene.h"

static thread_local std::string g_error;

int rt_fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return status;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return rt_fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : (e_ == hipErrorNoDevice ? RT_ENODEV : RT_EHIP), \
                           "%s failed: %s", #expr, hipGetErrorString(e_));                                \
    } while (0)

struct rtScene {
    int device = 0;
    DevScene dev{};
    void* buffers[16] = {nullptr};
    int n_buffers = 0;
    // scratch owned by the handle, grown on demand
    float* d_pattern = nullptr; size_t pattern_cap = 0;
    float* d_partial = nullptr; size_t partial_cap = 0;
    float* d_accum = nullptr;   size_t accum_cap = 0;
    unsigned long long* d_counters = nullptr;
    hipStream_t last_stream = nullptr;
    uint64_t last_paths = 0;
};

namespace {

template <typename T>
int upload(rtScene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = v.size() * sizeof(T);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    s->buffers[s->n_buffers++] = p;
    if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RT_OK;
}

template <typename T>
int ensure(T** buf, size_t* cap, size_t count) {
    if (*cap >= count && *buf) return RT_OK;
    if (*buf) { HIP_TRY(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc((void**)buf, count * sizeof(T)));
    *cap = count;
    return RT_OK;
}

uint32_t fold_seed(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u); }

void camera_to_dev(const rtCamera* c, RtCameraDev* d) {
    static_assert(sizeof(rtCamera) == sizeof(RtCameraDev), "camera records must match");
    std::memcpy(d, c, sizeof(*d));
}

uint32_t pick_chunks(uint32_t n_tiles, uint32_t n_samples) {
    // enough waves to keep 256 CUs busy for several rounds, but at least 4 pattern columns per lane
    const char* env = std::getenv("RT_AMD_CHUNKS");
    uint32_t chunks;
    if (env && std::atoi(env) > 0) chunks = (uint32_t)std::atoi(env);
    else chunks = (32768u + n_tiles - 1u) / n_tiles;
    uint32_t cap = n_samples / 4u;
    if (cap < 1u) cap = 1u;
    if (chunks > cap) chunks = cap;
    if (chunks < 1u) chunks = 1u;
    return chunks;
}

int fill_params(const float background[3], int width, int height, int max_depth, uint64_t seed, RtRenderParams* P) {
    if (width < 2 || height < 2) return rt_fail(RT_EINVAL, "width and height must be at least 2 (the reference divides by width-1 and height-1)");
    if ((uint64_t)width * (uint64_t)height > 0x7FFFFFFFull / 3ull) return rt_fail(RT_EINVAL, "image too large");
    if (max_depth < 0 || max_depth > 62) return rt_fail(RT_EINVAL, "max_depth must be in 0..62");
    if (!background) return rt_fail(RT_EINVAL, "background is null");
    std::memset(P, 0, sizeof(*P));
    P->width = width; P->height = height; P->max_depth = max_depth;
    P->tiles_x = (uint32_t)(width + 7) / 8u; P->tiles_y = (uint32_t)(height + 7) / 8u;
    P->seed32 = fold_seed(seed);
    P->background[0] = background[0]; P->background[1] = background[1]; P->background[2] = background[2];
    P->image_floats = (size_t)3 * (size_t)width * (size_t)height;
    const char* lt = std::getenv("RT_AMD_LEAF_THRESHOLD");
    const char* sh = std::getenv("RT_AMD_SHADE_THRESHOLD");
    P->leaf_threshold = lt && std::atoi(lt) > 0 ? (uint32_t)std::atoi(lt) : 16u;
    P->shade_threshold = sh && std::atoi(sh) > 0 ? (uint32_t)std::atoi(sh) : 16u;
    return RT_OK;
}

} // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(int* count) {
    if (!count) return rt_fail(RT_EINVAL, "rt_device_count: null argument");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return RT_OK;
    if (e != hipSuccess) return rt_fail(RT_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    *count = n;
    return RT_OK;
}

int rt_scene_create(const rtSceneDesc* desc, int device, rtScene** out) {
    if (!desc || !out) return rt_fail(RT_EINVAL, "rt_scene_create: null argument");
    *out = nullptr;
    HostScene H;
    int st = rt_flatten_scene(desc, &H);
    if (st != RT_OK) return st;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return rt_fail(RT_ENODEV, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return rt_fail(RT_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    rtScene* s = new (std::nothrow) rtScene();
    if (!s) return rt_fail(RT_ENOMEM, "out of host memory");
    s->device = device;
    st = RT_OK;
    if (st == RT_OK) st = upload(s, H.nodes, &s->dev.nodes);
    if (st == RT_OK) st = upload(s, H.prims, &s->dev.prims);
    if (st == RT_OK) st = upload(s, H.xforms, &s->dev.xforms);
    if (st == RT_OK) st = upload(s, H.materials, &s->dev.materials);
    if (st == RT_OK) st = upload(s, H.textures, &s->dev.textures);
    if (st == RT_OK) st = upload(s, H.tri_attr, &s->dev.tri_attr);
    if (st == RT_OK) st = upload(s, H.images, &s->dev.images);
    if (st == RT_OK) st = upload(s, H.image_data, &s->dev.image_data);
    if (st == RT_OK) st = upload(s, H.perlin_vec, &s->dev.perlin_vec);
    if (st == RT_OK) st = upload(s, H.perlin_perm, &s->dev.perlin_perm);
    if (st == RT_OK) st = upload(s, H.prim_top, &s->dev.prim_top);
    if (st == RT_OK) st = upload(s, H.prim_sub, &s->dev.prim_sub);
    if (st == RT_OK) {
        hipError_t e2 = hipMalloc((void**)&s->d_counters, 8 * sizeof(unsigned long long));
        if (e2 != hipSuccess) st = rt_fail(RT_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e2));
    }
    if (st != RT_OK) { rt_scene_destroy(s); return st; }
    s->dev.n_nodes = (uint32_t)(H.nodes.size() / 2);
    s->dev.n_prims = (uint32_t)(H.prims.size() / 3);
    s->dev.surface_root = H.surface_root;
    s->dev.medium_root = H.medium_root;
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(rtScene* s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->device);
    for (int i = 0; i < s->n_buffers; ++i) (void)hipFree(s->buffers[i]);
    if (s->d_pattern) (void)hipFree(s->d_pattern);
    if (s->d_partial) (void)hipFree(s->d_partial);
    if (s->d_accum) (void)hipFree(s->d_accum);
    if (s->d_counters) (void)hipFree(s->d_counters);
    delete s;
    return RT_OK;
}

int rt_render_accumulate(rtScene* s, const rtCamera* camera, const float* pattern, int n, int sample_first,
                         int sample_stride, const float background[3], int width, int height, int max_depth,
                         uint64_t seed, uint32_t flags, float* d_accum, void* stream_) {
    if (!s || !camera || !pattern || !d_accum) return rt_fail(RT_EINVAL, "rt_render_accumulate: null argument");
    if (n <= 0 || sample_first < 0 || sample_first >= n || sample_stride <= 0)
        return rt_fail(RT_EINVAL, "rt_render_accumulate: need n > 0, 0 <= sample_first < n, sample_stride > 0");
    RtRenderParams P;
    int st = fill_params(background, width, height, max_depth, seed, &P);
    if (st != RT_OK) return st;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(s->device));
    P.sample_first = sample_first;
    P.sample_stride = sample_stride;
    P.n_shard_samples = (uint32_t)((n - sample_first + sample_stride - 1) / sample_stride);
    P.n_chunks = pick_chunks(P.tiles_x * P.tiles_y, P.n_shard_samples);
    if ((uint64_t)P.tiles_x * P.tiles_y * P.n_chunks > 0x3FFFFFFFull) return rt_fail(RT_EINVAL, "launch too large");
    st = ensure(&s->d_pattern, &s->pattern_cap, (size_t)2 * (size_t)n);
    if (st != RT_OK) return st;
    st = ensure(&s->d_partial, &s->partial_cap, P.image_floats * P.n_chunks);
    if (st != RT_OK) return st;
    HIP_TRY(hipMemcpyAsync(s->d_pattern, pattern, (size_t)2 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
    bool count = (flags & RT_FLAG_COUNTERS) != 0;
    if (count) HIP_TRY(hipMemsetAsync(s->d_counters, 0, 8 * sizeof(unsigned long long), stream));
    RtCameraDev cam;
    camera_to_dev(camera, &cam);
    rt_launch_render(&s->dev, &cam, s->d_pattern, &P, s->d_partial, s->d_counters, count, stream);
    HIP_TRY(hipGetLastError());
    rt_launch_reduce(s->d_partial, P.n_chunks, P.image_floats, d_accum, stream);
    HIP_TRY(hipGetLastError());
    s->last_stream = stream;
    s->last_paths = (uint64_t)width * (uint64_t)height * (uint64_t)P.n_shard_samples;
    return RT_OK;
}

int rt_finalize(rtScene* s, const float* d_accum, int n_total, int width, int height, float* d_out, void* stream_) {
    if (!s || !d_accum || !d_out) return rt_fail(RT_EINVAL, "rt_finalize: null argument");
    if (n_total <= 0 || width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_finalize: bad size");
    HIP_TRY(hipSetDevice(s->device));
    float inv_n = (float)(1.0 / (double)n_total); // a / b::Int = a * Float32(1/b), src/vec3.jl:31
    rt_launch_finalize(d_accum, inv_n, (size_t)3 * (size_t)width * (size_t)height, d_out, (hipStream_t)stream_);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_get_counters(rtScene* s, rtStats* stats) {
    if (!s || !stats) return rt_fail(RT_EINVAL, "rt_get_counters: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    std::memset(stats, 0, sizeof(*stats));
    stats->paths = s->last_paths;
    stats->rays = c[0];
    stats->node_visits = c[1];
    stats->prim_tests = c[2];
    stats->node_wave_steps = c[3];
    stats->prim_wave_steps = c[4];
    stats->bounce_wave_steps = c[5];
    return RT_OK;
}

int rt_render(rtScene* s, const rtCamera* camera, const float* pattern, int n, const float background[3], int width,
              int height, int max_depth, uint64_t seed, uint32_t flags, float* out, rtStats* stats) {
    if (!s || !out) return rt_fail(RT_EINVAL, "rt_render: null argument");
    if (width < 2 || height < 2) return rt_fail(RT_EINVAL, "width and height must be at least 2");
    HIP_TRY(hipSetDevice(s->device));
    size_t image_floats = (size_t)3 * (size_t)width * (size_t)height;
    int st = ensure(&s->d_accum, &s->accum_cap, image_floats);
    if (st != RT_OK) return st;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    st = rt_render_accumulate(s, camera, pattern, n, 0, 1, background, width, height, max_depth, seed, flags, s->d_accum, nullptr);
    if (st == RT_OK) st = rt_finalize(s, s->d_accum, n, width, height, s->d_accum, nullptr);
    if (st != RT_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return st; }
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipMemcpy(out, s->d_accum, image_floats * sizeof(float), hipMemcpyDeviceToHost));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        if (flags & RT_FLAG_COUNTERS) {
            st = rt_get_counters(s, stats);
            if (st != RT_OK) return st;
        } else {
            std::memset(stats, 0, sizeof(*stats));
            stats->paths = (uint64_t)width * (uint64_t)height * (uint64_t)n;
        }
        stats->seconds = (double)ms * 1e-3;
    }
    return RT_OK;
}

int rt_trace_rays(rtScene* s, const float* rays, int n, float t_min, float t_max, uint64_t seed, int32_t* prim_id,
                  int32_t* sub_id, float* t, float* normal, float* uv, uint8_t* front) {
    if (!s || (n > 0 && !rays)) return rt_fail(RT_EINVAL, "rt_trace_rays: null argument");
    if (n < 0) return rt_fail(RT_EINVAL, "rt_trace_rays: negative count");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    size_t N = (size_t)n;
    // one device block: rays(7) prim(1) sub(1) t(1) normal(3) uv(2) as 4-byte words, then the front bytes
    char* d = nullptr;
    size_t words = N * 15;
    HIP_TRY(hipMalloc((void**)&d, words * 4 + N));
    float* d_rays = (float*)d;
    int32_t* d_prim = (int32_t*)(d_rays + 7 * N);
    int32_t* d_sub = d_prim + N;
    float* d_t = (float*)(d_sub + N);
    float* d_normal = d_t + N;
    float* d_uv = d_normal + 3 * N;
    uint8_t* d_front = (uint8_t*)(d_uv + 2 * N);
    int st = RT_OK;
    hipError_t e = hipMemcpy(d_rays, rays, 7 * N * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rt_launch_trace_rays(&s->dev, d_rays, n, t_min, t_max, fold_seed(seed), d_prim, d_sub, d_t, d_normal, d_uv, d_front, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && prim_id) e = hipMemcpy(prim_id, d_prim, N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && sub_id) e = hipMemcpy(sub_id, d_sub, N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && t) e = hipMemcpy(t, d_t, N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && normal) e = hipMemcpy(normal, d_normal, 3 * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && uv) e = hipMemcpy(uv, d_uv, 2 * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && front) e = hipMemcpy(front, d_front, N, hipMemcpyDeviceToHost);
    if (e != hipSuccess) st = rt_fail(RT_EHIP, "rt_trace_rays: %s", hipGetErrorString(e));
    (void)hipFree(d);
    return st;
}

int rt_trace_paths(rtScene* s, const rtCamera* camera, const float* pattern, int n, const float background[3],
                   int width, int height, int max_depth, uint64_t seed, const int32_t* ids, int n_paths,
                   float* radiance) {
    if (!s || !camera || !pattern || (n_paths > 0 && (!ids || !radiance))) return rt_fail(RT_EINVAL, "rt_trace_paths: null argument");
    if (n <= 0 || n_paths < 0) return rt_fail(RT_EINVAL, "rt_trace_paths: bad count");
    if (n_paths == 0) return RT_OK;
    RtRenderParams P;
    int st = fill_params(background, width, height, max_depth, seed, &P);
    if (st != RT_OK) return st;
    for (int k = 0; k < n_paths; ++k)
        if (ids[3 * k] < 0 || ids[3 * k] >= width || ids[3 * k + 1] < 0 || ids[3 * k + 1] >= height || ids[3 * k + 2] < 0 || ids[3 * k + 2] >= n)
            return rt_fail(RT_EINVAL, "rt_trace_paths: path id %d out of range", k);
    HIP_TRY(hipSetDevice(s->device));
    st = ensure(&s->d_pattern, &s->pattern_cap, (size_t)2 * (size_t)n);
    if (st != RT_OK) return st;
    size_t N = (size_t)n_paths;
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, N * 6 * 4));
    int32_t* d_ids = (int32_t*)d;
    float* d_rad = (float*)(d_ids + 3 * N);
    hipError_t e = hipMemcpy(s->d_pattern, pattern, (size_t)2 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ids, ids, 3 * N * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        RtCameraDev cam;
        camera_to_dev(camera, &cam);
        rt_launch_trace_paths(&s->dev, &cam, s->d_pattern, &P, d_ids, n_paths, d_rad, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(radiance, d_rad, 3 * N * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) st = rt_fail(RT_EHIP, "rt_trace_paths: %s", hipGetErrorString(e));
    (void)hipFree(d);
    return st;
}

} // extern "C"
