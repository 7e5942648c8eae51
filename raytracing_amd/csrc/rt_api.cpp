// rt_api.cpp -- the extern "C" entry points of librt_amd.so (include/rt_amd.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
    HostScene host;              // the staged scene: its leaves and tables stay, for node sets built for a wider reach
    uint4* d_nodes = nullptr;    // room for two node sets
    uint4* d_pairs = nullptr;    // the tree of surfaces as pair nodes, room for two sets (null: the scene has none)
    AccelDeviceStats build_stats{0u, 0u, 0.0f}; // of the last build of the trees on the device
    AccelDeviceWorkspace* build_ws = nullptr;   // device memory of the tree builder, kept for rebuilds
    void* buffers[24] = {nullptr};
    int n_buffers = 0;
    // scratch owned by the handle, grown on demand
    float* d_pattern = nullptr; size_t pattern_cap = 0;
    float4* d_radiance = nullptr; size_t radiance_cap = 0; // one entry per path of the batch in flight
    float4* d_pool = nullptr;   size_t pool_cap = 0;   // path state planes of the wavefront renderer
    uint32_t* d_redo = nullptr; size_t redo_cap = 0;   // slots on hold (PathPool::redo)
    float4* d_pool2 = nullptr;  size_t pool2_cap = 0;  // second pool, for the compaction at the end of a frame
    uint32_t* d_counts = nullptr; size_t counts_cap = 0; // ... and its counts per 256 slots
    uint32_t* d_sync = nullptr;                        // [0] slot dispenser head, [1..] active-slot counts per bounce
    uint32_t* h_sync = nullptr;                        // pinned mirror of the activity flag
    uint32_t h_init[RT_SYNC_WORDS];                    // staging of the initial sync array
    int n_cus = 0;
    float* d_accum = nullptr;   size_t accum_cap = 0;
    unsigned long long* d_counters = nullptr;
    uint64_t last_paths = 0;
    hipEvent_t* events = nullptr;                      // 3 per bounce of a batch, RT_FLAG_TIMING only
    uint32_t n_events = 0;                             // ... of which this many have been created
    hipStream_t side = nullptr;                        // rt_redo_kernel runs here, beside the first extend kernel of a batch
    hipEvent_t fork = nullptr, join = nullptr;
    double last_extend_s = 0.0, last_shade_s = 0.0;
    uint64_t last_extend_launches = 0, last_shade_launches = 0;
};

#define SYNC_WORDS RT_SYNC_WORDS // word 0: slot dispenser; words 2..: per-bounce activity flags; then the path dispensers
#define BOUNCE_BATCH 16u   // bounces enqueued between two looks at the active count
#define RT_LARGE_POOL_SLOTS (8u << 20) // a pool with more slots than this is looked at more often (render_frame)
#define RT_TAIL_SMALL_POOL (64u << 10) // ... and one that is running empty, until it has shrunk to this
#define RT_POOL_SLOTS (4u << 20)        // slots of the path pool: frames of fewer than 4 * RT_POOL_SLOTS_LARGE paths
#define RT_POOL_SLOTS_LARGE (32u << 20) // ... and longer frames (rt_render_accumulate)

namespace {

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v && std::atoi(v) > 0 ? std::atoi(v) : fallback;
}

// the node sets of s->host into d_nodes (allocated once, for two sets) and what the kernels need to know about them
int upload_nodes(rtScene* s) {
    const HostScene& H = s->host;
    const size_t capacity = (size_t)16 * H.nodes_per_layout;
    if (!s->d_nodes) HIP_TRY(hipMalloc((void**)&s->d_nodes, capacity ? capacity * sizeof(uint4) : 16));
    if (!s->d_pairs && H.pairs_per_layout) HIP_TRY(hipMalloc((void**)&s->d_pairs, (size_t)32 * H.pairs_per_layout * sizeof(uint4)));
    uint32_t depth = H.tree_depth;
    if (!H.host_trees) {
        // every node of every layout is written by the build: the trees of a layout fill it without gaps
        int st = RT_OK;
        if (!s->build_ws) st = accel_device_workspace_create(H.set_items[0].size(), H.device_trees.size(), &s->build_ws);
        if (st == RT_OK)
            st = accel_device_build(s->build_ws, H.set_items, H.n_sets, H.device_trees, H.nodes_per_layout, H.node_scale, s->d_nodes,
                                    H.pairs_per_layout, s->d_pairs, &s->build_stats);
        if (st != RT_OK) return st;
        depth = s->build_stats.max_depth;
    } else {
        if (H.nodes.size() != (size_t)H.n_sets * 8u * H.nodes_per_layout || H.nodes.size() > capacity)
            return rt_fail(RT_EINVAL, "internal: node sets of unexpected size");
        if (!H.nodes.empty()) HIP_TRY(hipMemcpy(s->d_nodes, H.nodes.data(), H.nodes.size() * sizeof(uint4), hipMemcpyHostToDevice));
        if (H.pairs.size() != (size_t)H.n_sets * 16u * H.pairs_per_layout) return rt_fail(RT_EINVAL, "internal: pair nodes of unexpected size");
        if (!H.pairs.empty()) HIP_TRY(hipMemcpy(s->d_pairs, H.pairs.data(), H.pairs.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    // rt_extend_kernel walks the pair nodes with a stack in LDS when the scene has them and the tree is no deeper than
    // the stack (a lane pushes at most one entry per level)
    s->dev.pairs = s->d_pairs;
    s->dev.n_pairs = H.pairs_per_layout;
    // (and when a record's byte offset fits 32 bits: 16 layouts of 32-byte records)
    s->dev.use_pairs = (H.pairs_per_layout && H.pairs_per_layout < (1u << 23) && depth <= RT_PAIR_STACK_DEEP && !std::getenv("RT_AMD_NO_PAIRS")) ? 1u : 0u;
    s->dev.pair_depth = depth;
    s->dev.nodes = s->d_nodes;
    s->dev.n_nodes = H.nodes_per_layout;
    s->dev.far_set = H.n_sets == 2 ? 8u * H.nodes_per_layout : 0u;
    for (int i = 0; i < 3; ++i) s->dev.centre[i] = H.centre[i];
    for (int i = 0; i < 3; ++i) s->dev.middle[i] = H.middle[i];
    s->dev.radius2 = H.radius * H.radius;
    return RT_OK;
}

// Makes sure the node boxes hold for rays that start up to `needed` from the origin (|coordinate|): the boxes of
// objects taken out of nested trees are padded in proportion to that (rt_build_nodes).  Scenes are built for rays that
// start inside them; a camera far outside asks for more, and a later call from close by gets the tighter boxes back.
int ensure_reach(rtScene* s, double needed) {
    HostScene& H = s->host;
    if (!(needed >= 0.0) || !std::isfinite(needed)) needed = 0.0;
    const double target = 2.0 * std::max(needed, H.scene_reach);
    if (H.reach >= needed && H.reach <= 64.0 * target) return RT_OK;
    HIP_TRY(hipDeviceSynchronize());
    rt_build_nodes(&H, target);
    return upload_nodes(s);
}

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

// largest |coordinate| a camera ray can start from (src/camera.jl:38-42: origin + lens_radius * (u, v combination))
double camera_reach(const rtCamera* c) {
    double m = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double v = std::fabs((double)c->origin[i]) + std::fabs((double)c->lens_radius) * (std::fabs((double)c->u[i]) + std::fabs((double)c->v[i]));
        if (std::isfinite(v)) m = std::max(m, v);
    }
    return m;
}

uint32_t fold_seed(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u); }

void camera_to_dev(const rtCamera* c, RtCameraDev* d) {
    static_assert(sizeof(rtCamera) == sizeof(RtCameraDev), "camera records must match");
    std::memcpy(d, c, sizeof(*d));
}

int fill_params(const float background[3], int width, int height, int max_depth, uint64_t seed, RtRenderParams* P) {
    if (width < 2 || height < 2) return rt_fail(RT_EINVAL, "width and height must be at least 2 (the reference divides by width-1 and height-1)");
    if ((uint64_t)width * (uint64_t)height > 0x7FFFFFFFull / 3ull) return rt_fail(RT_EINVAL, "image too large");
    if (max_depth < 0 || max_depth > 62) return rt_fail(RT_EINVAL, "max_depth must be in 0..62");
    if (!background) return rt_fail(RT_EINVAL, "background is null");
    std::memset(P, 0, sizeof(*P));
    P->width = width; P->height = height; P->max_depth = max_depth;
    P->n_pixels = (uint32_t)width * (uint32_t)height;
    P->seed32 = fold_seed(seed);
    P->background[0] = background[0]; P->background[1] = background[1]; P->background[2] = background[2];
    P->image_floats = (size_t)3 * (size_t)width * (size_t)height;
    P->leaf_threshold = (uint32_t)env_int("RT_AMD_LEAF_THRESHOLD", 24);
    P->blocked_threshold = (uint32_t)env_int("RT_AMD_BLOCKED_THRESHOLD", 20);
    P->fetch_threshold = (uint32_t)env_int("RT_AMD_FETCH_THRESHOLD", 28);
    P->step_dropout = (uint32_t)env_int("RT_AMD_STEP_DROPOUT", 12);
    P->count_mode = (uint32_t)env_int("RT_AMD_COUNT_MODE", 0);
    P->box_spread = std::getenv("RT_AMD_BOX_SPREAD") ? (uint32_t)std::min(10, std::atoi(std::getenv("RT_AMD_BOX_SPREAD"))) : 10u;
    P->regroup = std::getenv("RT_AMD_NO_REGROUP") ? 0u : 1u;
    if (!std::getenv("RT_AMD_NO_BIN")) P->regroup |= 2u;
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

static int static_scratch(int device, unsigned char** out);

int rt_scene_create(const rtSceneDesc* desc, int device, rtScene** out) {
    if (!desc || !out) return rt_fail(RT_EINVAL, "rt_scene_create: null argument");
    *out = nullptr;
    rtScene* s = new (std::nothrow) rtScene();
    if (!s) return rt_fail(RT_ENOMEM, "out of host memory");
    HostScene& H = s->host;
    H.host_trees = std::getenv("RT_AMD_HOST_BUILD") != nullptr; // default: the trees are built on the device
    int st = rt_flatten_scene(desc, &H);
    if (st != RT_OK) { delete s; return st; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) { delete s; return rt_fail(RT_ENODEV, "no HIP device available (%s)", hipGetErrorString(e)); }
    if (device < 0 || device >= n) { delete s; return rt_fail(RT_EINVAL, "device %d out of range (0..%d)", device, n - 1); }
    e = hipSetDevice(device);
    if (e != hipSuccess) { delete s; return rt_fail(RT_EHIP, "hipSetDevice failed: %s", hipGetErrorString(e)); }
    s->device = device;
    {   // The address of the static scratch array on this device is asked for here, at set-up: the runtime brings the
        // array's code object to a device when something of it is first used there (and takes the array's 53 MiB then),
        // so that rt_render_accumulate_static finds it resident and takes no device memory even on its first call.  A
        // failure is not this call's: the static entry point reports it if it is ever used.
        unsigned char* unused = nullptr;
        (void)static_scratch(device, &unused);
    }
    st = upload_nodes(s);
    if (st == RT_OK) st = upload(s, H.prims, &s->dev.prims);
    if (st == RT_OK) st = upload(s, H.prim_inv, &s->dev.prim_inv);
    if (st == RT_OK) st = upload(s, H.prim_box, &s->dev.prim_box);
    if (st == RT_OK) st = upload(s, H.bucket_base, &s->dev.bucket_base);
    if (st == RT_OK) st = upload(s, H.xforms, &s->dev.xforms);
    if (st == RT_OK) st = upload(s, H.materials, &s->dev.materials);
    if (st == RT_OK) st = upload(s, H.textures, &s->dev.textures);
    if (st == RT_OK) st = upload(s, H.tri_attr, &s->dev.tri_attr);
    if (st == RT_OK) st = upload(s, H.images, &s->dev.images);
    if (st == RT_OK) st = upload(s, H.image_data, &s->dev.image_data);
    if (st == RT_OK) st = upload(s, H.perlin_vec, &s->dev.perlin_vec);
    if (st == RT_OK) st = upload(s, H.perlin_perm, &s->dev.perlin_perm);
    if (st == RT_OK) st = upload(s, H.media, &s->dev.media);
    if (st == RT_OK) st = upload(s, H.media_group, &s->dev.media_group);
    if (st == RT_OK) st = upload(s, H.prim_top, &s->dev.prim_top);
    if (st == RT_OK) st = upload(s, H.prim_sub, &s->dev.prim_sub);
    if (st == RT_OK) {
        hipError_t e2 = hipMalloc((void**)&s->d_counters, 8 * sizeof(unsigned long long));
        if (e2 != hipSuccess) st = rt_fail(RT_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e2));
    }
    if (st == RT_OK) {
        hipError_t e3 = hipMalloc((void**)&s->d_sync, SYNC_WORDS * sizeof(uint32_t));
        if (e3 == hipSuccess) e3 = hipHostMalloc((void**)&s->h_sync, SYNC_WORDS * sizeof(uint32_t), hipHostMallocDefault);
        hipDeviceProp_t prop;
        if (e3 == hipSuccess) e3 = hipGetDeviceProperties(&prop, device);
        if (e3 != hipSuccess) st = rt_fail(RT_ENOMEM, "scene scratch allocation failed: %s", hipGetErrorString(e3));
        else s->n_cus = prop.multiProcessorCount;
    }
    if (st != RT_OK) { rt_scene_destroy(s); return st; }
    s->dev.node_scale = H.node_scale;
    s->dev.inv_node_scale = 1.0f / H.node_scale;
    s->dev.n_prims = (uint32_t)(H.prims.size() / 3);
    s->dev.surface_root = H.surface_root;
    s->dev.surface_size = H.surface_size;
    s->dev.n_media = (uint32_t)H.media.size();
    s->dev.n_buckets = (uint32_t)H.bucket_base.size();
    s->dev.first_group_base = H.bucket_base.empty() ? 0u : H.bucket_base[0];
    s->dev.nested = H.nested ? 1u : 0u;
    s->dev.textured = H.textured ? 1u : 0u;
    s->dev.classes = 0u;
    for (size_t k = 0; k + 2 < H.prims.size(); k += 3) {
        const uint32_t head = H.prims[k].x, kind = head & 15u;
        if (head & DP_MEDIUM_FLAG) continue; // a medium's boundary is tested by the shade kernel, not by the walk
        if (kind == DP_SPHERE) s->dev.classes |= RT_CL_SPHERE;
        else if (kind == DP_MOVING_SPHERE) s->dev.classes |= RT_CL_SPHERE | RT_CL_MOVING;
        else if (kind >= DP_XY_RECT && kind <= DP_YZ_RECT) s->dev.classes |= RT_CL_RECT;
        else if (kind == DP_BOX) s->dev.classes |= RT_CL_BOX;
        else if (kind == DP_TRIANGLE) s->dev.classes |= RT_CL_TRI;
        else s->dev.classes |= RT_CL_ALL; // (a nested tree: its kernel variant carries every class)
        if (kind != DP_TRIANGLE && (head >> 8)) s->dev.classes |= RT_CL_XFORM;
    }
    if (std::getenv("RT_AMD_ALL_CLASSES")) s->dev.classes = RT_CL_ALL; // the general kernel whatever the scene holds (measurements, tests)
    *out = s;
    return RT_OK;
}

int rt_scene_layout(const rtSceneDesc* desc, rtLayout* info, uint32_t* nodes, size_t node_capacity, int32_t* prim_top,
                    size_t prim_capacity) {
    if (!desc || !info) return rt_fail(RT_EINVAL, "rt_scene_layout: null argument");
    HostScene H;
    int st = rt_flatten_scene(desc, &H);
    if (st != RT_OK) return st;
    info->nodes_per_layout = H.nodes_per_layout;
    info->n_prims = (uint32_t)(H.prims.size() / 3);
    info->surface_root = H.surface_root; info->surface_size = H.surface_size;
    info->medium_root = H.medium_root; info->medium_size = H.medium_size;
    info->nested = H.nested ? 1u : 0u;
    info->node_scale = H.node_scale;
    info->n_sets = H.n_sets;
    for (int i = 0; i < 3; ++i) info->centre[i] = H.centre[i];
    info->radius = H.radius;
    if (nodes && node_capacity >= H.nodes.size()) std::memcpy(nodes, H.nodes.data(), H.nodes.size() * sizeof(uint4));
    if (prim_top && prim_capacity >= H.prim_top.size()) std::memcpy(prim_top, H.prim_top.data(), H.prim_top.size() * sizeof(int32_t));
    return RT_OK;
}

int rt_scene_layout_pairs(const rtSceneDesc* desc, uint32_t* pairs_per_layout, uint32_t* depth, uint32_t* pairs, size_t capacity) {
    if (!desc || !pairs_per_layout) return rt_fail(RT_EINVAL, "rt_scene_layout_pairs: null argument");
    HostScene H;
    int st = rt_flatten_scene(desc, &H);
    if (st != RT_OK) return st;
    *pairs_per_layout = H.pairs_per_layout;
    if (depth) *depth = H.tree_depth;
    if (pairs && capacity >= H.pairs.size() / 2u) std::memcpy(pairs, H.pairs.data(), H.pairs.size() * sizeof(uint4));
    return RT_OK;
}

int rt_scene_pairs(rtScene* s, uint32_t* pairs_per_layout, uint32_t* in_use, uint32_t* pairs, size_t capacity) {
    if (!s || !pairs_per_layout) return rt_fail(RT_EINVAL, "rt_scene_pairs: null argument");
    const HostScene& H = s->host;
    *pairs_per_layout = H.pairs_per_layout;
    if (in_use) *in_use = s->dev.use_pairs;
    const size_t n_records = (size_t)H.n_sets * 8u * H.pairs_per_layout;
    if (pairs && capacity >= n_records && n_records) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(pairs, s->d_pairs, n_records * 2u * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_scene_nodes(rtScene* s, rtLayout* info, uint32_t* nodes, size_t node_capacity, int32_t* prim_top, size_t prim_capacity,
                   rtBuildInfo* build) {
    if (!s || !info) return rt_fail(RT_EINVAL, "rt_scene_nodes: null argument");
    const HostScene& H = s->host;
    info->nodes_per_layout = H.nodes_per_layout;
    info->n_prims = (uint32_t)(H.prims.size() / 3);
    info->surface_root = H.surface_root; info->surface_size = H.surface_size;
    info->medium_root = H.medium_root; info->medium_size = H.medium_size;
    info->nested = H.nested ? 1u : 0u;
    info->node_scale = H.node_scale;
    info->n_sets = H.n_sets;
    for (int i = 0; i < 3; ++i) info->centre[i] = H.centre[i];
    info->radius = H.radius;
    const size_t n_nodes = (size_t)H.n_sets * 8u * H.nodes_per_layout;
    if (nodes && node_capacity >= n_nodes && n_nodes) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(nodes, s->d_nodes, n_nodes * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    if (prim_top && prim_capacity >= H.prim_top.size()) std::memcpy(prim_top, H.prim_top.data(), H.prim_top.size() * sizeof(int32_t));
    if (build) {
        build->on_device = H.host_trees ? 0u : 1u;
        build->n_trees = s->build_stats.n_trees;
        build->max_depth = s->build_stats.max_depth;
        build->build_ms = s->build_stats.build_ms;
    }
    return RT_OK;
}

int rt_scene_destroy(rtScene* s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->device);
    for (int i = 0; i < s->n_buffers; ++i) (void)hipFree(s->buffers[i]);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_pairs) (void)hipFree(s->d_pairs);
    accel_device_workspace_destroy(s->build_ws);
    if (s->d_pattern) (void)hipFree(s->d_pattern);
    if (s->d_radiance) (void)hipFree(s->d_radiance);
    if (s->d_pool) (void)hipFree(s->d_pool);
    if (s->d_redo) (void)hipFree(s->d_redo);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->fork) (void)hipEventDestroy(s->fork);
    if (s->join) (void)hipEventDestroy(s->join);
    if (s->d_pool2) (void)hipFree(s->d_pool2);
    if (s->d_counts) (void)hipFree(s->d_counts);
    if (s->d_sync) (void)hipFree(s->d_sync);
    if (s->h_sync) (void)hipHostFree(s->h_sync);
    if (s->events) {
        for (uint32_t i = 0; i < s->n_events; ++i) (void)hipEventDestroy(s->events[i]);
        delete[] s->events;
    }
    if (s->d_accum) (void)hipFree(s->d_accum);
    if (s->d_counters) (void)hipFree(s->d_counters);
    delete s;
    return RT_OK;
}

// ---- the frame loop shared by rt_render_accumulate (buffers owned by the handle, grown on demand) and
// rt_render_accumulate_static (buffers carved out of the file-scope scratch array; nothing is allocated)
struct Workspace {
    float* pattern;          // 2 * n floats
    float4* pool;            // 5 planes of n_slots entries
    uint32_t* redo;          // n_slots entries
    float4* radiance;        // cols * n_pixels entries
    uint32_t* sync;          // SYNC_WORDS words
    uint32_t n_slots;        // slots of the path pool
    uint32_t cols;           // pattern columns per batch
    bool may_allocate;       // end-of-frame compaction (second pool), side stream, timing events: only where allocating is allowed
};

// n_slots and cols for a frame: slots_wanted slots (a multiple of 16 Ki, so that the paths the slots start with cover
// whole rounds of the path dispensers; fewer if the frame has fewer paths) and as many pattern columns per batch as
// radiance_entries float4 entries hold.
static int plan_frame(const RtRenderParams& P, uint32_t n_shard_samples, uint32_t slots_wanted, uint64_t radiance_entries,
                      uint32_t* n_slots, uint32_t* cols_out) {
    const uint64_t total_paths = (uint64_t)P.n_pixels * n_shard_samples;
    const uint32_t slot_quantum = RT_DISP_UNIT * RT_DISP_MAX;
    uint32_t slots = slots_wanted / slot_quantum * slot_quantum;
    if (slots < slot_quantum) slots = slot_quantum;
    if ((uint64_t)slots > total_paths) slots = (uint32_t)((total_paths + 1023ull) / 1024ull * 1024ull);
    uint64_t cols = radiance_entries / (uint64_t)P.n_pixels;
    uint64_t max_cols = (0xFFFFFFFFull - slots - 4096ull) / P.n_pixels; // path numbers are 32-bit
    if (cols > max_cols) cols = max_cols;
    if (cols > 0xFFFFFFull) cols = 0xFFFFFFull; // the column inside the batch travels in 24 bits of the depth word
    if (cols > n_shard_samples) cols = n_shard_samples;
    if (cols < 1) return rt_fail(RT_EINVAL, "image too large for the radiance buffer");
    *n_slots = slots;
    *cols_out = (uint32_t)cols;
    return RT_OK;
}

static int render_frame(rtScene* s, const rtCamera* camera, const float* pattern, int n, RtRenderParams& P, uint32_t n_shard_samples,
                        const Workspace& ws, uint32_t flags, float* d_accum, hipStream_t stream) {
    int st = RT_OK;
    PathPool pool;
    pool.n_slots = ws.n_slots;
    pool.q = ws.pool;
    pool.redo = ws.redo;
    const uint32_t cols = ws.cols;
    const bool overlap = ws.may_allocate && s->side && s->fork && s->join && !std::getenv("RT_AMD_NO_REDO_OVERLAP");
    HIP_TRY(hipMemcpyAsync(ws.pattern, pattern, (size_t)2 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
    bool count = (flags & RT_FLAG_COUNTERS) != 0;
    if (count) HIP_TRY(hipMemsetAsync(s->d_counters, 0, 8 * sizeof(unsigned long long), stream));
    bool timing = (flags & RT_FLAG_TIMING) != 0;
    s->last_extend_s = s->last_shade_s = 0.0;
    s->last_extend_launches = s->last_shade_launches = 0;
    RtCameraDev cam;
    camera_to_dev(camera, &cam);
    uint32_t extend_blocks = (uint32_t)(s->n_cus > 0 ? s->n_cus : 256) * (uint32_t)env_int("RT_AMD_EXTEND_BLOCKS_PER_CU", 6);
    const uint32_t frame_slots = pool.n_slots;
    for (uint32_t k0 = 0; k0 < n_shard_samples; k0 += cols) {
        pool.n_slots = frame_slots;
        pool.q = ws.pool;
        bool tail = false; // the device reported that no unstarted path is left
        P.k_begin = k0;
        P.k_end = k0 + cols < n_shard_samples ? k0 + cols : n_shard_samples;
        P.n_paths = (P.k_end - P.k_begin) * P.n_pixels;
        // sync[0] = slot dispenser, sync[2..] = activity flag of each bounce, then the path dispensers.  The slots
        // start with paths 0 .. n_slots-1, which is the same number of whole units from every dispenser.
        const uint32_t shade_blocks = (pool.n_slots + 1023u) / 1024u;
        P.n_disp = shade_blocks < RT_DISP_MAX ? shade_blocks : RT_DISP_MAX;
        const uint32_t n_units = (P.n_paths + RT_DISP_UNIT - 1u) / RT_DISP_UNIT;
        P.disp_limit = (n_units + P.n_disp - 1u) / P.n_disp * RT_DISP_UNIT;
        uint32_t disp_start = P.disp_limit; // nothing left to hand out
        if (P.n_paths > pool.n_slots) disp_start = pool.n_slots / P.n_disp;
        std::memset(s->h_init, 0, SYNC_WORDS * sizeof(uint32_t));
        for (uint32_t c = 0; c < RT_DISP_MAX; ++c) s->h_init[RT_DISP_BASE + c * RT_DISP_STRIDE] = disp_start;
        HIP_TRY(hipMemcpyAsync(ws.sync, s->h_init, SYNC_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        // with max_depth == 0 trace() returns black without looking at the scene: no path ever writes its radiance
        if (P.max_depth == 0) HIP_TRY(hipMemsetAsync(ws.radiance, 0, (size_t)P.n_paths * sizeof(float4), stream));
        rt_launch_init(&cam, ws.pattern, &P, &pool, stream);
        HIP_TRY(hipGetLastError());
        // Bounce loop: the device reports after every bounce how many slots still hold a ray; the host enqueues
        // bounces in batches and stops after the first batch that ends with none.
        // Bounces enqueued before the host looks again (and, once the unstarted paths have run out, compacts the pool).
        // While paths are left: 16, or 8 while the pool is large (a launch over 32 Mi slots takes milliseconds, so the look
        // costs nothing).  Once they have run out, a fifth of the slots empties with every bounce, and a launch over
        // emptied slots costs nearly what one over full slots does (the kernels still visit them): the pool is then looked
        // at and compacted every 4 bounces until it is small.  To get there in time, the batch in which the paths will run
        // out is cut to end where they do: the dispensers' counters say how many are left, the batch before how many a
        // bounce starts.
        const uint32_t batch_large = (uint32_t)std::min(std::max(env_int("RT_AMD_LARGE_POOL_BATCH", 8), 1), (int)BOUNCE_BATCH);
        const uint32_t batch_tail = (uint32_t)std::min(std::max(env_int("RT_AMD_TAIL_BATCH", 4), 1), (int)BOUNCE_BATCH);
        uint64_t unstarted = P.n_paths > pool.n_slots ? (uint64_t)P.n_paths - pool.n_slots : 0ull;
        double started_per_bounce = 0.0;
        const bool predict = ws.may_allocate && unstarted != 0ull && !std::getenv("RT_AMD_NO_TAIL_PREDICTION");
        for (uint64_t bounce = 0;;) {
            uint32_t batch = pool.n_slots > RT_LARGE_POOL_SLOTS ? batch_large : BOUNCE_BATCH;
            if (predict && tail && pool.n_slots > RT_TAIL_SMALL_POOL) batch = batch_tail;
            if (predict && !tail && started_per_bounce > 0.0) {
                const double bounces_left = (double)unstarted / started_per_bounce + 1.0;
                if (bounces_left < (double)batch) batch = (uint32_t)bounces_left;
            }
            HIP_TRY(hipMemsetAsync(ws.sync + 2, 0, BOUNCE_BATCH * sizeof(uint32_t), stream));
            // The slots on hold are settled between two batches.  rt_extend_kernel leaves them alone and rt_redo_kernel
            // touches nothing else, so the two run side by side (the redo is a few lone lanes walking the tree); the
            // batch's first shade kernel waits for it.  Before a compaction, which moves the slots, it runs in line.
            const bool compacting = ws.may_allocate && tail && pool.n_slots > 4096u && !std::getenv("RT_AMD_NO_COMPACT");
            bool beside = false;
            if (bounce != 0) {
                if (!compacting && overlap) {
                    HIP_TRY(hipEventRecord(s->fork, stream));
                    HIP_TRY(hipStreamWaitEvent(s->side, s->fork, 0));
                    rt_launch_redo(&s->dev, &P, &pool, ws.sync, s->side);
                    HIP_TRY(hipEventRecord(s->join, s->side));
                    beside = true;
                } else {
                    rt_launch_redo(&s->dev, &P, &pool, ws.sync, stream);
                }
            }
            // No unstarted path left: the slots run dry over the next (up to max_depth) bounces.  The ones that still
            // hold a ray are moved together, so that the launches shrink with them.
            if (compacting) {
                st = ensure(&s->d_pool2, &s->pool2_cap, (size_t)5 * frame_slots);
                if (st == RT_OK) st = ensure(&s->d_counts, &s->counts_cap, (size_t)frame_slots / 256 + 2);
                if (st != RT_OK) return st;
                float4* to = pool.q == ws.pool ? s->d_pool2 : ws.pool;
                rt_launch_compact(&pool, to, s->d_counts, ws.sync, stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(s->h_sync + 1, ws.sync + RT_LIVE_SLOTS, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                pool.n_slots = RT_COMPACT_SLOTS(s->h_sync[1], pool.n_slots);
                pool.q = to;
            }
            for (uint32_t b = 0; b < batch; ++b)
                rt_launch_bounce(&s->dev, &cam, ws.pattern, &P, &pool, extend_blocks, ws.radiance, ws.sync,
                                 ws.sync + 2u + b, s->d_counters, count, timing ? s->events + 3u * b : nullptr, stream,
                                 (b == 0u && beside) ? s->join : nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(s->h_sync, ws.sync + 1u + batch, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(s->h_sync + 2, ws.sync + RT_TAIL_FLAG, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            if (predict && !tail)
                HIP_TRY(hipMemcpy2DAsync(s->h_sync + 8, sizeof(uint32_t), ws.sync + RT_DISP_BASE, RT_DISP_STRIDE * sizeof(uint32_t),
                                         sizeof(uint32_t), P.n_disp, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (predict && !tail) {
                uint64_t left = 0ull;
                for (uint32_t c = 0; c < P.n_disp; ++c) {
                    const uint32_t lim = rt_dispenser_limit(P.n_paths, P.n_disp, c);
                    if (s->h_sync[8u + c] < lim) left += lim - s->h_sync[8u + c];
                }
                started_per_bounce = left < unstarted ? (double)(unstarted - left) / (double)batch : 0.0;
                unstarted = left;
            }
            if (timing) {
                for (uint32_t b = 0; b < batch; ++b) {
                    float ms_e = 0.0f, ms_s = 0.0f;
                    HIP_TRY(hipEventElapsedTime(&ms_e, s->events[3u * b], s->events[3u * b + 1u]));
                    HIP_TRY(hipEventElapsedTime(&ms_s, s->events[3u * b + 1u], s->events[3u * b + 2u]));
                    s->last_extend_s += (double)ms_e * 1e-3;
                    s->last_shade_s += (double)ms_s * 1e-3;
                }
                s->last_extend_launches += batch;
                s->last_shade_launches += batch;
            }
            if (s->h_sync[0] == 0u) break;
            tail = s->h_sync[2] != 0u || (predict && unstarted == 0ull); // (the device's flag, or every counter at its limit)
            bounce += batch;
            if (bounce > ((uint64_t)P.n_paths / pool.n_slots + 2ull) * 64ull + 4096ull)
                return rt_fail(RT_EHIP, "bounce loop did not terminate");
        }
        if (P.n_paths > frame_slots) {
            // every path of the batch must have been handed to a slot: the counters are read back and held against
            // their limits (a path left behind would leave a stale entry in the radiance buffer, silently)
            HIP_TRY(hipMemcpy2DAsync(s->h_sync + 8, sizeof(uint32_t), ws.sync + RT_DISP_BASE, RT_DISP_STRIDE * sizeof(uint32_t),
                                     sizeof(uint32_t), P.n_disp, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (uint32_t c = 0; c < P.n_disp; ++c)
                if (s->h_sync[8u + c] < rt_dispenser_limit(P.n_paths, P.n_disp, c))
                    return rt_fail(RT_EHIP, "internal: path dispenser %u stopped at %u of %u", c, s->h_sync[8u + c],
                                   rt_dispenser_limit(P.n_paths, P.n_disp, c));
        }
        rt_launch_sum(&P, ws.radiance, d_accum, k0 == 0u ? 1 : 0, stream);
        HIP_TRY(hipGetLastError());
    }
    s->last_paths = (uint64_t)P.n_pixels * n_shard_samples;
    return RT_OK;
}

// checks shared by the two accumulate entry points; *empty: the shard holds no pattern column (d_accum was zeroed)
static int accumulate_prologue(const char* who, rtScene* s, const rtCamera* camera, const float* pattern, int n, int sample_first,
                               int sample_stride, const float background[3], int width, int height, int max_depth, uint64_t seed,
                               float* d_accum, hipStream_t stream, RtRenderParams* P, uint32_t* n_shard_samples, bool* empty) {
    *empty = false;
    if (!s || !camera || !pattern || !d_accum) return rt_fail(RT_EINVAL, "%s: null argument", who);
    if (n <= 0 || sample_first < 0 || sample_stride <= 0)
        return rt_fail(RT_EINVAL, "%s: need n > 0, sample_first >= 0, sample_stride > 0", who);
    int st = fill_params(background, width, height, max_depth, seed, P);
    if (st != RT_OK) return st;
    HIP_TRY(hipSetDevice(s->device));
    if (sample_first >= n) { // an empty shard (more shards than pattern columns) contributes a zero sum
        HIP_TRY(hipMemsetAsync(d_accum, 0, P->image_floats * sizeof(float), stream));
        s->last_paths = 0;
        *empty = true;
        return RT_OK;
    }
    st = ensure_reach(s, camera_reach(camera));
    if (st != RT_OK) return st;
    P->sample_first = sample_first;
    P->sample_stride = sample_stride;
    *n_shard_samples = (uint32_t)((n - sample_first + sample_stride - 1) / sample_stride);
    return RT_OK;
}

int rt_render_accumulate(rtScene* s, const rtCamera* camera, const float* pattern, int n, int sample_first,
                         int sample_stride, const float background[3], int width, int height, int max_depth,
                         uint64_t seed, uint32_t flags, float* d_accum, void* stream_) {
    RtRenderParams P;
    uint32_t n_shard_samples = 0;
    bool empty = false;
    hipStream_t stream = (hipStream_t)stream_;
    int st = accumulate_prologue("rt_render_accumulate", s, camera, pattern, n, sample_first, sample_stride, background, width,
                                 height, max_depth, seed, d_accum, stream, &P, &n_shard_samples, &empty);
    if (st != RT_OK || empty) return st;
    // Slots in flight: enough rays per launch to fill the machine many times over.  With 4 Mi slots the four live planes of
    // the path pool (64 bytes per slot) just stay inside the 256 MB Infinity Cache between the kernel that writes them
    // and the one that reads them; throughput falls by 4 % right above that.  It comes back at 12 Mi and goes beyond at 24-48
    // Mi: every launch of rt_extend_kernel ends with its waves running empty (a quarter of the step iterations of a 4 Mi
    // launch are taken after the slots ran out, with half the lanes holding a ray), and a launch eight times as long
    // has an eighth of those ends -- extend -11 %, shade +9 % (its 112 bytes per slot now come from HBM), +3 % in all.  A
    // larger pool drains for longer once the unstarted paths have run out, so a frame has to be long enough for it, and
    // the end of a launch is as long as the longest walk, so the tree has to be deep enough for there to be something to
    // win: cover1 (485 spheres under an open sky) +24 %, teapot +4 %, cu_cover2 +3 %, cover2 +-0, but the Cornell boxes
    // (a dozen primitives, walks of a few steps) -6 ... -11 % at any frame length.  32 Mi slots for frames of 128 Mi paths
    // or more (cu_cover2 800x800: 210 pattern columns) over a tree at least eight levels deep, 4 Mi otherwise.
    // The radiance of every path is kept until its pixel is summed, so the frame is cut into batches of pattern
    // columns whose radiance buffer fits the budget (default 12 GiB of the 288 GB).
    Workspace ws{};
    const uint64_t budget = (uint64_t)env_int("RT_AMD_RADIANCE_MIB", 12 * 1024) * 1024ull * 1024ull;
    const uint64_t frame_paths = (uint64_t)P.n_pixels * n_shard_samples;
    const int slots_default = frame_paths >= 4ull * RT_POOL_SLOTS_LARGE && s->dev.pair_depth >= 8u ? (int)RT_POOL_SLOTS_LARGE : (int)RT_POOL_SLOTS;
    st = plan_frame(P, n_shard_samples, (uint32_t)env_int("RT_AMD_SLOTS", slots_default), budget / sizeof(float4), &ws.n_slots, &ws.cols);
    if (st != RT_OK) return st;
    st = ensure(&s->d_pattern, &s->pattern_cap, (size_t)2 * (size_t)n);
    if (st != RT_OK) return st;
    st = ensure(&s->d_pool, &s->pool_cap, (size_t)5 * ws.n_slots);
    if (st != RT_OK) return st;
    st = ensure(&s->d_radiance, &s->radiance_cap, (size_t)ws.cols * P.n_pixels);
    if (st != RT_OK) return st;
    st = ensure(&s->d_redo, &s->redo_cap, (size_t)ws.n_slots);
    if (st != RT_OK) return st;
    if (!(s->side && s->fork && s->join) && !std::getenv("RT_AMD_NO_REDO_OVERLAP")) {
        // each handle is created once and kept; a failure part-way leaves the others as they are (null = not created)
        if (!s->side) HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        if (!s->fork) HIP_TRY(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
        if (!s->join) HIP_TRY(hipEventCreateWithFlags(&s->join, hipEventDisableTiming));
    }
    if ((flags & RT_FLAG_TIMING) && s->n_events < 3u * BOUNCE_BATCH) {
        if (!s->events) {
            s->events = new (std::nothrow) hipEvent_t[3u * BOUNCE_BATCH]();
            if (!s->events) return rt_fail(RT_ENOMEM, "out of host memory");
        }
        for (; s->n_events < 3u * BOUNCE_BATCH; ++s->n_events) HIP_TRY(hipEventCreate(&s->events[s->n_events]));
    }
    ws.pattern = s->d_pattern; ws.pool = s->d_pool; ws.redo = s->d_redo; ws.radiance = s->d_radiance; ws.sync = s->d_sync;
    ws.may_allocate = true;
    return render_frame(s, camera, pattern, n, P, n_shard_samples, ws, flags, d_accum, stream);
}

// ---- rt_render_accumulate_static: the same frame loop out of a fixed scratch array, for callers that cannot allocate.
// The array is device memory of the code object: every GPU the library runs on holds its own copy, at its own
// address, from the moment the code object is loaded there until the process ends.  Host code never uses the array's name
// as an address (on the host it is a stand-in); the address on a device is asked for once, with that device current
// (static_scratch), and kept.
#define RT_STATIC_SYNC_BYTES (((size_t)RT_SYNC_WORDS * sizeof(uint32_t) + 255u) / 256u * 256u)
#define RT_STATIC_PATTERN_BYTES ((size_t)2 * RT_STATIC_MAX_COLUMNS * sizeof(float))
#define RT_STATIC_REDO_BYTES ((size_t)RT_STATIC_SLOTS * sizeof(uint32_t))
#define RT_STATIC_POOL_BYTES ((size_t)5 * RT_STATIC_SLOTS * sizeof(float4))
#define RT_STATIC_RADIANCE_BYTES ((size_t)RT_STATIC_MAX_PIXELS * sizeof(float4))
#define RT_STATIC_SCRATCH_BYTES (RT_STATIC_SYNC_BYTES + RT_STATIC_PATTERN_BYTES + RT_STATIC_REDO_BYTES + RT_STATIC_POOL_BYTES + RT_STATIC_RADIANCE_BYTES)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

#define RT_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES]; // per device: where `scratch` lives there; null = not asked yet
static std::mutex g_scratch_mutex;

// the address of `scratch` on `device`, which must be the current device
static int static_scratch(int device, unsigned char** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) return rt_fail(RT_EINVAL, "device %d: only devices 0..%d can use the static scratch", device, RT_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return rt_fail(RT_EHIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

int rt_render_accumulate_static(rtScene* s, const rtCamera* camera, const float* pattern, int n, int sample_first,
                                int sample_stride, const float background[3], int width, int height, int max_depth,
                                uint64_t seed, float* d_accum, void* stream_) {
    if (n > (int)RT_STATIC_MAX_COLUMNS)
        return rt_fail(RT_EINVAL, "rt_render_accumulate_static: %d pattern columns, the static scratch holds %u at most", n, RT_STATIC_MAX_COLUMNS);
    if (width > 0 && height > 0 && (uint64_t)width * (uint64_t)height > RT_STATIC_MAX_PIXELS)
        return rt_fail(RT_EINVAL, "rt_render_accumulate_static: %d x %d pixels, the static scratch holds %u at most", width, height, RT_STATIC_MAX_PIXELS);
    RtRenderParams P;
    uint32_t n_shard_samples = 0;
    bool empty = false;
    hipStream_t stream = (hipStream_t)stream_;
    int st = accumulate_prologue("rt_render_accumulate_static", s, camera, pattern, n, sample_first, sample_stride, background,
                                 width, height, max_depth, seed, d_accum, stream, &P, &n_shard_samples, &empty);
    if (st != RT_OK || empty) return st;
    unsigned char* base = nullptr;
    st = static_scratch(s->device, &base); // (the device is current: accumulate_prologue set it)
    if (st != RT_OK) return st;
    Workspace ws{};
    ws.sync = (uint32_t*)base;
    ws.pattern = (float*)(base + RT_STATIC_SYNC_BYTES);
    ws.redo = (uint32_t*)(base + RT_STATIC_SYNC_BYTES + RT_STATIC_PATTERN_BYTES);
    ws.pool = (float4*)(base + RT_STATIC_SYNC_BYTES + RT_STATIC_PATTERN_BYTES + RT_STATIC_REDO_BYTES);
    ws.radiance = (float4*)(base + RT_STATIC_SYNC_BYTES + RT_STATIC_PATTERN_BYTES + RT_STATIC_REDO_BYTES + RT_STATIC_POOL_BYTES);
    ws.may_allocate = false;
    st = plan_frame(P, n_shard_samples, RT_STATIC_SLOTS, RT_STATIC_MAX_PIXELS, &ws.n_slots, &ws.cols);
    if (st != RT_OK) return st;
    // Everything the kernels expect to start from a known state is written before it is read: the sync words are copied
    // in whole before every column batch, rt_init_kernel writes every slot of the pool, the list of slots on hold is
    // counted from zero in the sync words, and every path of a batch writes its radiance entry before the sum reads it.
    return render_frame(s, camera, pattern, n, P, n_shard_samples, ws, 0u, d_accum, stream);
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

int rt_finalize_rgb8(rtScene* s, const float* d_accum, int n_total, int width, int height, uint8_t* d_rgb8, void* stream_) {
    if (!s || !d_accum || !d_rgb8) return rt_fail(RT_EINVAL, "rt_finalize_rgb8: null argument");
    if (n_total <= 0 || width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_finalize_rgb8: bad size");
    HIP_TRY(hipSetDevice(s->device));
    float inv_n = (float)(1.0 / (double)n_total);
    rt_launch_finalize_rgb8(d_accum, inv_n, width, height, d_rgb8, (hipStream_t)stream_);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_get_counters(rtScene* s, rtStats* stats) {
    if (!s || !stats) return rt_fail(RT_EINVAL, "rt_get_counters: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize()); // not the stream of the last call: the caller may have destroyed it since
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
    stats->extend_seconds = s->last_extend_s;
    stats->shade_seconds = s->last_shade_s;
    stats->extend_launches = s->last_extend_launches;
    stats->shade_launches = s->last_shade_launches;
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct EventGuard { // the two events are released on every way out
        hipEvent_t &a, &b;
        ~EventGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } guard{e0, e1};
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    st = rt_render_accumulate(s, camera, pattern, n, 0, 1, background, width, height, max_depth, seed, flags, s->d_accum, nullptr);
    if (st == RT_OK) st = rt_finalize(s, s->d_accum, n, width, height, s->d_accum, nullptr);
    if (st != RT_OK) return st;
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipMemcpy(out, s->d_accum, image_floats * sizeof(float), hipMemcpyDeviceToHost));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (stats) {
        if (flags & (RT_FLAG_COUNTERS | RT_FLAG_TIMING)) {
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
    {
        double m = 0.0;
        for (size_t k = 0; k < N; ++k)
            for (int i = 0; i < 3; ++i) {
                const double v = std::fabs((double)rays[7 * k + i]);
                if (std::isfinite(v)) m = std::max(m, v);
            }
        int st0 = ensure_reach(s, m);
        if (st0 != RT_OK) return st0;
    }
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
    st = ensure_reach(s, camera_reach(camera));
    if (st != RT_OK) return st;
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
