// rt_group.cpp -- one host process driving several GPUs: the scene replicated per device, the pattern columns dealt to
// the devices (device g of G traces columns g, g + G, ...), one host thread per device, the per-device accumulators summed
// on the first device -- by RCCL (ncclReduce over xGMI; librccl is loaded when the first multi-device group needs it) or,
// on request, by peer copies and an add kernel in device order -- then finalised and copied out.  The reference renders
// on one device from one process (main.jl:34-38); this is what lets that one call use every GPU of a node.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"
#include "rt_kernels.h"

namespace {

// the few RCCL entry points the reduction needs, bound at run time
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

const Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib || !g_rccl.error.empty()) return &g_rccl;
    // RT_AMD_RCCL_LIB names the library to load instead of the usual places (an installation elsewhere; the tests
    // name a file that does not exist to see RT_ERCCL)
    const char* named = std::getenv("RT_AMD_RCCL_LIB");
    const char* usual[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    if (named && *named) {
        g_rccl.lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.lib) { const char* e = dlerror(); why = e ? e : "unknown error"; } // (dlerror() clears what it returns: ask once)
    } else {
        for (const char* name : usual) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.lib) break;
            const char* e = dlerror();
            if (why.empty()) why = e ? e : "unknown error";
        }
    }
    if (!g_rccl.lib) {
        g_rccl.error = std::string(named && *named ? named : "librccl.so.1") + " could not be loaded: " + why;
        return &g_rccl;
    }
    bool ok = true;
    auto bind = [&ok](void* lib, const char* sym) { void* p = dlsym(lib, sym); if (!p) ok = false; return p; };
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))bind(g_rccl.lib, "ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))bind(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))bind(g_rccl.lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))bind(g_rccl.lib, "ncclGroupEnd");
    g_rccl.Reduce = (decltype(g_rccl.Reduce))bind(g_rccl.lib, "ncclReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))bind(g_rccl.lib, "ncclGetErrorString");
    if (!ok) {
        g_rccl.error = "librccl.so.1 lacks an entry point the reduction needs";
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
    }
    return &g_rccl;
}

// What differs between the two element types of render(F, ...) (main.jl:21): the records, the entry points of one
// scene on one GPU, the kernel that adds two accumulators and the RCCL data type.
struct Float32Path {
    typedef float real;
    typedef rtScene Scene;
    typedef rtSceneDesc Desc;
    typedef rtCamera Camera;
    static const char* name() { return "rt_group_render"; }
    static int create(const Desc* d, int device, Scene** s) { return rt_scene_create(d, device, s); }
    static int destroy(Scene* s) { return rt_scene_destroy(s); }
    static int accumulate(Scene* s, const Camera* c, const real* pattern, int n, int first, int stride, const real* bg, int w, int h, int depth,
                          uint64_t seed, uint32_t flags, real* accum, hipStream_t stream) {
        return rt_render_accumulate(s, c, pattern, n, first, stride, bg, w, h, depth, seed, flags & ~RT_FLAG_TIMING, accum, stream);
    }
    static int finalize(Scene* s, const real* accum, int n, int w, int h, real* out, hipStream_t stream) { return rt_finalize(s, accum, n, w, h, out, stream); }
    static void add(real* dst, const real* src, size_t n, hipStream_t stream) { rt_launch_add(dst, src, n, stream); }
    static ncclDataType_t nccl_type() { return ncclFloat; }
};
struct Float64Path {
    typedef double real;
    typedef rtScene64 Scene;
    typedef rtSceneDesc64 Desc;
    typedef rtCamera64 Camera;
    static const char* name() { return "rt_group_render_f64"; }
    static int create(const Desc* d, int device, Scene** s) { return rt_scene_create_f64(d, device, s); }
    static int destroy(Scene* s) { return rt_scene_destroy_f64(s); }
    static int accumulate(Scene* s, const Camera* c, const real* pattern, int n, int first, int stride, const real* bg, int w, int h, int depth,
                          uint64_t seed, uint32_t, real* accum, hipStream_t stream) {
        return rt_render_accumulate_f64(s, c, pattern, n, first, stride, bg, w, h, depth, seed, accum, stream);
    }
    static int finalize(Scene* s, const real* accum, int n, int w, int h, real* out, hipStream_t stream) { return rt_finalize_f64(s, accum, n, w, h, out, stream); }
    static void add(real* dst, const real* src, size_t n, hipStream_t stream) { rt64_launch_add(dst, src, n, stream); }
    static ncclDataType_t nccl_type() { return ncclDouble; }
};

// What a member's worker thread is asked to do for one frame (the arguments of rt_group_render).
template <class P>
struct Frame {
    const typename P::Camera* camera = nullptr;
    const typename P::real* pattern = nullptr;
    const typename P::real* background = nullptr;
    int n = 0, width = 0, height = 0, max_depth = 0;
    uint64_t seed = 0;
    uint32_t flags = 0;
};

template <class P>
struct Group {
    std::vector<int> devices;
    std::vector<typename P::Scene*> scenes;
    std::vector<hipStream_t> streams;
    std::vector<typename P::real*> accum; // per member: 3 * H * W numbers on its device
    std::vector<size_t> accum_cap;
    typename P::real* stage = nullptr;    // on the first device: where a peer's accumulator lands before it is added
    size_t stage_cap = 0;
    std::vector<ncclComm_t> comms;        // one per member once the group has reduced over RCCL
    bool distinct = true;                 // no device appears twice (RCCL needs that)
    hipEvent_t e0 = nullptr, e1 = nullptr; // on the first device, made once: the frame's first and last kernel
    // One worker thread per member, started with the group and parked between frames (a group of one has none: the
    // caller's thread renders).  A frame is handed over by raising `generation` under the mutex; every worker runs its
    // member's share, stores status and message, and the last one to finish wakes the caller.
    std::vector<std::thread> workers;
    std::mutex mutex;
    std::condition_variable wake, done;
    uint64_t generation = 0;
    size_t running = 0;
    bool quit = false;
    Frame<P> frame;
    std::vector<int> status;
    std::vector<std::string> message;
};

#define GROUP_HIP(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return rt_fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// a member's share of the pattern columns into its own accumulator, and the wait for it
template <class P>
void member_frame(Group<P>* g, size_t k) {
    const Frame<P>& f = g->frame;
    int st = P::accumulate(g->scenes[k], f.camera, f.pattern, f.n, (int)k, (int)g->devices.size(), f.background, f.width, f.height, f.max_depth,
                           f.seed, f.flags, g->accum[k], g->streams[k]);
    if (st == RT_OK && hipStreamSynchronize(g->streams[k]) != hipSuccess) st = rt_fail(RT_EHIP, "hipStreamSynchronize failed");
    g->status[k] = st;
    g->message[k] = st != RT_OK ? rt_last_error() : ""; // (the message lives in the thread that failed)
}

template <class P>
void worker_loop(Group<P>* g, size_t k) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(g->mutex);
            g->wake.wait(lock, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
        }
        member_frame(g, k);
        std::lock_guard<std::mutex> lock(g->mutex);
        if (--g->running == 0) g->done.notify_all();
    }
}

// Leaves no work of the group in flight: a caller that gets an error may free or reuse its buffers at once.
template <class P>
void drain(Group<P>* g) {
    for (size_t k = 0; k < g->devices.size(); ++k) {
        if (!g->streams[k]) continue;
        (void)hipSetDevice(g->devices[k]);
        (void)hipStreamSynchronize(g->streams[k]);
    }
    (void)hipGetLastError();
    (void)hipSetDevice(g->devices[0]);
}

template <class P, class H> // H: the handle type of the C ABI (a Group<P> under its public name)
int group_destroy(H* g) {
    if (!g) return RT_OK;
    {
        std::lock_guard<std::mutex> lock(g->mutex);
        g->quit = true;
    }
    g->wake.notify_all();
    for (std::thread& t : g->workers) if (t.joinable()) t.join();
    const Rccl* R = g->comms.empty() ? nullptr : rccl();
    for (size_t k = 0; k < g->devices.size(); ++k) {
        (void)hipSetDevice(g->devices[k]);
        if (k < g->comms.size() && g->comms[k] && R && R->CommDestroy) (void)R->CommDestroy(g->comms[k]);
        if (k < g->accum.size() && g->accum[k]) (void)hipFree(g->accum[k]);
        if (k < g->streams.size() && g->streams[k]) (void)hipStreamDestroy(g->streams[k]);
        if (k < g->scenes.size() && g->scenes[k]) (void)P::destroy(g->scenes[k]);
    }
    if (!g->devices.empty()) {
        (void)hipSetDevice(g->devices[0]);
        if (g->stage) (void)hipFree(g->stage);
        if (g->e0) (void)hipEventDestroy(g->e0);
        if (g->e1) (void)hipEventDestroy(g->e1);
    }
    delete g;
    return RT_OK;
}

template <class P, class H>
int group_create(const typename P::Desc* desc, const int* devices, int n_devices, H** out) {
    if (!desc || !out) return rt_fail(RT_EINVAL, "rt_group_create: null argument");
    *out = nullptr;
    if (n_devices < 0 || (n_devices > 0 && !devices)) return rt_fail(RT_EINVAL, "rt_group_create: bad device list");
    int visible = 0;
    hipError_t e = hipGetDeviceCount(&visible);
    if (e != hipSuccess || visible == 0) return rt_fail(RT_ENODEV, "no HIP device available (%s)", hipGetErrorString(e));
    H* g = new (std::nothrow) H();
    if (!g) return rt_fail(RT_ENOMEM, "out of host memory");
    if (n_devices == 0) for (int d = 0; d < visible; ++d) g->devices.push_back(d);
    else g->devices.assign(devices, devices + n_devices);
    for (size_t a = 0; a < g->devices.size(); ++a) {
        if (g->devices[a] < 0 || g->devices[a] >= visible) {
            const int bad = g->devices[a];
            g->devices.clear();
            delete g;
            return rt_fail(RT_EINVAL, "device %d out of range (0..%d)", bad, visible - 1);
        }
        for (size_t b = 0; b < a; ++b) if (g->devices[a] == g->devices[b]) g->distinct = false;
    }
    const size_t G = g->devices.size();
    g->scenes.assign(G, nullptr);
    g->streams.assign(G, nullptr);
    g->accum.assign(G, nullptr);
    g->accum_cap.assign(G, 0);
    g->status.assign(G, RT_OK);
    g->message.assign(G, std::string());
    auto give_up = [&](int st) {
        const std::string msg = rt_last_error();
        group_destroy<P, H>(g);
        return rt_fail(st, "%s", msg.c_str());
    };
    for (size_t k = 0; k < G; ++k) {
        int st = P::create(desc, g->devices[k], &g->scenes[k]); // (leaves that device current)
        if (st == RT_OK && hipStreamCreateWithFlags(&g->streams[k], hipStreamNonBlocking) != hipSuccess)
            st = rt_fail(RT_EHIP, "hipStreamCreate failed on device %d", g->devices[k]);
        if (st != RT_OK) return give_up(st);
    }
    // The members see each other's memory where the hardware lets them (xGMI between the GPUs of a node): the peer
    // copies of the reduction then go device to device instead of through the host.  Not being able to is no error.
    for (size_t a = 0; a < G; ++a)
        for (size_t b = 0; b < G; ++b) {
            if (g->devices[a] == g->devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->devices[a], g->devices[b]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(g->devices[a]) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t pe = hipDeviceEnablePeerAccess(g->devices[b], 0);
            if (pe != hipSuccess) (void)hipGetLastError(); // (hipErrorPeerAccessAlreadyEnabled: another group or the host did it)
        }
    if (hipSetDevice(g->devices[0]) != hipSuccess || hipEventCreate(&g->e0) != hipSuccess || hipEventCreate(&g->e1) != hipSuccess)
        return give_up(rt_fail(RT_EHIP, "hipEventCreate failed on device %d", g->devices[0]));
    if (G > 1) {
        try {
            for (size_t k = 0; k < G; ++k) g->workers.emplace_back(worker_loop<P>, static_cast<Group<P>*>(g), k);
        } catch (...) {
            return give_up(rt_fail(RT_ENOMEM, "could not start a host thread per device"));
        }
    }
    *out = g;
    return RT_OK;
}

template <class P>
int group_reduce(Group<P>* g, size_t image_floats) {
    typedef typename P::real real;
    const size_t G = g->devices.size();
    const char* how = std::getenv("RT_AMD_REDUCE"); // "peer": copies and adds in member order instead of RCCL
    const bool peer = (how && std::strcmp(how, "peer") == 0) || !g->distinct;
    if (G > 1 && peer) {
        GROUP_HIP(hipSetDevice(g->devices[0]));
        if (g->stage_cap < image_floats) {
            if (g->stage) { GROUP_HIP(hipFree(g->stage)); g->stage = nullptr; g->stage_cap = 0; }
            GROUP_HIP(hipMalloc((void**)&g->stage, image_floats * sizeof(real)));
            g->stage_cap = image_floats;
        }
        for (size_t k = 1; k < G; ++k) {
            GROUP_HIP(hipMemcpyPeerAsync(g->stage, g->devices[0], g->accum[k], g->devices[k], image_floats * sizeof(real), g->streams[0]));
            P::add(g->accum[0], g->stage, image_floats, g->streams[0]);
            GROUP_HIP(hipGetLastError());
        }
    } else if (G > 1 || std::getenv("RT_AMD_FORCE_RCCL")) {
        const Rccl* R = rccl();
        if (!R->lib) return rt_fail(RT_ERCCL, "%s (RT_AMD_REDUCE=peer sums with peer copies instead)", R->error.c_str());
        if (g->comms.empty()) {
            g->comms.assign(G, nullptr);
            ncclResult_t r = R->CommInitAll(g->comms.data(), (int)G, g->devices.data());
            if (r != ncclSuccess) { g->comms.clear(); return rt_fail(RT_ERCCL, "ncclCommInitAll failed: %s", R->GetErrorString(r)); }
        }
        ncclResult_t r = R->GroupStart();
        hipError_t he = hipSuccess;
        for (size_t k = 0; k < G && r == ncclSuccess && he == hipSuccess; ++k) {
            he = hipSetDevice(g->devices[k]);
            if (he == hipSuccess) r = R->Reduce(g->accum[k], g->accum[k], image_floats, P::nccl_type(), ncclSum, 0, g->comms[k], g->streams[k]);
        }
        ncclResult_t r2 = R->GroupEnd(); // (always closed, whatever happened inside)
        if (r == ncclSuccess) r = r2;
        if (he != hipSuccess) return rt_fail(RT_EHIP, "hipSetDevice failed: %s", hipGetErrorString(he));
        if (r != ncclSuccess) return rt_fail(RT_ERCCL, "ncclReduce failed: %s", R->GetErrorString(r));
        for (size_t k = 1; k < G; ++k) { GROUP_HIP(hipSetDevice(g->devices[k])); GROUP_HIP(hipStreamSynchronize(g->streams[k])); }
    }
    return RT_OK;
}

template <class P>
int group_frame(Group<P>* g, const typename P::Camera* camera, const typename P::real* pattern, int n, const typename P::real* background,
                int width, int height, int max_depth, uint64_t seed, uint32_t flags, typename P::real* out, rtStats* stats) {
    typedef typename P::real real;
    const size_t G = g->devices.size();
    const size_t image_floats = (size_t)3 * (size_t)width * (size_t)height;
    for (size_t k = 0; k < G; ++k) {
        if (g->accum_cap[k] >= image_floats) continue;
        GROUP_HIP(hipSetDevice(g->devices[k]));
        if (g->accum[k]) { GROUP_HIP(hipFree(g->accum[k])); g->accum[k] = nullptr; g->accum_cap[k] = 0; }
        GROUP_HIP(hipMalloc((void**)&g->accum[k], image_floats * sizeof(real)));
        g->accum_cap[k] = image_floats;
    }
    GROUP_HIP(hipSetDevice(g->devices[0]));
    GROUP_HIP(hipEventRecord(g->e0, g->streams[0]));
    g->frame.camera = camera; g->frame.pattern = pattern; g->frame.background = background;
    g->frame.n = n; g->frame.width = width; g->frame.height = height; g->frame.max_depth = max_depth;
    g->frame.seed = seed; g->frame.flags = flags;
    if (G == 1) {
        member_frame(g, 0);
    } else {
        std::unique_lock<std::mutex> lock(g->mutex);
        g->running = G;
        ++g->generation;
        g->wake.notify_all();
        g->done.wait(lock, [&] { return g->running == 0; });
    }
    for (size_t k = 0; k < G; ++k)
        if (g->status[k] != RT_OK) return rt_fail(g->status[k], "device %d (member %zu of %zu): %s", g->devices[k], k, G, g->message[k].c_str());
    int st = group_reduce(g, image_floats); // the sum of the members' accumulators, on the first device
    if (st != RT_OK) return st;
    GROUP_HIP(hipSetDevice(g->devices[0]));
    st = P::finalize(g->scenes[0], g->accum[0], n, width, height, g->accum[0], g->streams[0]);
    if (st != RT_OK) return st;
    GROUP_HIP(hipEventRecord(g->e1, g->streams[0]));
    GROUP_HIP(hipMemcpyAsync(out, g->accum[0], image_floats * sizeof(real), hipMemcpyDeviceToHost, g->streams[0]));
    GROUP_HIP(hipStreamSynchronize(g->streams[0]));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        float ms = 0.0f;
        GROUP_HIP(hipEventElapsedTime(&ms, g->e0, g->e1));
        stats->seconds = (double)ms * 1e-3; // first device's clock: from its first kernel to the finalised image
        stats->paths = (uint64_t)width * (uint64_t)height * (uint64_t)n;
    }
    return RT_OK;
}

template <class P>
int group_render(Group<P>* g, const typename P::Camera* camera, const typename P::real* pattern, int n, const typename P::real* background,
                 int width, int height, int max_depth, uint64_t seed, uint32_t flags, typename P::real* out, rtStats* stats) {
    if (!g || !camera || !pattern || !background || !out) return rt_fail(RT_EINVAL, "%s: null argument", P::name());
    if (width < 2 || height < 2) return rt_fail(RT_EINVAL, "width and height must be at least 2");
    if ((uint64_t)width * (uint64_t)height > 0x7FFFFFFFull / 3ull) return rt_fail(RT_EINVAL, "image too large");
    if (flags & RT_FLAG_COUNTERS) return rt_fail(RT_EINVAL, "%s: RT_FLAG_COUNTERS is per scene (use rt_render)", P::name());
    const int st = group_frame(g, camera, pattern, n, background, width, height, max_depth, seed, flags, out, stats);
    if (st != RT_OK) {
        // whatever failed, nothing of this frame stays queued on any member's stream (the message survives the drain)
        const std::string msg = rt_last_error();
        drain(g);
        return rt_fail(st, "%s", msg.c_str());
    }
    return RT_OK;
}

} // namespace

struct rtGroup : Group<Float32Path> {};
struct rtGroup64 : Group<Float64Path> {};

extern "C" {

int rt_rccl_available(void) {
    const Rccl* R = rccl();
    if (!R->lib) return rt_fail(RT_ERCCL, "%s", R->error.c_str());
    return RT_OK;
}

int rt_group_destroy(rtGroup* g) { return group_destroy<Float32Path, rtGroup>(g); }
int rt_group_create(const rtSceneDesc* desc, const int* devices, int n_devices, rtGroup** out) {
    return group_create<Float32Path, rtGroup>(desc, devices, n_devices, out);
}
int rt_group_size(const rtGroup* g, int* n_devices) {
    if (!g || !n_devices) return rt_fail(RT_EINVAL, "rt_group_size: null argument");
    *n_devices = (int)g->devices.size();
    return RT_OK;
}
int rt_group_render(rtGroup* g, const rtCamera* camera, const float* pattern, int n, const float background[3], int width,
                    int height, int max_depth, uint64_t seed, uint32_t flags, float* out, rtStats* stats) {
    return group_render<Float32Path>(g, camera, pattern, n, background, width, height, max_depth, seed, flags, out, stats);
}

// the same for render(Float64, ...): the members are Float64 scenes, the accumulators doubles (ncclDouble)
int rt_group_destroy_f64(rtGroup64* g) { return group_destroy<Float64Path, rtGroup64>(g); }
int rt_group_create_f64(const rtSceneDesc64* desc, const int* devices, int n_devices, rtGroup64** out) {
    return group_create<Float64Path, rtGroup64>(desc, devices, n_devices, out);
}
int rt_group_size_f64(const rtGroup64* g, int* n_devices) {
    if (!g || !n_devices) return rt_fail(RT_EINVAL, "rt_group_size_f64: null argument");
    *n_devices = (int)g->devices.size();
    return RT_OK;
}
int rt_group_render_f64(rtGroup64* g, const rtCamera64* camera, const double* pattern, int n, const double background[3], int width,
                        int height, int max_depth, uint64_t seed, uint32_t flags, double* out, rtStats* stats) {
    return group_render<Float64Path>(g, camera, pattern, n, background, width, height, max_depth, seed, flags, out, stats);
}

} // extern "C"
