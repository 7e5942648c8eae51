// rt_kernels.h -- launch interface between the C ABI (rt_api.cpp) and the gfx950 kernels (rt_kernels.hip).
#ifndef RT_KERNELS_H
#define RT_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rt_device.h"

struct RtCameraDev { // same field order as rtCamera (src/camera.jl:3-14)
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius, t0, t1;
};

#define RT_DISP_MAX 64u    // path dispensers at most
#define RT_DISP_UNIT 256u  // consecutive paths dealt to one dispenser
#define RT_DISP_BASE 64u   // first dispenser word in the sync array
#define RT_DISP_STRIDE 32u // words between dispensers (128 bytes)
#define RT_SLOT_DISP 128u  // slot dispensers of the extend kernel (each hands out one contiguous share of the slots)
#define RT_SLOT_DISP_BASE (RT_DISP_BASE + RT_DISP_MAX * RT_DISP_STRIDE)
#define RT_REDO_COUNT (RT_SLOT_DISP_BASE + RT_SLOT_DISP * RT_DISP_STRIDE) // entries in PathPool::redo; + 1: finished workgroups of rt_redo_kernel
#define RT_TAIL_FLAG (RT_REDO_COUNT + 2u)  // set once a workgroup found no unstarted path left: slots now only run dry
#define RT_LIVE_SLOTS (RT_REDO_COUNT + 3u) // written by rt_launch_compact: slots that hold a ray
#define RT_SYNC_WORDS (RT_REDO_COUNT + RT_DISP_STRIDE)

// Local indices path dispenser c (of n_disp) has to hand out for a batch of n_paths paths: v < limit <=> the path
// ((v / RT_DISP_UNIT) * n_disp + c) * RT_DISP_UNIT + v % RT_DISP_UNIT is below n_paths.  The units are dealt round-robin,
// so the dispensers differ by at most one unit, and only the last unit of the batch is ragged.
__host__ __device__ inline uint32_t rt_dispenser_limit(uint32_t n_paths, uint32_t n_disp, uint32_t c) {
    const uint32_t n_units = (n_paths + RT_DISP_UNIT - 1u) / RT_DISP_UNIT;
    if (n_units == 0u) return 0u;
    uint32_t lim = (n_units / n_disp + (c < n_units % n_disp ? 1u : 0u)) * RT_DISP_UNIT;
    if (c == (n_units - 1u) % n_disp) lim -= n_units * RT_DISP_UNIT - n_paths;
    return lim;
}

struct PathPool { // per-slot path state, five float4 planes of n_slots entries each (see rt_kernels.hip)
    float4* q;
    uint32_t* redo; // n_slots entries: slots on hold, whose ray rt_redo_kernel walks again
    uint32_t n_slots;
};

struct RtRenderParams {
    int width, height, max_depth;
    int sample_first, sample_stride; // shard sample k is pattern column sample_first + k*sample_stride
    uint32_t k_begin, k_end;         // batch of shard samples rendered by the current bounce loop
    uint32_t n_pixels;               // width * height
    uint32_t n_paths;                // (k_end - k_begin) * n_pixels
    uint32_t seed32;
    float background[3];
    uint32_t n_disp, disp_limit;     // path dispensers in use; local index at which a dispenser has run dry
    uint32_t leaf_threshold;         // wave vote: test queued leaves once this many lanes hold one
    uint32_t blocked_threshold;      // ... or once this many lanes cannot step before their leaves are tested
    uint32_t fetch_threshold;        // extend kernel: refill the wave once this many lanes are idle
    uint32_t step_dropout;           // extend kernel: leave the node-stepping loop after this many lanes dropped out
    uint32_t box_spread;             // extend kernel: a leaf-block run with up to this many boxes (at most 10) has their six
                                     // sides tested by six lanes each instead of one after the other; 0: never
    uint32_t count_mode;             // counting variant only: 0 = every leaf test; 1 = only rectangles and boxes, and the
                                     // leaf-block runs that hold one; 2 = only spheres, and the runs that hold one
    uint32_t regroup;                // shade kernel: bit 0 = shade the workgroup's slots in order of material type,
                                     // bit 1 = write the rays that go on in order of direction octant and origin cell
    size_t image_floats;             // 3 * width * height
};

extern "C" {
void rt_launch_init(const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P, const PathPool* pool,
                    hipStream_t stream);
// one bounce of every slot: extend (closest hits) then shade (scatter / next path).  d_sync holds the slot dispensers
// (zero before the first bounce) and the path dispensers; *d_active is set to 1 if any slot still holds a ray after the
// bounce.
void rt_launch_bounce(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                      const PathPool* pool, uint32_t extend_blocks, float4* d_radiance, uint32_t* d_sync, uint32_t* d_active,
                      unsigned long long* d_counters, bool count, hipEvent_t* timing, hipStream_t stream,
                      hipEvent_t before_shade = nullptr); // before_shade: the shade kernel waits for this event (rt_redo_kernel on another stream)
// settles the slots that the shade kernel put on hold since the last call; to be enqueued between batches of bounces
void rt_launch_redo(const DevScene* S, const RtRenderParams* P, const PathPool* pool, uint32_t* d_sync, hipStream_t stream);
// The slots of `from` that hold a ray, moved to the front of the pool `to` (same order), which has
// RT_COMPACT_SLOTS(their number, from->n_slots) slots, the others marked empty; d_sync[RT_LIVE_SLOTS] receives their
// number.  d_counts: scratch, from->n_slots / 256 + 2 words.
#define RT_COMPACT_SLOTS(live, n_slots) (((live) + 1023u) / 1024u * 1024u < 1024u ? 1024u : (((live) + 1023u) / 1024u * 1024u > (n_slots) ? (n_slots) : ((live) + 1023u) / 1024u * 1024u))
// For the end of a frame, when no unstarted path is left and the slots run dry over up to max_depth bounces.
void rt_launch_compact(const PathPool* from, float4* to, uint32_t* d_counts, uint32_t* d_sync, hipStream_t stream);
void rt_launch_sum(const RtRenderParams* P, const float4* d_radiance, float* d_accum, int first_batch, hipStream_t stream);
void rt_launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream); // dst[i] += src[i]
void rt64_launch_add(double* d_dst, const double* d_src, size_t n, hipStream_t stream); // the same in double (rt_f64.hip)
void rt_launch_finalize(const float* d_accum, float inv_n, size_t image_floats, float* d_out, hipStream_t stream);
void rt_launch_finalize_rgb8(const float* d_accum, float inv_n, int width, int height, uint8_t* d_out, hipStream_t stream);
void rt_launch_trace_rays(const DevScene* S, const float* d_rays, int n, float t_min, float t_max, uint32_t seed32,
                          int32_t* prim_id, int32_t* sub_id, float* t, float* normal, float* uv, uint8_t* front,
                          hipStream_t stream);
void rt_launch_trace_paths(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                           const int32_t* d_ids, int n_paths, float* d_radiance, hipStream_t stream);
}
#endif
