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

struct RtRenderParams {
    int width, height, max_depth;
    int sample_first, sample_stride; // pattern columns of this shard: sample_first + k*sample_stride
    uint32_t n_shard_samples;        // number of such columns
    uint32_t n_chunks;               // waves per pixel tile; chunk c takes k = c, c + n_chunks, ...
    uint32_t tiles_x, tiles_y;       // 8x8-pixel tiles
    uint32_t seed32;
    float background[3];
    uint32_t leaf_threshold, shade_threshold; // wave vote: run the leaf / shade block once this many lanes wait for it
    size_t image_floats;             // 3 * width * height
};

extern "C" {
void rt_launch_render(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                      float* d_partial, unsigned long long* d_counters, bool count, hipStream_t stream);
void rt_launch_reduce(const float* d_partial, uint32_t n_chunks, size_t image_floats, float* d_accum, hipStream_t stream);
void rt_launch_finalize(const float* d_accum, float inv_n, size_t image_floats, float* d_out, hipStream_t stream);
void rt_launch_trace_rays(const DevScene* S, const float* d_rays, int n, float t_min, float t_max, uint32_t seed32,
                          int32_t* prim_id, int32_t* sub_id, float* t, float* normal, float* uv, uint8_t* front,
                          hipStream_t stream);
void rt_launch_trace_paths(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                           const int32_t* d_ids, int n_paths, float* d_radiance, hipStream_t stream);
}
#endif
