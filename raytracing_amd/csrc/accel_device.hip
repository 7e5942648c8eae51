// accel_device.hip -- the device trees built on the GPU (see accel_device.h).
//
// The reference builds one median-split tree per object type on the host (src/BVH.jl:65-96; meshes
// src/mesh.jl:217-283).  Which boxes a ray walks does not change which primitive is its closest hit, so the device
// walks trees of its own, and here it also builds them, by the surface-area heuristic, top down, level by level (the
// ranges of a level are independent: each gets a workgroup or a wave anywhere on the device; the host reads one count
// per level):
//   ranges of more than 64 leaves (accel_split_blocks_kernel: a workgroup per range of more than 2048,
//     accel_split_waves_kernel: a wave per smaller one): box and centroid bounds, 32 bins per axis (LDS atomics on
//     order-preserving integer images of the floats), the cheapest of the 3 x 31 bin boundaries, a stable partition by
//     prefix sums / ballots;
//   ranges of up to 64 leaves (accel_sweep_kernel), a wave each with a leaf per lane, down to the leaves: sorted along
//     every axis in turn
//     (ranks by comparison, ds_permute), every split position costed from prefix and suffix boxes (wave scans).
// The arithmetic, the order of the candidates and the tie-breaks are those of the host builder (accel_build.cpp),
// which stays for hosts without a GPU: both give the same tree, node for node (tests/test_gpu_device_build.py).
// accel_layout_kernel then writes the tree eight times, once per octant of ray directions, in depth-first order with
// the near child first (the stackless layout of accel_build.h): a thread per (node, octant) finds the node's place by
// walking up to the root.
// Measured and left out: bottom-up clustering in Morton order (parallel locally-ordered clustering, Meister & Bittner,
// TVCG 2018) builds cu_cover2's tree in 1 ms too, but rt_extend_kernel then takes 12-19 % longer per frame, whatever the
// search radius (8 ... all clusters).
#include "accel_device.h"

#include <algorithm>
#include <cmath>

#include "../../include/rt_amd.h"
#include "accel_half.h"
#include "rt_internal.h"

namespace {

constexpr uint32_t BUILD_THREADS = 1024;

struct WorkNode {
    float b[6];
    int32_t child[2]; // child[0] holds the lower centroid along `axis`; -1 in leaves
    int32_t parent;
    uint32_t size;    // nodes in the subtree, this one included
    uint32_t axis;
    uint32_t leaf;    // leaf word; 0 in interior nodes
};

struct Job {
    uint32_t item_first, n_items; // the tree's leaves in the item array (set offset included)
    uint32_t layout_first;        // first node of the tree inside a layout
    uint32_t node_first;          // first of the tree's 2 n - 1 work nodes
    uint32_t set;
    uint32_t pairs;               // also written as pair nodes
    uint32_t pad[2];
};

struct Range {
    uint32_t lo, hi;  // items [lo, hi) of the tree
    uint32_t node;    // the node that holds them
    uint32_t alloc;   // small ranges: first of the 2 (hi - lo) - 2 nodes reserved for the subtree
    uint32_t job;     // the tree
};

constexpr int BINS = 32;
constexpr uint32_t SWEEP_LIMIT = 64; // ranges up to this size try every split position
constexpr uint32_t WAVE_LIMIT = 2048; // larger ranges are split by the whole workgroup, smaller ones by a wave

__device__ inline double item_centroid(const AccelItem& it, int a) {
    double c = 0.5 * ((double)it.b[a] + (double)it.b[3 + a]);
    if (!(c == c)) c = 0.0;
    if (c > 1e30) c = 1e30;
    if (c < -1e30) c = -1e30;
    return c;
}

// b := the box of b and x; comparisons that a NaN bound fails leave b alone (as the host builder's Box::grow)
__device__ inline void grow(float* b, const float* x) {
    for (int a = 0; a < 3; ++a) {
        if (x[a] < b[a]) b[a] = x[a];
        if (x[3 + a] > b[3 + a]) b[3 + a] = x[3 + a];
    }
}
__device__ inline void reset(float* b) {
    for (int a = 0; a < 3; ++a) { b[a] = INFINITY; b[3 + a] = -INFINITY; }
}
__device__ inline double half_area(const float* b) {
    double e[3];
    for (int a = 0; a < 3; ++a) {
        e[a] = (double)b[3 + a] - (double)b[a];
        if (!(e[a] >= 0.0)) return 0.0; // empty or NaN
        if (e[a] > 1e30) e[a] = 1e30;   // unbounded boxes still compare
    }
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}

// integer images of floats and doubles whose unsigned order is the numbers' order (for atomic min / max in LDS)
__device__ inline uint32_t image32(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float number32(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }
__device__ inline unsigned long long image64(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double number64(unsigned long long e) {
    return __longlong_as_double((long long)((e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e));
}

__device__ inline void write_leaf(WorkNode* nodes, uint32_t node, uint32_t parent, const AccelItem& it) {
    WorkNode nd;
    for (int a = 0; a < 6; ++a) nd.b[a] = it.b[a]; // NaN / infinite bounds exactly as given
    nd.child[0] = nd.child[1] = -1;
    nd.parent = (int32_t)parent;
    nd.size = 1u;
    nd.axis = 0u;
    nd.leaf = it.leaf;
    nodes[node] = nd;
}

__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline double readlane_f64(double v, int lane) { // lane: the same in every lane of the wave
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ inline double shfl_f64(double v, int lane) {
    const int lo = __shfl(__double2loint(v), lane), hi = __shfl(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// The items of the wave, one per lane, lanes [first, last) sorted by centroid along `axis`, stably: a lane's new place
// is the number of items of the range that must come before its own.  The other lanes keep their items.
__device__ inline void wave_sort(AccelItem& it, int lane, int first, int last, int axis) {
    const double key = item_centroid(it, axis);
    int place = lane;
    if (lane >= first && lane < last) place = first;
    for (int j = first; j < last; ++j) {
        const double kj = readlane_f64(key, j);
        if (lane >= first && lane < last && (kj < key || (kj == key && j < lane))) place++;
    }
    for (int c = 0; c < 6; ++c) it.b[c] = __int_as_float(__builtin_amdgcn_ds_permute(place << 2, __float_as_int(it.b[c])));
    it.leaf = (uint32_t)__builtin_amdgcn_ds_permute(place << 2, (int)it.leaf);
}

// The subtree over items [r.lo, r.hi) (2 .. SWEEP_LIMIT = 64 of them) below node r.node, by one wave with an item per
// lane: at every node the range is sorted along each axis in turn and every split position is costed
// (accel_build.cpp, sweep_split: the same sorts, the same sums, the first of the cheapest positions).
// stack: 3 * 64 words of LDS of the wave's own.
__device__ void build_small(const AccelItem* __restrict__ items, WorkNode* __restrict__ nodes, Range r, uint32_t* stack) {
    const int lane = (int)(threadIdx.x & 63u);
    const int total = (int)(r.hi - r.lo);
    AccelItem it;
    reset(it.b);
    it.leaf = 0u;
    if (lane < total) it = items[r.lo + (uint32_t)lane];
    uint32_t top = 1u, alloc = r.alloc;
    if (lane == 0) { stack[0] = 0u; stack[1] = (uint32_t)total; stack[2] = r.node; }
    wave_sync();
    while (top) {
        --top;
        const int first = (int)stack[3u * top], last = (int)stack[3u * top + 1u];
        const uint32_t node = stack[3u * top + 2u];
        wave_sync(); // (the entry is read by every lane before anything is pushed over it)
        const int n = last - first;
        const bool in = lane >= first && lane < last;
        float box[6];
        reset(box);
        if (in) grow(box, it.b);
        for (int off = 32; off > 0; off >>= 1) {
            float other[6];
            for (int c = 0; c < 6; ++c) other[c] = __shfl_xor(box[c], off);
            grow(box, other);
        }
        double best = INFINITY;
        int axis = -1, mid = 0;
        for (int a = 0; a < 3; ++a) {
            wave_sort(it, lane, first, last, a);
            // boxes of the items up to and including this lane's, and from this lane's on
            float left[6], right[6];
            reset(left); reset(right);
            if (in) { grow(left, it.b); grow(right, it.b); }
            for (int off = 1; off < 64; off <<= 1) {
                float lo_[6], hi_[6];
                for (int c = 0; c < 6; ++c) { lo_[c] = __shfl_up(left[c], off); hi_[c] = __shfl_down(right[c], off); }
                if (in && lane - off >= first) grow(left, lo_);
                if (in && lane + off < last) grow(right, hi_);
            }
            const double left_area = half_area(left), right_area = half_area(right);
            const double before = shfl_f64(left_area, lane > 0 ? lane - 1 : 0); // items [first, lane)
            const int k = lane - first;                                          // split: k items go left
            double cost = INFINITY;
            if (in && k >= 1) cost = before * (double)k + right_area * (double)(n - k);
            // the cheapest position, the first of them among equals
            int at = k;
            for (int off = 32; off > 0; off >>= 1) {
                const double oc = shfl_f64(cost, lane ^ off);
                const int oa = __shfl_xor(at, off);
                if (oc < cost || (oc == cost && oa < at)) { cost = oc; at = oa; }
            }
            if (cost < best) { best = cost; axis = a; mid = first + at; }
        }
        if (axis < 0) { axis = 0; mid = first + n / 2; }
        if (axis != 2) wave_sort(it, lane, first, last, axis);
        const uint32_t c0 = alloc, c1 = alloc + 1u;
        alloc += 2u;
        if (lane == 0) {
            WorkNode nd;
            for (int c = 0; c < 6; ++c) nd.b[c] = box[c];
            nd.child[0] = (int32_t)c0;
            nd.child[1] = (int32_t)c1;
            nd.parent = nodes[node].parent; // set when the node was handed out
            nd.size = 2u * (uint32_t)n - 1u;
            nd.axis = (uint32_t)axis;
            nd.leaf = 0u;
            nodes[node] = nd;
        }
        const int cl[2] = {first, mid}, ch[2] = {mid, last};
        const uint32_t cn[2] = {c0, c1};
        for (int s = 0; s < 2; ++s) {
            if (ch[s] - cl[s] == 1) {
                if (lane == cl[s]) write_leaf(nodes, cn[s], node, it);
            } else {
                if (lane == 0) {
                    nodes[cn[s]].parent = (int32_t)node;
                    stack[3u * top] = (uint32_t)cl[s]; stack[3u * top + 1u] = (uint32_t)ch[s]; stack[3u * top + 2u] = cn[s];
                }
                ++top;
            }
        }
        wave_sync();
    }
}

__device__ inline void wave_sync_global() { // as wave_sync, for what the wave's lanes hand each other through global memory
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// min / max over the wave, NaN-skipping like grow()
__device__ inline void wave_box(float* box) {
    for (int off = 32; off > 0; off >>= 1) {
        float other[6];
        for (int c = 0; c < 6; ++c) other[c] = __shfl_xor(box[c], off);
        grow(box, other);
    }
}
__device__ inline void wave_bounds(double* lo, double* hi) {
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; ++a) {
            const double ol = shfl_f64(lo[a], (int)(threadIdx.x & 63u) ^ off), oh = shfl_f64(hi[a], (int)(threadIdx.x & 63u) ^ off);
            if (ol < lo[a]) lo[a] = ol;
            if (oh > hi[a]) hi[a] = oh;
        }
}

struct Counters { uint32_t* n_alloc; uint32_t* n_small; uint32_t* n_next; };

// the two children of a split node: leaves are written, ranges go to the list of small ranges or to the next level
__device__ inline void hand_out_children(WorkNode* nodes, const AccelItem* items, const Range& r, uint32_t mid, int axis, const float* box,
                                         Range* small, Range* nxt, const Counters& C) {
    const uint32_t m = r.hi - r.lo;
    const uint32_t c0 = atomicAdd(C.n_alloc, 2u), c1 = c0 + 1u;
    WorkNode nd;
    for (int a = 0; a < 6; ++a) nd.b[a] = box[a];
    nd.child[0] = (int32_t)c0;
    nd.child[1] = (int32_t)c1;
    nd.parent = nodes[r.node].parent; // set when the node was handed out
    nd.size = 2u * m - 1u;
    nd.axis = (uint32_t)(axis < 0 ? 0 : axis);
    nd.leaf = 0u;
    nodes[r.node] = nd;
    const uint32_t cl[2] = {r.lo, mid}, ch[2] = {mid, r.hi}, cn[2] = {c0, c1};
    for (int s = 0; s < 2; ++s) {
        const uint32_t cm = ch[s] - cl[s];
        nodes[cn[s]].parent = (int32_t)r.node;
        if (cm == 1u) write_leaf(nodes, cn[s], r.node, items[cl[s]]);
        else if (cm <= SWEEP_LIMIT) small[atomicAdd(C.n_small, 1u)] = Range{cl[s], ch[s], cn[s], atomicAdd(C.n_alloc, 2u * cm - 2u), r.job};
        else nxt[atomicAdd(C.n_next, 1u)] = Range{cl[s], ch[s], cn[s], 0u, r.job};
    }
}

#define BIN_WORDS 7 // six bounds and a count

// Binned split of a range of SWEEP_LIMIT < m <= WAVE_LIMIT items by one wave (accel_build.cpp, binned_split): bounds by
// shuffles, bins in LDS of the wave's own, one lane per bin boundary for the costs, a stable partition by ballots.
__device__ void split_by_wave(AccelItem* items, AccelItem* temp, WorkNode* nodes, Range r, uint32_t (*bins)[BINS][BIN_WORDS],
                              Range* small, Range* nxt, const Counters& C) {
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t m = r.hi - r.lo;
    float box[6];
    double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    reset(box);
    for (uint32_t k = r.lo + (uint32_t)lane; k < r.hi; k += 64u) {
        const AccelItem it = items[k];
        grow(box, it.b);
        for (int a = 0; a < 3; ++a) {
            const double c = item_centroid(it, a);
            if (c < cmin[a]) cmin[a] = c;
            if (c > cmax[a]) cmax[a] = c;
        }
    }
    wave_box(box);
    wave_bounds(cmin, cmax);
    for (int i = lane; i < 3 * BINS; i += 64) {
        for (int c = 0; c < 6; ++c) bins[i / BINS][i % BINS][c] = image32(c < 3 ? INFINITY : -INFINITY);
        bins[i / BINS][i % BINS][6] = 0u;
    }
    wave_sync();
    double scale[3];
    bool open[3];
    for (int a = 0; a < 3; ++a) {
        const double ext = cmax[a] - cmin[a];
        open[a] = ext > 0.0;
        scale[a] = open[a] ? (double)BINS / ext : 0.0;
    }
    for (uint32_t k = r.lo + (uint32_t)lane; k < r.hi; k += 64u) {
        const AccelItem it = items[k];
        for (int a = 0; a < 3; ++a) {
            if (!open[a]) continue;
            int b = (int)((item_centroid(it, a) - cmin[a]) * scale[a]);
            b = min(max(b, 0), BINS - 1);
            for (int c = 0; c < 3; ++c) {
                if (it.b[c] == it.b[c]) atomicMin(&bins[a][b][c], image32(it.b[c]));
                if (it.b[3 + c] == it.b[3 + c]) atomicMax(&bins[a][b][3 + c], image32(it.b[3 + c]));
            }
            atomicAdd(&bins[a][b][6], 1u);
        }
    }
    wave_sync();
    double best = INFINITY;
    int axis = -1, best_bin = 0;
    for (int a = 0; a < 3; ++a) {
        if (!open[a]) continue;
        // lane b < BINS holds bin b: boxes and counts of the bins up to and including it, and from it on
        float left[6], right[6];
        uint32_t lc = 0u, rc = 0u;
        reset(left); reset(right);
        if (lane < BINS) {
            float bb[6];
            for (int c = 0; c < 6; ++c) bb[c] = number32(bins[a][lane][c]);
            grow(left, bb); grow(right, bb);
            lc = rc = bins[a][lane][6];
        }
        for (int off = 1; off < BINS; off <<= 1) {
            float lo_[6], hi_[6];
            for (int c = 0; c < 6; ++c) { lo_[c] = __shfl_up(left[c], off); hi_[c] = __shfl_down(right[c], off); }
            const uint32_t lcu = (uint32_t)__shfl_up((int)lc, off), rcd = (uint32_t)__shfl_down((int)rc, off);
            if (lane < BINS && lane - off >= 0) { grow(left, lo_); lc += lcu; }
            if (lane + off < BINS) { grow(right, hi_); rc += rcd; }
        }
        const double left_area = half_area(left), right_area = half_area(right);
        const double before_area = shfl_f64(left_area, lane > 0 ? lane - 1 : 0);
        const uint32_t before_count = (uint32_t)__shfl((int)lc, lane > 0 ? lane - 1 : 0);
        double cost = INFINITY;
        if (lane >= 1 && lane < BINS && before_count != 0u && rc != 0u) cost = before_area * (double)before_count + right_area * (double)rc;
        int at = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const double oc = shfl_f64(cost, lane ^ off);
            const int oa = __shfl_xor(at, off);
            if (oc < cost || (oc == cost && oa < at)) { cost = oc; at = oa; }
        }
        if (cost < best) { best = cost; axis = a; best_bin = at; }
    }
    uint32_t mid = r.lo + m / 2u; // every centroid coincides: halve the range
    if (axis >= 0) {
        uint32_t total = 0u;
        for (uint32_t k0 = r.lo; k0 < r.hi; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            bool below = false;
            if (k < r.hi) {
                int b = (int)((item_centroid(items[k], axis) - cmin[axis]) * scale[axis]);
                b = min(max(b, 0), BINS - 1);
                below = b < best_bin;
            }
            total += (uint32_t)__popcll(__ballot(below));
        }
        uint32_t at_low = r.lo, at_high = r.lo + total;
        for (uint32_t k0 = r.lo; k0 < r.hi; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            bool below = false, here = k < r.hi;
            AccelItem it;
            if (here) {
                it = items[k];
                int b = (int)((item_centroid(it, axis) - cmin[axis]) * scale[axis]);
                b = min(max(b, 0), BINS - 1);
                below = b < best_bin;
            }
            const unsigned long long mb = __ballot(below), mh = __ballot(here && !below), lt = (1ull << lane) - 1ull;
            if (below) temp[at_low + (uint32_t)__popcll(mb & lt)] = it;
            else if (here) temp[at_high + (uint32_t)__popcll(mh & lt)] = it;
            at_low += (uint32_t)__popcll(mb);
            at_high += (uint32_t)__popcll(mh);
        }
        wave_sync_global();
        for (uint32_t k = r.lo + (uint32_t)lane; k < r.hi; k += 64u) items[k] = temp[k];
        wave_sync_global();
        mid = r.lo + total;
        if (mid == r.lo || mid == r.hi) mid = r.lo + m / 2u;
    }
    if (lane == 0) hand_out_children(nodes, items, r, mid, axis, box, small, nxt, C);
    wave_sync_global();
}

// ctr[0] = ranges listed for the next level, ctr[1] = small ranges listed, ctr[2 + j] = nodes handed out in tree j
__device__ inline Counters counters_of(uint32_t* ctr, uint32_t job) { return Counters{ctr + 2u + job, ctr + 1u, ctr}; }

// The roots.  A workgroup per tree: the tree's items are copied (the build permutes them), node 0 is the root.
__global__ void __launch_bounds__(256) accel_start_kernel(const AccelItem* __restrict__ items_in, const Job* __restrict__ jobs,
                                                          WorkNode* __restrict__ work, AccelItem* __restrict__ items_all,
                                                          Range* __restrict__ queue, Range* __restrict__ small, uint32_t* __restrict__ ctr) {
    const Job job = jobs[blockIdx.x];
    const uint32_t n = job.n_items;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) items_all[job.item_first + i] = items_in[job.item_first + i];
    if (threadIdx.x != 0u || n == 0u) return;
    WorkNode* nodes = work + job.node_first;
    nodes[0].parent = -1;
    ctr[2u + blockIdx.x] = n <= SWEEP_LIMIT ? 2u * n - 1u : 1u;
    if (n == 1u) write_leaf(nodes, 0u, 0xFFFFFFFFu, items_in[job.item_first]);
    else if (n <= SWEEP_LIMIT) small[atomicAdd(ctr + 1u, 1u)] = Range{0u, n, 0u, 1u, blockIdx.x};
    else queue[atomicAdd(ctr, 1u)] = Range{0u, n, 0u, 0u, blockIdx.x};
}

// One level of the ranges of more than WAVE_LIMIT items: a workgroup per listed range (binned split, accel_build.cpp
// binned_split; the smaller ranges of the list are accel_split_waves_kernel's).
__global__ void __launch_bounds__(BUILD_THREADS) accel_split_blocks_kernel(const Job* __restrict__ jobs, WorkNode* __restrict__ work,
                                                                           AccelItem* __restrict__ items_all, AccelItem* __restrict__ temp_all,
                                                                           const Range* __restrict__ cur, Range* __restrict__ nxt,
                                                                           Range* __restrict__ small, uint32_t* __restrict__ ctr) {
    __shared__ uint32_t box_image[6];
    __shared__ unsigned long long centroid_image[6];
    __shared__ uint32_t bin_box[3][BINS][BIN_WORDS];
    __shared__ uint32_t scan[BUILD_THREADS];
    __shared__ int split_axis, split_bin;
    const Range r = cur[blockIdx.x];
    const uint32_t m = r.hi - r.lo, tid = threadIdx.x;
    if (m <= WAVE_LIMIT) return;
    const Job job = jobs[r.job];
    AccelItem* items = items_all + job.item_first;
    AccelItem* temp = temp_all + job.item_first;
    WorkNode* nodes = work + job.node_first;
    if (tid < 6u) {
        box_image[tid] = image32(tid < 3u ? INFINITY : -INFINITY);
        centroid_image[tid] = image64(tid < 3u ? INFINITY : -INFINITY);
    }
    for (uint32_t i = tid; i < 3u * BINS; i += BUILD_THREADS) {
        for (int a = 0; a < 6; ++a) bin_box[i / BINS][i % BINS][a] = image32(a < 3 ? INFINITY : -INFINITY);
        bin_box[i / BINS][i % BINS][6] = 0u;
    }
    __syncthreads();
    {   // bounds: per thread, per wave (shuffles), then one atomic per wave and bound
        float tb[6];
        double tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
        reset(tb);
        for (uint32_t k = r.lo + tid; k < r.hi; k += BUILD_THREADS) {
            const AccelItem it = items[k];
            grow(tb, it.b);
            for (int a = 0; a < 3; ++a) {
                const double c = item_centroid(it, a);
                if (c < tl[a]) tl[a] = c;
                if (c > th[a]) th[a] = c;
            }
        }
        wave_box(tb);
        wave_bounds(tl, th);
        if ((tid & 63u) == 0u)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&box_image[a], image32(tb[a]));
                atomicMax(&box_image[3 + a], image32(tb[3 + a]));
                atomicMin(&centroid_image[a], image64(tl[a]));
                atomicMax(&centroid_image[3 + a], image64(th[a]));
            }
    }
    __syncthreads();
    double cmin[3], scale[3];
    bool open[3];
    for (int a = 0; a < 3; ++a) {
        cmin[a] = number64(centroid_image[a]);
        const double ext = number64(centroid_image[3 + a]) - cmin[a];
        open[a] = ext > 0.0;
        scale[a] = open[a] ? (double)BINS / ext : 0.0;
    }
    for (uint32_t k = r.lo + tid; k < r.hi; k += BUILD_THREADS) {
        const AccelItem it = items[k];
        for (int a = 0; a < 3; ++a) {
            if (!open[a]) continue;
            int b = (int)((item_centroid(it, a) - cmin[a]) * scale[a]);
            b = min(max(b, 0), BINS - 1);
            for (int c = 0; c < 3; ++c) {
                if (it.b[c] == it.b[c]) atomicMin(&bin_box[a][b][c], image32(it.b[c]));
                if (it.b[3 + c] == it.b[3 + c]) atomicMax(&bin_box[a][b][3 + c], image32(it.b[3 + c]));
            }
            atomicAdd(&bin_box[a][b][6], 1u);
        }
    }
    __syncthreads();
    if (tid == 0u) { // the cheapest bin boundary
        double best = INFINITY;
        int axis = -1, best_bin = 0;
        for (int a = 0; a < 3; ++a) {
            if (!open[a]) continue;
            double ra[BINS];
            uint32_t rc[BINS];
            float rb[6];
            reset(rb);
            uint32_t c = 0u;
            for (int b = BINS - 1; b >= 1; --b) {
                float bb[6];
                for (int k = 0; k < 6; ++k) bb[k] = number32(bin_box[a][b][k]);
                grow(rb, bb);
                c += bin_box[a][b][6];
                ra[b] = half_area(rb);
                rc[b] = c;
            }
            float lb[6];
            reset(lb);
            uint32_t lc = 0u;
            for (int b = 1; b < BINS; ++b) {
                float bb[6];
                for (int k = 0; k < 6; ++k) bb[k] = number32(bin_box[a][b - 1][k]);
                grow(lb, bb);
                lc += bin_box[a][b - 1][6];
                if (lc == 0u || rc[b] == 0u) continue;
                const double cost = half_area(lb) * (double)lc + ra[b] * (double)rc[b];
                if (cost < best) { best = cost; axis = a; best_bin = b; }
            }
        }
        split_axis = axis;
        split_bin = best_bin;
    }
    __syncthreads();
    const int axis = split_axis;
    uint32_t mid = r.lo + m / 2u; // every centroid coincides: halve the range
    if (axis >= 0) {
        // stable partition: items whose bin lies below the boundary first.  Each thread owns a run of places.
        const int best_bin = split_bin;
        const uint32_t run = (m + BUILD_THREADS - 1u) / BUILD_THREADS;
        const uint32_t r0 = min(r.lo + tid * run, r.hi), r1 = min(r0 + run, r.hi);
        uint32_t below = 0u;
        for (uint32_t k = r0; k < r1; ++k) {
            int b = (int)((item_centroid(items[k], axis) - cmin[axis]) * scale[axis]);
            b = min(max(b, 0), BINS - 1);
            below += b < best_bin ? 1u : 0u;
        }
        scan[tid] = below;
        __syncthreads();
        for (uint32_t off = 1; off < BUILD_THREADS; off <<= 1) {
            const uint32_t v = tid >= off ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const uint32_t total = scan[BUILD_THREADS - 1u];
        uint32_t at_low = r.lo + scan[tid] - below, at_high = r.lo + total + (r0 - r.lo) - (scan[tid] - below);
        for (uint32_t k = r0; k < r1; ++k) {
            const AccelItem it = items[k];
            int b = (int)((item_centroid(it, axis) - cmin[axis]) * scale[axis]);
            b = min(max(b, 0), BINS - 1);
            if (b < best_bin) temp[at_low++] = it;
            else temp[at_high++] = it;
        }
        __syncthreads();
        for (uint32_t k = r.lo + tid; k < r.hi; k += BUILD_THREADS) items[k] = temp[k];
        mid = r.lo + total;
        if (mid == r.lo || mid == r.hi) mid = r.lo + m / 2u;
    }
    __syncthreads();
    if (tid == 0u) {
        float box[6];
        for (int a = 0; a < 6; ++a) box[a] = number32(box_image[a]);
        hand_out_children(nodes, items, r, mid, axis, box, small, nxt, counters_of(ctr, r.job));
    }
}

// The same level, the listed ranges of up to WAVE_LIMIT items: a wave each.
__global__ void __launch_bounds__(256) accel_split_waves_kernel(const Job* __restrict__ jobs, WorkNode* __restrict__ work,
                                                                AccelItem* __restrict__ items_all, AccelItem* __restrict__ temp_all,
                                                                const Range* __restrict__ cur, uint32_t n_cur, Range* __restrict__ nxt,
                                                                Range* __restrict__ small, uint32_t* __restrict__ ctr) {
    __shared__ uint32_t wave_bins[4][3][BINS][BIN_WORDS];
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= n_cur) return;
    const Range r = cur[wave];
    if (r.hi - r.lo > WAVE_LIMIT) return;
    const Job job = jobs[r.job];
    split_by_wave(items_all + job.item_first, temp_all + job.item_first, work + job.node_first, r, wave_bins[threadIdx.x >> 6], small, nxt,
                  counters_of(ctr, r.job));
}

// The listed ranges of up to SWEEP_LIMIT items: a wave each, down to the leaves.
__global__ void __launch_bounds__(256) accel_sweep_kernel(const Job* __restrict__ jobs, WorkNode* __restrict__ work,
                                                          const AccelItem* __restrict__ items_all, const Range* __restrict__ small,
                                                          uint32_t n_small) {
    __shared__ uint32_t wave_stack[4][3u * SWEEP_LIMIT];
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= n_small) return;
    const Range r = small[wave];
    const Job job = jobs[r.job];
    build_small(items_all + job.item_first, work + job.node_first, r, wave_stack[threadIdx.x >> 6]);
}

// A thread per (node, octant): the node's place in the octant's depth-first order, near child first, is the number of
// nodes that come before it: one per ancestor, plus the near sibling's subtree wherever the way down took the far child.
__global__ void __launch_bounds__(256) accel_layout_kernel(const WorkNode* __restrict__ work, const Job* __restrict__ jobs, uint32_t j,
                                                           uint32_t nodes_per_layout, float scale, uint4* __restrict__ out_nodes,
                                                           uint32_t* __restrict__ max_depth) {
    const Job job = jobs[j];
    const WorkNode* nodes = work + job.node_first;
    const uint32_t n_nodes = 2u * job.n_items - 1u;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = idx >> 3, o = idx & 7u;
    if (v >= n_nodes) return;
    uint32_t pos = 0u, depth = 0u, c = v;
    while (c != 0u && depth < n_nodes) { // node 0 is the root
        const uint32_t p = (uint32_t)nodes[c].parent;
        if (p >= n_nodes) return; // (a node without a way to the root cannot occur; it would leave its place unwritten)
        const uint32_t near = (uint32_t)nodes[p].child[(o >> nodes[p].axis) & 1u];
        if (c != near) pos += nodes[near].size;
        pos += 1u;
        c = p;
        depth++;
    }
    const WorkNode nd = nodes[v];
    const uint32_t base = (job.set * 8u + o) * nodes_per_layout, at = job.layout_first + pos;
    out_nodes[base + at] = accel_pack_node(nd.b, (int)o, scale, nd.leaf ? (RT_NODE_LEAF | nd.leaf) : base + at + nd.size);
    if (o == 0u && nd.leaf) atomicMax(max_depth, depth);
}

// The same walk up for the pair nodes (accel_emit_pairs): a thread per (interior node, octant); the record's place is
// the number of interior nodes before it in depth-first order, near child first -- a subtree of `size` nodes holds
// (size - 1) / 2 of them.
__global__ void __launch_bounds__(256) accel_pairs_kernel(const WorkNode* __restrict__ work, const Job* __restrict__ jobs, uint32_t j,
                                                          uint32_t pairs_per_layout, float scale, uint4* __restrict__ out_pairs) {
    const Job job = jobs[j];
    const WorkNode* nodes = work + job.node_first;
    const uint32_t n_nodes = 2u * job.n_items - 1u;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = idx >> 3, o = idx & 7u;
    if (v >= n_nodes) return;
    const WorkNode nd = nodes[v];
    const uint32_t base = (job.set * 8u + o) * pairs_per_layout;
    if (n_nodes == 1u) { // a single leaf: one record that holds it twice
        const uint4 rec = accel_pack_node(nd.b, (int)o, scale, RT_NODE_LEAF | nd.leaf);
        out_pairs[2u * base] = rec;
        out_pairs[2u * base + 1u] = rec;
        return;
    }
    if (nd.leaf) return;
    uint32_t pos = 0u, depth = 0u, c = v;
    while (c != 0u && depth < n_nodes) {
        const uint32_t p = (uint32_t)nodes[c].parent;
        if (p >= n_nodes) return;
        const uint32_t near = (uint32_t)nodes[p].child[(o >> nodes[p].axis) & 1u];
        if (c != near) pos += (nodes[near].size - 1u) / 2u;
        pos += 1u;
        c = p;
        depth++;
    }
    const uint32_t side = (o >> nd.axis) & 1u;
    const WorkNode near = nodes[nd.child[side]], far = nodes[nd.child[1u - side]];
    const uint32_t at = base + pos, near_at = at + 1u, far_at = at + 1u + (near.size - 1u) / 2u;
    out_pairs[2u * at] = accel_pack_node(near.b, (int)o, scale, near.leaf ? (RT_NODE_LEAF | near.leaf) : near_at);
    out_pairs[2u * at + 1u] = accel_pack_node(far.b, (int)o, scale, far.leaf ? (RT_NODE_LEAF | far.leaf) : far_at);
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

} // namespace

// Everything a build needs on the device, for up to two node sets of n_items leaves in n_trees trees: allocated once
// with the scene, so that a later build for another reach allocates nothing.
struct AccelDeviceWorkspace {
    size_t n_items = 0, n_trees = 0;
    DevBuf<AccelItem> items_in, items, temp;
    DevBuf<Job> jobs;
    DevBuf<WorkNode> work;
    DevBuf<Range> queue_a, queue_b, small;
    DevBuf<uint32_t> depth, counters; // counters: see counters_of()
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~AccelDeviceWorkspace() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return rt_fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int accel_device_workspace_create(size_t n_items, size_t n_trees, AccelDeviceWorkspace** out) {
    if (!out) return rt_fail(RT_EINVAL, "accel_device_workspace_create: null argument");
    *out = nullptr;
    if (4ull * n_items > 0x7FFFFFFFull) return rt_fail(RT_EUNSUPPORTED, "too many primitives for the device tree builder");
    AccelDeviceWorkspace* ws = new (std::nothrow) AccelDeviceWorkspace();
    if (!ws) return rt_fail(RT_ENOMEM, "out of host memory");
    ws->n_items = n_items;
    ws->n_trees = n_trees;
    const size_t n_all = 2 * n_items;
    hipError_t e = ws->items_in.alloc(n_all);
    if (e == hipSuccess) e = ws->items.alloc(n_all);
    if (e == hipSuccess) e = ws->temp.alloc(n_all);
    if (e == hipSuccess) e = ws->jobs.alloc(2 * n_trees);
    if (e == hipSuccess) e = ws->work.alloc(2 * n_all);
    if (e == hipSuccess) e = ws->queue_a.alloc(n_all);
    if (e == hipSuccess) e = ws->queue_b.alloc(n_all);
    if (e == hipSuccess) e = ws->small.alloc(n_all);
    if (e == hipSuccess) e = ws->depth.alloc(1);
    if (e == hipSuccess) e = ws->counters.alloc(2 + 2 * n_trees);
    if (e == hipSuccess) e = hipEventCreate(&ws->e0);
    if (e == hipSuccess) e = hipEventCreate(&ws->e1);
    if (e != hipSuccess) {
        delete ws;
        return rt_fail(e == hipErrorOutOfMemory ? RT_ENOMEM : RT_EHIP, "device tree builder: %s", hipGetErrorString(e));
    }
    *out = ws;
    return RT_OK;
}

void accel_device_workspace_destroy(AccelDeviceWorkspace* ws) { delete ws; }

int accel_device_build(AccelDeviceWorkspace* ws, const std::vector<AccelItem>* items, uint32_t n_sets,
                       const std::vector<AccelDeviceTree>& trees, uint32_t nodes_per_layout, float scale, uint4* d_nodes,
                       uint32_t pairs_per_layout, uint4* d_pairs, AccelDeviceStats* stats) {
    if (stats) *stats = AccelDeviceStats{0u, 0u, 0.0f};
    if (!ws || !items || n_sets < 1u || n_sets > 2u || !d_nodes) return rt_fail(RT_EINVAL, "accel_device_build: bad argument");
    const size_t n_items = items[0].size();
    if (n_sets == 2u && items[1].size() != n_items) return rt_fail(RT_EINVAL, "accel_device_build: the node sets differ in size");
    if (n_items > ws->n_items || trees.size() > ws->n_trees) return rt_fail(RT_EINVAL, "accel_device_build: workspace too small");
    std::vector<Job> jobs;
    size_t n_work = 0;
    for (uint32_t s = 0; s < n_sets; ++s)
        for (const AccelDeviceTree& t : trees) {
            if (t.n_items == 0u) continue;
            if ((size_t)t.item_first + t.n_items > n_items || (uint64_t)t.layout_first + 2ull * t.n_items - 1ull > nodes_per_layout)
                return rt_fail(RT_EINVAL, "accel_device_build: a tree does not fit its layout");
            Job j{};
            j.item_first = (uint32_t)(s * n_items) + t.item_first;
            j.n_items = t.n_items;
            j.layout_first = t.layout_first;
            j.node_first = (uint32_t)n_work;
            j.set = s;
            j.pairs = t.pairs && d_pairs && pairs_per_layout >= (t.n_items > 1u ? t.n_items - 1u : 1u) ? 1u : 0u;
            n_work += 2u * (size_t)t.n_items - 1u;
            jobs.push_back(j);
        }
    if (jobs.empty()) return RT_OK;
    for (uint32_t s = 0; s < n_sets; ++s)
        HIP_TRY(hipMemcpy(ws->items_in.p + s * n_items, items[s].data(), n_items * sizeof(AccelItem), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ws->jobs.p, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(ws->depth.p, 0, sizeof(uint32_t)));
    HIP_TRY(hipMemset(ws->counters.p, 0, (2 + jobs.size()) * sizeof(uint32_t)));
    HIP_TRY(hipEventRecord(ws->e0, nullptr));
    hipLaunchKernelGGL(accel_start_kernel, dim3((uint32_t)jobs.size()), dim3(256), 0, nullptr, ws->items_in.p, ws->jobs.p, ws->work.p, ws->items.p,
                       ws->queue_a.p, ws->small.p, ws->counters.p);
    // level by level: the ranges of a level are independent, so each gets a workgroup or a wave of its own anywhere on
    // the device; the host reads how many the level listed for the next one
    Range *cur = ws->queue_a.p, *nxt = ws->queue_b.p;
    uint32_t counts[2] = {0u, 0u}; // {ranges of the next level, small ranges so far}
    for (uint32_t level = 0;; ++level) {
        HIP_TRY(hipMemcpy(counts, ws->counters.p, sizeof(counts), hipMemcpyDeviceToHost));
        const uint32_t n_cur = counts[0];
        if (n_cur == 0u) break;
        if (n_cur > n_sets * n_items || level > 4096u) return rt_fail(RT_EHIP, "internal: the device tree builder lost count of its ranges");
        HIP_TRY(hipMemset(ws->counters.p, 0, sizeof(uint32_t)));
        hipLaunchKernelGGL(accel_split_blocks_kernel, dim3(n_cur), dim3(BUILD_THREADS), 0, nullptr, ws->jobs.p, ws->work.p, ws->items.p, ws->temp.p,
                           cur, nxt, ws->small.p, ws->counters.p);
        hipLaunchKernelGGL(accel_split_waves_kernel, dim3((n_cur + 3u) / 4u), dim3(256), 0, nullptr, ws->jobs.p, ws->work.p, ws->items.p, ws->temp.p,
                           cur, n_cur, nxt, ws->small.p, ws->counters.p);
        std::swap(cur, nxt);
    }
    if (counts[1] > n_sets * n_items) return rt_fail(RT_EHIP, "internal: the device tree builder lost count of its ranges");
    if (counts[1])
        hipLaunchKernelGGL(accel_sweep_kernel, dim3((counts[1] + 3u) / 4u), dim3(256), 0, nullptr, ws->jobs.p, ws->work.p, ws->items.p, ws->small.p,
                           counts[1]);
    for (uint32_t j = 0; j < (uint32_t)jobs.size(); ++j) {
        const uint32_t threads = (2u * jobs[j].n_items - 1u) * 8u;
        hipLaunchKernelGGL(accel_layout_kernel, dim3((threads + 255u) / 256u), dim3(256), 0, nullptr, ws->work.p, ws->jobs.p, j, nodes_per_layout,
                           scale, d_nodes, ws->depth.p);
        if (jobs[j].pairs)
            hipLaunchKernelGGL(accel_pairs_kernel, dim3((threads + 255u) / 256u), dim3(256), 0, nullptr, ws->work.p, ws->jobs.p, j,
                               pairs_per_layout, scale, d_pairs);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ws->e1, nullptr));
    HIP_TRY(hipEventSynchronize(ws->e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ws->e0, ws->e1));
    uint32_t depth = 0;
    HIP_TRY(hipMemcpy(&depth, ws->depth.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stats) { stats->n_trees = (uint32_t)jobs.size(); stats->max_depth = depth; stats->build_ms = ms; }
    return RT_OK;
}
