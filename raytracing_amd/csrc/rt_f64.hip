// rt_f64.hip -- the Float64 mode of the path: render(Float64, ...) (main.jl:21,46) on gfx950.
//
// The Float32 path (rt_kernels.hip) owes its speed to an arithmetic contract derived for Float32 operation by
// operation (half-precision culling boxes with proven margins, two node sets, hits held against the reference's leaf
// boxes).  None of it carries over to another element type, so this mode takes the other road: it walks THE
// REFERENCE'S OWN TREES (one per type group, src/scene_manager.jl:25-38, as the host built them, src/BVH.jl:65-96,
// src/mesh.jl:217-283) in the reference's own order (src/BVH.jl:22-53, src/mesh.jl:115-163, right child first) with
// the reference's own slab test (src/bounding_box.jl:20-67) in double, group after group with the bound handed on as
// hit(::SceneManager) hands it on (src/scene_manager.jl:40-53).  Every intermediate value is then the reference's by
// construction, and a picture is the CPU path's bit for bit with no tolerance to derive.
//
// Shape on the GPU: one lane = one path (one pattern column at one pixel).  Waves are persistent: a lane whose path has
// ended takes the next unstarted path (one atomic per wave and refill), so a wave stays full until the frame runs
// out.  Every bounce walks the trees with a per-lane stack in LDS (node ids, conflict-free: entry e of lane l at
// e * blockDim + l).  Paths write their radiance to a per-path buffer; rt64_sum_kernel adds them per pixel in
// pattern-column order, which is the order of the sample loop of draw! (src/Ray_Tracing.jl:82-89).
// FP64 vector rate on this part is half the packed-FP32 rate and registers hold half as many values; the scene tables
// stay L2-resident.  No FMA contraction (the file is compiled with -ffp-contract=off like the rest of the library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace {

typedef double real;
#define R64_INF (__builtin_inf())
#define R64_BLOCK 128
#define R64_STACK 40 // stack entries per lane: 20 KB of LDS per workgroup, eight workgroups (four waves per SIMD) per CU

#define HIP_TRY64(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return rt_fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : (e_ == hipErrorNoDevice ? RT_ENODEV : RT_EHIP), \
                           "%s failed: %s", #expr, hipGetErrorString(e_));                                \
    } while (0)

// ------------------------------------------------------------------ device tables
struct __align__(64) Node64 { // src/BVH.jl:5-10 (one 64-byte line: four 16-byte loads)
    real box[6];
    int32_t left, right; // 1-based node ids / object positions, as the reference stores them
    int32_t leaf, first; // first: the object at position `left` of a leaf of a tree over objects
};
// What the walk of a tree over objects reads of a node: the box narrowed to float and the links, half a Node64.  The
// float box decides the reference's box test wherever the rounding of its planes cannot matter (box_verdict32); where
// it can, the walk reads the node's Node64 and decides as before.  inner node: left, right > 0, the children's ids;
// leaf of one object (every leaf of a tree the reference builds): left = the object, right = -1 - its Leaf64::code, so
// that the walk need not read the record's head; leaf of the objects at positions l..r: left = l, right = r with the
// top bit set.  Tree over meshes (src/mesh.jl:250-305): a leaf of either level keeps left
// and holds right = -1 - r.
struct __align__(32) Node32 {
    float box[6];
    int32_t left, right;
};
// What a walk needs of one object, in one record: a plain object as it is, a Transform around a plain object as that
// object with the matrix that takes the ray to it.  The walk reaches a leaf's test after two dependent fetches (node,
// this record) instead of four (node, object list, object, transform and wrapped object).
struct Leaf64 {
    int32_t kind;     // the plain object's kind; for everything else (medium, tree, mesh) the object's own kind
    int32_t material;
    int32_t xf;       // transform index, -1: none
    int32_t code;     // what a walk must know before it reads the rest (a leaf's Node32 carries it too): bits 0-3 kind, bit 4 a
                      // transform, bits 5-7 the kind of a medium's boundary
    real p[12];       // a medium: its boundary's numbers, and in p[11] the medium's own (-1 / density)
    real inv[12];     // src/transform.jl:65: inv_xform, 3x4 column-major (a medium: of the Transform around its boundary)
};
struct Tree64 {
    int32_t node_base, n_nodes, obj_base, n_objects, is_mesh, pad0, pad1, pad2;
};
struct Mesh64 {
    int32_t vertex_base, index_base, material_base, n_triangles;
};
struct Image64 {
    int32_t width, height;
    uint32_t offset, pad;
};
struct Scene64 {
    const rtObject64* objects;
    const Leaf64* fused;       // per object
    const int32_t* top_pos;
    const rtTransform64* transforms;
    const rtMaterial64* materials;
    const rtTexture64* textures;
    const Image64* images;
    const uint8_t* texels;
    const rtPerlin64* perlins;
    const Mesh64* meshes;
    const real* vertices;      // 8 per vertex
    const int32_t* indices;    // 4 per triangle
    const int32_t* tri_ids;
    const int32_t* mesh_materials;
    const Node64* nodes;
    const Node32* nodes32;     // node for node beside `nodes`
    const int32_t* tree_objects;
    const Tree64* trees;       // the type groups of the SceneManager in order, then the nested trees
    const uint8_t* mat_uv;     // per material: 1 if its texture can read (u, v) (an image texture, possibly through checkers)
    int32_t n_buckets, n_groups;
    int32_t all_uv, pad;       // 1: every hit carries its (u, v) (the parity hook reports them); 0: only where a texture reads them
};
struct Frame64 {
    rtCamera64 cam;
    const real* pattern; // 2 x n
    int32_t n, width, height, max_depth;
    int32_t first, stride;       // this call's columns: first, first + stride, ...
    int32_t batch_first, batch_cols; // positions within those columns that this launch traces
    real bg[3];
    uint32_t seed32;
    unsigned long long n_paths;  // batch_cols * width * height
};

// ------------------------------------------------------------------ src/vec3.jl
struct V3 {
    real x, y, z;
};
__device__ inline V3 v3(real x, real y, real z) { V3 a; a.x = x; a.y = y; a.z = z; return a; }
__device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ inline V3 operator*(V3 a, real b) { return v3(a.x * b, a.y * b, a.z * b); }
__device__ inline V3 operator*(real a, V3 b) { return v3(a * b.x, a * b.y, a * b.z); }
__device__ inline V3 div_real(V3 a, real b) { return a * (1.0 / b); } // src/vec3.jl:31: a * F(1/b)
__device__ inline real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline real length(V3 a) { return sqrt(dot(a, a)); }
__device__ inline V3 normalize(V3 a) { return div_real(a, length(a)); }
__device__ inline bool almost_zero(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; } // isapprox(x, 0): x == 0

struct Ray {
    V3 o, d;
    real time;
};
__device__ inline V3 at(const Ray& r, real t) { return r.o + r.d * t; }

// ------------------------------------------------------------------ deterministic math (IEEE basic operations only)
__device__ inline real det_log(real x) { // x in (0,1): x = 2^k m, m in [sqrt(1/2), sqrt(2)); atanh series
    if (x == 0.0) return -R64_INF;
    const real ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const real Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(bits >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    bits = ((unsigned long long)hx << 32) | (bits & 0xffffffffull);
    real m = __longlong_as_double((long long)bits);
    real f = m - 1.0;
    real hfsq = 0.5 * f * f;
    real s = f / (2.0 + f);
    real z = s * s;
    real w = z * z;
    real t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    real t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    real Rr = t2 + t1;
    real dk = (real)k;
    return (((s * (hfsq + Rr) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}
__device__ inline real det_sin(real x) { // quadrant by a two-part pi/2, then the sine / cosine kernels
    const real TWO_OVER_PI = 6.36619772367581382433e-01;
    const real PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    real n = floor(x * TWO_OVER_PI + 0.5);
    real r = (x - n * PIO2_1) - n * PIO2_1T;
    long long q = (long long)n;
    real z = r * r;
    real sn, cs;
    {
        const real S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                   S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        real p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        sn = r + (r * z) * (S1 + z * p);
    }
    {
        const real C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                   C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        real p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        cs = 1.0 - (0.5 * z - z * p);
    }
    switch ((int)(q & 3)) {
        case 0: return sn;
        case 1: return cs;
        case 2: return -sn;
        default: return -cs;
    }
}
__device__ inline real det_atan01(real z) { // z in [0,1]: three half-angle steps, then the odd series
    z = z / (1.0 + sqrt(1.0 + z * z));
    z = z / (1.0 + sqrt(1.0 + z * z));
    z = z / (1.0 + sqrt(1.0 + z * z));
    real w = z * z;
    real p = 1.0 / 19.0;
    p = 1.0 / 17.0 - w * p;
    p = 1.0 / 15.0 - w * p;
    p = 1.0 / 13.0 - w * p;
    p = 1.0 / 11.0 - w * p;
    p = 1.0 / 9.0 - w * p;
    p = 1.0 / 7.0 - w * p;
    p = 1.0 / 5.0 - w * p;
    p = 1.0 / 3.0 - w * p;
    p = 1.0 - w * p;
    return 8.0 * (z * p);
}
__device__ inline real det_atan2(real y, real x) {
    const real PI = 3.14159265358979323846;
    real ax = fabs(x), ay = fabs(y);
    real a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax >= ay) a = det_atan01(ay / ax);
    else a = 0.5 * PI - det_atan01(ax / ay);
    if (__double_as_longlong(x) < 0) a = PI - a;
    if (__double_as_longlong(y) < 0) a = -a;
    return a;
}
__device__ inline real det_acos(real c) {
    real s = sqrt((1.0 - c) * (1.0 + c));
    return det_atan2(s, c);
}

// ------------------------------------------------------------------ random numbers: pcg4d(pixel, sample, counter, seed)
__device__ inline void pcg4d(uint32_t v[4]) {
    for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
// rand(Float64): 53 bits from two words, 27 from the first and 26 from the second
__device__ inline real u64_to_01(uint32_t hi, uint32_t lo) {
    return ((real)(hi >> 5) * 67108864.0 + (real)(lo >> 6)) * (1.0 / 9007199254740992.0);
}
enum { PURPOSE_TRIAL = 0, PURPOSE_SCALAR = 1, PURPOSE_MEDIUM = 2 };
enum { CAMERA_BOUNCE = 63 };
struct Ctx {
    uint32_t pixel, sample, seed32, bounce;
    __device__ uint32_t counter(uint32_t purpose, uint32_t k) const { return (bounce << 26) | (purpose << 24) | (k & 0xFFFFFFu); }
    // numbers 0 and 1 of draw k
    __device__ void draw2(uint32_t purpose, uint32_t k, real& a, real& b) const {
        uint32_t v[4] = {pixel, sample, counter(purpose, 2 * k), seed32};
        pcg4d(v);
        a = u64_to_01(v[0], v[1]);
        b = u64_to_01(v[2], v[3]);
    }
    // number 2 of draw k
    __device__ real draw_third(uint32_t purpose, uint32_t k) const {
        uint32_t v[4] = {pixel, sample, counter(purpose, 2 * k + 1), seed32};
        pcg4d(v);
        return u64_to_01(v[0], v[1]);
    }
};
__device__ V3 random_in_unit_sphere(const Ctx& c) { // src/vec3.jl:87-94
    for (uint32_t k = 0;; ++k) {
        real a, b;
        c.draw2(PURPOSE_TRIAL, k, a, b);
        real d = c.draw_third(PURPOSE_TRIAL, k);
        V3 p = v3(a * 2.0 + -1.0, b * 2.0 + -1.0, d * 2.0 + -1.0);
        if (!(dot(p, p) > 1.0)) return p;
    }
}
__device__ V3 random_in_unit_disk(const Ctx& c) { // src/vec3.jl:96-105
    for (uint32_t k = 0;; ++k) {
        real a, b;
        c.draw2(PURPOSE_TRIAL, k, a, b);
        V3 p = v3(a * 2.0 - 1.0, b * 2.0 - 1.0, 0.0);
        if (!(dot(p, p) > 1.0)) return p;
    }
}
__device__ real rand_scalar(const Ctx& c) {
    real a, b;
    c.draw2(PURPOSE_SCALAR, 0, a, b);
    return a;
}

// ------------------------------------------------------------------ src/hit_record.jl
struct Hit {
    real t;
    V3 p, n;
    real u, v;
    int32_t material, prim, sub;
    bool front;
};
__device__ inline Hit miss() {
    Hit h;
    h.t = R64_INF;
    h.p = v3(R64_INF, R64_INF, R64_INF);
    h.n = v3(0.0, 0.0, 0.0);
    h.u = 0.0; h.v = 0.0;
    h.material = -1; h.prim = -1; h.sub = 0;
    h.front = false;
    return h;
}

// ------------------------------------------------------------------ src/bounding_box.jl:20-67 (the CPU slab test)
__device__ inline bool hit_aabb(const real* b, const Ray& r, real t_min, real t_max) {
    real tmin = (b[0] - r.o.x) / r.d.x;
    real tmax = (b[3] - r.o.x) / r.d.x;
    if (tmin > tmax) { real s = tmin; tmin = tmax; tmax = s; }
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;

    tmin = (b[1] - r.o.y) / r.d.y;
    tmax = (b[4] - r.o.y) / r.d.y;
    if (tmin > tmax) { real s = tmin; tmin = tmax; tmax = s; }
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;

    tmin = (b[2] - r.o.z) / r.d.z;
    tmax = (b[5] - r.o.z) / r.d.z;
    if (tmin > tmax) { real s = tmin; tmin = tmax; tmax = s; }
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;
    return true;
}

// The same decision from reciprocal distances, where that is certain.  The box test only decides which subtrees are
// walked, so all that matters is its outcome: with inv = 1 / d rounded once per ray, (b - o) * inv differs from the
// reference's (b - o) / d by less than 3.2 ulp, the minima and maxima of such values keep that bound, and a
// comparison of two of them has the reference's outcome whenever they are further apart than 2^-50 of their
// magnitudes.  Every one of the reference's three `t_max <= t_min` decisions is taken that way; the first that is
// not certain (grazing rays, infinities and NaNs of axis-parallel rays) hands the whole test to the divisions above.
// Six IEEE double divisions per node become six multiplications.
__device__ inline bool hit_aabb_fast(const real* b, const Ray& r, const V3& inv, real t_min0, real t_max0) {
    const real TOL = 8.881784197001252e-16; // 2^-50
    real t_min = t_min0, t_max = t_max0;
    real lo = (b[0] - r.o.x) * inv.x, hi = (b[3] - r.o.x) * inv.x;
    if (lo > hi) { real s = lo; lo = hi; hi = s; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    real gap = t_max - t_min, tol = TOL * (fabs(t_max) + fabs(t_min)) + 1e-300;
    if (gap < -tol) return false;
    if (!(gap > tol)) return hit_aabb(b, r, t_min0, t_max0);

    lo = (b[1] - r.o.y) * inv.y; hi = (b[4] - r.o.y) * inv.y;
    if (lo > hi) { real s = lo; lo = hi; hi = s; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    gap = t_max - t_min; tol = TOL * (fabs(t_max) + fabs(t_min)) + 1e-300;
    if (gap < -tol) return false;
    if (!(gap > tol)) return hit_aabb(b, r, t_min0, t_max0);

    lo = (b[2] - r.o.z) * inv.z; hi = (b[5] - r.o.z) * inv.z;
    if (lo > hi) { real s = lo; lo = hi; hi = s; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    gap = t_max - t_min; tol = TOL * (fabs(t_max) + fabs(t_min)) + 1e-300;
    if (gap < -tol) return false;
    if (!(gap > tol)) return hit_aabb(b, r, t_min0, t_max0);
    return true;
}
__device__ inline V3 reciprocal(V3 d) { return v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z); }

// box_verdict32: the reference's box test decided on a node's float box (Node32) -- 1 it passes, 0 it fails, -1 not
// certain (then the walk takes the planes in double through hit_aabb_fast).
struct RayF { // what box_verdict32 takes of a ray: origin and reciprocal direction as floats, and per axis 2^-22 |1/d|
    float ox, oy, oz, ix, iy, iz, kx, ky, kz;
};
__device__ inline float slack_of(real inv) { // Inf where float arithmetic on this axis says nothing (or the axis is parallel to the planes)
    const real a = fabs(inv);
    return (a >= 7.888609052210118e-31 && a <= 1.2676506002282294e+30) ? (float)(a * 2.384185791015625e-07) : __builtin_inff(); // 2^-100 .. 2^100
}
__device__ inline RayF ray_f32(const Ray& r, const V3& inv) {
    RayF f;
    f.ox = (float)r.o.x; f.oy = (float)r.o.y; f.oz = (float)r.o.z;
    f.ix = (float)inv.x; f.iy = (float)inv.y; f.iz = (float)inv.z;
    f.kx = slack_of(inv.x); f.ky = slack_of(inv.y); f.kz = slack_of(inv.z);
    return f;
}
// The whole test in float.  A distance (b - o) * (1/d) computed from the float plane, the float origin and the float
// reciprocal is off from the reference's (b - o) / d by at most 2^-24 (|b| + |o|) |1/d| (the two narrowings) plus four
// roundings of 2^-24 relative; the bound of the walk narrowed to float is off by 2^-24 relative.  The reference's three
// decisions `t_max <= t_min` are one: its interval only shrinks from axis to axis, so one of them holds if and only if
// the last one does.  That one is certain when the two ends are further apart than
//     e + 2^-20 (|t_max| + |t_min|) + 2^-100,   e = max over the axes of 2^-22 (max(|b|, 1) + |o|) |1/d|,
// twice and more what the ends can be off by together (planes count as at least 1: that covers planes below the normal
// float range, which are off by up to 2^-150 instead; 2^-100 covers products below it).  Overflow gives Inf or NaN on
// both sides of a comparison only where the true values are beyond the float range too, and then the answer is "not
// certain" or the order is the true one.  NaN distances drop out of min and max as they drop out of the reference's
// comparisons; a NaN beside a number needs an infinite plane, origin or direction, and then e is Inf or NaN.
__device__ inline int box_verdict32(const Node32& q, const RayF& f, real t_min, real t_max) {
    float lo = (q.box[0] - f.ox) * f.ix, hi = (q.box[3] - f.ox) * f.ix;
    float tmax = fminf((float)t_max, fmaxf(lo, hi));
    float tmin = fmaxf((float)t_min, fminf(lo, hi));
    lo = (q.box[1] - f.oy) * f.iy; hi = (q.box[4] - f.oy) * f.iy;
    tmax = fminf(tmax, fmaxf(lo, hi));
    tmin = fmaxf(tmin, fminf(lo, hi));
    lo = (q.box[2] - f.oz) * f.iz; hi = (q.box[5] - f.oz) * f.iz;
    tmax = fminf(tmax, fmaxf(lo, hi));
    tmin = fmaxf(tmin, fminf(lo, hi));
    float e = (fmaxf(fmaxf(fabsf(q.box[0]), fabsf(q.box[3])), 1.0f) + fabsf(f.ox)) * f.kx;
    e = fmaxf(e, (fmaxf(fmaxf(fabsf(q.box[1]), fabsf(q.box[4])), 1.0f) + fabsf(f.oy)) * f.ky);
    e = fmaxf(e, (fmaxf(fmaxf(fabsf(q.box[2]), fabsf(q.box[5])), 1.0f) + fabsf(f.oz)) * f.kz);
    const float gap = tmax - tmin, tol = e + 9.5367431640625e-07f * (fabsf(tmax) + fabsf(tmin)) + 7.888609052210118e-31f;
    if (gap < -tol) return 0;
    if (!(gap > tol)) return -1;
    return 1;
}

// ------------------------------------------------------------------ primitives
__device__ inline void get_sphere_uv(V3 p, real& u, real& v) { // src/sphere.jl:104-108
    const real PI = 3.14159265358979323846;
    const real TWO_PI = 2.0 * 3.14159265358979323846;
    real cy = -p.y;
    cy = cy < -1.0 ? -1.0 : (cy > 1.0 ? 1.0 : cy);
    real theta = det_acos(cy);
    real phi = det_atan2(-p.z, p.x) + PI;
    u = phi / TWO_PI;
    v = theta / PI;
}
// want_uv: the reference computes (u, v) -- an acos and an atan -- for every candidate hit; nothing but an image texture
// reads them, so the render kernels leave them 0 for every other material (radiance unchanged), the hook asks for all
__device__ Hit hit_sphere(V3 center, real radius, int material, const Ray& r, real t_min, real t_max, bool want_uv) { // src/sphere.jl:45-102
    V3 oc = r.o - center;
    real a = dot(r.d, r.d);
    real half_b = dot(oc, r.d);
    real c = dot(oc, oc) - radius * radius;
    real discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return miss();
    real sqrtd = sqrt(discriminant);
    real root = (-half_b - sqrtd);
    if (root < t_min * a || t_max * a < root) {
        root = (-half_b + sqrtd);
        if (root < t_min * a || t_max * a < root) return miss();
    }
    root /= a;
    Hit h = miss();
    h.t = root;
    h.p = at(r, root);
    V3 normal = div_real(h.p - center, radius);
    h.front = dot(r.d, normal) < 0.0;
    if (want_uv) get_sphere_uv(normal, h.u, h.v);
    h.n = (real)(-1 + ((int)h.front << 1)) * normal;
    h.material = material;
    return h;
}
// src/aarect.jl:49-119; axis = the constant axis (2: xy_rect, 1: xz_rect, 0: yz_rect)
__device__ Hit hit_rect(int axis, real a0, real a1, real b0, real b1, real k, int material, const Ray& r, real t_min, real t_max, bool want_uv) {
    real ok, dk, oa, da, ob, db;
    if (axis == 2) { ok = r.o.z; dk = r.d.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
    else if (axis == 1) { ok = r.o.y; dk = r.d.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
    else { ok = r.o.x; dk = r.d.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
    real t = (k - ok) / dk;
    if (t < t_min || t_max < t) return miss();
    real a = oa + t * da;
    real b = ob + t * db;
    if (a < a0 || a1 < a || b < b0 || b1 < b) return miss();
    Hit h = miss();
    h.t = t;
    if (axis == 2) h.p = v3(a, b, k);
    else if (axis == 1) h.p = v3(a, k, b);
    else h.p = v3(k, a, b);
    V3 normal = axis == 2 ? v3(0, 0, 1) : (axis == 1 ? v3(0, 1, 0) : v3(1, 0, 0));
    h.front = dot(r.d, normal) < 0.0;
    if (want_uv) {
        h.u = (a - a0) / (a1 - a0);
        h.v = (b - b0) / (b1 - b0);
    }
    h.n = (real)(-1 + 2 * (int)h.front) * normal;
    h.material = material;
    return h;
}
__device__ Hit hit_box(const real* p, int material, const Ray& r, real t_min, real t_max, bool want_uv) { // src/box.jl:9-24,43-83
    Hit closest = miss();
    for (int s = 0; s < 6; ++s) {
        Hit h;
        switch (s) {
            case 0: h = hit_rect(2, p[0], p[3], p[1], p[4], p[5], material, r, t_min, t_max, want_uv); break;
            case 1: h = hit_rect(2, p[0], p[3], p[1], p[4], p[2], material, r, t_min, t_max, want_uv); break;
            case 2: h = hit_rect(1, p[0], p[3], p[2], p[5], p[4], material, r, t_min, t_max, want_uv); break;
            case 3: h = hit_rect(1, p[0], p[3], p[2], p[5], p[1], material, r, t_min, t_max, want_uv); break;
            case 4: h = hit_rect(0, p[1], p[4], p[2], p[5], p[3], material, r, t_min, t_max, want_uv); break;
            default: h = hit_rect(0, p[1], p[4], p[2], p[5], p[0], material, r, t_min, t_max, want_uv); break;
        }
        if (h.t < closest.t) {
            closest = h;
            closest.sub = s;
            t_max = h.t;
        }
    }
    return closest;
}
// Sphere, MovingSphere, the three rectangles, Box
__device__ Hit hit_basic(const Scene64& S, int kind, int material, const real* p, const Ray& r, real t_min, real t_max) {
    const bool want_uv = S.all_uv || S.mat_uv[material];
    struct { int kind, material; } o = {kind, material};
    switch (o.kind) {
        case RT_SPHERE: return hit_sphere(v3(p[0], p[1], p[2]), p[3], o.material, r, t_min, t_max, want_uv);
        case RT_MOVING_SPHERE: { // src/sphere.jl:24-26
            V3 c0 = v3(p[0], p[1], p[2]), c1 = v3(p[4], p[5], p[6]);
            V3 center = c0 + ((r.time - p[3]) / (p[7] - p[3])) * (c1 - c0);
            return hit_sphere(center, p[8], o.material, r, t_min, t_max, want_uv);
        }
        case RT_XY_RECT: return hit_rect(2, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max, want_uv);
        case RT_XZ_RECT: return hit_rect(1, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max, want_uv);
        case RT_YZ_RECT: return hit_rect(0, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max, want_uv);
        case RT_BOX: return hit_box(p, o.material, r, t_min, t_max, want_uv);
    }
    return miss();
}
// the distance alone, for the walk: (u, v) never decide it
__device__ inline real hit_basic_t(int kind, const real* p, const Ray& r, real t_min, real t_max) {
    switch (kind) {
        case RT_SPHERE: return hit_sphere(v3(p[0], p[1], p[2]), p[3], 0, r, t_min, t_max, false).t;
        case RT_MOVING_SPHERE: { // src/sphere.jl:24-26
            V3 c0 = v3(p[0], p[1], p[2]), c1 = v3(p[4], p[5], p[6]);
            V3 center = c0 + ((r.time - p[3]) / (p[7] - p[3])) * (c1 - c0);
            return hit_sphere(center, p[8], 0, r, t_min, t_max, false).t;
        }
        case RT_XY_RECT: return hit_rect(2, p[0], p[1], p[2], p[3], p[4], 0, r, t_min, t_max, false).t;
        case RT_XZ_RECT: return hit_rect(1, p[0], p[1], p[2], p[3], p[4], 0, r, t_min, t_max, false).t;
        case RT_YZ_RECT: return hit_rect(0, p[0], p[1], p[2], p[3], p[4], 0, r, t_min, t_max, false).t;
        case RT_BOX: return hit_box(p, 0, r, t_min, t_max, false).t;
    }
    return R64_INF;
}
// src/transform.jl:76-98
__device__ inline V3 apply_point(const real* T, V3 v) {
    return v3(T[0] * v.x + T[3] * v.y + T[6] * v.z + T[9], T[1] * v.x + T[4] * v.y + T[7] * v.z + T[10],
              T[2] * v.x + T[5] * v.y + T[8] * v.z + T[11]);
}
__device__ inline V3 apply_vec(const real* T, V3 v) {
    return v3(T[0] * v.x + T[3] * v.y + T[6] * v.z, T[1] * v.x + T[4] * v.y + T[7] * v.z, T[2] * v.x + T[5] * v.y + T[8] * v.z);
}
// a basic object, or one inside a Transform (src/transform.jl:104-116)
__device__ Hit hit_simple(const Scene64& S, int oi, const Ray& r, real t_min, real t_max) {
    const rtObject64& o = S.objects[oi];
    if (o.kind != RT_TRANSFORM) return hit_basic(S, o.kind, o.material, o.p, r, t_min, t_max);
    const rtTransform64& T = S.transforms[o.aux];
    Ray rr;
    rr.o = apply_point(T.inv_xform, r.o);
    rr.d = apply_vec(T.inv_xform, r.d);
    rr.time = r.time;
    const rtObject64& ch = S.objects[o.child];
    Hit h = hit_basic(S, ch.kind, ch.material, ch.p, rr, t_min, t_max);
    if (h.t == R64_INF) return h;
    h.p = at(r, h.t);
    h.n = apply_vec(T.xform, h.n);
    return h;
}
__device__ Hit hit_medium(const Scene64& S, int oi, const Ray& r, real t_min, real t_max, const Ctx& c) { // src/constant_medium.jl:29-65
    const rtObject64& o = S.objects[oi];
    real t1 = hit_simple(S, o.child, r, -R64_INF, R64_INF).t;
    if (t1 == R64_INF) return miss();
    real t2 = hit_simple(S, o.child, r, t1 + 0.0001, R64_INF).t;
    if (t2 == R64_INF) return miss();
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return miss();
    if (t1 < 0.0) t1 = 0.0;
    real ray_length = length(r.d);
    real distance_inside_boundary = (t2 - t1) * ray_length;
    real a, b;
    c.draw2(PURPOSE_MEDIUM, (uint32_t)oi, a, b);
    real hit_distance = o.p[0] * det_log(a);
    if (hit_distance > distance_inside_boundary) return miss();
    Hit h = miss();
    h.t = t1 + hit_distance / ray_length;
    h.p = at(r, h.t);
    h.n = v3(1, 0, 0);
    h.material = o.material;
    h.front = false;
    return h;
}

// src/triangle.jl:73-91
__device__ real hit_triangle_geom(V3 v1, V3 v2, V3 v3_, const Ray& r, real t_min, real t_max, V3& normal, real& u, real& v) {
    V3 e1 = v2 - v1;
    V3 e2 = v3_ - v1;
    V3 origin_v1 = v1 - r.o;
    normal = cross(e1, e2);
    V3 s = cross(origin_v1, r.d);
    real c = 1.0 / dot(normal, r.d);
    u = dot(s, e2) * c;
    v = dot(s, e1) * -c;
    real t = dot(normal, origin_v1) * c;
    if (u < 0.0 || v < 0.0 || u + v > 1.0 || t < t_min || t_max < t) {
        u = -1.0; v = -1.0;
        return R64_INF;
    }
    return t;
}
// src/mesh.jl:53-59 + src/triangle.jl:95-121; col is 1-based
__device__ Hit hit_mesh_triangle(const Scene64& S, const Mesh64& m, int col, const Ray& r, real t_min, real t_max) {
    const int32_t* id = S.indices + 4 * ((size_t)m.index_base + (size_t)(col - 1));
    const real* a = S.vertices + 8 * ((size_t)m.vertex_base + (size_t)id[0]);
    const real* b = S.vertices + 8 * ((size_t)m.vertex_base + (size_t)id[1]);
    const real* cc = S.vertices + 8 * ((size_t)m.vertex_base + (size_t)id[2]);
    V3 normal;
    real u, v;
    real t = hit_triangle_geom(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]), v3(cc[0], cc[1], cc[2]), r, t_min, t_max, normal, u, v);
    if (t == R64_INF) return miss();
    real w = 1.0 - u - v;
    V3 vn = (v3(a[3], a[4], a[5]) * w + v3(b[3], b[4], b[5]) * u) + v3(cc[3], cc[4], cc[5]) * v;
    real vu = (a[6] * w + b[6] * u) + cc[6] * v;
    real vv = (a[7] * w + b[7] * u) + cc[7] * v;
    // src/triangle.jl:110-115: the flag holds when the interpolated normal is exactly zero, so a mesh that carries
    // normals is shaded with the geometric one
    bool normal_flag = almost_zero(vn);
    bool uv_flag = vu >= 0.0 && vv >= 0.0;
    if (normal_flag) normal = vn;
    if (uv_flag) { u = vu; v = vv; }
    Hit h = miss();
    h.t = t;
    h.p = at(r, t);
    h.front = dot(r.d, normal) < 0.0;
    h.n = (real)(-1 + ((int)h.front << 1)) * normalize(normal);
    h.material = S.mesh_materials[m.material_base + id[3]];
    h.u = u; h.v = v;
    h.sub = S.tri_ids[(size_t)m.index_base + (size_t)(col - 1)];
    return h;
}

// ------------------------------------------------------------------ trees
// While the trees are walked a hit is only what decides the walk: its distance, which object gave it and the bound
// that object was tested with.  The record the reference builds for every candidate (point, normal, uv, material:
// src/hit_record.jl) is built once, for the hit that is left at the end, by running that object's test again with the
// same ray and the same bound (complete_hit): the same instructions on the same inputs, so the same bits.  It keeps a
// walk's live state at a few registers per level of nesting instead of a full record each, which is what decides
// how many waves a SIMD holds -- and the walk is bound by the latency of its node fetches, not by arithmetic.
struct Lite {
    real t;           // Inf: no hit
    real bound;       // t_max of the test that found it
    real inner_bound; // nested tree: t_max of the inner object's test
    int32_t oi;       // the object of the walked tree (for a tree of meshes: the mesh)
    int32_t inner;    // nested tree: the object inside it; tree of meshes: the column of the triangle (1-based)
};
__device__ inline Lite lite_miss() {
    Lite l;
    l.t = R64_INF; l.bound = R64_INF; l.inner_bound = R64_INF; l.oi = -1; l.inner = -1;
    return l;
}

// stk: the calling lane's column of the stack in LDS (entry e at stk[e * R64_BLOCK]); sp0: entries already in use
// GROUPS: the scene has nested trees (BVHTree objects); a scene without them is walked by code that does not hold the
// registers a second, inner walk needs (128 instead of 168: four waves per SIMD instead of three)
template <bool NESTED, bool GROUPS>
__device__ Lite hit_tree(const Scene64& S, int tree, const Ray& r, const RayF& rf, real t_min, real t_max, const Ctx& c, uint32_t* stk, int sp0);

// the ray in the frame of a Transform (src/transform.jl:105-107)
__device__ inline Ray to_object_frame(const rtTransform64& T, const Ray& r) {
    Ray rr;
    rr.o = apply_point(T.inv_xform, r.o);
    rr.d = apply_vec(T.inv_xform, r.d);
    rr.time = r.time;
    return rr;
}

// hit_medium's distance from the medium's one record (its boundary's numbers and transform are in it): the same
// operations on the same values, so the same distance
__device__ inline real hit_medium_t(const Leaf64& L, int code, int oi, const Ray& r, real t_min, real t_max, const Ctx& c) { // src/constant_medium.jl:29-65
    const int boundary = code >> 5;
    Ray rr = r;
    if (code & 16) {
        rr.o = apply_point(L.inv, r.o);
        rr.d = apply_vec(L.inv, r.d);
    }
    real t1 = hit_basic_t(boundary, L.p, rr, -R64_INF, R64_INF);
    if (t1 == R64_INF) return R64_INF;
    real t2 = hit_basic_t(boundary, L.p, rr, t1 + 0.0001, R64_INF);
    if (t2 == R64_INF) return R64_INF;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return R64_INF;
    if (t1 < 0.0) t1 = 0.0;
    real ray_length = length(r.d);
    real distance_inside_boundary = (t2 - t1) * ray_length;
    real a, b;
    c.draw2(PURPOSE_MEDIUM, (uint32_t)oi, a, b);
    real hit_distance = L.p[11] * det_log(a);
    if (hit_distance > distance_inside_boundary) return R64_INF;
    return t1 + hit_distance / ray_length;
}

// distance of the hit of object oi (Inf: none); for a nested tree also which object inside it and with what bound
template <bool NESTED, bool GROUPS>
// code: the object's Leaf64::code (from the leaf's node, so the record's head is not read)
__device__ inline real hit_object_t(const Scene64& S, int oi, int code, const Ray& r, real t_min, real t_max, const Ctx& c, uint32_t* stk, int sp,
                                    int32_t& inner, real& inner_bound) {
    const Leaf64& L = S.fused[oi];
    const int kind = code & 15;
    if (kind >= RT_SPHERE && kind <= RT_BOX) { // a plain object, possibly inside a Transform: the distance carries over
        if (!(code & 16)) return hit_basic_t(kind, L.p, r, t_min, t_max);
        Ray rr;
        rr.o = apply_point(L.inv, r.o);
        rr.d = apply_vec(L.inv, r.d);
        rr.time = r.time;
        return hit_basic_t(kind, L.p, rr, t_min, t_max);
    }
    if (!NESTED && kind == RT_CONSTANT_MEDIUM) return hit_medium_t(L, code, oi, r, t_min, t_max, c);
    const rtObject64& o = S.objects[oi];
    if (!NESTED) {
        const bool wrapped = o.kind == RT_TRANSFORM && S.objects[o.child].kind == RT_BVH;
        if (GROUPS && (o.kind == RT_BVH || wrapped)) {
            Lite l;
            if (wrapped) {
                Ray rr = to_object_frame(S.transforms[o.aux], r);
                l = hit_tree<true, GROUPS>(S, S.n_buckets + S.objects[o.child].aux, rr, ray_f32(rr, reciprocal(rr.d)), t_min, t_max, c, stk, sp);
            } else {
                l = hit_tree<true, GROUPS>(S, S.n_buckets + o.aux, r, ray_f32(r, reciprocal(r.d)), t_min, t_max, c, stk, sp);
            }
            inner = l.oi;
            inner_bound = l.bound;
            return l.t;
        }
    }
    return hit_simple(S, oi, r, t_min, t_max).t;
}

template <bool NESTED, bool GROUPS>
__device__ Lite hit_tree(const Scene64& S, int tree, const Ray& r, const RayF& rf, real t_min, real t_max, const Ctx& c, uint32_t* stk, int sp0) { // src/BVH.jl:22-53
    const Tree64 T = S.trees[tree];
    Lite closest = lite_miss();
    if (T.n_nodes == 0) return closest;
    int n = sp0;
    stk[(n++) * R64_BLOCK] = 1u;
    while (n > sp0) {
        const size_t at_node = (size_t)T.node_base + stk[(--n) * R64_BLOCK] - 1u;
        const Node32 node = S.nodes32[at_node];
        int pass = box_verdict32(node, rf, t_min, t_max);
        if (pass < 0) pass = hit_aabb_fast(S.nodes[at_node].box, r, reciprocal(r.d), t_min, t_max) ? 1 : 0; // (rare: the reciprocals are not kept for it)
        if (!pass) continue;
        if (node.right < 0) {
            // src/hittable.jl:21-31 over objects[left:right]
            Lite lh = lite_miss();
            real tm = t_max;
            const bool one = node.right >= -256;
            const int first = one ? 0 : node.left, last = one ? 0 : (int)((uint32_t)node.right & 0x7fffffffu);
            for (int i = first; i <= last; ++i) {
                int oi = one ? node.left : S.tree_objects[T.obj_base + i - 1];
                const int code = one ? -1 - node.right : S.fused[oi].code;
                int32_t inner = -1;
                real inner_bound = R64_INF;
                real t = hit_object_t<NESTED, GROUPS>(S, oi, code, r, t_min, tm, c, stk, n, inner, inner_bound);
                if (t < lh.t) {
                    lh.t = t; lh.bound = tm; lh.oi = oi; lh.inner = inner; lh.inner_bound = inner_bound;
                    tm = t;
                }
            }
            if (lh.t < closest.t) { t_max = lh.t; closest = lh; }
        } else {
            stk[(n++) * R64_BLOCK] = (uint32_t)node.left;
            stk[(n++) * R64_BLOCK] = (uint32_t)node.right;
        }
    }
    return closest;
}

// src/mesh.jl:115-163: the CPU walk of a tree over meshes (t_max is not narrowed)
__device__ Lite hit_mesh_tree(const Scene64& S, int tree, const Ray& r, const RayF& rf, real t_min, real t_max, uint32_t* stk) {
    const Tree64 T = S.trees[tree];
    Lite closest = lite_miss();
    if (T.n_nodes == 0) return closest;
    int n = 1;
    stk[0] = 1u;
    int active_mesh_index = 0;
    while (n > 0) {
        if (active_mesh_index == n) {
            active_mesh_index = 0;
            n -= 1;
            continue;
        }
        const size_t at_node = (size_t)T.node_base + stk[(n - 1) * R64_BLOCK] - 1u;
        const Node32 q = S.nodes32[at_node];
        n -= 1;
        int pass = box_verdict32(q, rf, t_min, t_max);
        if (pass < 0) pass = hit_aabb_fast(S.nodes[at_node].box, r, reciprocal(r.d), t_min, t_max) ? 1 : 0;
        if (!pass) continue;
        struct { int left, right; bool leaf; } node = {q.left, q.right < 0 ? -1 - q.right : q.right, q.right < 0}; // (a leaf of a tree over meshes: right = -1 - r)
        if (node.leaf) {
            if (active_mesh_index == 0) {
                stk[(n++) * R64_BLOCK] = (uint32_t)node.right;
                active_mesh_index = n;
                stk[(n++) * R64_BLOCK] = (uint32_t)node.left;
            } else {
                int oi = S.tree_objects[T.obj_base + (int)stk[(active_mesh_index - 1) * R64_BLOCK] - 1];
                const Mesh64 m = S.meshes[S.objects[oi].aux];
                Lite lh = lite_miss(); // src/mesh.jl:61-71
                real tm = t_max;
                for (int i = node.left; i <= node.right; ++i) {
                    real t = hit_mesh_triangle(S, m, i, r, t_min, tm).t;
                    if (t < lh.t) { lh.t = t; lh.bound = tm; lh.oi = oi; lh.inner = i; tm = t; }
                }
                if (lh.t < closest.t) closest = lh;
            }
        } else {
            stk[(n++) * R64_BLOCK] = (uint32_t)node.left;
            stk[(n++) * R64_BLOCK] = (uint32_t)node.right;
        }
    }
    return closest;
}

// The record of the hit that is left: the winning object's test once more (see Lite).
__device__ Hit complete_hit(const Scene64& S, const Ray& r, real t_min, const Lite& l, const Ctx& c) {
    if (l.t == R64_INF) return miss();
    const rtObject64& o = S.objects[l.oi];
    Hit h;
    if (o.kind == RT_MESH) {
        h = hit_mesh_triangle(S, S.meshes[o.aux], l.inner, r, t_min, l.bound);
    } else if (o.kind == RT_CONSTANT_MEDIUM) {
        h = hit_medium(S, l.oi, r, t_min, l.bound, c);
    } else if (o.kind == RT_BVH) {
        h = hit_simple(S, l.inner, r, t_min, l.inner_bound);
        h.sub = l.inner;
    } else if (o.kind == RT_TRANSFORM && S.objects[o.child].kind == RT_BVH) { // src/transform.jl:104-116 around the tree
        const rtTransform64& T = S.transforms[o.aux];
        h = hit_simple(S, l.inner, to_object_frame(T, r), t_min, l.inner_bound);
        h.sub = l.inner;
        h.p = at(r, h.t);
        h.n = apply_vec(T.xform, h.n);
    } else {
        h = hit_simple(S, l.oi, r, t_min, l.bound);
    }
    h.prim = S.top_pos[l.oi];
    return h;
}

// src/scene_manager.jl:40-53: the closest over the type groups; the bound moves only when a later group's hit replaces
// the one held
template <bool GROUPS>
__device__ Lite hit_scene_lite(const Scene64& S, const Ray& r, real t_min, real t_max, const Ctx& c, uint32_t* stk) {
    Lite a = lite_miss();
    real closest_t = t_max;
    const RayF rf = ray_f32(r, reciprocal(r.d));
    for (int b = 0; b < S.n_buckets; ++b) {
        Lite h = S.trees[b].is_mesh ? hit_mesh_tree(S, b, r, rf, t_min, closest_t, stk) : hit_tree<false, GROUPS>(S, b, r, rf, t_min, closest_t, c, stk, 0);
        if (b == 0) { a = h; continue; }
        if (a.t <= h.t) { /* keep a */ }
        else { closest_t = h.t; a = h; }
    }
    return a;
}
__device__ Hit hit_scene(const Scene64& S, const Ray& r, real t_min, real t_max, const Ctx& c, uint32_t* stk) {
    return complete_hit(S, r, t_min, hit_scene_lite<true>(S, r, t_min, t_max, c, stk), c);
}

// ------------------------------------------------------------------ src/perlin.jl, src/texture.jl
__device__ real perlin_noise(const rtPerlin64& P, V3 p) { // src/perlin.jl:48-85
    real fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    real uu = u * u * (3.0 - 2.0 * u);
    real vv = v * v * (3.0 - 2.0 * v);
    real ww = w * w * (3.0 - 2.0 * w);
    real accum = 0.0;
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                int idx = P.perm_x[(i + ii + 1) & 255] ^ P.perm_y[(j + jj + 1) & 255] ^ P.perm_z[(k + kk + 1) & 255];
                V3 cv = v3(P.ranvec[3 * idx], P.ranvec[3 * idx + 1], P.ranvec[3 * idx + 2]);
                V3 weight_v = v3(u - (real)ii, v - (real)jj, w - (real)kk);
                real fi = (real)ii, fj = (real)jj, fk = (real)kk;
                accum += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                         (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(cv, weight_v);
            }
    return accum;
}
__device__ real perlin_noise_scalar(const rtPerlin64& P, V3 p) { // src/perlin.jl:48-68,87-97 (Perlin3{F}: ranvec[3 i] is the scalar)
    real fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    real accum = 0.0;
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                int idx = P.perm_x[(i + ii + 1) & 255] ^ P.perm_y[(j + jj + 1) & 255] ^ P.perm_z[(k + kk + 1) & 255];
                real fi = (real)ii, fj = (real)jj, fk = (real)kk;
                accum += (fi * u + (1.0 - fi) * (1.0 - u)) * (fj * v + (1.0 - fj) * (1.0 - v)) *
                         (fk * w + (1.0 - fk) * (1.0 - w)) * P.ranvec[3 * idx];
            }
    return accum;
}
__device__ real turbulence(const rtPerlin64& P, V3 p) { // src/perlin.jl:35-46, depth 7
    real accum = 0.0, weight = 1.0;
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(P, p);
        weight *= 0.5;
        p = p * 2.0;
    }
    return fabs(accum);
}
__device__ V3 texture_value(const Scene64& S, int ti, real u, real v, V3 p) { // src/texture.jl:42-100
    // a checker hands on to one of its two textures, which the host lists before it
    for (int guard = 0; guard < 16 && S.textures[ti].kind == RT_TEX_CHECKER; ++guard) {
        real sines = det_sin(10.0 * p.x) * det_sin(10.0 * p.y) * det_sin(10.0 * p.z);
        ti = sines < 0.0 ? S.textures[ti].a : S.textures[ti].b;
    }
    const rtTexture64& T = S.textures[ti];
    switch (T.kind) {
        case RT_TEX_SOLID: return v3(T.color[0], T.color[1], T.color[2]);
        case RT_TEX_IMAGE: {
            const Image64 im = S.images[T.a];
            if (im.width == 0 || im.height == 0) return v3(0, 1, 1);
            u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            v = 1.0 - v;
            int i = (int)(v * (real)im.height);
            int j = (int)(u * (real)im.width);
            if (i == 0) i = 1;
            if (j == 0) j = 1;
            const uint8_t* px = S.texels + im.offset + 3 * ((size_t)(i - 1) * im.width + (j - 1));
            return v3((real)px[0] / 255.0, (real)px[1] / 255.0, (real)px[2] / 255.0);
        }
        case RT_TEX_NOISE_MARBLE: {
            real arg = T.scale * p.z + 10.0 * turbulence(S.perlins[T.a], p);
            real val = 0.5 * (1.0 + det_sin(arg));
            return v3(val, val, val);
        }
        case RT_TEX_NOISE_SCALAR: {
            real val = perlin_noise_scalar(S.perlins[T.a], T.scale * p);
            return v3(1.0, 1.0, 1.0) * val;
        }
    }
    return v3(0, 0, 0);
}

// ------------------------------------------------------------------ src/material.jl:23-38, src/hittable.jl:36-83
__device__ inline V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }
__device__ inline V3 refract(V3 direction, V3 normal, real etai_over_etat) {
    real cos_theta = dot(-direction, normal);
    V3 perp = etai_over_etat * (direction + cos_theta * normal);
    V3 parallel = (-sqrt(fabs(1.0 - dot(perp, perp)))) * normal;
    return perp + parallel;
}
__device__ inline real schlick(real cos_theta, real eta) {
    real k = (1.0 - eta) / (1.0 + eta);
    real r0 = k * k;
    real x = 1.0 - cos_theta;
    real x2 = x * x;
    real x5 = (x2 * x2) * x;
    return r0 + (1.0 - r0) * x5;
}
__device__ void scatter(const Scene64& S, const Ray& ray, const Hit& h, const Ctx& c, Ray& scattered, V3& attenuation) {
    const rtMaterial64& m = S.materials[h.material];
    if (m.type == RT_LAMBERTIAN) {
        V3 dir = h.n + normalize(random_in_unit_sphere(c));
        if (almost_zero(dir)) dir = h.n;
        scattered.o = h.p; scattered.d = dir; scattered.time = ray.time;
        attenuation = texture_value(S, m.texture, h.u, h.v, h.p);
    } else if (m.type == RT_METAL) {
        V3 reflected = reflect(normalize(ray.d), h.n);
        scattered.o = h.p; scattered.d = reflected + m.fuzz * random_in_unit_sphere(c); scattered.time = ray.time;
        bool keep = dot(scattered.d, h.n) > 0.0;
        V3 a = texture_value(S, m.texture, h.u, h.v, h.p);
        attenuation = keep ? a : v3(0, 0, 0); // Bool * Vec3: false is a strong zero
    } else if (m.type == RT_DIELECTRIC) {
        real ratio = h.front ? (1.0 / m.eta) : m.eta;
        V3 unit_dir = normalize(ray.d);
        real cos_theta = fmin(dot(-unit_dir, h.n), 1.0);
        real sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        V3 dir;
        if ((ratio * sin_theta > 1.0) || (schlick(cos_theta, m.eta) > rand_scalar(c))) dir = reflect(unit_dir, h.n);
        else dir = refract(unit_dir, h.n, ratio);
        scattered.o = h.p; scattered.d = dir; scattered.time = ray.time;
        attenuation = v3(1.0, 1.0, 1.0);
    } else if (m.type == RT_ISOTROPIC) {
        scattered.o = h.p; scattered.d = random_in_unit_sphere(c); scattered.time = ray.time;
        attenuation = texture_value(S, m.texture, h.u, h.v, h.p);
    } else {
        scattered.o = v3(0, 0, 0); scattered.d = v3(0, 0, 0); scattered.time = 0.0;
        attenuation = v3(0, 0, 0);
    }
}
__device__ V3 emit(const Scene64& S, const Hit& h) {
    const rtMaterial64& m = S.materials[h.material];
    if (m.type == RT_EMISSIVE) return texture_value(S, m.texture, h.u, h.v, h.p);
    return v3(0, 0, 0);
}

// src/camera.jl:38-42
__device__ Ray get_ray(const rtCamera64& cam, real s, real t, Ctx& c) {
    c.bounce = CAMERA_BOUNCE;
    V3 u = v3(cam.u[0], cam.u[1], cam.u[2]), v = v3(cam.v[0], cam.v[1], cam.v[2]);
    V3 origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    V3 llc = v3(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    V3 rd = cam.lens_radius * random_in_unit_disk(c);
    V3 offset_origin = (u * rd.x + v * rd.y) + origin;
    real time = rand_scalar(c) * (cam.t1 - cam.t0) + cam.t0;
    Ray r;
    r.o = offset_origin;
    r.d = ((llc + s * hor) + t * ver) - offset_origin;
    r.time = time;
    return r;
}

// One path's state between bounces (src/Ray_Tracing.jl:11-29 unrolled into a loop a lane can leave and re-enter)
struct Path {
    Ray ray;
    V3 acc, color;
    int depth;
    Ctx c;
};
__device__ inline void start_path(const Frame64& F, int i, int j, int col, Path& P) { // src/Ray_Tracing.jl:83-88
    real dx = F.pattern[2 * col], dy = F.pattern[2 * col + 1];
    real s = ((real)i + dx) / (real)(F.width - 1);
    real t = ((real)j + dy) / (real)(F.height - 1);
    P.c.pixel = (uint32_t)j * (uint32_t)F.width + (uint32_t)i;
    P.c.sample = (uint32_t)col;
    P.c.seed32 = F.seed32;
    P.ray = get_ray(F.cam, s, t, P.c);
    P.acc = v3(1.0, 1.0, 1.0);
    P.color = v3(0.0, 0.0, 0.0);
    P.depth = F.max_depth;
    P.c.bounce = 0;
}
// The part of one iteration of the loop of trace() that follows hit(scene, ray); true when the path has ended
// (P.color is its radiance)
__device__ inline bool shade(const Scene64& S, const Frame64& F, Path& P, const Hit& h) {
    if (h.t == R64_INF) {
        P.color = P.color + P.acc * v3(F.bg[0], F.bg[1], F.bg[2]);
        return true;
    }
    Ray scattered;
    V3 attenuation;
    scatter(S, P.ray, h, P.c, scattered, attenuation);
    P.ray = scattered;
    P.color = P.color + P.acc * emit(S, h);
    if (almost_zero(attenuation)) return true;
    P.acc = P.acc * attenuation;
    P.depth -= 1;
    P.c.bounce += 1;
    return P.depth <= 0;
}
// one whole iteration (the parity hook's loop)
__device__ inline bool bounce(const Scene64& S, const Frame64& F, Path& P, uint32_t* stk) {
    if (P.depth <= 0) return true;
    return shade(S, F, P, hit_scene(S, P.ray, 0.001, R64_INF, P.c, stk));
}

// ------------------------------------------------------------------ kernels
// A frame is traced wavefront by wavefront over a pool of slots, a path per slot: rt64_extend_kernel finds the closest
// hit of every slot's ray (the walk: registers for a ray and the hits in flight, nothing of the path), rt64_shade_kernel
// completes the hit, scatters, and hands a slot whose path has ended the next unstarted path (one atomic per wave).
// Apart they hold two to three times as many waves per SIMD as one kernel doing both, and the walk is bound by the
// latency of its node fetches.  Slot state, plane by plane (plane p of slot s at [p * n_slots + s]): reals 0-2 origin,
// 3-5 direction, 6 time, 7-9 acc, 10-12 color, 13-15 the hit (t, bound, inner bound); ints 0 depth left (0: wants a
// path, -1: none left for it), 1 pixel, 2 pattern column, 3-4 the hit (object, inner), 5-6 the path's number.
enum { PD_O = 0, PD_D = 3, PD_TIME = 6, PD_ACC = 7, PD_COLOR = 10, PD_T = 13, PD_BOUND = 14, PD_INNER_BOUND = 15, PD_PLANES = 16 };
enum { PI_DEPTH = 0, PI_PIXEL = 1, PI_SAMPLE = 2, PI_OI = 3, PI_INNER = 4, PI_PATH_LO = 5, PI_PATH_HI = 6, PI_PLANES = 7 };
struct Pool64 {
    real* d;
    int32_t* i;
    uint32_t n_slots, pad;
};

template <bool GROUPS>
__global__ __launch_bounds__(R64_BLOCK) __attribute__((amdgpu_waves_per_eu(GROUPS ? 3 : 4, GROUPS ? 3 : 4)))
void rt64_extend_kernel(Scene64 S, Frame64 F, Pool64 pool) {
    __shared__ uint32_t stack[R64_STACK * R64_BLOCK];
    const uint32_t slot = blockIdx.x * R64_BLOCK + threadIdx.x;
    if (slot >= pool.n_slots) return;
    const size_t N = pool.n_slots;
    const int depth = pool.i[PI_DEPTH * N + slot];
    if (depth <= 0) return;
    Ray r;
    r.o = v3(pool.d[(PD_O + 0) * N + slot], pool.d[(PD_O + 1) * N + slot], pool.d[(PD_O + 2) * N + slot]);
    r.d = v3(pool.d[(PD_D + 0) * N + slot], pool.d[(PD_D + 1) * N + slot], pool.d[(PD_D + 2) * N + slot]);
    r.time = pool.d[PD_TIME * N + slot];
    Ctx c;
    c.pixel = (uint32_t)pool.i[PI_PIXEL * N + slot];
    c.sample = (uint32_t)pool.i[PI_SAMPLE * N + slot];
    c.seed32 = F.seed32;
    c.bounce = (uint32_t)(F.max_depth - depth);
    Lite l = hit_scene_lite<GROUPS>(S, r, 0.001, R64_INF, c, stack + threadIdx.x);
    pool.d[PD_T * N + slot] = l.t;
    pool.d[PD_BOUND * N + slot] = l.bound;
    pool.d[PD_INNER_BOUND * N + slot] = l.inner_bound;
    pool.i[PI_OI * N + slot] = l.oi;
    pool.i[PI_INNER * N + slot] = l.inner;
}

// radiance: 3 reals per path of the batch, path = column position in the batch * pixels + pixel (pixel = j * width + i).
// live: incremented by the number of slots that leave this launch with a path (the host stops at 0).
__global__ __launch_bounds__(R64_BLOCK) void rt64_shade_kernel(Scene64 S, Frame64 F, Pool64 pool, unsigned long long* next_path, real* radiance,
                                                               uint32_t* live) {
    const uint32_t slot = blockIdx.x * R64_BLOCK + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const size_t N = pool.n_slots;
    const bool in_pool = slot < pool.n_slots;
    int depth = in_pool ? pool.i[PI_DEPTH * N + slot] : -1;
    Path P;
    bool want = depth == 0;
    if (depth > 0) {
        P.ray.o = v3(pool.d[(PD_O + 0) * N + slot], pool.d[(PD_O + 1) * N + slot], pool.d[(PD_O + 2) * N + slot]);
        P.ray.d = v3(pool.d[(PD_D + 0) * N + slot], pool.d[(PD_D + 1) * N + slot], pool.d[(PD_D + 2) * N + slot]);
        P.ray.time = pool.d[PD_TIME * N + slot];
        P.acc = v3(pool.d[(PD_ACC + 0) * N + slot], pool.d[(PD_ACC + 1) * N + slot], pool.d[(PD_ACC + 2) * N + slot]);
        P.color = v3(pool.d[(PD_COLOR + 0) * N + slot], pool.d[(PD_COLOR + 1) * N + slot], pool.d[(PD_COLOR + 2) * N + slot]);
        P.depth = depth;
        P.c.pixel = (uint32_t)pool.i[PI_PIXEL * N + slot];
        P.c.sample = (uint32_t)pool.i[PI_SAMPLE * N + slot];
        P.c.seed32 = F.seed32;
        P.c.bounce = (uint32_t)(F.max_depth - depth);
        Lite l;
        l.t = pool.d[PD_T * N + slot];
        l.bound = pool.d[PD_BOUND * N + slot];
        l.inner_bound = pool.d[PD_INNER_BOUND * N + slot];
        l.oi = pool.i[PI_OI * N + slot];
        l.inner = pool.i[PI_INNER * N + slot];
        Hit h = complete_hit(S, P.ray, 0.001, l, P.c);
        if (shade(S, F, P, h)) {
            unsigned long long mine = ((unsigned long long)(uint32_t)pool.i[PI_PATH_HI * N + slot] << 32) | (uint32_t)pool.i[PI_PATH_LO * N + slot];
            radiance[3 * mine] = P.color.x;
            radiance[3 * mine + 1] = P.color.y;
            radiance[3 * mine + 2] = P.color.z;
            want = true;
        }
    }
    bool fresh = false;
    if (want) {
        // the lanes of the wave without a path take the next ones together: one atomic
        unsigned long long mask = __ballot(1);
        unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(next_path, (unsigned long long)__popcll(mask));
        unsigned lo = __shfl((unsigned)(base & 0xffffffffull), (int)leader);
        unsigned hi = __shfl((unsigned)(base >> 32), (int)leader);
        unsigned long long mine = (((unsigned long long)hi << 32) | lo) + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        if (mine < F.n_paths) {
            const unsigned long long pixels = (unsigned long long)F.width * (unsigned long long)F.height;
            unsigned long long pos = mine / pixels, pixel = mine % pixels;
            int col = F.first + (F.batch_first + (int)pos) * F.stride;
            start_path(F, (int)(pixel % (unsigned long long)F.width), (int)(pixel / (unsigned long long)F.width), col, P);
            pool.i[PI_PIXEL * N + slot] = (int32_t)P.c.pixel;
            pool.i[PI_SAMPLE * N + slot] = (int32_t)P.c.sample;
            pool.i[PI_PATH_LO * N + slot] = (int32_t)(uint32_t)(mine & 0xffffffffull);
            pool.i[PI_PATH_HI * N + slot] = (int32_t)(uint32_t)(mine >> 32);
            fresh = true;
        } else {
            pool.i[PI_DEPTH * N + slot] = -1;
            depth = -1;
        }
    }
    const bool alive = in_pool && (fresh || (depth > 0 && !want));
    if (alive) {
        pool.d[(PD_O + 0) * N + slot] = P.ray.o.x; pool.d[(PD_O + 1) * N + slot] = P.ray.o.y; pool.d[(PD_O + 2) * N + slot] = P.ray.o.z;
        pool.d[(PD_D + 0) * N + slot] = P.ray.d.x; pool.d[(PD_D + 1) * N + slot] = P.ray.d.y; pool.d[(PD_D + 2) * N + slot] = P.ray.d.z;
        pool.d[PD_TIME * N + slot] = P.ray.time;
        pool.d[(PD_ACC + 0) * N + slot] = P.acc.x; pool.d[(PD_ACC + 1) * N + slot] = P.acc.y; pool.d[(PD_ACC + 2) * N + slot] = P.acc.z;
        pool.d[(PD_COLOR + 0) * N + slot] = P.color.x; pool.d[(PD_COLOR + 1) * N + slot] = P.color.y; pool.d[(PD_COLOR + 2) * N + slot] = P.color.z;
        pool.i[PI_DEPTH * N + slot] = P.depth;
    }
    unsigned long long alive_mask = __ballot(alive);
    if (lane == 0 && alive_mask) atomicAdd(live, (uint32_t)__popcll(alive_mask));
}

// The slots that still hold a path, moved to the front of a smaller pool (dst, whose depth plane the host has set to
// -1): once the last path of a batch has been handed out a pool only empties, and without this every launch until the
// longest path ends runs nearly all of its waves for a few lanes each.  Runs between a shade and an extend launch, so
// the hit planes hold nothing that is still needed.  A path's arithmetic does not depend on the slot it sits in.
__global__ __launch_bounds__(R64_BLOCK) void rt64_compact_kernel(Pool64 src, Pool64 dst, uint32_t* taken) {
    const uint32_t slot = blockIdx.x * R64_BLOCK + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const size_t N = src.n_slots, M = dst.n_slots;
    const int depth = slot < src.n_slots ? src.i[PI_DEPTH * N + slot] : -1;
    const bool alive = depth > 0;
    const unsigned long long mask = __ballot(alive);
    if (!mask) return;
    const unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(taken, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (!alive) return;
    const size_t to = (size_t)base + (size_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (to >= M) return; // (the host sizes dst by a count taken when at least as many slots held a path)
    for (int p = PD_O; p < PD_T; ++p) dst.d[p * M + to] = src.d[p * N + slot];
    dst.i[PI_DEPTH * M + to] = depth;
    dst.i[PI_PIXEL * M + to] = src.i[PI_PIXEL * N + slot];
    dst.i[PI_SAMPLE * M + to] = src.i[PI_SAMPLE * N + slot];
    dst.i[PI_PATH_LO * M + to] = src.i[PI_PATH_LO * N + slot];
    dst.i[PI_PATH_HI * M + to] = src.i[PI_PATH_HI * N + slot];
}

// accum (Julia Array{RGB{Float64}}(H, W) order) += the batch's radiances, column after column: the sample loop of draw!
__global__ void rt64_sum_kernel(const real* radiance, int batch_cols, int width, int height, int first_batch, real* accum) {
    int pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= width * height) return;
    int i = pixel % width, j = pixel / width;
    size_t out = ((size_t)i * height + (size_t)(height - 1 - j)) * 3;
    real cx = 0.0, cy = 0.0, cz = 0.0;
    if (!first_batch) { cx = accum[out]; cy = accum[out + 1]; cz = accum[out + 2]; }
    size_t pixels = (size_t)width * height;
    for (int k = 0; k < batch_cols; ++k) {
        const real* r = radiance + 3 * ((size_t)k * pixels + (size_t)pixel);
        cx = cx + r[0]; cy = cy + r[1]; cz = cz + r[2];
    }
    accum[out] = cx; accum[out + 1] = cy; accum[out + 2] = cz;
}
__global__ void rt64_add_kernel(real* dst, const real* src, size_t n) { // the sum of two accumulators (rt_group_render_f64)
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = dst[k] + src[k];
}
__global__ void rt64_finalize_kernel(const real* accum, real inv_n, size_t count, real* out) { // src/Ray_Tracing.jl:90-91
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt(accum[k] * inv_n);
}

// parity hooks
__global__ __launch_bounds__(R64_BLOCK) void rt64_trace_rays_kernel(Scene64 S, const real* rays, int n, real t_min, real t_max, uint32_t seed32,
                                                                  int32_t* prim_id, int32_t* sub_id, real* t, real* normal, real* uv, uint8_t* front) {
    __shared__ uint32_t stack[R64_STACK * R64_BLOCK];
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const real* q = rays + 7 * (size_t)k;
    Ray r;
    r.o = v3(q[0], q[1], q[2]); r.d = v3(q[3], q[4], q[5]); r.time = q[6];
    Ctx c;
    c.pixel = (uint32_t)k; c.sample = 0; c.seed32 = seed32; c.bounce = 0;
    Hit h = hit_scene(S, r, t_min, t_max, c, stack + threadIdx.x);
    bool hit = !(h.t == R64_INF);
    if (prim_id) prim_id[k] = hit ? h.prim : -1;
    if (sub_id) sub_id[k] = hit ? h.sub : -1;
    if (t) t[k] = h.t;
    if (normal) { normal[3 * k] = h.n.x; normal[3 * k + 1] = h.n.y; normal[3 * k + 2] = h.n.z; }
    if (uv) { uv[2 * k] = h.u; uv[2 * k + 1] = h.v; }
    if (front) front[k] = h.front ? 1 : 0;
}
__global__ __launch_bounds__(R64_BLOCK) void rt64_trace_paths_kernel(Scene64 S, Frame64 F, const int32_t* ids, int n_paths, real* radiance) {
    __shared__ uint32_t stack[R64_STACK * R64_BLOCK];
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_paths) return;
    Path P;
    start_path(F, ids[3 * k], ids[3 * k + 1], ids[3 * k + 2], P);
    while (!bounce(S, F, P, stack + threadIdx.x)) {}
    radiance[3 * k] = P.color.x; radiance[3 * k + 1] = P.color.y; radiance[3 * k + 2] = P.color.z;
}

inline uint32_t fold_seed(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u); }

} // namespace

// ------------------------------------------------------------------ host side
struct rtScene64 {
    int device = 0;
    Scene64 dev{};
    std::vector<void*> buffers;
    real* d_pattern = nullptr; size_t pattern_cap = 0;
    real* d_radiance = nullptr; size_t radiance_cap = 0; // in reals
    real* d_accum = nullptr; size_t accum_cap = 0;
    unsigned long long* d_next = nullptr;
    real* d_pool = nullptr; size_t pool_cap = 0;      // the slots' real planes
    int32_t* d_pool_i = nullptr; size_t pool_i_cap = 0; // ... and int planes
    uint32_t* d_live = nullptr;                        // per half pool R64_SYNC_EVERY counters of slots alive after a shade launch
    uint32_t* h_live = nullptr;                        // pinned mirror
    uint32_t* d_taken = nullptr;                       // per pool R64_MAX_COMPACTIONS counters of rt64_compact_kernel
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr}; // the streams of the pools after the first
    hipEvent_t fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t read[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}}; // counters of a batch are on the host
    int n_cus = 0;
};

namespace {

template <typename T>
int upload64(rtScene64* s, const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    size_t bytes = v.size() * sizeof(T);
    HIP_TRY64(hipMalloc(&p, bytes ? bytes : 16));
    s->buffers.push_back(p);
    if (bytes) HIP_TRY64(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RT_OK;
}

struct Staged64 {
    std::vector<rtObject64> objects;
    std::vector<Leaf64> fused;
    std::vector<int32_t> top_pos;
    std::vector<rtTransform64> transforms;
    std::vector<rtMaterial64> materials;
    std::vector<rtTexture64> textures;
    std::vector<Image64> images;
    std::vector<uint8_t> texels;
    std::vector<rtPerlin64> perlins;
    std::vector<Mesh64> meshes;
    std::vector<real> vertices;
    std::vector<int32_t> indices, tri_ids, mesh_materials;
    std::vector<Node64> nodes;
    std::vector<Node32> nodes32;
    std::vector<int32_t> tree_objects;
    std::vector<Tree64> trees;
    std::vector<uint8_t> mat_uv;
};

// a plain object (what hit_basic takes)
bool is_basic(int kind) { return kind >= RT_SPHERE && kind <= RT_BOX; }

// Walks a tree from node 1 the way the kernels do and checks everything they will index; returns the largest number of
// stack entries a walk can hold (0 on error, message set).
int check_tree(const rtSceneDesc64& D, const rtBvh64& T, bool nested, int* status) {
    auto fail = [&](int st, const char* msg) { *status = rt_fail(st, "rt_scene_create_f64: %s", msg); return 0; };
    if (T.n_nodes <= 0 || T.n_objects <= 0 || !T.objects || !T.node_bounds || !T.node_left || !T.node_right || !T.node_leaf)
        return fail(RT_EINVAL, "empty or incomplete tree record");
    for (int i = 0; i < T.n_objects; ++i)
        if (T.objects[i] < 0 || T.objects[i] >= D.n_objects) return fail(RT_EINVAL, "tree object index out of range");
    std::vector<uint8_t> seen((size_t)T.n_nodes, 0);
    struct Item { int id, depth; bool in_mesh; int mesh_pos; };
    std::vector<Item> todo;
    todo.push_back(Item{1, 1, false, 0});
    int deepest = 1;
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (it.id < 1 || it.id > T.n_nodes) return fail(RT_EINVAL, "BVH node id out of range");
        if (seen[(size_t)it.id - 1]) return fail(RT_EINVAL, "BVH links form a cycle or share a node");
        seen[(size_t)it.id - 1] = 1;
        if (it.depth > deepest) deepest = it.depth;
        uint8_t leaf = T.node_leaf[it.id - 1];
        if (leaf > 1) return fail(RT_EINVAL, "reached an unused BVH node slot");
        int l = T.node_left[it.id - 1], r = T.node_right[it.id - 1];
        if (!leaf) {
            todo.push_back(Item{l, it.depth + 1, it.in_mesh, it.mesh_pos});
            todo.push_back(Item{r, it.depth + 1, it.in_mesh, it.mesh_pos});
            continue;
        }
        if (T.kind == 0) {
            if (l < 1 || r > T.n_objects || l > r) return fail(RT_EINVAL, "BVH leaf range out of bounds");
            for (int i = l; i <= r; ++i) {
                const rtObject64& o = D.objects[T.objects[i - 1]];
                if (o.kind == RT_MESH) return fail(RT_EUNSUPPORTED, "a Mesh is only supported in a type group of meshes");
                if (nested && !is_basic(o.kind)) return fail(RT_EUNSUPPORTED, "only plain objects are supported inside a nested BVHTree");
            }
        } else if (!it.in_mesh) { // a mesh-level leaf: left = the root of the mesh's triangles, right = the mesh's position
            if (r < 1 || r > T.n_objects) return fail(RT_EINVAL, "mesh position out of range");
            const rtObject64& o = D.objects[T.objects[r - 1]];
            if (o.kind != RT_MESH || o.aux < 0 || o.aux >= D.n_meshes) return fail(RT_EINVAL, "mesh tree over a non-mesh object");
            todo.push_back(Item{l, it.depth + 2, true, r});
        } else { // a triangle-level leaf: columns left..right of the mesh's index matrix
            const rtMesh64& M = D.meshes[D.objects[T.objects[it.mesh_pos - 1]].aux];
            if (l < 1 || r > M.n_triangles || l > r) return fail(RT_EINVAL, "triangle range out of bounds");
        }
    }
    return deepest + 2;
}

// Leaf64::code of object oi (the description has been checked)
int leaf_code(const rtSceneDesc64& D, int oi) {
    const rtObject64& o = D.objects[oi];
    if (o.kind == RT_TRANSFORM && is_basic(D.objects[o.child].kind)) return D.objects[o.child].kind | 16;
    if (o.kind == RT_CONSTANT_MEDIUM) {
        const rtObject64* b = &D.objects[o.child];
        int wrapped = 0;
        if (b->kind == RT_TRANSFORM) { wrapped = 16; b = &D.objects[b->child]; }
        return RT_CONSTANT_MEDIUM | wrapped | (b->kind << 5);
    }
    return o.kind;
}

int stage_scene64(const rtSceneDesc64& D, Staged64& H, int* stack_need) {
    auto fail = [&](int st, const char* msg) { return rt_fail(st, "rt_scene_create_f64: %s", msg); };
    if (D.n_objects <= 0 || D.n_top <= 0 || D.n_buckets <= 0 || !D.objects || !D.top || !D.buckets) return fail(RT_EINVAL, "scene has no objects");
    if (D.n_transforms < 0 || D.n_materials < 0 || D.n_textures < 0 || D.n_images < 0 || D.n_perlins < 0 || D.n_meshes < 0 || D.n_groups < 0)
        return fail(RT_EINVAL, "negative count");
    if ((D.n_transforms && !D.transforms) || (D.n_materials && !D.materials) || (D.n_textures && !D.textures) || (D.n_images && !D.images) ||
        (D.n_perlins && !D.perlins) || (D.n_meshes && !D.meshes) || (D.n_groups && !D.groups))
        return fail(RT_EINVAL, "scene description has a null table with a non-zero count");
    // textures, materials
    for (int i = 0; i < D.n_textures; ++i) {
        const rtTexture64& t = D.textures[i];
        switch (t.kind) {
            case RT_TEX_SOLID: break;
            case RT_TEX_CHECKER:
                if (t.a < 0 || t.a >= i || t.b < 0 || t.b >= i) return fail(RT_EINVAL, "a checker's textures must be listed before it");
                break;
            case RT_TEX_IMAGE: if (t.a < 0 || t.a >= D.n_images) return fail(RT_EINVAL, "image index out of range"); break;
            case RT_TEX_NOISE_MARBLE: case RT_TEX_NOISE_SCALAR:
                if (t.a < 0 || t.a >= D.n_perlins) return fail(RT_EINVAL, "perlin index out of range");
                break;
            default: return fail(RT_EUNSUPPORTED, "unknown texture kind");
        }
    }
    for (int i = 0; i < D.n_materials; ++i) {
        const rtMaterial64& m = D.materials[i];
        if (m.type < RT_LAMBERTIAN || m.type > RT_ISOTROPIC) return fail(RT_EUNSUPPORTED, "unknown material type");
        if (m.texture < 0 || m.texture >= D.n_textures) return fail(RT_EINVAL, "material texture index out of range");
    }
    for (int i = 0; i < D.n_perlins; ++i)
        for (int k = 0; k < 256; ++k)
            if ((D.perlins[i].perm_x[k] | D.perlins[i].perm_y[k] | D.perlins[i].perm_z[k]) & ~255)
                return fail(RT_EINVAL, "perlin permutation entry outside 0..255");
    for (int i = 0; i < D.n_images; ++i) {
        const rtImage& im = D.images[i];
        if (im.width < 0 || im.height < 0 || (im.width > 0 && im.height > 0 && !im.rgb)) return fail(RT_EINVAL, "bad image");
        Image64 d{im.width, im.height, (uint32_t)H.texels.size(), 0u};
        size_t bytes = (size_t)3 * (size_t)im.width * (size_t)im.height;
        if (H.texels.size() + bytes > 0xffffffffull) return fail(RT_EUNSUPPORTED, "more than 4 GiB of texels");
        if (bytes) H.texels.insert(H.texels.end(), im.rgb, im.rgb + bytes);
        H.images.push_back(d);
    }
    // objects: the nestings the kernels know (those of the Float32 path)
    for (int i = 0; i < D.n_objects; ++i) {
        const rtObject64& o = D.objects[i];
        auto valid = [&](int k) { return k >= 0 && k < D.n_objects; };
        bool needs_material = true;
        switch (o.kind) {
            case RT_SPHERE: case RT_MOVING_SPHERE: case RT_XY_RECT: case RT_XZ_RECT: case RT_YZ_RECT: case RT_BOX: break;
            case RT_TRANSFORM: {
                needs_material = false;
                if (o.aux < 0 || o.aux >= D.n_transforms) return fail(RT_EINVAL, "transform index out of range");
                if (!valid(o.child)) return fail(RT_EINVAL, "Transform child index out of range");
                int ck = D.objects[o.child].kind;
                if (!is_basic(ck) && ck != RT_BVH) return fail(RT_EUNSUPPORTED, "unsupported nesting of Transform / ConstantMedium");
                break;
            }
            case RT_CONSTANT_MEDIUM: {
                if (!valid(o.child)) return fail(RT_EINVAL, "ConstantMedium boundary index out of range");
                const rtObject64* b = &D.objects[o.child];
                if (b->kind == RT_TRANSFORM) {
                    if (!valid(b->child)) return fail(RT_EINVAL, "Transform child index out of range");
                    b = &D.objects[b->child];
                }
                if (b->kind != RT_SPHERE && b->kind != RT_BOX)
                    return fail(RT_EUNSUPPORTED, "ConstantMedium boundary must be a Sphere or a Box (optionally inside a Transform)");
                break;
            }
            case RT_BVH:
                needs_material = false;
                if (o.aux < 0 || o.aux >= D.n_groups) return fail(RT_EINVAL, "group index out of range");
                break;
            case RT_MESH:
                needs_material = false;
                if (o.aux < 0 || o.aux >= D.n_meshes) return fail(RT_EINVAL, "mesh index out of range");
                break;
            default: return fail(RT_EUNSUPPORTED, "unknown object kind");
        }
        if (needs_material && (o.material < 0 || o.material >= D.n_materials)) return fail(RT_EINVAL, "material index out of range");
    }
    for (int i = 0; i < D.n_meshes; ++i) {
        const rtMesh64& M = D.meshes[i];
        if (M.n_triangles < 0 || M.n_vertices < 0 || M.n_materials < 0) return fail(RT_EINVAL, "mesh with a negative count");
        if ((M.n_triangles && !M.indices) || (M.n_vertices && !M.vertices) || (M.n_materials && !M.materials))
            return fail(RT_EINVAL, "mesh has a null table with a non-zero count");
        for (int t = 0; t < M.n_triangles; ++t) {
            const int32_t* ix = M.indices + 4 * (size_t)t;
            for (int k = 0; k < 3; ++k) if (ix[k] < 0 || ix[k] >= M.n_vertices) return fail(RT_EINVAL, "vertex index out of range");
            if (ix[3] < 0 || ix[3] >= M.n_materials) return fail(RT_EINVAL, "mesh material id out of range");
        }
        for (int k = 0; k < M.n_materials; ++k)
            if (M.materials[k] < 0 || M.materials[k] >= D.n_materials) return fail(RT_EINVAL, "material index out of range");
        Mesh64 d{(int32_t)(H.vertices.size() / 8), (int32_t)(H.indices.size() / 4), (int32_t)H.mesh_materials.size(), M.n_triangles};
        H.vertices.insert(H.vertices.end(), M.vertices, M.vertices + (size_t)8 * M.n_vertices);
        H.indices.insert(H.indices.end(), M.indices, M.indices + (size_t)4 * M.n_triangles);
        for (int t = 0; t < M.n_triangles; ++t) H.tri_ids.push_back(M.tri_ids ? M.tri_ids[t] : t);
        H.mesh_materials.insert(H.mesh_materials.end(), M.materials, M.materials + M.n_materials);
        H.meshes.push_back(d);
    }
    // trees: the type groups, then the nested ones
    int need_main = 0, need_nested = 0;
    for (int k = 0; k < D.n_buckets + D.n_groups; ++k) {
        bool nested = k >= D.n_buckets;
        const rtBvh64& T = nested ? D.groups[k - D.n_buckets] : D.buckets[k];
        if (nested && T.kind != 0) return fail(RT_EUNSUPPORTED, "nested tree of meshes is not supported");
        int status = RT_OK;
        int need = check_tree(D, T, nested, &status);
        if (!need) return status;
        if (nested) need_nested = std::max(need_nested, need); else need_main = std::max(need_main, need);
        Tree64 d{(int32_t)H.nodes.size(), T.n_nodes, (int32_t)H.tree_objects.size(), T.n_objects, T.kind != 0 ? 1 : 0, 0, 0, 0};
        for (int i = 0; i < T.n_nodes; ++i) {
            Node64 n;
            memcpy(n.box, T.node_bounds + 6 * (size_t)i, sizeof(n.box));
            n.left = T.node_left[i]; n.right = T.node_right[i];
            n.leaf = T.node_leaf[i] == 1 ? 1 : 0;
            n.first = (n.leaf && T.kind == 0 && n.left >= 1 && n.left <= T.n_objects) ? T.objects[n.left - 1] : 0;
            H.nodes.push_back(n);
            Node32 q;
            for (int a = 0; a < 6; ++a) q.box[a] = (float)n.box[a]; // to nearest: what box_verdict32's bound assumes
            q.left = n.left; q.right = n.right;
            if (n.leaf && T.kind == 0) {
                if (n.left == n.right) { q.left = n.first; q.right = -1 - leaf_code(D, n.first); }
                else q.right = (int32_t)(0x80000000u | (uint32_t)n.right);
            } else if (n.leaf) { // tree over meshes: left stays, right = -1 - r for either kind of leaf
                q.right = -1 - n.right;
            }
            H.nodes32.push_back(q);
        }
        H.tree_objects.insert(H.tree_objects.end(), T.objects, T.objects + T.n_objects);
        H.trees.push_back(d);
    }
    *stack_need = need_main + need_nested;
    H.objects.assign(D.objects, D.objects + D.n_objects);
    for (int i = 0; i < D.n_objects; ++i) {
        const rtObject64& o = D.objects[i];
        Leaf64 L;
        memset(&L, 0, sizeof(L));
        L.kind = o.kind; L.material = o.material; L.xf = -1;
        memcpy(L.p, o.p, sizeof(L.p));
        if (o.kind == RT_TRANSFORM && is_basic(D.objects[o.child].kind)) {
            const rtObject64& ch = D.objects[o.child];
            L.kind = ch.kind; L.material = ch.material; L.xf = o.aux;
            memcpy(L.p, ch.p, sizeof(L.p));
            memcpy(L.inv, D.transforms[o.aux].inv_xform, sizeof(L.inv));
        } else if (o.kind == RT_CONSTANT_MEDIUM) { // the boundary's numbers beside the medium's own (hit_medium_t)
            const rtObject64* b = &D.objects[o.child];
            if (b->kind == RT_TRANSFORM) {
                L.xf = b->aux;
                memcpy(L.inv, D.transforms[b->aux].inv_xform, sizeof(L.inv));
                b = &D.objects[b->child];
            }
            memcpy(L.p, b->p, sizeof(L.p));
            L.p[11] = o.p[0];
        }
        L.code = leaf_code(D, i);
        H.fused.push_back(L);
    }
    H.top_pos.assign((size_t)D.n_objects, -1);
    for (int i = 0; i < D.n_top; ++i) {
        if (D.top[i] < 0 || D.top[i] >= D.n_objects) return fail(RT_EINVAL, "top-level object index out of range");
        H.top_pos[(size_t)D.top[i]] = i;
    }
    if (D.n_transforms) H.transforms.assign(D.transforms, D.transforms + D.n_transforms);
    if (D.n_materials) H.materials.assign(D.materials, D.materials + D.n_materials);
    if (D.n_textures) H.textures.assign(D.textures, D.textures + D.n_textures);
    if (D.n_perlins) H.perlins.assign(D.perlins, D.perlins + D.n_perlins);
    // which textures read (u, v): image textures, and checkers over them (a checker's textures precede it)
    std::vector<uint8_t> tex_uv((size_t)D.n_textures, 0);
    for (int i = 0; i < D.n_textures; ++i) {
        const rtTexture64& t = D.textures[i];
        tex_uv[(size_t)i] = t.kind == RT_TEX_IMAGE || (t.kind == RT_TEX_CHECKER && (tex_uv[(size_t)t.a] || tex_uv[(size_t)t.b]));
    }
    for (int i = 0; i < D.n_materials; ++i) H.mat_uv.push_back(tex_uv[(size_t)D.materials[i].texture]);
    return RT_OK;
}

template <typename T>
int grow64(T** buf, size_t* cap, size_t count) {
    if (*cap >= count && *buf) return RT_OK;
    if (*buf) { HIP_TRY64(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    HIP_TRY64(hipMalloc((void**)buf, (count ? count : 1) * sizeof(T)));
    *cap = count;
    return RT_OK;
}

int check_frame64(const char* who, rtScene64* s, const rtCamera64* cam, const real* pattern, int n, const real* bg, int width, int height, int max_depth) {
    if (!s || !cam || !pattern || !bg) return rt_fail(RT_EINVAL, "%s: null argument", who);
    if (n <= 0 || width <= 1 || height <= 1 || max_depth < 0) return rt_fail(RT_EINVAL, "%s: bad size (n %d, %d x %d, depth %d)", who, n, width, height, max_depth);
    if ((unsigned long long)width * (unsigned long long)height > 0x7fffffffull / 3ull) return rt_fail(RT_EINVAL, "%s: image too large", who);
    return RT_OK;
}

// the radiance buffer holds this many column batches' worth of paths at most (bytes)
const size_t R64_RADIANCE_BUDGET = (size_t)8 << 30;

// slots of the pool (156 bytes each); bounce launches between two looks at the live counter
const uint32_t R64_MAX_SLOTS = 3u << 19; // 1.5 Mi: the best of 1, 1.5, 2, 3 and 4 Mi with four pools
const int R64_SYNC_EVERY = 8;
const int R64_MAX_COMPACTIONS = 64; // per pool and batch of columns (each takes a quarter off or more: 1.5 Mi slots are down to one workgroup's after 33)
const int R64_MAX_POOLS = 4, R64_POOLS_DEFAULT = 3; // pools of slots, each on a stream of its own (three: with the caller's
                                                    // stream that leaves one of the four hardware queues of a process to others; a fifth stream made four pools lose a quarter)

int accumulate64_run(rtScene64* s, const rtCamera64* cam, const real* pattern, int n, int first, int stride, const real* bg, int width, int height,
                     int max_depth, uint64_t seed, real* d_accum, hipStream_t stream) {
    size_t pixels = (size_t)width * (size_t)height;
    int count = first < n ? (n - first + stride - 1) / stride : 0;
    if (count == 0 || max_depth <= 0) { // no column of this shard, or trace() returns zero(Vec3) at once
        HIP_TRY64(hipMemsetAsync(d_accum, 0, 3 * pixels * sizeof(real), stream));
        return RT_OK;
    }
    int st = grow64(&s->d_pattern, &s->pattern_cap, (size_t)2 * (size_t)n);
    if (st != RT_OK) return st;
    HIP_TRY64(hipMemcpyAsync(s->d_pattern, pattern, (size_t)2 * n * sizeof(real), hipMemcpyHostToDevice, stream));
    size_t per_col = 3 * pixels * sizeof(real);
    int cols_per_batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, R64_RADIANCE_BUDGET / per_col));
    if (const char* e = getenv("RT_AMD_F64_BATCH_COLUMNS")) // for tests: several batches on a small frame
        if (atoi(e) > 0) cols_per_batch = std::min(cols_per_batch, atoi(e));
    st = grow64(&s->d_radiance, &s->radiance_cap, (size_t)cols_per_batch * 3 * pixels);
    if (st != RT_OK) return st;
    if (!s->d_next) HIP_TRY64(hipMalloc((void**)&s->d_next, sizeof(unsigned long long)));
    if (!s->d_live) HIP_TRY64(hipMalloc((void**)&s->d_live, 2 * R64_MAX_POOLS * R64_SYNC_EVERY * sizeof(uint32_t)));
    if (!s->h_live) HIP_TRY64(hipHostMalloc((void**)&s->h_live, 2 * R64_MAX_POOLS * R64_SYNC_EVERY * sizeof(uint32_t)));
    if (!s->d_taken) HIP_TRY64(hipMalloc((void**)&s->d_taken, R64_MAX_POOLS * R64_MAX_COMPACTIONS * sizeof(uint32_t)));
    const bool compacting = !getenv("RT_AMD_F64_NO_COMPACT"); // for measurements
    int sixteenths = 12;
    if (const char* e = getenv("RT_AMD_F64_COMPACT_AT")) if (atoi(e) >= 1 && atoi(e) <= 16) sixteenths = atoi(e);
    uint32_t max_slots = R64_MAX_SLOTS;
    if (const char* e = getenv("RT_AMD_F64_SLOTS")) // for tests and tuning
        if (atoi(e) > 0) max_slots = (uint32_t)atoi(e);
    Frame64 F;
    F.cam = *cam;
    F.pattern = s->d_pattern;
    F.n = n; F.width = width; F.height = height; F.max_depth = max_depth;
    F.first = first; F.stride = stride;
    F.bg[0] = bg[0]; F.bg[1] = bg[1]; F.bg[2] = bg[2];
    F.seed32 = fold_seed(seed);
    // Several pools on a stream each: a launch of the extend kernel ends with the few workgroups whose rays walk longest
    // and the rest of the GPU idle; the other pools' kernels fill that time.
    int n_pools = R64_POOLS_DEFAULT;
    if (const char* e = getenv("RT_AMD_F64_POOLS"))
        if (atoi(e) >= 1 && atoi(e) <= R64_MAX_POOLS) n_pools = atoi(e);
    for (int h = 1; h < n_pools; ++h)
        if (!s->side[h]) {
            HIP_TRY64(hipStreamCreateWithFlags(&s->side[h], hipStreamNonBlocking));
            HIP_TRY64(hipEventCreateWithFlags(&s->join[h], hipEventDisableTiming));
        }
    if (!s->fork) HIP_TRY64(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
    for (int h = 0; h < n_pools; ++h)
        for (int q = 0; q < 2; ++q)
            if (!s->read[h][q]) HIP_TRY64(hipEventCreateWithFlags(&s->read[h][q], hipEventDisableTiming));
    for (int b0 = 0; b0 < count; b0 += cols_per_batch) {
        int cols = std::min(cols_per_batch, count - b0);
        F.batch_first = b0; F.batch_cols = cols;
        F.n_paths = (unsigned long long)cols * (unsigned long long)pixels;
        // a pool an eighth of the batch (small frames: the launches at the end of a batch, when few slots still hold a
        // path, cost less), between 256 Ki and max_slots
        unsigned long long want_slots = std::max<unsigned long long>(F.n_paths / 8ull, 256ull << 10);
        want_slots = std::min<unsigned long long>(want_slots, F.n_paths);
        uint32_t total_slots = (uint32_t)std::min<unsigned long long>((want_slots + R64_BLOCK - 1) / R64_BLOCK * R64_BLOCK, max_slots);
        const int pools = total_slots >= 64u * R64_BLOCK ? n_pools : 1;
        const uint32_t per = (total_slots / pools + R64_BLOCK - 1) / R64_BLOCK * R64_BLOCK;
        // every pool twice: the second copy is where a pool is compacted to (and back from) once the batch runs out of paths
        st = grow64(&s->d_pool, &s->pool_cap, (size_t)PD_PLANES * per * pools * 2);
        if (st == RT_OK) st = grow64(&s->d_pool_i, &s->pool_i_cap, (size_t)PI_PLANES * per * pools * 2);
        if (st != RT_OK) return st;
        Pool64 pool[R64_MAX_POOLS], spare[R64_MAX_POOLS];
        hipStream_t on[R64_MAX_POOLS];
        dim3 grid[R64_MAX_POOLS];
        int compactions[R64_MAX_POOLS];
        on[0] = stream;
        for (int h = 1; h < pools; ++h) on[h] = s->side[h];
        for (int h = 0; h < pools; ++h) {
            pool[h].n_slots = per;
            pool[h].pad = 0;
            pool[h].d = s->d_pool + (size_t)h * PD_PLANES * per;
            pool[h].i = s->d_pool_i + (size_t)h * PI_PLANES * per;
            spare[h] = pool[h];
            spare[h].d += (size_t)pools * PD_PLANES * per;
            spare[h].i += (size_t)pools * PI_PLANES * per;
            grid[h] = dim3((per + R64_BLOCK - 1) / R64_BLOCK);
            compactions[h] = 0;
        }
        const dim3 block(R64_BLOCK);
        HIP_TRY64(hipMemsetAsync(s->d_next, 0, sizeof(unsigned long long), stream));
        HIP_TRY64(hipMemsetAsync(s->d_taken, 0, R64_MAX_POOLS * R64_MAX_COMPACTIONS * sizeof(uint32_t), stream));
        if (pools > 1) { // the side streams start after what the caller's stream holds so far
            HIP_TRY64(hipEventRecord(s->fork, stream));
            for (int h = 1; h < pools; ++h) HIP_TRY64(hipStreamWaitEvent(s->side[h], s->fork, 0));
        }
        for (int h = 0; h < pools; ++h) {
            HIP_TRY64(hipMemsetAsync(pool[h].i, 0, (size_t)per * sizeof(int32_t), on[h])); // depth plane: every slot wants a path
            hipLaunchKernelGGL(rt64_shade_kernel, grid[h], block, 0, on[h], s->dev, F, pool[h], s->d_next, s->d_radiance,
                               s->d_live + (h * 2 + 1) * R64_SYNC_EVERY); // the first paths (its count is not read)
        }
        // every path ends within max_depth bounces of its start, and a launch that starts none ends the frame soon after:
        // the loop is bounded by the paths there are
        const unsigned long long most = (F.n_paths / per + 2ull) * (unsigned long long)(max_depth + 1) + 8ull;
        // The host looks at a pool's counters one batch of launches late: batch i + 1 is in the stream before the host
        // waits for batch i, so a stream never runs dry while the host reads (the batch launched after a pool's last
        // live path finds every slot idle and costs a few empty launches).  Counters: [pool][batch parity][launch].
        bool done[R64_MAX_POOLS];
        for (int h = 0; h < R64_MAX_POOLS; ++h) done[h] = h >= pools;
        auto all_done = [&]() { for (int h = 0; h < pools; ++h) if (!done[h]) return false; return true; };
        auto live_at = [&](uint32_t* base, int h, int parity) { return base + (h * 2 + parity) * R64_SYNC_EVERY; };
        for (unsigned long long it = 0, batch = 0; !all_done(); it += R64_SYNC_EVERY, ++batch) {
            if (it > most) return rt_fail(RT_EHIP, "rt_render_f64: the frame did not end within %llu bounces", most);
            const int parity = (int)(batch & 1ull);
            for (int h = 0; h < pools; ++h)
                if (!done[h]) HIP_TRY64(hipMemsetAsync(live_at(s->d_live, h, parity), 0, R64_SYNC_EVERY * sizeof(uint32_t), on[h]));
            for (int k = 0; k < R64_SYNC_EVERY; ++k)
                for (int h = 0; h < pools; ++h) {
                    if (done[h]) continue;
                    if (s->dev.n_groups > 0) hipLaunchKernelGGL(rt64_extend_kernel<true>, grid[h], block, 0, on[h], s->dev, F, pool[h]);
                    else hipLaunchKernelGGL(rt64_extend_kernel<false>, grid[h], block, 0, on[h], s->dev, F, pool[h]);
                    hipLaunchKernelGGL(rt64_shade_kernel, grid[h], block, 0, on[h], s->dev, F, pool[h], s->d_next, s->d_radiance,
                                       live_at(s->d_live, h, parity) + k);
                }
            HIP_TRY64(hipGetLastError());
            for (int h = 0; h < pools; ++h) {
                if (done[h]) continue;
                HIP_TRY64(hipMemcpyAsync(live_at(s->h_live, h, parity), live_at(s->d_live, h, parity), R64_SYNC_EVERY * sizeof(uint32_t),
                                         hipMemcpyDeviceToHost, on[h]));
                HIP_TRY64(hipEventRecord(s->read[h][parity], on[h]));
            }
            if (batch == 0) continue;
            for (int h = 0; h < pools; ++h) { // the batch before this one
                if (done[h]) continue;
                HIP_TRY64(hipEventSynchronize(s->read[h][parity ^ 1]));
                const uint32_t held = live_at(s->h_live, h, parity ^ 1)[R64_SYNC_EVERY - 1];
                done[h] = held == 0;
                // Fewer paths than slots after a launch: a slot asked for a path and there was none, so the batch has
                // handed out its last and the pool only empties from here on -- `held`, a batch of launches old, bounds
                // what it holds now.  At three quarters or less the paths move to the front of the pool's other copy
                // and the launches shrink to it (after the batch already in the stream).
                const uint32_t fit = (held + R64_BLOCK - 1) / R64_BLOCK * R64_BLOCK;
                if (compacting && held > 0 && (unsigned long long)held * 16ull <= (unsigned long long)pool[h].n_slots * (unsigned long long)sixteenths &&
                    fit < pool[h].n_slots && compactions[h] < R64_MAX_COMPACTIONS) {
                    Pool64 to = spare[h];
                    to.n_slots = fit;
                    HIP_TRY64(hipMemsetAsync(to.i + (size_t)PI_DEPTH * fit, 0xFF, (size_t)fit * sizeof(int32_t), on[h])); // depth -1: no path, none to come
                    hipLaunchKernelGGL(rt64_compact_kernel, grid[h], block, 0, on[h], pool[h], to, s->d_taken + h * R64_MAX_COMPACTIONS + compactions[h]);
                    HIP_TRY64(hipGetLastError());
                    spare[h].d = pool[h].d; spare[h].i = pool[h].i;
                    pool[h] = to;
                    grid[h] = dim3(fit / R64_BLOCK);
                    ++compactions[h];
                }
            }
        }
        for (int h = 1; h < pools; ++h) { // the caller's stream goes on after the side streams' last kernels
            HIP_TRY64(hipEventRecord(s->join[h], s->side[h]));
            HIP_TRY64(hipStreamWaitEvent(stream, s->join[h], 0));
        }
        hipLaunchKernelGGL(rt64_sum_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, (const real*)s->d_radiance, cols, width,
                           height, b0 == 0 ? 1 : 0, d_accum);
        HIP_TRY64(hipGetLastError());
    }
    return RT_OK;
}

// A frame that fails part-way has kernels on the pools' streams that the caller's stream no longer waits for: they are
// drained before the error is handed back, so that the caller may free or reuse the accumulator at once.
int accumulate64(rtScene64* s, const rtCamera64* cam, const real* pattern, int n, int first, int stride, const real* bg, int width, int height,
                 int max_depth, uint64_t seed, real* d_accum, hipStream_t stream) {
    int st = accumulate64_run(s, cam, pattern, n, first, stride, bg, width, height, max_depth, seed, d_accum, stream);
    if (st != RT_OK) (void)hipDeviceSynchronize();
    return st;
}

} // namespace

extern "C" void rt64_launch_add(double* d_dst, const double* d_src, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(rt64_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_dst, d_src, n);
}

extern "C" int rt_scene_create_f64(const rtSceneDesc64* desc, int device, rtScene64** out) {
    if (!desc || !out) return rt_fail(RT_EINVAL, "rt_scene_create_f64: null argument");
    *out = nullptr;
    // the description is checked before the device is looked for: a bad scene is reported as such on any host
    Staged64 H;
    int stack_need = 0;
    int st = stage_scene64(*desc, H, &stack_need);
    if (st != RT_OK) return st;
    if (stack_need > R64_STACK) return rt_fail(RT_EUNSUPPORTED, "rt_scene_create_f64: trees %d levels deep need more than the %d stack entries of a lane", stack_need, R64_STACK);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return rt_fail(RT_ENODEV, "rt_scene_create_f64: no HIP device");
    if (device < 0 || device >= count) return rt_fail(RT_EINVAL, "rt_scene_create_f64: device %d out of range (0..%d)", device, count - 1);
    HIP_TRY64(hipSetDevice(device));
    rtScene64* s = new (std::nothrow) rtScene64();
    if (!s) return rt_fail(RT_ENOMEM, "rt_scene_create_f64: out of host memory");
    s->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->n_cus = prop.multiProcessorCount;
    if (s->n_cus <= 0) s->n_cus = 256;
    Scene64& d = s->dev;
#define UP64(field, vec)                                              \
    do {                                                              \
        st = upload64(s, vec, &d.field);                              \
        if (st != RT_OK) { rt_scene_destroy_f64(s); return st; }      \
    } while (0)
    UP64(objects, H.objects); UP64(fused, H.fused); UP64(top_pos, H.top_pos); UP64(transforms, H.transforms); UP64(materials, H.materials);
    UP64(textures, H.textures); UP64(images, H.images); UP64(texels, H.texels); UP64(perlins, H.perlins);
    UP64(meshes, H.meshes); UP64(vertices, H.vertices); UP64(indices, H.indices); UP64(tri_ids, H.tri_ids);
    UP64(mesh_materials, H.mesh_materials); UP64(nodes, H.nodes); UP64(nodes32, H.nodes32); UP64(tree_objects, H.tree_objects); UP64(trees, H.trees);
    UP64(mat_uv, H.mat_uv);
#undef UP64
    d.n_buckets = desc->n_buckets;
    d.n_groups = desc->n_groups;
    d.all_uv = 0;
    d.pad = 0;
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_destroy_f64(rtScene64* s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (void* p : s->buffers) (void)hipFree(p);
    if (s->d_pattern) (void)hipFree(s->d_pattern);
    if (s->d_radiance) (void)hipFree(s->d_radiance);
    if (s->d_accum) (void)hipFree(s->d_accum);
    if (s->d_next) (void)hipFree(s->d_next);
    if (s->d_pool) (void)hipFree(s->d_pool);
    if (s->d_pool_i) (void)hipFree(s->d_pool_i);
    if (s->d_live) (void)hipFree(s->d_live);
    if (s->h_live) (void)hipHostFree(s->h_live);
    if (s->d_taken) (void)hipFree(s->d_taken);
    for (int h = 0; h < 4; ++h) {
        if (s->side[h]) (void)hipStreamDestroy(s->side[h]);
        if (s->join[h]) (void)hipEventDestroy(s->join[h]);
        for (int q = 0; q < 2; ++q)
            if (s->read[h][q]) (void)hipEventDestroy(s->read[h][q]);
    }
    if (s->fork) (void)hipEventDestroy(s->fork);
    delete s;
    return RT_OK;
}

extern "C" int rt_render_accumulate_f64(rtScene64* s, const rtCamera64* camera, const double* pattern, int n, int sample_first, int sample_stride,
                                        const double background[3], int width, int height, int max_depth, uint64_t seed, double* d_accum, void* stream) {
    int st = check_frame64("rt_render_accumulate_f64", s, camera, pattern, n, background, width, height, max_depth);
    if (st != RT_OK) return st;
    if (!d_accum || sample_first < 0 || sample_stride < 1) return rt_fail(RT_EINVAL, "rt_render_accumulate_f64: bad shard or null accumulator");
    HIP_TRY64(hipSetDevice(s->device));
    return accumulate64(s, camera, pattern, n, sample_first, sample_stride, background, width, height, max_depth, seed, d_accum, (hipStream_t)stream);
}

extern "C" int rt_finalize_f64(rtScene64* s, const double* d_accum, int n_total, int width, int height, double* d_out, void* stream) {
    if (!s || !d_accum || !d_out || n_total <= 0 || width <= 0 || height <= 0) return rt_fail(RT_EINVAL, "rt_finalize_f64: bad argument");
    HIP_TRY64(hipSetDevice(s->device));
    size_t count = (size_t)3 * (size_t)width * (size_t)height;
    real inv_n = (real)(1.0 / (double)n_total); // src/vec3.jl:31 with an Int divisor
    hipLaunchKernelGGL(rt64_finalize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_accum, inv_n, count, d_out);
    HIP_TRY64(hipGetLastError());
    return RT_OK;
}

// rt64_finalize_kernel + clamp01nan + N0f8 (round to nearest even of x * 255, in double); out is H x W x 3 bytes, row 0 = top
__global__ void __launch_bounds__(256) rt64_finalize_rgb8_kernel(const real* __restrict__ accum, real inv_n, int width, int height,
                                                                 uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * (size_t)height) return;
    int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * (size_t)width);
    const real* a = accum + ((size_t)x * (size_t)height + (size_t)y) * 3u;
    for (int c = 0; c < 3; ++c) {
        real v = __builtin_sqrt(a[c] * inv_n);
        if (!(v == v)) v = 0.0;
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        out[i * 3u + (size_t)c] = (uint8_t)__builtin_rint(v * 255.0);
    }
}

extern "C" int rt_finalize_rgb8_f64(rtScene64* s, const double* d_accum, int n_total, int width, int height, uint8_t* d_rgb8, void* stream) {
    if (!s || !d_accum || !d_rgb8 || n_total <= 0 || width <= 0 || height <= 0) return rt_fail(RT_EINVAL, "rt_finalize_rgb8_f64: bad argument");
    HIP_TRY64(hipSetDevice(s->device));
    size_t pixels = (size_t)width * (size_t)height;
    real inv_n = (real)(1.0 / (double)n_total);
    hipLaunchKernelGGL(rt64_finalize_rgb8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_accum, inv_n, width, height, d_rgb8);
    HIP_TRY64(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_render_f64(rtScene64* s, const rtCamera64* camera, const double* pattern, int n, const double background[3], int width, int height,
                             int max_depth, uint64_t seed, uint32_t flags, double* out, rtStats* stats) {
    (void)flags;
    int st = check_frame64("rt_render_f64", s, camera, pattern, n, background, width, height, max_depth);
    if (st != RT_OK) return st;
    if (!out) return rt_fail(RT_EINVAL, "rt_render_f64: null output");
    HIP_TRY64(hipSetDevice(s->device));
    size_t count = (size_t)3 * (size_t)width * (size_t)height;
    st = grow64(&s->d_accum, &s->accum_cap, count);
    if (st != RT_OK) return st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY64(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return rt_fail(RT_EHIP, "rt_render_f64: hipEventCreate failed"); }
    (void)hipEventRecord(e0, nullptr);
    st = accumulate64(s, camera, pattern, n, 0, 1, background, width, height, max_depth, seed, s->d_accum, nullptr);
    if (st == RT_OK) st = rt_finalize_f64(s, s->d_accum, n, width, height, s->d_accum, nullptr);
    (void)hipEventRecord(e1, nullptr);
    hipError_t e = st == RT_OK ? hipMemcpy(out, s->d_accum, count * sizeof(real), hipMemcpyDeviceToHost) : hipSuccess;
    float ms = 0.0f;
    if (st == RT_OK && e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (st != RT_OK) return st;
    if (e != hipSuccess) return rt_fail(RT_EHIP, "rt_render_f64: %s", hipGetErrorString(e));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->seconds = ms * 1e-3;
        stats->paths = (uint64_t)width * (uint64_t)height * (uint64_t)n;
    }
    return RT_OK;
}

extern "C" int rt_trace_rays_f64(rtScene64* s, const double* rays, int n, double t_min, double t_max, uint64_t seed, int32_t* prim_id, int32_t* sub_id,
                                 double* t, double* normal, double* uv, uint8_t* front) {
    if (!s || !rays || n < 0) return rt_fail(RT_EINVAL, "rt_trace_rays_f64: bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY64(hipSetDevice(s->device));
    real *d_rays = nullptr, *d_t = nullptr, *d_n = nullptr, *d_uv = nullptr;
    int32_t *d_prim = nullptr, *d_sub = nullptr;
    uint8_t* d_front = nullptr;
    int st = RT_OK;
    auto body = [&]() -> int {
        HIP_TRY64(hipMalloc((void**)&d_rays, (size_t)7 * n * sizeof(real)));
        HIP_TRY64(hipMalloc((void**)&d_t, (size_t)n * sizeof(real)));
        HIP_TRY64(hipMalloc((void**)&d_n, (size_t)3 * n * sizeof(real)));
        HIP_TRY64(hipMalloc((void**)&d_uv, (size_t)2 * n * sizeof(real)));
        HIP_TRY64(hipMalloc((void**)&d_prim, (size_t)n * sizeof(int32_t)));
        HIP_TRY64(hipMalloc((void**)&d_sub, (size_t)n * sizeof(int32_t)));
        HIP_TRY64(hipMalloc((void**)&d_front, (size_t)n));
        HIP_TRY64(hipMemcpy(d_rays, rays, (size_t)7 * n * sizeof(real), hipMemcpyHostToDevice));
        Scene64 with_uv = s->dev;
        with_uv.all_uv = 1;
        hipLaunchKernelGGL(rt64_trace_rays_kernel, dim3((unsigned)((n + R64_BLOCK - 1) / R64_BLOCK)), dim3(R64_BLOCK), 0, nullptr, with_uv,
                           (const real*)d_rays, n, t_min, t_max, fold_seed(seed), d_prim, d_sub, d_t, d_n, d_uv, d_front);
        HIP_TRY64(hipGetLastError());
        HIP_TRY64(hipDeviceSynchronize());
        if (prim_id) HIP_TRY64(hipMemcpy(prim_id, d_prim, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (sub_id) HIP_TRY64(hipMemcpy(sub_id, d_sub, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (t) HIP_TRY64(hipMemcpy(t, d_t, (size_t)n * sizeof(real), hipMemcpyDeviceToHost));
        if (normal) HIP_TRY64(hipMemcpy(normal, d_n, (size_t)3 * n * sizeof(real), hipMemcpyDeviceToHost));
        if (uv) HIP_TRY64(hipMemcpy(uv, d_uv, (size_t)2 * n * sizeof(real), hipMemcpyDeviceToHost));
        if (front) HIP_TRY64(hipMemcpy(front, d_front, (size_t)n, hipMemcpyDeviceToHost));
        return RT_OK;
    };
    st = body();
    (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_uv); (void)hipFree(d_prim); (void)hipFree(d_sub); (void)hipFree(d_front);
    return st;
}

extern "C" int rt_trace_paths_f64(rtScene64* s, const rtCamera64* camera, const double* pattern, int n, const double background[3], int width,
                                  int height, int max_depth, uint64_t seed, const int32_t* ids, int n_paths, double* radiance) {
    int st = check_frame64("rt_trace_paths_f64", s, camera, pattern, n, background, width, height, max_depth);
    if (st != RT_OK) return st;
    if (!ids || !radiance || n_paths < 0) return rt_fail(RT_EINVAL, "rt_trace_paths_f64: bad argument");
    if (n_paths == 0) return RT_OK;
    for (int k = 0; k < n_paths; ++k) {
        int i = ids[3 * k], j = ids[3 * k + 1], c = ids[3 * k + 2];
        if (i < 0 || i >= width || j < 0 || j >= height || c < 0 || c >= n) return rt_fail(RT_EINVAL, "rt_trace_paths_f64: path id out of range");
    }
    HIP_TRY64(hipSetDevice(s->device));
    st = grow64(&s->d_pattern, &s->pattern_cap, (size_t)2 * (size_t)n);
    if (st != RT_OK) return st;
    HIP_TRY64(hipMemcpy(s->d_pattern, pattern, (size_t)2 * n * sizeof(real), hipMemcpyHostToDevice));
    int32_t* d_ids = nullptr;
    real* d_rad = nullptr;
    auto body = [&]() -> int {
        HIP_TRY64(hipMalloc((void**)&d_ids, (size_t)3 * n_paths * sizeof(int32_t)));
        HIP_TRY64(hipMalloc((void**)&d_rad, (size_t)3 * n_paths * sizeof(real)));
        HIP_TRY64(hipMemcpy(d_ids, ids, (size_t)3 * n_paths * sizeof(int32_t), hipMemcpyHostToDevice));
        Frame64 F;
        F.cam = *camera;
        F.pattern = s->d_pattern;
        F.n = n; F.width = width; F.height = height; F.max_depth = max_depth;
        F.first = 0; F.stride = 1; F.batch_first = 0; F.batch_cols = n;
        F.bg[0] = background[0]; F.bg[1] = background[1]; F.bg[2] = background[2];
        F.seed32 = fold_seed(seed);
        F.n_paths = (unsigned long long)n_paths;
        hipLaunchKernelGGL(rt64_trace_paths_kernel, dim3((unsigned)((n_paths + R64_BLOCK - 1) / R64_BLOCK)), dim3(R64_BLOCK), 0, nullptr, s->dev, F,
                           (const int32_t*)d_ids, n_paths, d_rad);
        HIP_TRY64(hipGetLastError());
        HIP_TRY64(hipDeviceSynchronize());
        HIP_TRY64(hipMemcpy(radiance, d_rad, (size_t)3 * n_paths * sizeof(real), hipMemcpyDeviceToHost));
        return RT_OK;
    };
    st = body();
    (void)hipFree(d_ids); (void)hipFree(d_rad);
    return st;
}
