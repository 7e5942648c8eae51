// rt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the path-tracing loop of
// rafaellg79/raytracing: draw_kernel / cu_trace / cu_hit / scatter / emit
// (src/Ray_Tracing.jl:51-69,102-139, src/cuscene_manager.jl:34-52, src/cuBVH.jl:32-67, src/hittable.jl:36-83).
//
// Arithmetic contract: every primitive intersection, hit completion and scatter formula is evaluated in the
// operation order of the reference's Float32 code with IEEE add/mul/div/sqrt and NO fused multiply-add (this
// file is compiled with -ffp-contract=off), so hit distances, primitive choices and scattered rays are
// bit-identical to a plain CPU evaluation of the same formulas.  Only the box test of the device tree, which merely
// culls, uses reciprocal multiplication, and it is widened so that it never rejects a box the exact test accepts;
// what the reference's own box tests cull on top of that is taken away again, exactly (ref_reaches).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "rt_device.h"
#include "rt_kernels.h"

#define INF __builtin_inff()

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ f3 operator*(float a, f3 b) { return mk3(a * b.x, a * b.y, a * b.z); }
// src/vec3.jl:46: (x*x + y*y) + z*z, left to right
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// src/vec3.jl:57 with :31: a * F(1/length(a))
__device__ __forceinline__ f3 normalize3(f3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
__device__ __forceinline__ bool is_zero3(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
__device__ __forceinline__ f3 at3(f3 o, f3 d, float t) { return o + d * t; }

// ------------------------------------------------------------------ deterministic math (same sequences as
// the checker uses on the CPU: only IEEE basic operations, so results are reproducible bit for bit)
__device__ __forceinline__ float det_logf(float x) {
    if (x == 0.0f) return -INF;
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = __float_as_uint(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    int k = (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float m = __uint_as_float(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __noinline__ float det_sinf(float x) {
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2 = 1.57079632679489661923;
    double xd = (double)x;
    double n = floor(xd * TWO_OVER_PI + 0.5);
    float r = (float)(xd - n * PIO2);
    long long q = (long long)n;
    float z = r * r;
    float sn, cs;
    {
        const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f,
                    S4 = 2.7557314297e-06f, S5 = -2.5050759689e-08f, S6 = 1.5896910177e-10f;
        float p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        sn = r + (r * z) * (S1 + z * p);
    }
    {
        const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f,
                    C4 = -2.7557314297e-07f, C5 = 2.0875723372e-09f, C6 = -1.1359647598e-11f;
        float p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        cs = 1.0f - (0.5f * z - z * p);
    }
    switch ((int)(q & 3)) {
        case 0: return sn;
        case 1: return cs;
        case 2: return -sn;
        default: return -cs;
    }
}

__device__ __forceinline__ double det_atan01(double z) {
    z = z / (1.0 + sqrt(1.0 + z * z));
    z = z / (1.0 + sqrt(1.0 + z * z));
    double w = z * z;
    double p = 1.0 / 17.0;
    p = 1.0 / 15.0 - w * p;
    p = 1.0 / 13.0 - w * p;
    p = 1.0 / 11.0 - w * p;
    p = 1.0 / 9.0 - w * p;
    p = 1.0 / 7.0 - w * p;
    p = 1.0 / 5.0 - w * p;
    p = 1.0 / 3.0 - w * p;
    p = 1.0 - w * p;
    return 4.0 * (z * p);
}
__device__ __noinline__ double det_atan2d(double y, double x) {
    const double PI = 3.14159265358979323846;
    double ax = fabs(x), ay = fabs(y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax >= ay) a = det_atan01(ay / ax);
    else a = 0.5 * PI - det_atan01(ax / ay);
    if (__builtin_signbit(x)) a = PI - a;
    if (__builtin_signbit(y)) a = -a;
    return a;
}
// src/sphere.jl:104-108
__device__ __forceinline__ void sphere_uv(f3 p, float& u, float& v) {
    const float PI_F = 3.14159265358979323846f;
    const float TWO_PI_F = (float)(2.0 * 3.14159265358979323846);
    float cy = -p.y;
    cy = cy < -1.0f ? -1.0f : (cy > 1.0f ? 1.0f : cy);
    double cd = (double)cy;
    float theta = (float)det_atan2d(sqrt((1.0 - cd) * (1.0 + cd)), cd);
    float phi = (float)det_atan2d((double)(-p.z), (double)p.x) + PI_F;
    u = phi / TWO_PI_F;
    v = theta / PI_F;
}

// ------------------------------------------------------------------ counter-based random numbers
// pcg4d (Jarzynski & Olano, JCGT 2020) of (pixel, sample, counter, seed); counter = bounce | purpose | index.
struct u4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ u4 pcg4d(u4 v) {
    v.x = v.x * 1664525u + 1013904223u;
    v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u;
    v.w = v.w * 1664525u + 1013904223u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    return v;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

#define PURPOSE_TRIAL 0u
#define PURPOSE_SCALAR 1u
#define PURPOSE_MEDIUM 2u
#define CAMERA_BOUNCE 63u

struct RngKey {
    uint32_t pixel, sample, seed32, bounce;
};
__device__ __forceinline__ u4 rng_draw(const RngKey& k, uint32_t purpose, uint32_t idx) {
    return pcg4d(u4{k.pixel, k.sample, (k.bounce << 26) | (purpose << 24) | (idx & 0xFFFFFFu), k.seed32});
}
// src/vec3.jl:87-94
__device__ __forceinline__ f3 random_in_unit_sphere(const RngKey& k) {
    for (uint32_t trial = 0;; ++trial) {
        u4 r = rng_draw(k, PURPOSE_TRIAL, trial);
        f3 p = mk3(u01(r.x) * 2.0f + -1.0f, u01(r.y) * 2.0f + -1.0f, u01(r.z) * 2.0f + -1.0f);
        if (!(dot3(p, p) > 1.0f)) return p;
    }
}
// src/vec3.jl:96-105
__device__ __forceinline__ f3 random_in_unit_disk(const RngKey& k) {
    for (uint32_t trial = 0;; ++trial) {
        u4 r = rng_draw(k, PURPOSE_TRIAL, trial);
        f3 p = mk3(u01(r.x) * 2.0f - 1.0f, u01(r.y) * 2.0f - 1.0f, 0.0f);
        if (!(dot3(p, p) > 1.0f)) return p;
    }
}

// ------------------------------------------------------------------ transforms (src/transform.jl:76-98)
__device__ __forceinline__ f3 apply_point(const float* T, f3 v) {
    return mk3(T[0] * v.x + T[3] * v.y + T[6] * v.z + T[9], T[1] * v.x + T[4] * v.y + T[7] * v.z + T[10],
               T[2] * v.x + T[5] * v.y + T[8] * v.z + T[11]);
}
__device__ __forceinline__ f3 apply_vec(const float* T, f3 v) {
    return mk3(T[0] * v.x + T[3] * v.y + T[6] * v.z, T[1] * v.x + T[4] * v.y + T[7] * v.z,
               T[2] * v.x + T[5] * v.y + T[8] * v.z);
}

// ------------------------------------------------------------------ primitive tests: distance only
struct Geom { // the 9 geometry floats of a primitive record
    float g[9];
};
__device__ __forceinline__ void load_prim(const DevScene& S, uint32_t prim, uint4& head, Geom& G) {
    const uint4* p = S.prims + 3u * prim;
    head = p[0];
    uint4 a = p[1], b = p[2];
    G.g[0] = __uint_as_float(head.w);
    G.g[1] = __uint_as_float(a.x); G.g[2] = __uint_as_float(a.y); G.g[3] = __uint_as_float(a.z); G.g[4] = __uint_as_float(a.w);
    G.g[5] = __uint_as_float(b.x); G.g[6] = __uint_as_float(b.y); G.g[7] = __uint_as_float(b.z); G.g[8] = __uint_as_float(b.w);
}

// src/sphere.jl:45-65: the root test is done scaled by a = dot(d,d), the division comes last
__device__ __forceinline__ float sphere_t(f3 center, float radius, f3 o, f3 d, float t_min, float t_max) {
    f3 oc = o - center;
    float a = dot3(d, d);
    float half_b = dot3(oc, d);
    float c = dot3(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return INF;
    float sqrtd = sqrtf(disc);
    float root = (-half_b - sqrtd);
    if (root < t_min * a || t_max * a < root) {
        root = (-half_b + sqrtd);
        if (root < t_min * a || t_max * a < root) return INF;
    }
    return root / a;
}
// src/aarect.jl:49-95; (ok,dk) constant axis, (oa,da),(ob,db) the two in-plane axes
__device__ __forceinline__ float rect_t(float ok, float dk, float oa, float da, float ob, float db, float a0, float a1,
                                        float b0, float b1, float k, float t_min, float t_max) {
    float t = (k - ok) / dk;
    if (t < t_min || t_max < t) return INF;
    float a = oa + t * da;
    float b = ob + t * db;
    if (a < a0 || a1 < a || b < b0 || b1 < b) return INF;
    return t;
}
// src/box.jl:9-24,43-83: six rectangles in the order front, back, top, bottom, right, left; strict <
__device__ __forceinline__ float box_t(const float* g, f3 o, f3 d, float t_min, float t_max, uint32_t& side) {
    float best = INF;
    float t;
    t = rect_t(o.z, d.z, o.x, d.x, o.y, d.y, g[0], g[3], g[1], g[4], g[5], t_min, t_max);
    if (t < best) { best = t; side = 0; t_max = t; }
    t = rect_t(o.z, d.z, o.x, d.x, o.y, d.y, g[0], g[3], g[1], g[4], g[2], t_min, t_max);
    if (t < best) { best = t; side = 1; t_max = t; }
    t = rect_t(o.y, d.y, o.x, d.x, o.z, d.z, g[0], g[3], g[2], g[5], g[4], t_min, t_max);
    if (t < best) { best = t; side = 2; t_max = t; }
    t = rect_t(o.y, d.y, o.x, d.x, o.z, d.z, g[0], g[3], g[2], g[5], g[1], t_min, t_max);
    if (t < best) { best = t; side = 3; t_max = t; }
    t = rect_t(o.x, d.x, o.y, d.y, o.z, d.z, g[1], g[4], g[2], g[5], g[3], t_min, t_max);
    if (t < best) { best = t; side = 4; t_max = t; }
    t = rect_t(o.x, d.x, o.y, d.y, o.z, d.z, g[1], g[4], g[2], g[5], g[0], t_min, t_max);
    if (t < best) { best = t; side = 5; t_max = t; }
    return best;
}
// src/triangle.jl:73-91
__device__ __forceinline__ float triangle_t(const float* g, f3 o, f3 d, float t_min, float t_max, f3& normal,
                                            float& u, float& v) {
    f3 v1 = mk3(g[0], g[1], g[2]);
    f3 e1 = mk3(g[3], g[4], g[5]) - v1;
    f3 e2 = mk3(g[6], g[7], g[8]) - v1;
    f3 origin_v1 = v1 - o;
    normal = cross3(e1, e2);
    f3 s = cross3(origin_v1, d);
    float c = 1.0f / dot3(normal, d);
    u = dot3(s, e2) * c;
    v = dot3(s, e1) * -c;
    float t = dot3(normal, origin_v1) * c;
    if (u < 0.0f || v < 0.0f || u + v > 1.0f || t < t_min || t_max < t) return INF;
    return t;
}

// Distance to a sphere, rectangle or box for a ray that is already in the primitive's frame (ro, rd).  One code path
// per class so that a wave holding a mix of primitives does not run one copy of the arithmetic per kind: spheres and
// moving spheres share sphere_t, the three rectangle kinds and the six sides of a box share rect_t (ray components and
// bounds are selected per lane).  CL: the classes of primitives the scene holds (RT_CL_*); the code of the others is left out.
template <uint32_t CL = RT_CL_ALL>
__device__ __forceinline__ float framed_t(uint32_t kind, const float* g, f3 ro, f3 rd, float time, float t_min, float t_max, uint32_t& side) {
    if ((CL & RT_CL_SPHERE) && (!(CL & (RT_CL_RECT | RT_CL_BOX)) || kind <= DP_MOVING_SPHERE)) {
        f3 center = mk3(g[0], g[1], g[2]);
        float radius = g[3];
        if ((CL & RT_CL_MOVING) && kind == DP_MOVING_SPHERE) { // src/sphere.jl:24-26
            f3 c1 = mk3(g[4], g[5], g[6]);
            center = center + ((time - g[3]) / (g[7] - g[3])) * (c1 - center);
            radius = g[8];
        }
        return sphere_t(center, radius, ro, rd, t_min, t_max);
    }
    if (!(CL & (RT_CL_RECT | RT_CL_BOX))) return INF;
    // rectangles (src/aarect.jl:49-95) and boxes (src/box.jl:9-24,43-83: front, back, top, bottom, right, left; strict <)
    const bool isbox = (CL & RT_CL_BOX) && (!(CL & RT_CL_RECT) || kind == DP_BOX);
    const uint32_t axis = isbox ? 2u : (kind == DP_XY_RECT ? 2u : (kind == DP_XZ_RECT ? 1u : 0u)); // constant axis
    float ok = axis == 2u ? ro.z : (axis == 1u ? ro.y : ro.x), dk = axis == 2u ? rd.z : (axis == 1u ? rd.y : rd.x);
    float oa = axis == 0u ? ro.y : ro.x, da = axis == 0u ? rd.y : rd.x;
    float ob = axis == 2u ? ro.y : ro.z, db = axis == 2u ? rd.y : rd.z;
    float best = rect_t(ok, dk, oa, da, ob, db, g[0], isbox ? g[3] : g[1], isbox ? g[1] : g[2], isbox ? g[4] : g[3],
                        isbox ? g[5] : g[4], t_min, t_max);
    if (isbox) {
        if (best < INF) t_max = best; else best = INF;
        float t;
        t = rect_t(ro.z, rd.z, ro.x, rd.x, ro.y, rd.y, g[0], g[3], g[1], g[4], g[2], t_min, t_max);
        if (t < best) { best = t; side = 1; t_max = t; }
        t = rect_t(ro.y, rd.y, ro.x, rd.x, ro.z, rd.z, g[0], g[3], g[2], g[5], g[4], t_min, t_max);
        if (t < best) { best = t; side = 2; t_max = t; }
        t = rect_t(ro.y, rd.y, ro.x, rd.x, ro.z, rd.z, g[0], g[3], g[2], g[5], g[1], t_min, t_max);
        if (t < best) { best = t; side = 3; t_max = t; }
        t = rect_t(ro.x, rd.x, ro.y, rd.y, ro.z, rd.z, g[1], g[4], g[2], g[5], g[3], t_min, t_max);
        if (t < best) { best = t; side = 4; t_max = t; }
        t = rect_t(ro.x, rd.x, ro.y, rd.y, ro.z, rd.z, g[1], g[4], g[2], g[5], g[0], t_min, t_max);
        if (t < best) { best = t; side = 5; t_max = t; }
    }
    return best;
}

// Distance to a surface primitive.  Tinv: the inverse matrix of the Transform around the primitive (used when field != 0)
template <uint32_t CL = RT_CL_ALL>
__device__ __forceinline__ float surface_t(const float* Tinv, uint32_t kind, uint32_t field, const float* g, f3 o, f3 d,
                                           float time, float t_min, float t_max, uint32_t& side, uint32_t& xf) {
    side = 0;
    if ((CL & RT_CL_TRI) && (CL == RT_CL_TRI || kind == DP_TRIANGLE)) {
        f3 nrm;
        float u, v;
        return triangle_t(g, o, d, t_min, t_max, nrm, u, v);
    }
    f3 ro = o, rd = d;
    if ((CL & RT_CL_XFORM) && field) { // Transform: src/transform.jl:104-105
        ro = apply_point(Tinv, o);
        rd = apply_vec(Tinv, d);
        xf = field;
    }
    return framed_t<CL>(kind, g, ro, rd, time, t_min, t_max, side);
}

// ------------------------------------------------------------------ closest hit over the whole scene
struct Closest {
    float t;
    uint32_t prim; // primitive index
    uint32_t sx;   // box side (bits 0-2) | (transform index + 1) << 3 | material type << 29
    uint32_t rank; // tie-break: higher wins among equal t
};
struct Counters {
    uint32_t rays, nodes, prims;
    uint32_t node_waves, prim_waves, bounce_waves; // wave-level iterations, counted by the first active lane
};
__device__ __forceinline__ uint32_t first_active_lane() {
    unsigned long long m = __ballot(1);
    return (uint32_t)(__ffsll((long long)m) - 1) == (threadIdx.x & 63u) ? 1u : 0u;
}

// The ray as the box test wants it: origin times the power of two the node boxes were scaled by, and 1/d divided by
// it (scaling by a power of two is exact, so the distances come out as with unscaled boxes), used for both planes of an axis at once
// (a node stores the two planes of an axis in one word).  The reciprocal is the
// hardware approximation (1 ulp): the box test only culls, and its far distance is widened by more than that.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
struct BoxRay {
    f3 i;      // 1/d divided by the node scale
    f3 cn, cf; // -(origin * node scale) * i, lowered / raised by the rounding slack of that product (near / far planes)
};
__device__ __forceinline__ void box_frame(const DevScene& S, f3 o, f3 d, BoxRay& R) {
    R.i = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)) * S.inv_node_scale;
    f3 oi = (o * S.node_scale) * R.i;
    // plane distance = plane * i - oi in one fused operation; the rounding of oi (half an ulp of it, however small the
    // distance) is taken out of the near distances and added to the far ones: 2^-22 |oi| each
    f3 slack = mk3(fabsf(oi.x), fabsf(oi.y), fabsf(oi.z)) * 2.384185791015625e-07f;
    R.cn = mk3(-oi.x - slack.x, -oi.y - slack.y, -oi.z - slack.z);
    R.cf = mk3(-oi.x + slack.x, -oi.y + slack.y, -oi.z + slack.z);
}

// The reference's own box test (src/bounding_box.jl:20-67) for the one box per primitive whose outcome is not implied
// by the device tree: the box of the leaf that holds the primitive.  The device tree only ever culls less than the
// reference's trees; this test takes away what the reference culls on top of that -- a thin box (a rectangle's box is
// 0.0002 thick) seen from far away, whose two plane distances round to the same float, fails `t_max <= t_min`, and so
// does a ray that grazes a box within rounding of an edge.  The test passes iff min(t_max, far distances) >
// max(t_min, near distances), NaN distances ignored, every distance computed as (plane - origin) / direction.
//
// Fast path, decisive for all but knife-edge rays: the distances from reciprocals (each within 2^-22 of the
// reference's, relative), moved inward by more than that, must leave every far distance above every near distance of
// the OTHER axes (and above t_min); the two distances of one axis -- a few ulp apart when the box is thin -- are told
// apart exactly instead: their numerators (plane - origin), which are the reference's own, differ by three ulp or more,
// and then the quotients by the same direction component cannot round to one float.  Anything else (including every
// non-finite or denormal distance, which fails these comparisons) goes through the reference's sequence itself.
// MISSES: the caller's boxes are often missed outright (the candidates of a walk, as opposed to the box of a hit that
// was found): an interval that is empty by more than the error of the fast distances is turned down on the fast path too.
// near_up (when the test passes): a distance not below the one at which the reference finds the ray entering the box
// (its t_min after the three axes), exact where the reference's own sequence ran.
template <bool MISSES>
__device__ __forceinline__ bool ref_box_pass(float4 b0, float4 b1, f3 o, f3 d, float t_min, float t_max, float& near_up) {
    const f3 mn = mk3(b0.x, b0.y, b0.z), mx = mk3(b0.w, b1.x, b1.y);
    near_up = t_min;
    if (MISSES && t_min == 0.001f) {
        // A ray that starts inside the box, further than g from every plane (the ray of nearly every path inside the
        // box of a scene-wide medium): per axis one plane lies behind the origin and one ahead, |plane - origin| > g as
        // the reference computes it, so every near distance is negative and every far distance is above
        // g / |direction component| >= 0.0011: the test comes down to t_max > t_min.
        const float g = 0.0011f * fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
        if (o.x - mn.x > g && mx.x - o.x > g && o.y - mn.y > g && mx.y - o.y > g && o.z - mn.z > g && mx.z - o.z > g && g < INF)
            return t_max > t_min;
    }
    {
        const float C = 4e-7f, SEP = 3.6e-7f;
        const v2f nx = (v2f){mn.x, mx.x} - (v2f){o.x, o.x}, ny = (v2f){mn.y, mx.y} - (v2f){o.y, o.y}, nz = (v2f){mn.z, mx.z} - (v2f){o.z, o.z};
        const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
        const v2f qx = nx * (v2f){ix, ix}, qy = ny * (v2f){iy, iy}, qz = nz * (v2f){iz, iz};
        float lx = fminf(qx.x, qx.y), hx = fmaxf(qx.x, qx.y), ly = fminf(qy.x, qy.y), hy = fmaxf(qy.x, qy.y);
        float lz = fminf(qz.x, qz.y), hz = fmaxf(qz.x, qz.y);
        if (MISSES) {
            const float T0 = fmaxf(fmaxf(lx, ly), fmaxf(lz, t_min)), T1 = fminf(fminf(hx, hy), fminf(hz, t_max));
            if (T1 - T0 < -C * (fabsf(T1) + fabsf(T0)) && t_min > 1e-30f) return false;
        }
        lx = __builtin_fmaf(C, fabsf(lx), lx); ly = __builtin_fmaf(C, fabsf(ly), ly); lz = __builtin_fmaf(C, fabsf(lz), lz);
        hx = __builtin_fmaf(-C, fabsf(hx), hx); hy = __builtin_fmaf(-C, fabsf(hy), hy); hz = __builtin_fmaf(-C, fabsf(hz), hz);
        const bool cross = (hx > fmaxf(fmaxf(ly, lz), t_min)) && (hy > fmaxf(fmaxf(lx, lz), t_min)) && (hz > fmaxf(fmaxf(lx, ly), t_min)) &&
                           (t_max > fmaxf(fmaxf(lx, ly), fmaxf(lz, t_min)));
        const bool apart = (__builtin_fmaf(-SEP, fmaxf(fabsf(nx.x), fabsf(nx.y)), nx.y - nx.x) > 0.0f) &&
                           (__builtin_fmaf(-SEP, fmaxf(fabsf(ny.x), fabsf(ny.y)), ny.y - ny.x) > 0.0f) &&
                           (__builtin_fmaf(-SEP, fmaxf(fabsf(nz.x), fabsf(nz.y)), nz.y - nz.x) > 0.0f);
        if (cross && apart && t_min > 1e-30f) { near_up = fmaxf(fmaxf(lx, ly), fmaxf(lz, t_min)); return true; }
    }
    float lo, hi;
    lo = (mn.x - o.x) / d.x; hi = (mx.x - o.x) / d.x;
    if (lo > hi) { float w = lo; lo = hi; hi = w; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    if (t_max <= t_min) return false;
    lo = (mn.y - o.y) / d.y; hi = (mx.y - o.y) / d.y;
    if (lo > hi) { float w = lo; lo = hi; hi = w; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    if (t_max <= t_min) return false;
    lo = (mn.z - o.z) / d.z; hi = (mx.z - o.z) / d.z;
    if (lo > hi) { float w = lo; lo = hi; hi = w; }
    if (hi < t_max) t_max = hi;
    if (lo > t_min) t_min = lo;
    near_up = t_min;
    return !(t_max <= t_min);
}
template <bool MISSES>
__device__ __forceinline__ bool ref_box_pass(float4 b0, float4 b1, f3 o, f3 d, float t_min, float t_max) {
    float unused;
    return ref_box_pass<MISSES>(b0, b1, o, d, t_min, t_max, unused);
}
// Does the reference reach primitive `prim` with the ray (o, d) as cast?  xf: index + 1 of the Transform around the
// nested tree the primitive belongs to, if any.  t_max: the bound the reference has at that moment -- the one the ray
// was cast with when the question is only whether the box is passed at all (hit_stands says when that is enough), the
// closest distance so far of the primitive's tree when the reference's sequence is played (front_sequence).
// b0, b1: the first two words of the primitive's record in prim_box (loaded by the caller, as early as it can)
// near_up: upper bound of the distance at which the ray enters the innermost of those boxes.  skip_outer: the outer box
// is known to have been passed (the walk is already inside that nested tree).
template <bool MISSES>
__device__ __forceinline__ bool ref_reaches(const DevScene& S, uint32_t prim, float4 b0, float4 b1, uint32_t xf, f3 o, f3 d,
                                            float t_min, float t_max, float& near_up, bool skip_outer = false) {
    const float4* b = S.prim_box + 4u * prim;
    const uint32_t flags = __float_as_uint(b1.z);
    near_up = t_min;
    if ((flags & RT_BOX_OUTER) && !(skip_outer && (flags & RT_BOX_INNER)) && !ref_box_pass<MISSES>(b0, b1, o, d, t_min, t_max, near_up)) return false;
    if (flags & RT_BOX_INNER) {
        if (xf) { // src/transform.jl:104-105
            const float* T = S.xforms + 24u * (xf - 1u) + 12;
            f3 no = apply_point(T, o), nd = apply_vec(T, d);
            o = no; d = nd;
        }
        float inner;
        if (!ref_box_pass<MISSES>(b[2], b[3], o, d, t_min, t_max, inner)) return false;
        near_up = fmaxf(near_up, inner);
    }
    return true;
}
template <bool MISSES>
__device__ __forceinline__ bool ref_reaches(const DevScene& S, uint32_t prim, float4 b0, float4 b1, uint32_t xf, f3 o, f3 d,
                                            float t_min, float t_max) {
    float unused;
    return ref_reaches<MISSES>(S, prim, b0, b1, xf, o, d, t_min, t_max, unused);
}
// t_min as the reference's box test leaves it after the three axes (the distance at which the ray enters the box, or
// the t_min it came with), for a box the test passes
__device__ __forceinline__ float ref_box_near(float4 b0, float4 b1, f3 o, f3 d, float t_min) {
    float lo, hi;
    lo = (b0.x - o.x) / d.x; hi = (b0.w - o.x) / d.x;
    if (lo > hi) lo = hi;
    if (lo > t_min) t_min = lo;
    lo = (b0.y - o.y) / d.y; hi = (b1.x - o.y) / d.y;
    if (lo > hi) lo = hi;
    if (lo > t_min) t_min = lo;
    lo = (b0.z - o.z) / d.z; hi = (b1.y - o.z) / d.z;
    if (lo > hi) lo = hi;
    if (lo > t_min) t_min = lo;
    return t_min;
}
// Does the hit `t` of primitive `prim` stand as the reference's closest hit, given that it is the closest of all the
// hits the primitives report?  It does if the reference reaches the primitive and the hit does not lie in front of the
// primitive's own box: then whatever the reference met before lies further away and cannot have culled that box, and
// nothing met later is closer.  A sphere's hit can lie in front of its box (the sphere test reports points up to
// sqrt(1.4e-6) |o - c| off the sphere); the other shapes report points of their box up to the last bit.
__device__ __forceinline__ float ref_entry(const DevScene& S, uint32_t prim, uint32_t xf, f3 o, f3 d, float t_min) {
    // the distance at which the ray enters the primitive's box(es), as the reference computes it (src/bounding_box.jl:20-67)
    const float4* b = S.prim_box + 4u * prim;
    const uint32_t flags = __float_as_uint(b[1].z);
    float near = t_min;
    if (flags & RT_BOX_OUTER) near = ref_box_near(b[0], b[1], o, d, near);
    if (flags & RT_BOX_INNER) {
        if (xf) {
            const float* T = S.xforms + 24u * (xf - 1u) + 12;
            f3 no = apply_point(T, o), nd = apply_vec(T, d);
            o = no; d = nd;
        }
        near = fmaxf(near, ref_box_near(b[2], b[3], o, d, t_min));
    }
    return near;
}
__device__ __forceinline__ bool hit_stands(const DevScene& S, uint32_t prim, uint32_t xf, f3 o, f3 d, float t_min, float t_max, float t) {
    const float4* b = S.prim_box + 4u * prim;
    const uint32_t kind = S.prims[3u * prim].x & 15u;
    float near_up;
    if (!ref_reaches<false>(S, prim, b[0], b[1], xf, o, d, t_min, t_max, near_up)) return false;
    if (!(kind <= DP_MOVING_SPHERE && t < near_up)) return true;
    // in front of the upper bound: now the entry distances as the reference computes them
    return !(t < ref_entry(S, prim, xf, o, d, t_min));
}
template <bool MISSES>
__device__ __forceinline__ bool ref_reaches(const DevScene& S, uint32_t prim, uint32_t xf, f3 o, f3 d, float t_min, float t_max) {
    const float4* b = S.prim_box + 4u * prim;
    return ref_reaches<MISSES>(S, prim, b[0], b[1], xf, o, d, t_min, t_max);
}

// Traversal of the flattened SceneManager.  Equivalent to hit(::SceneManager, ray, t_min, t_max)
// (src/scene_manager.jl:40-53) over per-type trees traversed as in src/BVH.jl:22-53 / src/mesh.jl:115-163:
// the closest hit wins; among exactly equal distances the reference keeps the earlier type group and, inside
// a group, the leaf its right-child-first stack order reaches first -- encoded here as `rank`.
//
// Surfaces and constant media are walked one after the other: first the tree of surfaces (the Walker below, or the
// state machine of rt_extend_kernel), then the list of media (media_closest), so that the (expensive, stochastic)
// medium test runs with the closest surface already known.
//
// The surface walk is a per-lane state machine with two blocks, walker_step (fetch one node, box test) and
// walker_leaf (test the pending leaf primitive).  Kernels run these blocks under a wave-wide vote: at every
// iteration the 64 lanes of the wave execute the one block that most lanes are waiting for, instead of each lane
// dragging the whole wave through its own control flow.
struct Walker {
    f3 o, d;                   // ray in the frame being traversed
    BoxRay R;                  // ... in the form the box test wants
    float time, t_min, t_max;  // t_max shrinks to the closest hit found so far
    f3 so, sd;                 // outer ray while inside a nested tree (a BVHTree inside a Transform)
    uint32_t ret, ret_end, inst_rank, inst_xf; // ret: node to continue at after a nested tree, RT_NODE_NONE outside one
    uint32_t node, end;        // next node to visit; the walk of the current tree is over at node == end
    uint32_t leaf;             // pending leaf word: (primitive index + 1) | flags, 0 = none
    Closest best;
    float t_hi;                // t_max the ray was cast with
    uint32_t set;              // node_set() of the ray
};

// first node of the layout that orders children near side first for direction d (one layout per octant)
__device__ __forceinline__ uint32_t layout_base(const DevScene& S, f3 d) {
    // by sign bit, so that -0.0 (reciprocal -Inf) takes the layout whose planes are ordered for a negative component
    uint32_t octant = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    return octant * S.n_nodes;
}
// The node set for a ray that starts at o (as cast: outside any nested tree): 0 for the set of tight boxes, which only
// rays starting inside the scene's ball may walk, else the offset of the set of the reference's boxes.  A NaN origin
// takes the second set.
__device__ __forceinline__ uint32_t node_set(const DevScene& S, f3 o) {
    const float x = o.x - S.centre[0], y = o.y - S.centre[1], z = o.z - S.centre[2];
    return (x * x + y * y + z * z < S.radius2) ? 0u : S.far_set;
}
__device__ __forceinline__ void walker_start(Walker& W, const DevScene& S, f3 o, f3 d, float time, float t_min, float t_max) {
    W.o = o; W.d = d;
    box_frame(S, o, d, W.R);
    W.time = time; W.t_min = t_min; W.t_max = t_max;
    W.so = o; W.sd = d;
    W.ret = RT_NODE_NONE; W.ret_end = 0; W.inst_rank = 0; W.inst_xf = 0;
    W.set = node_set(S, o);
    W.node = S.surface_root + layout_base(S, d) + W.set;
    W.end = W.node + S.surface_size;
    W.leaf = 0;
    W.best.t = INF; W.best.prim = 0xFFFFFFFFu; W.best.sx = 0; W.best.rank = 0;
    W.t_hi = t_max;
}
__device__ __forceinline__ bool walker_finished(const Walker& W) { return W.leaf == 0u && W.node == W.end && W.ret == RT_NODE_NONE; }
__device__ __forceinline__ bool walker_wants_leaf(const Walker& W) { return W.leaf != 0u; }
__device__ __forceinline__ bool walker_wants_step(const Walker& W) { return W.leaf == 0u && !walker_finished(W); }

// (floor_t, floor_rank): a hit the reference does not reach (ref_reaches) and everything that ranks before it are
// left out when the walk is done again for the next closest hit; (-INF, 0) leaves out nothing.
__device__ __forceinline__ void consider(Closest& best, float& t_max, float t, uint32_t prim, uint32_t sx, uint32_t rank) {
    if (t < best.t || (t == best.t && t != INF && rank > best.rank)) {
        best.t = t;
        best.prim = prim;
        best.sx = sx;
        best.rank = rank;
        t_max = t;
    }
}
// The same for a walk that keeps only `held` = primitive | box side << 28 of its closest hit (all ones: none) beside the
// bound, which is that hit's distance: transform index, material type and rank are the primitive's own (no nested trees)
// and are read from its record when the hit is published; a tie in distance -- rare -- reads the held primitive's rank.
__device__ __forceinline__ void consider_packed(const DevScene& S, uint32_t& held, float& t_max, float t, uint32_t prim, uint32_t side,
                                                uint32_t rank) {
    bool take = t < t_max;
    if (t == t_max && t != INF && held != 0xFFFFFFFFu) take = rank > S.prims[3u * (held & RT_LEAF_PRIM_MASK)].z;
    if (take) {
        t_max = t;
        held = prim | (side << 28);
    }
}
__device__ __forceinline__ void consider(Closest& best, float& t_max, float t, uint32_t prim, uint32_t sx, uint32_t rank,
                                         float floor_t, uint32_t floor_rank) {
    if (t > floor_t || (t == floor_t && rank < floor_rank)) consider(best, t_max, t, prim, sx, rank);
}

// The two boundary probes of src/constant_medium.jl:30-35, hit(boundary, r, -Inf, Inf).t and
// hit(boundary, r, t1 + 0.0001, Inf).t, from ONE evaluation of the boundary: every candidate distance is computed
// with the same arithmetic as the shape's own hit function, then the two acceptance windows are applied.
__device__ __forceinline__ void sphere_probe(f3 center, float radius, f3 o, f3 d, float& t1, float& t2) {
    // src/sphere.jl:45-65 with (t_min, t_max) = (-Inf, Inf) and then (t1 + 0.0001, Inf)
    t1 = INF; t2 = INF;
    f3 oc = o - center;
    float a = dot3(d, d);
    float half_b = dot3(oc, d);
    float c = dot3(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return;
    float sqrtd = sqrtf(disc);
    float near_root = (-half_b - sqrtd), far_root = (-half_b + sqrtd);
    // first probe: the window is (-Inf * a, Inf * a); for a > 0 the near root always passes
    float lo = -INF * a, hi = INF * a;
    float root = near_root;
    if (root < lo || hi < root) {
        root = far_root;
        if (root < lo || hi < root) return;
    }
    t1 = root / a;
    // second probe
    float lo2 = (t1 + 0.0001f) * a;
    root = near_root;
    if (root < lo2 || hi < root) {
        root = far_root;
        if (root < lo2 || hi < root) return;
    }
    t2 = root / a;
}
__device__ __forceinline__ void box_probe(const float* g, f3 o, f3 d, float& t1, float& t2) {
    // src/box.jl:43-83 over the six sides (src/aarect.jl:49-95) for the two windows; the minimum over the
    // accepted sides does not depend on the order in which box.jl visits them
    float ts[6];
    ts[0] = rect_t(o.z, d.z, o.x, d.x, o.y, d.y, g[0], g[3], g[1], g[4], g[5], -INF, INF);
    ts[1] = rect_t(o.z, d.z, o.x, d.x, o.y, d.y, g[0], g[3], g[1], g[4], g[2], -INF, INF);
    ts[2] = rect_t(o.y, d.y, o.x, d.x, o.z, d.z, g[0], g[3], g[2], g[5], g[4], -INF, INF);
    ts[3] = rect_t(o.y, d.y, o.x, d.x, o.z, d.z, g[0], g[3], g[2], g[5], g[1], -INF, INF);
    ts[4] = rect_t(o.x, d.x, o.y, d.y, o.z, d.z, g[1], g[4], g[2], g[5], g[3], -INF, INF);
    ts[5] = rect_t(o.x, d.x, o.y, d.y, o.z, d.z, g[1], g[4], g[2], g[5], g[0], -INF, INF);
    t1 = INF;
#pragma unroll
    for (int q = 0; q < 6; ++q) if (ts[q] < t1) t1 = ts[q];
    t2 = INF;
    if (t1 == INF) return;
    float lo2 = t1 + 0.0001f;
#pragma unroll
    for (int q = 0; q < 6; ++q) if (!(ts[q] < lo2) && ts[q] < t2) t2 = ts[q];
}

// src/constant_medium.jl:29-65 for one medium primitive
__device__ __forceinline__ float medium_t(const DevScene& S, uint32_t head, const float* g, f3 o, f3 d, float time,
                                          float t_min, float t_max, const RngKey& key) {
    uint32_t kind = head & 15u, field = head >> 8;
    f3 ro = o, rd = d;
    if (field) {
        const float* T = S.xforms + 24u * (field - 1u) + 12;
        ro = apply_point(T, o);
        rd = apply_vec(T, d);
    }
    float t1, t2;
    if (kind == DP_SPHERE) sphere_probe(mk3(g[0], g[1], g[2]), g[3], ro, rd, t1, t2);
    else box_probe(g, ro, rd, t1, t2);
    if (t1 == INF) return INF;
    if (t2 == INF) return INF;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return INF;
    if (t1 < 0.0f) t1 = 0.0f;
    float ray_length = sqrtf(dot3(d, d));
    float distance_inside = (t2 - t1) * ray_length;
    u4 r = rng_draw(key, PURPOSE_MEDIUM, __float_as_uint(g[7]));
    float hit_distance = g[6] * det_logf(u01(r.x));
    if (hit_distance > distance_inside) return INF;
    return t1 + hit_distance / ray_length;
}

// One node: fetch + box test.  `hit`: the ray passes the node's box inside (t_min, t_max); `link`: the node's link word
// (leaf word with RT_NODE_LEAF set, or the skip link).  The walk continues at node + 1 after a leaf or a hit interior
// node and at `link` after a missed interior node; it is over when it reaches the end index of its tree.
// The node must belong to the layout of the ray's own octant: its words hold {near plane, far plane} per axis in the
// order the ray meets them, so there is no min/max per axis, and a NaN distance (ray parallel to the axis: Inf - Inf)
// only ever lands in the slot where it is ignored.
__device__ __forceinline__ void node_test(const DevScene& S, const BoxRay& R, float t_min, float t_max, uint32_t node,
                                          bool& hit, uint32_t& link) {
    const uint4 q = S.nodes[node];
    // Box test (src/bounding_box.jl:70-84 form).  It only culls: the stored boxes are rounded strictly outward, and
    // the far distance is widened by 8 ulp so that rounding can never reject a box that the exact slab test of
    // src/bounding_box.jl:20-67 accepts.  A ray parallel to an axis leaves that axis unconstrained (its distances come
    // out as -Inf / +Inf / NaN, and fmaxf / fminf drop the NaN).
    const v2h hx = __builtin_bit_cast(v2h, q.x), hy = __builtin_bit_cast(v2h, q.y), hz = __builtin_bit_cast(v2h, q.z);
    v2f a, b, c; // per axis: distances to the first and the second stored plane (half -> float inside the fused op)
    a.x = __builtin_fmaf((float)hx.x, R.i.x, R.cn.x); a.y = __builtin_fmaf((float)hx.y, R.i.x, R.cf.x);
    b.x = __builtin_fmaf((float)hy.x, R.i.y, R.cn.y); b.y = __builtin_fmaf((float)hy.y, R.i.y, R.cf.y);
    c.x = __builtin_fmaf((float)hz.x, R.i.z, R.cn.z); c.y = __builtin_fmaf((float)hz.y, R.i.z, R.cf.z);
    float tnear = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, t_min));
    float tfar = fminf(fminf(a.y, b.y), fminf(c.y, t_max));
    tfar = __builtin_fmaf(fabsf(tfar), 4.8e-7f, tfar);
    hit = tnear <= tfar;
    link = q.w;
}
__device__ __forceinline__ void node_step(const DevScene& S, const BoxRay& R, float t_min, float t_max, uint32_t& node,
                                          uint32_t& leaf) {
    bool hit;
    uint32_t link;
    node_test(S, R, t_min, t_max, node, hit, link);
    const bool is_leaf = (int32_t)link < 0;
    node = (hit || is_leaf) ? node + 1u : link;
    leaf = (hit && is_leaf) ? (link & ~RT_NODE_LEAF) : 0u;
}

// type group (position in the SceneManager tuple) of a tie-break rank: the groups hold descending ranges of ranks
__device__ __forceinline__ uint32_t group_of(const DevScene& S, uint32_t rank) {
    uint32_t g = 0u;
    while (g + 1u < S.n_buckets && rank < S.bucket_base[g]) ++g;
    return g;
}
// One step of the reduction of hit(::SceneManager) (src/scene_manager.jl:40-52): `x` is what type group `g` returned.
// The first group's result is taken as it is; a later one replaces the hit held only if it is closer, and only then
// does closest_t -- the bound the following groups are given -- change.
__device__ __forceinline__ void reduce_group(Closest& held, float& closest_t, const Closest& x, bool first_group) {
    if (first_group) held = x;
    else if (!(held.t <= x.t)) { held = x; closest_t = x.t; }
}
// The constant media of the scene for a ray whose closest surface is already in `best` (best.t = INF: none; best.rank
// must be the surface's rank).  The reference meets the objects of a scene in a fixed order -- type group after type
// group, inside a group in the order of its tree walk -- and tests each with the closest distance found so far as t_max
// (src/scene_manager.jl:40-53, src/BVH.jl:22-53); for a medium that bound cuts the path through the boundary
// (src/constant_medium.jl:36-44), so it matters what was found before it.  Ranks are that order (higher = met earlier):
// the media are taken in descending rank and the surface is put in where its rank falls, through the reference's own
// reduction: inside a group the bound is the closest medium of the group so far, between the groups it is closest_t,
// which the first group's hit never sets.  (Surfaces other than the closest one are left out: what they could add is a
// tighter bound, and a tighter bound only turns down hits that lose anyway, see below.)  Every lane of a wave meets
// the same medium in the same step.  (The media are few -- a scene-wide fog, a smoke box or two -- so they are a list.)
//
// Returns false when this is not certain to be the reference's answer: a medium was hit in front of its own leaf box
// (only a sphere boundary does that, from far away: the sphere test reports points up to sqrt(1.4e-6) |o - c| off the
// sphere).  The reference tests that box with the bound it has at that moment, and a bound tighter than the one used
// here -- set by a surface of an earlier group that is not the closest one -- can cull the box although the hit lies in
// front of the bound.  For a hit inside its box that cannot matter: a bound that culls the box lies in front of the hit.
// The caller then plays the reference's whole sequence (reference_sequence).
__device__ __forceinline__ bool media_closest(const DevScene& S, f3 o, f3 d, float time, float t_min, float t_max,
                                              const RngKey& key, Closest& best) {
    const Closest none = {INF, 0xFFFFFFFFu, 0u, 0u};
    Closest held = none, local = none;
    float closest_t = t_max;
    uint32_t group = 0xFFFFFFFFu;    // type group of the media being walked (kept by the host next to the list)
    bool surface_in = best.t == INF; // (no surface: nothing to put in)
    const bool surface_first = best.rank >= S.first_group_base; // the surface belongs to the first type group
    bool certain = true;
#pragma unroll 1
    for (uint32_t k = 0; k < S.n_media; ++k) {
        const uint32_t prim = S.media[k], g = S.media_group[k];
        uint4 head;
        Geom G;
        load_prim(S, prim, head, G);
        const float4* bx = S.prim_box + 4u * prim;
        const float4 b0 = bx[0], b1 = bx[1];
        if (!surface_in && best.rank > head.z) { // the surface's group comes before this medium's
            if (group != 0xFFFFFFFFu) reduce_group(held, closest_t, local, group == 0u);
            group = 0xFFFFFFFFu;
            reduce_group(held, closest_t, best, surface_first);
            surface_in = true;
        }
        if (g != group) {
            if (group != 0xFFFFFFFFu) reduce_group(held, closest_t, local, group == 0u);
            group = g;
            local = none;
        }
        const float lim = fminf(closest_t, local.t);
        float near_up;
        if (!ref_reaches<true>(S, prim, b0, b1, 0u, o, d, t_min, lim, near_up)) continue;
        const float t = medium_t(S, head.x, G.g, o, d, time, t_min, lim, key);
        if (!(t < local.t)) continue;
        // in front of the upper bound of the entry distance: now the distance as the reference computes it
        if ((head.x & 15u) == DP_SPHERE && t < near_up && t < ref_box_near(b0, b1, o, d, t_min)) certain = false;
        local.t = t; local.prim = prim; local.sx = ((head.x >> 8) << 3) | ((head.x & DP_TYPE_MASK) << 24); local.rank = head.z;
    }
    if (group != 0xFFFFFFFFu) reduce_group(held, closest_t, local, group == 0u);
    if (!surface_in) reduce_group(held, closest_t, best, surface_first);
    best = held;
    return certain;
}

// back out of a nested tree (its walk is over): restore the outer ray
__device__ __forceinline__ void walker_return(Walker& W, const DevScene& S) {
    W.o = W.so; W.d = W.sd;
    box_frame(S, W.o, W.d, W.R);
    W.node = W.ret; W.end = W.ret_end;
    W.ret = RT_NODE_NONE;
    W.inst_rank = 0; W.inst_xf = 0;
}

template <bool COUNT>
__device__ __forceinline__ void walker_step(Walker& W, const DevScene& S, Counters& cnt) {
    if (W.node == W.end && W.ret != RT_NODE_NONE) walker_return(W, S);
    if (W.node != W.end) {
        if (COUNT) cnt.nodes++;
        node_step(S, W.R, W.t_min, W.t_max, W.node, W.leaf);
    }
}

// One surface leaf for the ray (o, d) of the frame being walked.  A nested tree is entered (returns true, with the
// new frame in `W`-style arguments handled by the caller); any other primitive is tested against `best`.
template <bool FLOOR, uint32_t CL = RT_CL_ALL>
__device__ __forceinline__ void surface_leaf(const float* Tinv, uint32_t prim, const uint4& head, const Geom& G, f3 o, f3 d,
                                             float time, float t_min, float& t_max, uint32_t inst_rank, uint32_t inst_xf,
                                             Closest& best, float floor_t, uint32_t floor_rank) {
    uint32_t kind = head.x & 15u;
    uint32_t side = 0, xf = inst_xf;
    float t = surface_t<CL>(Tinv, kind, kind == DP_TRIANGLE ? 0u : head.x >> 8, G.g, o, d, time, t_min, t_max, side, xf);
    if (FLOOR) consider(best, t_max, t, prim, side | (xf << 3) | ((head.x & DP_TYPE_MASK) << 24), head.z + inst_rank, floor_t, floor_rank);
    else consider(best, t_max, t, prim, side | (xf << 3) | ((head.x & DP_TYPE_MASK) << 24), head.z + inst_rank);
}

// The pending leaf primitive.  Call only for lanes with walker_wants_leaf().
template <bool COUNT>
__device__ __forceinline__ void walker_leaf(Walker& W, const DevScene& S, Counters& cnt) {
    uint32_t prim = (W.leaf & RT_LEAF_PRIM_MASK) - 1u;
    W.leaf = 0u;
    uint4 head;
    Geom G;
    load_prim(S, prim, head, G);
    if (COUNT) cnt.prims++;
    if ((head.x & 15u) == DP_INSTANCE) {
        // Transform{BVHTree}: src/transform.jl:104-116 around src/BVH.jl:22-53
        uint32_t field = head.x >> 8;
        if (!ref_reaches<true>(S, prim, 0u, W.o, W.d, W.t_min, W.t_hi)) return; // the reference culls the whole nested tree
        W.so = W.o; W.sd = W.d;
        W.ret = W.node; W.ret_end = W.end;
        W.inst_rank = head.z;
        W.inst_xf = field;
        if (field) {
            const float* T = S.xforms + 24u * (field - 1u) + 12;
            f3 no = apply_point(T, W.o), nd = apply_vec(T, W.d);
            W.o = no; W.d = nd;
        }
        box_frame(S, W.o, W.d, W.R);
        W.node = __float_as_uint(G.g[0]) + layout_base(S, W.d) + W.set;
        W.end = W.node + __float_as_uint(G.g[1]);
        return;
    }
    surface_leaf<false>((const float*)(S.prim_inv + 3u * prim), prim, head, G, W.o, W.d, W.time, W.t_min, W.t_max, W.inst_rank,
                        W.inst_xf, W.best, 0.0f, 0u);
}

// Wave-wide vote: which block should the wave run now?  0 = step, 1 = leaf, 2 = shade, 3 = nothing left.
// A block runs when enough lanes wait for it (thresholds) or when nothing cheaper is pending.
struct Vote {
    uint32_t leaf_threshold, shade_threshold;
};
__device__ __forceinline__ int wave_vote(bool ws, bool wl, bool wh, const Vote& V) {
    uint32_t ns = (uint32_t)__popcll(__ballot(ws));
    uint32_t nl = (uint32_t)__popcll(__ballot(wl));
    uint32_t nh = (uint32_t)__popcll(__ballot(wh));
    if ((ns | nl | nh) == 0u) return 3;
    if (nh >= V.shade_threshold || (ns == 0u && nl == 0u)) return 2;
    if (nl >= V.leaf_threshold || ns == 0u) return 1;
    return 0;
}

// The hit the reference ends up with, found the way the reference finds it, over the whole scene (what front_sequence
// falls back to when its short list overflows).  The reference meets the primitives in a fixed order (rank, higher
// first).  Inside one type group (one tree) the closest distance so far is t_max for the boxes on the way to a primitive
// and for its own test, and a hit is taken if it is closer (src/BVH.jl:22-53); between the groups hit(::SceneManager)
// (src/scene_manager.jl:40-52) hands on `closest_t`, which its reduction only sets when a later group's hit replaces the
// one held -- the first group's own hit never becomes a bound.  So: again and again, the highest-ranked primitive below
// the last one taken that the reference reaches and hits with the bound of its group -- one walk of the device tree
// per hit the reference takes, a handful.  The device boxes hold every point at which their primitive can report a
// hit, so dropping the boxes that begin behind the largest bound in play loses nothing.
// MEDIA: the constant media take part (they are objects of the scene like any other, in type groups of their own); `key`
// then names the random stream of the ray (the draw of a medium is keyed by the medium, not by the order of the tests).
template <bool MEDIA>
__device__ __forceinline__ Closest reference_sequence(const DevScene& S, f3 o, f3 d, float time, float t_min, float t_hi, const RngKey& key) {
    const f3 wo = o, wd = d; // the ray as cast
    Closest held = {INF, 0xFFFFFFFFu, 0u, 0u};  // the reduction's accumulator
    Closest local = {INF, 0xFFFFFFFFu, 0u, 0u}; // closest of the group being walked
    float4 local_b0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), local_b1 = local_b0; // outer box of the primitive taken last
    bool local_nested = false;
    uint32_t below = 0xFFFFFFFFu; // ranks still to come
    uint32_t group = 0xFFFFFFFFu;
    float closest_t = t_hi, bound = t_hi;
    for (;;) {
        // closest_t as the groups after this one will get it
        const float handed_on = (group != 0xFFFFFFFFu && group != 0u && !(held.t <= local.t)) ? local.t : closest_t;
        const float widest = fmaxf(bound, handed_on);
        Closest next = {INF, 0xFFFFFFFFu, 0u, 0u};
        float4 next_b0 = local_b0, next_b1 = local_b1;
        bool next_nested = false;
        o = wo; d = wd;
        BoxRay R;
        box_frame(S, o, d, R);
        const uint32_t set = node_set(S, wo);
        uint32_t node = S.surface_root + layout_base(S, d) + set, end = node + S.surface_size;
        uint32_t ret = RT_NODE_NONE, ret_end = 0u, inst_rank = 0u, inst_xf = 0u;
        for (;;) {
            if (node == end) {
                if (ret == RT_NODE_NONE) break;
                o = wo; d = wd; // back out of the nested tree
                box_frame(S, o, d, R);
                node = ret; end = ret_end;
                ret = RT_NODE_NONE;
                inst_rank = 0u; inst_xf = 0u;
                continue;
            }
            uint32_t leaf;
            node_step(S, R, t_min, widest, node, leaf);
            if (leaf == 0u) continue;
            const uint32_t prim = (leaf & RT_LEAF_PRIM_MASK) - 1u;
            uint4 hd;
            Geom G;
            load_prim(S, prim, hd, G);
            const float4 b0 = S.prim_box[4u * prim], b1 = S.prim_box[4u * prim + 1u];
            const uint32_t rank = (leaf & RT_LEAF_INSTANCE) ? hd.z : hd.z + inst_rank; // (a nested tree lies in one group)
            const bool in_group = group_of(S, rank) == group;
            const float lim = in_group ? bound : handed_on;
            // the nested tree the reference is inside of once it has taken a hit there: its outer box was passed before
            const bool same_tree = in_group && local_nested && b0.x == local_b0.x && b0.y == local_b0.y && b0.z == local_b0.z &&
                                   b0.w == local_b0.w && b1.x == local_b1.x && b1.y == local_b1.y;
            if (leaf & RT_LEAF_INSTANCE) { // Transform{BVHTree}: src/transform.jl:104-116 around src/BVH.jl:22-53
                if (!same_tree && !ref_reaches<true>(S, prim, b0, b1, 0u, o, d, t_min, lim)) continue;
                const uint32_t field = hd.x >> 8;
                ret = node; ret_end = end;
                inst_rank = hd.z;
                inst_xf = field;
                if (field) {
                    const float* T = S.xforms + 24u * (field - 1u) + 12;
                    f3 no = apply_point(T, o), nd = apply_vec(T, d);
                    o = no; d = nd;
                }
                box_frame(S, o, d, R);
                node = __float_as_uint(G.g[0]) + layout_base(S, d) + set;
                end = node + __float_as_uint(G.g[1]);
                continue;
            }
            if (rank >= below || (next.prim != 0xFFFFFFFFu && rank <= next.rank)) continue;
            const uint32_t kind = hd.x & 15u;
            const uint32_t field = kind == DP_TRIANGLE ? 0u : hd.x >> 8;
            float unused;
            if (!ref_reaches<true>(S, prim, b0, b1, field ? field : inst_xf, wo, wd, t_min, lim, unused, same_tree)) continue;
            uint32_t side = 0, xf = inst_xf;
            const float t = surface_t((const float*)(S.prim_inv + 3u * prim), kind, field, G.g, o, d, time, t_min, lim, side, xf);
            if (!(t < INF) || (in_group && !(t < local.t))) continue; // strict: an equal distance keeps the earlier hit
            next.t = t; next.prim = prim; next.sx = side | (xf << 3) | ((hd.x & DP_TYPE_MASK) << 24); next.rank = rank;
            next_b0 = b0; next_b1 = b1;
            next_nested = (__float_as_uint(b1.z) & RT_BOX_INNER) != 0u;
        }
        if (MEDIA) {
            // the media between the last primitive taken and the surface found: the first one (highest rank) that the
            // reference reaches and that is hit with the bound of its group comes before that surface
            for (uint32_t k = 0; k < S.n_media; ++k) {
                const uint32_t prim = S.media[k];
                uint4 hd;
                Geom G;
                load_prim(S, prim, hd, G);
                const uint32_t rank = hd.z;
                if (rank >= below) continue;
                if (next.prim != 0xFFFFFFFFu && rank <= next.rank) break;
                const bool in_group = group_of(S, rank) == group;
                const float lim = in_group ? bound : handed_on;
                if (!ref_reaches<true>(S, prim, 0u, wo, wd, t_min, lim)) continue;
                const float t = medium_t(S, hd.x, G.g, wo, wd, time, t_min, lim, key);
                if (!(t < INF) || (in_group && !(t < local.t))) continue;
                next.t = t; next.prim = prim; next.sx = ((hd.x >> 8) << 3) | ((hd.x & DP_TYPE_MASK) << 24); next.rank = rank;
                next_nested = false;
                break;
            }
        }
        const uint32_t g = next.prim != 0xFFFFFFFFu ? group_of(S, next.rank) : 0xFFFFFFFFu;
        if (g != group) { // the group is done: its closest hit goes into the reduction
            if (group == 0u) held = local;
            else if (group != 0xFFFFFFFFu && !(held.t <= local.t)) { held = local; closest_t = local.t; }
            group = g;
            local.t = INF; local.prim = 0xFFFFFFFFu;
        }
        if (next.prim == 0xFFFFFFFFu) break;
        local = next; local_b0 = next_b0; local_b1 = next_b1; local_nested = next_nested;
        bound = local.t;
        below = local.rank;
    }
    return held;
}
// The closest hit behind (floor_t, floor_rank) in the order (distance, then higher rank first), no further than t_bound;
// none: t = INF.  One walk of the device tree with the usual shrinking bound.
__device__ __forceinline__ Closest next_hit_after(const DevScene& S, f3 wo, f3 wd, float time, float t_min, float t_bound,
                                                  float floor_t, uint32_t floor_rank) {
    Closest best = {INF, 0xFFFFFFFFu, 0u, 0u};
    float t_max = t_bound;
    f3 o = wo, d = wd;
    BoxRay R;
    box_frame(S, o, d, R);
    const uint32_t set = node_set(S, wo);
    uint32_t node = S.surface_root + layout_base(S, d) + set, end = node + S.surface_size;
    uint32_t ret = RT_NODE_NONE, ret_end = 0u, inst_rank = 0u, inst_xf = 0u;
    for (;;) {
        if (node == end) {
            if (ret == RT_NODE_NONE) break;
            o = wo; d = wd; // back out of the nested tree
            box_frame(S, o, d, R);
            node = ret; end = ret_end;
            ret = RT_NODE_NONE;
            inst_rank = 0u; inst_xf = 0u;
            continue;
        }
        uint32_t leaf;
        node_step(S, R, t_min, t_max, node, leaf);
        if (leaf == 0u) continue;
        const uint32_t prim = (leaf & RT_LEAF_PRIM_MASK) - 1u;
        uint4 hd;
        Geom G;
        load_prim(S, prim, hd, G);
        if (leaf & RT_LEAF_INSTANCE) { // Transform{BVHTree}: src/transform.jl:104-116 around src/BVH.jl:22-53
            if (!ref_reaches<true>(S, prim, 0u, o, d, t_min, INF)) continue; // hits in front of that box count too
            const uint32_t field = hd.x >> 8;
            ret = node; ret_end = end;
            inst_rank = hd.z;
            inst_xf = field;
            if (field) {
                const float* T = S.xforms + 24u * (field - 1u) + 12;
                f3 no = apply_point(T, o), nd = apply_vec(T, d);
                o = no; d = nd;
            }
            box_frame(S, o, d, R);
            node = __float_as_uint(G.g[0]) + layout_base(S, d) + set;
            end = node + __float_as_uint(G.g[1]);
        } else {
            surface_leaf<true>((const float*)(S.prim_inv + 3u * prim), prim, hd, G, o, d, time, t_min, t_max, inst_rank, inst_xf,
                               best, floor_t, floor_rank);
        }
    }
    return best;
}

// For a ray whose closest surface hit does not stand (hit_stands; about one ray in a hundred thousand), or one of whose
// media is hit in front of its own box: the hit the reference ends up with.  Only the hits at the front can matter.
// Let N be the closest hit that stands on its own (reached, not in front of its box) and T the largest of N's distance
// and the box-entry distances of the hits in front of their boxes found up to T.  A hit beyond T is never the last one
// taken (N is closer), and the bound it leaves behind, being beyond T, neither culls a box nor turns down a hit among
// those up to T.  So the surface hits up to T are listed (closest first, one cheap walk each), the constant media -- a
// handful, each a hit or not when nothing bounds it -- are put beside them, and the reference's sequence (see
// reference_sequence) is played over that short list, each medium tested again with the bound it meets there; if the
// list overflows, over the whole scene.
#define SETTLE_LIST 6
// first: the closest of all the hits the surfaces report (what the walk of the device tree found).  MEDIA: the scene's
// constant media take part, `key` names the ray's random stream.
// list: room for SETTLE_LIST entries of the calling lane's own (in LDS: a list indexed by a variable would live in
// scratch memory if it were a local array, and the kernels that settle hits are kept free of scratch so that their
// first launch on a queue does not make the runtime set scratch memory aside -- rt_render_accumulate_static must not
// cause device memory to be taken).
template <bool MEDIA>
__device__ __forceinline__ Closest front_sequence(const DevScene& S, f3 o, f3 d, float time, float t_min, float t_hi, Closest first,
                                                  const RngKey& key, Closest* list) {
    int n = 0;
    float stands_t = INF;     // distance of the closest hit that stands (N)
    float front = t_min;      // largest box-entry distance among the hits in front of their boxes
    uint32_t media_hit = 0u;  // media (positions in S.media) that are hit when nothing bounds them
    if (MEDIA) {
        if (S.n_media > 32u) return reference_sequence<true>(S, o, d, time, t_min, t_hi, key);
        for (uint32_t k = 0; k < S.n_media; ++k) {
            const uint32_t prim = S.media[k];
            uint4 hd;
            Geom G;
            load_prim(S, prim, hd, G);
            const float4 b0 = S.prim_box[4u * prim], b1 = S.prim_box[4u * prim + 1u];
            float near_up;
            if (!ref_reaches<true>(S, prim, b0, b1, 0u, o, d, t_min, t_hi, near_up)) continue;
            const float t = medium_t(S, hd.x, G.g, o, d, time, t_min, t_hi, key);
            if (!(t < INF)) continue;
            media_hit |= 1u << k;
            float near = near_up;
            bool in_front = (hd.x & 15u) == DP_SPHERE && t < near_up;
            if (in_front) { near = ref_box_near(b0, b1, o, d, t_min); in_front = t < near; }
            if (in_front) front = fmaxf(front, near);
            else stands_t = fminf(stands_t, t);
        }
    }
    float T = fminf(t_hi, fmaxf(stands_t, front));
    float floor_t = -INF;
    uint32_t floor_rank = 0u;
    for (bool start = true; first.t != INF || !start; start = false) {
        if (!start && stands_t < INF && !(T > floor_t)) break; // nothing in front reaches beyond the hit that stands: the list is complete
        const Closest c = start ? first : next_hit_after(S, o, d, time, t_min, T, floor_t, floor_rank);
        if (c.t == INF) break;
        floor_t = c.t; floor_rank = c.rank;
        const float4* b = S.prim_box + 4u * c.prim;
        float near_up;
        if (!ref_reaches<false>(S, c.prim, b[0], b[1], (c.sx >> 3) & 0x3FFFFFFu, o, d, t_min, t_hi, near_up)) continue;
        // in front of its own box?  near_up is an upper bound of the entry distance: below it, the reference's own
        bool in_front = (S.prims[3u * c.prim].x & 15u) <= DP_MOVING_SPHERE && c.t < near_up;
        if (in_front) {
            near_up = ref_entry(S, c.prim, (c.sx >> 3) & 0x3FFFFFFu, o, d, t_min);
            in_front = c.t < near_up;
        }
        if (n == SETTLE_LIST) return reference_sequence<MEDIA>(S, o, d, time, t_min, t_hi, key);
        list[n++] = c;
        if (in_front) front = fmaxf(front, near_up);
        else stands_t = fminf(stands_t, c.t);
        T = fminf(t_hi, fmaxf(stands_t, front));
    }
    // The reference's sequence over the list, highest rank first.  Inside one type group (one tree) the bound shrinks
    // with every hit taken (src/BVH.jl:36-47).  Between the groups, hit(::SceneManager) (src/scene_manager.jl:40-52)
    // hands on `closest_t`, which its reduction only sets when a later group's hit replaces the one held: the first
    // group's own hit never becomes a bound, and a group that loses leaves it as it was.
    Closest held = {INF, 0xFFFFFFFFu, 0u, 0u};  // the reduction's accumulator
    Closest local = {INF, 0xFFFFFFFFu, 0u, 0u}; // closest of the group being walked
    float closest_t = t_hi, bound = t_hi;
    uint32_t group = 0xFFFFFFFFu, below = 0xFFFFFFFFu;
    float4 local_b0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), local_b1 = local_b0;
    bool local_nested = false;
    for (;;) {
        int pick = -1;       // list entry with the highest rank still to come ...
        uint32_t pick_rank = 0u;
        for (int k = 0; k < n; ++k)
            if (list[k].rank < below && (pick < 0 || list[k].rank > pick_rank)) { pick = k; pick_rank = list[k].rank; }
        int medium = -1;     // ... or the medium that comes before it
        if (MEDIA)
            for (uint32_t k = 0; k < S.n_media; ++k) {
                const uint32_t rank = S.prims[3u * S.media[k]].z;
                if ((media_hit >> k & 1u) && rank < below && ((pick < 0 && medium < 0) || rank > pick_rank)) { medium = (int)k; pick_rank = rank; }
            }
        const bool any = pick >= 0 || medium >= 0;
        const uint32_t g = any ? group_of(S, pick_rank) : 0xFFFFFFFFu;
        if (g != group) { // the group is done: its closest hit goes into the reduction
            if (group == 0u) held = local;
            else if (group != 0xFFFFFFFFu && !(held.t <= local.t)) { held = local; closest_t = local.t; }
            group = g;
            local.t = INF; local.prim = 0xFFFFFFFFu;
            local_nested = false;
            bound = closest_t;
        }
        if (!any) break;
        below = pick_rank;
        if (MEDIA && medium >= 0) {
            const uint32_t prim = S.media[medium];
            uint4 hd;
            Geom G;
            load_prim(S, prim, hd, G);
            if (!ref_reaches<true>(S, prim, 0u, o, d, t_min, bound)) continue;
            const float t = medium_t(S, hd.x, G.g, o, d, time, t_min, bound, key);
            if (!(t < local.t)) continue;
            local.t = t; local.prim = prim; local.sx = ((hd.x >> 8) << 3) | ((hd.x & DP_TYPE_MASK) << 24); local.rank = hd.z;
            local_nested = false;
            bound = t;
            continue;
        }
        const Closest c = list[pick];
        const float4 b0 = S.prim_box[4u * c.prim], b1 = S.prim_box[4u * c.prim + 1u];
        const bool same_tree = local_nested && b0.x == local_b0.x && b0.y == local_b0.y && b0.z == local_b0.z && b0.w == local_b0.w &&
                               b1.x == local_b1.x && b1.y == local_b1.y;
        float unused;
        if (!ref_reaches<false>(S, c.prim, b0, b1, (c.sx >> 3) & 0x3FFFFFFu, o, d, t_min, bound, unused, same_tree)) continue;
        if (!(c.t <= bound) || !(c.t < local.t)) continue; // an equal distance keeps the earlier hit
        local = c; local_b0 = b0; local_b1 = b1;
        local_nested = (__float_as_uint(b1.z) & RT_BOX_INNER) != 0u;
        bound = c.t;
    }
    return held;
}
// The closest hit of the whole scene as the reference has it, from `best`, the closest surface the walk of the device
// tree found.  Short way, almost always: the surface stands (hit_stands) and the media go through the reference's
// reduction with it (media_closest).  Otherwise the reference's sequence is played over the hits at the front.
__device__ __forceinline__ void settle_hit(const DevScene& S, f3 o, f3 d, float time, float t_min, float t_hi, const RngKey& key, Closest& best,
                                           Closest* list) {
    if (best.t != INF && !hit_stands(S, best.prim, (best.sx >> 3) & 0x3FFFFFFu, o, d, t_min, t_hi, best.t)) {
        if (S.n_media == 0u) best = front_sequence<false>(S, o, d, time, t_min, t_hi, best, key, list);
        else best = front_sequence<true>(S, o, d, time, t_min, t_hi, best, key, list);
        return;
    }
    if (S.n_media == 0u) return;
    Closest with_media = best;
    if (media_closest(S, o, d, time, t_min, t_hi, key, with_media)) best = with_media;
    else best = front_sequence<true>(S, o, d, time, t_min, t_hi, best, key, list);
}

// closest hit of one ray per lane (used by the ray parity hook)
template <bool COUNT>
__device__ __forceinline__ void closest_hit(const DevScene& S, f3 o, f3 d, float time, float t_min, float t_max,
                                            const RngKey& key, Closest& best, Counters& cnt, bool active, Closest* list) {
    Walker W;
    walker_start(W, S, o, d, time, t_min, t_max);
    Vote V = {16u, 64u};
    for (;;) {
        bool ws = active && walker_wants_step(W), wl = active && walker_wants_leaf(W);
        int block = wave_vote(ws, wl, false, V);
        if (block == 3) break;
        if (block == 0) { if (ws) walker_step<COUNT>(W, S, cnt); }
        else { if (wl) walker_leaf<COUNT>(W, S, cnt); }
    }
    best = W.best;
    if (active) settle_hit(S, o, d, time, t_min, t_max, key, best, list);
}

// ------------------------------------------------------------------ hit completion (only for the winner)
struct HitRec { // src/hit_record.jl:4-12 with the material replaced by its index
    f3 p, n;
    float u, v;
    uint32_t material;
    bool front;
};

template <bool FULL_UV>
__device__ __forceinline__ void complete_hit(const DevScene& S, f3 o, f3 d, float time, const Closest& best,
                                             HitRec& h) {
    uint4 head;
    Geom G;
    load_prim(S, best.prim, head, G);
    const float* g = G.g;
    uint32_t kind = head.x & 15u;
    uint32_t side = best.sx & 7u, xf = (best.sx >> 3) & 0x03FFFFFFu;
    float t = best.t;
    h.material = head.y;
    if (head.x & DP_MEDIUM_FLAG) { // src/constant_medium.jl:62-64
        h.p = at3(o, d, t);
        h.n = mk3(1.0f, 0.0f, 0.0f);
        h.front = false;
        h.u = 0.0f; h.v = 0.0f;
        return;
    }
    f3 ro = o, rd = d;
    const float* T = nullptr;
    if (xf) {
        T = S.xforms + 24u * (xf - 1u);
        ro = apply_point(T + 12, o);
        rd = apply_vec(T + 12, d);
    }
    bool want_uv = FULL_UV || (S.materials[head.y].x & DM_NEEDS_UV);
    h.u = 0.0f; h.v = 0.0f;
    if (kind == DP_SPHERE || kind == DP_MOVING_SPHERE) { // src/sphere.jl:67-71
        f3 center;
        float radius;
        if (kind == DP_SPHERE) { center = mk3(g[0], g[1], g[2]); radius = g[3]; }
        else {
            f3 c0 = mk3(g[0], g[1], g[2]), c1 = mk3(g[4], g[5], g[6]);
            center = c0 + ((time - g[3]) / (g[7] - g[3])) * (c1 - c0);
            radius = g[8];
        }
        h.p = at3(ro, rd, t);
        f3 normal = (h.p - center) * (1.0f / radius);
        h.front = dot3(rd, normal) < 0.0f;
        if (want_uv) sphere_uv(normal, h.u, h.v);
        h.n = (h.front ? 1.0f : -1.0f) * normal;
    } else if (kind == DP_TRIANGLE) { // src/triangle.jl:101-121
        f3 normal;
        float u, v;
        triangle_t(g, ro, rd, -INF, INF, normal, u, v);
        const float* A = S.tri_attr + 16u * (head.x >> 8);
        float w = 1.0f - u - v;
        f3 vn = (mk3(A[0], A[1], A[2]) * w + mk3(A[3], A[4], A[5]) * u) + mk3(A[6], A[7], A[8]) * v;
        float vu = (A[9] * w + A[11] * u) + A[13] * v;
        float vv = (A[10] * w + A[12] * u) + A[14] * v;
        if (is_zero3(vn)) normal = vn; // the reference's inverted "has normal data" flag (src/triangle.jl:110-113)
        if (vu >= 0.0f && vv >= 0.0f) { u = vu; v = vv; }
        h.p = at3(ro, rd, t);
        h.front = dot3(rd, normal) < 0.0f;
        h.n = (h.front ? 1.0f : -1.0f) * normalize3(normal);
        h.u = u; h.v = v;
    } else { // rectangles and box sides: src/aarect.jl:49-119
        float a0, a1, b0, b1, k;
        uint32_t axis; // constant axis: 2 = xy_rect, 1 = xz_rect, 0 = yz_rect
        if (kind == DP_BOX) { // src/box.jl:13-22
            switch (side) {
                case 0: axis = 2; a0 = g[0]; a1 = g[3]; b0 = g[1]; b1 = g[4]; k = g[5]; break;
                case 1: axis = 2; a0 = g[0]; a1 = g[3]; b0 = g[1]; b1 = g[4]; k = g[2]; break;
                case 2: axis = 1; a0 = g[0]; a1 = g[3]; b0 = g[2]; b1 = g[5]; k = g[4]; break;
                case 3: axis = 1; a0 = g[0]; a1 = g[3]; b0 = g[2]; b1 = g[5]; k = g[1]; break;
                case 4: axis = 0; a0 = g[1]; a1 = g[4]; b0 = g[2]; b1 = g[5]; k = g[3]; break;
                default: axis = 0; a0 = g[1]; a1 = g[4]; b0 = g[2]; b1 = g[5]; k = g[0]; break;
            }
        } else {
            axis = kind == DP_XY_RECT ? 2u : (kind == DP_XZ_RECT ? 1u : 0u);
            a0 = g[0]; a1 = g[1]; b0 = g[2]; b1 = g[3]; k = g[4];
        }
        float oa, da, ob, db, dk;
        if (axis == 2) { oa = ro.x; da = rd.x; ob = ro.y; db = rd.y; dk = rd.z; }
        else if (axis == 1) { oa = ro.x; da = rd.x; ob = ro.z; db = rd.z; dk = rd.y; }
        else { oa = ro.y; da = rd.y; ob = ro.z; db = rd.z; dk = rd.x; }
        float a = oa + t * da;
        float b = ob + t * db;
        // dot(direction, +axis) < 0 with the zero products kept (src/aarect.jl:98-99)
        float dn = axis == 2 ? (rd.x * 0.0f + rd.y * 0.0f + dk * 1.0f)
                             : (axis == 1 ? (rd.x * 0.0f + dk * 1.0f + rd.z * 0.0f) : (dk * 1.0f + rd.y * 0.0f + rd.z * 0.0f));
        h.front = dn < 0.0f;
        float sgn = h.front ? 1.0f : -1.0f;
        if (axis == 2) { h.p = mk3(a, b, k); h.n = mk3(sgn * 0.0f, sgn * 0.0f, sgn * 1.0f); }
        else if (axis == 1) { h.p = mk3(a, k, b); h.n = mk3(sgn * 0.0f, sgn * 1.0f, sgn * 0.0f); }
        else { h.p = mk3(k, a, b); h.n = mk3(sgn * 1.0f, sgn * 0.0f, sgn * 0.0f); }
        h.u = (a - a0) / (a1 - a0);
        h.v = (b - b0) / (b1 - b0);
    }
    if (xf) { // src/transform.jl:111-115
        h.p = at3(o, d, t);
        h.n = apply_vec(T, h.n);
    }
}

// ------------------------------------------------------------------ textures (src/texture.jl:42-100, src/perlin.jl)
__device__ __noinline__ float perlin_noise(const DevScene& S, uint32_t pi, f3 p) { // src/perlin.jl:48-85
    const float* vec = S.perlin_vec + 768u * pi;
    const int32_t* perm = S.perlin_perm + 768u * pi;
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                int idx = perm[(i + ii + 1) & 255] ^ perm[256 + ((j + jj + 1) & 255)] ^ perm[512 + ((k + kk + 1) & 255)];
                f3 c = mk3(vec[3 * idx], vec[3 * idx + 1], vec[3 * idx + 2]);
                f3 weight_v = mk3(u - (float)ii, v - (float)jj, w - (float)kk);
                float fi = (float)ii, fj = (float)jj, fk = (float)kk;
                accum += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
                         (fk * ww + (1.0f - fk) * (1.0f - ww)) * dot3(c, weight_v);
            }
    return accum;
}
__device__ __noinline__ float perlin_noise_scalar(const DevScene& S, uint32_t pi, f3 p) { // src/perlin.jl:87-97
    const float* vec = S.perlin_vec + 768u * pi;
    const int32_t* perm = S.perlin_perm + 768u * pi;
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float accum = 0.0f;
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                int idx = perm[(i + ii + 1) & 255] ^ perm[256 + ((j + jj + 1) & 255)] ^ perm[512 + ((k + kk + 1) & 255)];
                float fi = (float)ii, fj = (float)jj, fk = (float)kk;
                accum += (fi * u + (1.0f - fi) * (1.0f - u)) * (fj * v + (1.0f - fj) * (1.0f - v)) *
                         (fk * w + (1.0f - fk) * (1.0f - w)) * vec[3 * idx];
            }
    return accum;
}
__device__ __forceinline__ float turbulence(const DevScene& S, uint32_t pi, f3 p) { // src/perlin.jl:35-46
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(S, pi, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}

// TEXTURED = false: every texture of the scene is a SolidColor (the host checked), so only that case is compiled in;
// the noise and image code would otherwise set the register budget of the whole shade kernel.
template <bool TEXTURED>
__device__ __forceinline__ f3 texture_value(const DevScene& S, uint32_t ti, float u, float v, f3 p) {
    if (!TEXTURED) {
        uint4 b = S.textures[2u * ti + 1u];
        return mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
    }
    // a checker nests two textures (src/texture.jl:46-53): resolve it iteratively
    for (int depth = 0; depth < 8; ++depth) {
        uint4 a = S.textures[2u * ti];
        uint4 b = S.textures[2u * ti + 1u];
        switch (a.x) {
            case 0: // SolidColor
                return mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            case 1: { // CheckerTexture
                float sines = det_sinf(10.0f * p.x) * det_sinf(10.0f * p.y) * det_sinf(10.0f * p.z);
                ti = sines < 0.0f ? a.y : a.z;
                continue;
            }
            case 2: { // ImageTexture, src/texture.jl:55-76
                DevImage im = S.images[a.y];
                if (im.width == 0 || im.height == 0) return mk3(0.0f, 1.0f, 1.0f);
                u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                v = 1.0f - v;
                int i = (int)(v * (float)im.height);
                int j = (int)(u * (float)im.width);
                if (i == 0) i = 1;
                if (j == 0) j = 1;
                const uint8_t* px = S.image_data + im.offset + 3u * ((uint32_t)(i - 1) * im.width + (uint32_t)(j - 1));
                return mk3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
            }
            case 3: { // NoiseTexture{Vec3}: marble, src/texture.jl:98-100
                float arg = __uint_as_float(b.w) * p.z + 10.0f * turbulence(S, a.y, p);
                float val = 0.5f * (1.0f + det_sinf(arg));
                return mk3(val, val, val);
            }
            default: { // NoiseTexture{F}, src/texture.jl:79-81
                float val = 1.0f * perlin_noise_scalar(S, a.y, __uint_as_float(b.w) * p);
                return mk3(val, val, val);
            }
        }
    }
    return mk3(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------ materials (src/material.jl:23-38, src/hittable.jl:36-83)
__device__ __forceinline__ f3 reflect3(f3 v, f3 n) { return v - (2.0f * dot3(v, n)) * n; }
__device__ __forceinline__ f3 refract3(f3 dir, f3 n, float ratio) {
    float cos_theta = dot3(-dir, n);
    f3 perp = ratio * (dir + cos_theta * n);
    f3 parallel = (-sqrtf(fabsf(1.0f - dot3(perp, perp)))) * n;
    return perp + parallel;
}
__device__ __forceinline__ float schlick(float cos_theta, float eta) {
    float k = (1.0f - eta) / (1.0f + eta);
    float r0 = k * k;
    float x = 1.0f - cos_theta;
    float x2 = x * x;
    float x5 = (x2 * x2) * x;
    return r0 + (1.0f - r0) * x5;
}

// State of the path a lane is tracing: the loop variables of trace() (src/Ray_Tracing.jl:11-29 / :51-69)
struct PathState {
    f3 acc, color;
    int depth;
};

// The body of the while loop of trace() after hit(scene, ray) returned `best` for the ray (o, d, time).
// Returns true when the path ended, with its radiance in `result`; otherwise (no, nd) is the scattered ray.
template <bool TEXTURED>
__device__ __forceinline__ bool shade(const DevScene& S, f3 o, f3 d, float time, const Closest& best, PathState& ps,
                                      f3 background, RngKey& key, f3& result, f3& no, f3& nd) {
    if (best.t == INF) { result = ps.color + ps.acc * background; return true; }
    HitRec h;
    complete_hit<false>(S, o, d, time, best, h);
    uint4 m = S.materials[h.material];
    uint32_t type = m.x & 0xFFu;
    // What several materials need is evaluated once for the whole wave, not once per material: the rejection loop of
    // random_in_unit_sphere (Lambertian, Metal, Isotropic draw the same numbers from the same counters) and the texture.
    f3 rs = mk3(0.0f, 0.0f, 0.0f);
    if (type == 1u || type == 2u || type == 5u) rs = random_in_unit_sphere(key);
    f3 tex = mk3(1.0f, 1.0f, 1.0f);
    if (type != 3u) tex = texture_value<TEXTURED>(S, m.y, h.u, h.v, h.p);
    f3 att;
    f3 emitted = mk3(0.0f, 0.0f, 0.0f);
    if (type == 1u) { // Lambertian
        nd = h.n + normalize3(rs);
        if (is_zero3(nd)) nd = h.n;
        att = tex;
    } else if (type == 2u) { // Metal
        f3 reflected = reflect3(normalize3(d), h.n);
        nd = reflected + __uint_as_float(m.z) * rs;
        att = dot3(nd, h.n) > 0.0f ? tex : mk3(0.0f, 0.0f, 0.0f);
    } else if (type == 3u) { // Dielectric
        float eta = __uint_as_float(m.w);
        float ratio = h.front ? (1.0f / eta) : eta;
        f3 unit_dir = normalize3(d);
        float cos_theta = fminf(dot3(-unit_dir, h.n), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool refl = ratio * sin_theta > 1.0f;
        if (!refl) refl = schlick(cos_theta, eta) > u01(rng_draw(key, PURPOSE_SCALAR, 0).x);
        nd = refl ? reflect3(unit_dir, h.n) : refract3(unit_dir, h.n, ratio);
        att = mk3(1.0f, 1.0f, 1.0f);
    } else if (type == 5u) { // Isotropic
        nd = rs;
        att = tex;
    } else { // Emissive: zero ray, zero attenuation (src/hittable.jl:72-74), emits on both faces (:77-83)
        nd = mk3(0.0f, 0.0f, 0.0f);
        att = mk3(0.0f, 0.0f, 0.0f);
        emitted = tex;
    }
    ps.color = ps.color + ps.acc * emitted;
    if (is_zero3(att)) { result = ps.color; return true; }
    ps.acc = ps.acc * att;
    ps.depth -= 1;
    key.bounce += 1;
    if (ps.depth <= 0) { result = ps.color; return true; }
    no = h.p;
    return false;
}

// get_ray (src/camera.jl:38-42) for pixel column i, row j (from the bottom), pattern column n
__device__ __forceinline__ void camera_ray(const RtCameraDev& cam, const float* pattern, int i, int j, int n, int width,
                                           int height, RngKey& key, f3& o, f3& d, float& time) {
    float dx = pattern[2 * n], dy = pattern[2 * n + 1];
    float s = ((float)i + dx) / (float)(width - 1); // src/Ray_Tracing.jl:85-86
    float t = ((float)j + dy) / (float)(height - 1);
    key.bounce = CAMERA_BOUNCE;
    f3 cu = mk3(cam.u[0], cam.u[1], cam.u[2]), cv = mk3(cam.v[0], cam.v[1], cam.v[2]);
    f3 origin = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    f3 llc = mk3(cam.llc[0], cam.llc[1], cam.llc[2]);
    f3 hor = mk3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    f3 ver = mk3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    f3 rd = cam.lens_radius * random_in_unit_disk(key);
    o = (cu * rd.x + cv * rd.y) + origin;
    time = u01(rng_draw(key, PURPOSE_SCALAR, 0).x) * (cam.t1 - cam.t0) + cam.t0;
    d = ((llc + s * hor) + t * ver) - o;
}

// ------------------------------------------------------------------ kernels
// The frame is rendered wavefront style.  A *path* is one trace() call: pattern column k of pixel r (r = row-
// major pixel index, j * width + i, j counted from the bottom like the loops of draw!).  Paths are numbered
// p = (k - k_begin) * n_pixels + r inside the batch of columns [k_begin, k_end) being rendered.  A pool of
// *slots* (PathPool, five float4 planes in HBM) holds the paths in flight; every bounce is two launches:
//   rt_extend_kernel  closest hit of every slot's ray (persistent waves; a lane fetches the next slot as soon as
//                     its ray is finished; node stepping and leaf tests run under a wave-wide vote)
//   rt_shade_kernel   hit completion, emission, scatter; a slot whose path ended writes the path's radiance to
//                     radiance[p] and takes the next unstarted path.  The slots of a workgroup (SHADE_BLOCK) are regrouped
//                     by material type in LDS so that a wave shades one material, and the rays that go on are written
//                     back in order of direction octant and origin cell, so that a wave of rt_extend_kernel gets
//                     rays that walk the same nodes.
// rt_sum_kernel then adds each pixel's radiances in column order, the order of the sample loop of draw!
// (src/Ray_Tracing.jl:82-89), so the image does not depend on scheduling.
// plane 0: {origin, time}        plane 1: {direction, bits(depth | flag)}   depth == 0: the slot holds no ray
// plane 2: {acc, bits(pixel)}        plane 3: {t, bits(prim), bits(side|xform|type), bits(rank)}: the closest hit of the
// device tree, t = Inf: none.   plane 4: {color, -}, only valid (and only touched) when DEPTH_HAS_COLOR is set in the depth word
// Material types travel as 1..5, HIT_MISS = nothing was hit.
// Path dispensers (words of the sync array): counter c hands out local indices v; units of DISP_UNIT consecutive paths
// are dealt round-robin to the counters, so every counter covers the whole batch evenly.
#define DISP_BASE RT_DISP_BASE
#define DISP_STRIDE RT_DISP_STRIDE
#define DISP_UNIT RT_DISP_UNIT
// depth word of a slot: remaining depth (bits 0-5), pattern column of the path inside the batch (bits 6-29), colour flag
#define DEPTH_MASK 0x3Fu
#define DEPTH_COLUMN_SHIFT 6
#define DEPTH_COLUMN_MASK 0xFFFFFFu
#define DEPTH_HAS_COLOR 0x40000000u
#define DEPTH_HOLD 0x80000000u       // the slot's hit is being settled (rt_redo_kernel): rt_extend_kernel leaves the slot alone
// Index of a slot's entry in a plane of a pool of n slots.  Planes, not one 64-byte record per slot: rt_extend_kernel
// reads two planes and writes one, and a record per slot makes it pull the whole record through the caches (measured:
// extend +3 %, shade +4 %, the frame 1497 -> 1444 Mpaths/s).
#define QPI(n, plane, slot) ((size_t)(plane) * (n) + (slot))
#define QP(pool, plane, slot) (pool).q[QPI((pool).n_slots, plane, slot)]
#define HIT_MISS 6u
#define HIT_WAIT 7u                  // shade kernel: the slot waits for rt_redo_kernel, nothing to shade yet
#define RANK_SETTLED 0x80000000u     // bit of the rank word of a hit record: rt_redo_kernel has settled the hit
#define EXTEND_CHUNK 128u

// random-number key of the path at pixel r (row-major) and column kk of the batch; returns the pattern column
__device__ __forceinline__ int path_key(const RtRenderParams& P, uint32_t kk, uint32_t r, RngKey& key) {
    int n = P.sample_first + (int)(P.k_begin + kk) * P.sample_stride;
    key.pixel = r;
    key.sample = (uint32_t)n;
    key.seed32 = P.seed32;
    return n;
}

__device__ __forceinline__ uint32_t dispensed_path(const RtRenderParams& P, uint32_t c, uint32_t v) {
    return ((v / DISP_UNIT) * P.n_disp + c) * DISP_UNIT + (v % DISP_UNIT);
}
__device__ __forceinline__ uint32_t dispenser_limit(const RtRenderParams& P, uint32_t c) { return rt_dispenser_limit(P.n_paths, P.n_disp, c); }

// camera ray of path p into slot `slot`
__device__ __forceinline__ void start_path(const RtCameraDev& cam, const float* pattern, const RtRenderParams& P,
                                           PathPool& pool, uint32_t slot, uint32_t p) {
    // the only place where a path number is taken apart (two divisions per path, none per bounce)
    uint32_t kk = p / P.n_pixels;
    uint32_t r = p - kk * P.n_pixels;
    int j = (int)(r / (uint32_t)P.width);
    int i = (int)(r - (uint32_t)j * (uint32_t)P.width);
    RngKey key;
    int n = path_key(P, kk, r, key);
    f3 o, d;
    float time;
    camera_ray(cam, pattern, i, j, n, P.width, P.height, key, o, d, time);
    QP(pool, 0, slot) = make_float4(o.x, o.y, o.z, time);
    QP(pool, 1, slot) = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)P.max_depth | (kk << DEPTH_COLUMN_SHIFT)));
    QP(pool, 2, slot) = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(r));
}
// the same into registers (planes 0-2 of the slot)
__device__ __forceinline__ void start_path_regs(const RtCameraDev& cam, const float* pattern, const RtRenderParams& P, uint32_t p,
                                                float4& w0, float4& w1, float4& w2) {
    uint32_t kk = p / P.n_pixels;
    uint32_t r = p - kk * P.n_pixels;
    int j = (int)(r / (uint32_t)P.width);
    int i = (int)(r - (uint32_t)j * (uint32_t)P.width);
    RngKey key;
    int n = path_key(P, kk, r, key);
    f3 o, d;
    float time;
    camera_ray(cam, pattern, i, j, n, P.width, P.height, key, o, d, time);
    w0 = make_float4(o.x, o.y, o.z, time);
    w1 = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)P.max_depth | (kk << DEPTH_COLUMN_SHIFT)));
    w2 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(r));
}

// the first n_slots paths of the batch go to the slots directly (path p = slot)
__global__ void __launch_bounds__(256) rt_init_kernel(RtCameraDev cam, const float* __restrict__ pattern, RtRenderParams P,
                                                      PathPool pool) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= pool.n_slots) return;
    float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    QP(pool, 3, slot) = z;
    if (slot < P.n_paths && P.max_depth > 0) {
        start_path(cam, pattern, P, pool, slot, slot);
    } else {
        QP(pool, 0, slot) = z;
        QP(pool, 1, slot) = z; // depth 0: no ray
        QP(pool, 2, slot) = z;
    }
}

// Closest surface of every slot's ray.  Persistent waves; each lane is a small state machine:
//   * it steps through BVH nodes (hot loop) and collects the leaves it meets in a three-entry queue, so that it
//     keeps stepping (with a t_max that is at most three leaves stale) instead of stopping at every leaf;
//   * queued leaves are tested in the leaf block, which the wave runs when enough lanes hold leaves or too many
//     lanes cannot step any more (queue full, tree finished with leaves pending, nested tree waiting to be entered);
//   * a lane whose walk is over publishes its hit and fetches the next slot in the retire/fetch block.
// Which block runs is a wave-wide vote.  Constant media are left to rt_shade_kernel.  NESTED: the scene holds nested
// trees (Transform{BVHTree}); without them the state for the outer ray is not carried.
#define LEAF_QUEUE 3u
#ifndef PAIR_LEAVES
#define PAIR_LEAVES 3u // leaves a lane can have waiting in the walk over pair nodes (a step adds up to two)
#endif
// CL: the classes of surface primitives the scene holds (RT_CL_*; the leaf block carries only their code).  BLOCKS:
// workgroups per CU the kernel is compiled for (registers per lane = 512 / BLOCKS, rounded down to a multiple of 8).
// STACK: entries of a lane's stack in the walk over pair nodes (the tree must be no deeper): with 16 instead of
// RT_PAIR_STACK the two stacks take 19 KB of LDS per workgroup and eight workgroups fit a CU instead of six.
template <bool COUNT, bool NESTED, bool PAIRS, uint32_t CL = RT_CL_ALL, int BLOCKS = (NESTED ? 4 : 6), uint32_t STACK = RT_PAIR_STACK>
__global__ void __launch_bounds__(256, BLOCKS) rt_extend_kernel(DevScene S, RtRenderParams P, PathPool pool, uint32_t* __restrict__ head,
                                                        unsigned long long* __restrict__ counters) {
    const uint32_t lane = threadIdx.x & 63u;
    const float t_min = 0.001f;
    Counters cnt = {0, 0, 0, 0, 0, 0};
    uint32_t cur = 0, cur_end = 0; // wave-uniform cursor into the chunk of slots this wave owns
    const uint32_t slot_share = ((pool.n_slots + RT_SLOT_DISP - 1u) / RT_SLOT_DISP + EXTEND_CHUNK - 1u) / EXTEND_CHUNK * EXTEND_CHUNK;
    const uint32_t disp_first = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % RT_SLOT_DISP;
    uint32_t disp_tries = 0;
    bool exhausted = false;
    // A lane without a ray has slot == NO_SLOT, an empty leaf queue and node == end: it looks like a lane whose walk is
    // over, so the votes below need no test of their own for it.
    constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
    uint32_t slot = NO_SLOT;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);
    BoxRay R; // the ray in the form the box test wants
    box_frame(S, o, d, R);
    float time = 0.0f, t_max = INF;
    f3 so = o, sd = d; // outer ray while inside a nested tree
    uint32_t ret = RT_NODE_NONE, ret_end = 0, inst_rank = 0, inst_xf = 0;
    uint32_t node = 0, end = 0; // next node; the walk of the current tree is over at node == end
    uint32_t set = 0;           // node_set() of the ray (only kept where nested trees are entered)
    uint32_t q0 = 0, q1 = 0, q2 = 0, nq = 0; // leaf queue: q0 newest; entry nq - 1 is the oldest
    // PAIRS (scenes without nested trees whose tree is no deeper than RT_PAIR_STACK): the walk reads pair nodes -- the
    // boxes of both children of an interior node in one record -- and keeps the far child on a stack in LDS while it is in
    // the near one's subtree: a third of the dependent fetches of the stackless walk (no fetch for a leaf, none for a
    // box that is missed).  `node` is the record to read next, 0xFFFFFFFF (= `end`) when the walk is over.  The leaves
    // found wait on a second small stack in LDS (a step can find two): in which order a lane's leaves are tested does
    // not change its closest hit.  Both stacks take their stores unconditionally at the top; only the counts say what stays.
    __shared__ uint32_t pair_stack[PAIRS ? STACK : 1u][256];
    __shared__ uint32_t leaf_stack[PAIRS ? PAIR_LEAVES : 1u][256];
    // (not in the variant that also carries the triangle code: there the stage's live values no longer fit the registers)
    constexpr bool SPREAD_BOXES = PAIRS && (CL & RT_CL_BOX) != 0u && (CL & RT_CL_TRI) == 0u;
    __shared__ __attribute__((aligned(16))) float box_data[SPREAD_BOXES ? 4u : 1u][SPREAD_BOXES ? 10u : 1u][20]; // per wave: the boxes a leaf-block run spreads over its lanes (16 words each, four of padding against bank conflicts)
    uint32_t sp = 0;
    Closest best = {INF, 0xFFFFFFFFu, 0u, 0u};
    // PACKED (the walk over pair nodes: no nested trees): the closest hit is kept as one word beside t_max (consider_packed)
    constexpr bool PACKED = PAIRS && !NESTED;
    uint32_t held = 0xFFFFFFFFu;
    for (;;) {
        // ---- end of a nested tree with nothing pending: back to the outer ray (cheap, no vote needed)
        if (NESTED && nq == 0u && node == end && ret != RT_NODE_NONE) {
            o = so; d = sd;
            box_frame(S, o, d, R);
            node = ret; end = ret_end;
            ret = RT_NODE_NONE;
            inst_rank = 0; inst_xf = 0;
        }
        // the votes: each comparison is voted on by itself (its lane mask as the comparison leaves it) and the masks are
        // combined by scalar instructions; a vote on a combined condition would cost two vector instructions more
        const bool over = nq == 0u && node == end;  // the walk is over, or there is no ray
        const bool fin = over && slot != NO_SLOT;
        const bool wl = nq != 0u;
        const bool ws = node != end && (PAIRS ? nq <= PAIR_LEAVES - 2u : nq < LEAF_QUEUE) && !(NESTED && nq != 0u && (q0 & RT_LEAF_INSTANCE));
        const unsigned long long m_nq0 = __builtin_amdgcn_ballot_w64(nq == 0u), m_end = __builtin_amdgcn_ballot_w64(node == end);
        const unsigned long long m_room = __builtin_amdgcn_ballot_w64(PAIRS ? nq <= PAIR_LEAVES - 2u : nq < LEAF_QUEUE);
        const unsigned long long m_enter = NESTED ? __builtin_amdgcn_ballot_w64((q0 & RT_LEAF_INSTANCE) != 0u) & ~m_nq0 : 0ull;
        const unsigned long long m_wl = ~m_nq0, m_ws = ~m_end & m_room & ~m_enter;
        const uint32_t n_need = (uint32_t)__popcll(m_nq0 & m_end);
        const uint32_t nl = (uint32_t)__popcll(m_wl);
        const uint32_t ns = (uint32_t)__popcll(m_ws);

        // ---- retire finished rays and fetch new ones once enough lanes are idle
        if ((n_need >= P.fetch_threshold && !exhausted) || (nl | ns) == 0u) {
            if (COUNT) cnt.bounce_waves += first_active_lane();
            if (fin) {
                // publish the hit for the shade kernel (the material type is bits 29-31 of sx)
                if (PACKED) {
                    best.t = INF; best.prim = 0xFFFFFFFFu; best.sx = 0u; best.rank = 0u;
                    if (held != 0xFFFFFFFFu) {
                        best.t = t_max;
                        best.prim = held & RT_LEAF_PRIM_MASK;
                        const uint4 hd = S.prims[3u * best.prim];
                        best.sx = (held >> 28) | (((hd.x & 15u) == DP_TRIANGLE ? 0u : hd.x >> 8) << 3) | ((hd.x & DP_TYPE_MASK) << 24);
                        best.rank = hd.z;
                    }
                }
                QP(pool, 3, slot) = make_float4(best.t, __uint_as_float(best.prim), __uint_as_float(best.sx), __uint_as_float(best.rank));
                if (COUNT) cnt.rays++;
                slot = NO_SLOT;
            }
            if (!exhausted) {
                unsigned long long m = __builtin_amdgcn_ballot_w64(slot == NO_SLOT);
                if (cur >= cur_end) { // take the next chunk of slots
                    // RT_SLOT_DISP counters on separate cache lines, each handing out one contiguous share of the slots
                    // (a single counter serialises thousands of atomics per launch); a wave starts at "its" counter and
                    // moves on to the next few when that share is used up
                    uint32_t base = 0xFFFFFFFFu, limit = 0u;
                    if (lane == 0u) {
                        for (; disp_tries < 4u; ++disp_tries) {
                            const uint32_t c = (disp_first + disp_tries) % RT_SLOT_DISP;
                            const uint32_t lo = c * slot_share;
                            if (lo >= pool.n_slots) continue;
                            const uint32_t size = min(slot_share, pool.n_slots - lo);
                            uint32_t* ctr = head + RT_SLOT_DISP_BASE + c * RT_DISP_STRIDE;
                            const uint32_t v = atomicAdd(ctr, EXTEND_CHUNK);
                            if (v >= size) continue;
                            base = lo + v;
                            limit = lo + size;
                            break;
                        }
                    }
                    // (lane 0's values for the wave, in scalar registers: the cursor into the chunk is the same in every lane)
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    limit = (uint32_t)__builtin_amdgcn_readfirstlane((int)limit);
                    disp_tries = (uint32_t)__builtin_amdgcn_readfirstlane((int)disp_tries);
                    if (base == 0xFFFFFFFFu) { exhausted = true; cur = cur_end = 0; }
                    else { cur = base; cur_end = min(base + EXTEND_CHUNK, limit); }
                }
                if (!exhausted) {
                    if (slot == NO_SLOT) {
                        uint32_t sidx = cur + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (sidx < cur_end) {
                            float4 p1 = QP(pool, 1, sidx), p0 = QP(pool, 0, sidx);
                            asm volatile("" : "+v"(p0.x), "+v"(p1.x)); // both loads in flight before the depth is looked at
                            if (((__float_as_uint(p1.w) & (DEPTH_MASK | DEPTH_HOLD)) - 1u) < DEPTH_MASK) { // holds a ray, not on hold
                                slot = sidx;
                                o = mk3(p0.x, p0.y, p0.z); d = mk3(p1.x, p1.y, p1.z);
                                box_frame(S, o, d, R);
                                time = p0.w;
                                t_max = INF;
                                set = node_set(S, o);
                                if (PAIRS) {
                                    const uint32_t octant = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
                                    node = ((set ? 8u : 0u) + octant) * S.n_pairs;
                                    end = 0xFFFFFFFFu;
                                    sp = 0u;
                                } else {
                                    node = S.surface_root + layout_base(S, d) + set;
                                    end = node + S.surface_size;
                                }
                                nq = 0u;
                                ret = RT_NODE_NONE;
                                inst_rank = 0; inst_xf = 0;
                                best.t = INF; best.prim = 0xFFFFFFFFu; best.sx = 0u; best.rank = 0u;
                                held = 0xFFFFFFFFu;
                            }
                        }
                    }
                    cur = min(cur + (uint32_t)__popcll(m), cur_end);
                }
            }
            if (exhausted && __builtin_amdgcn_ballot_w64(slot != NO_SLOT) == 0ull) break;
            continue;
        }

        // ---- queued leaves: once enough lanes hold one, or too many lanes cannot step before theirs are tested
        const uint32_t n_blocked = (uint32_t)__popcll(m_wl & ~m_ws);
        if (nl >= P.leaf_threshold || n_blocked >= P.blocked_threshold || ns == 0u) {
            // SPREAD: the boxes of this run, if they are few, are not tested by their own lanes one side after the other
            // (six exact rectangle tests for a handful of lanes while the rest of the wave waits) but by six lanes each,
            // any lanes: lane j of the wave takes side j % 6 of box j / 6, gets the box and its lane's ray through
            // ds_bpermute, runs the one rectangle test, and the box's lane reads the six distances back and takes the
            // closest in the order of src/box.jl:43-83 (strict <, so the first of equal distances stays) -- the same six
            // results combined the same way.
            constexpr bool SPREAD = SPREAD_BOXES;
            if (COUNT && PAIRS && P.count_mode == 3u) {
                // lanes without a ray (or done with theirs) and lanes with a second leaf waiting that is not a box
                const bool idle = over;
                const bool second = wl && nq >= 2u && !(leaf_stack[nq - 2u][threadIdx.x] & RT_LEAF_BOX);
                const uint32_t ni = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(idle)), n2 = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(second));
                if (first_active_lane()) { cnt.prims += ni; cnt.nodes += n2; cnt.node_waves += min(ni, n2); cnt.prim_waves++; }
            }
            uint32_t e = 0u;
            if (wl) {
                if (PAIRS) e = leaf_stack[nq - 1u][threadIdx.x];
                else e = nq == 3u ? q2 : (nq == 2u ? q1 : q0);
                nq -= 1u;
            }
            const unsigned long long m_box = SPREAD ? __builtin_amdgcn_ballot_w64((e & RT_LEAF_BOX) != 0u) : 0ull; // (e is 0 without a leaf)
            const uint32_t n_box = (uint32_t)__popcll(m_box);
            const bool spread = SPREAD && n_box != 0u && n_box <= P.box_spread; // (wave-uniform)
            const bool box_lane = spread && wl && (e & RT_LEAF_BOX);
            const uint32_t wave = threadIdx.x >> 6;
            const uint32_t my_box = SPREAD ? __builtin_amdgcn_mbcnt_hi((uint32_t)(m_box >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_box, 0u)) : 0u;
            if (wl) {
                if (COUNT && P.count_mode == 0u) { cnt.prims++; cnt.prim_waves += first_active_lane(); }
                uint32_t prim = (e & RT_LEAF_PRIM_MASK) - 1u;
                uint4 hd;
                Geom G;
                load_prim(S, prim, hd, G);
                if (PACKED) asm volatile("" : "+v"(hd.y)); // (keeps the record's three 16-byte loads whole: split around the unused word, their registers need six copies to line up)
                if (COUNT && (P.count_mode == 1u || P.count_mode == 2u)) { // one class only: its lane tests, and the runs that hold a lane of it
                    const uint32_t kind = hd.x & 15u;
                    const bool mine = P.count_mode == 1u ? (kind >= DP_XY_RECT && kind <= DP_BOX) : (kind <= DP_MOVING_SPHERE);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) cnt.prims++;
                    if (m != 0ull && (uint32_t)(__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) == lane) cnt.prim_waves++;
                }
                float Tinv[12] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                if ((CL & RT_CL_XFORM) && (e & RT_LEAF_XFORM)) { // issued together with the loads of the primitive
                    float4 m0 = S.prim_inv[3u * prim], m1 = S.prim_inv[3u * prim + 1u], m2 = S.prim_inv[3u * prim + 2u];
                    Tinv[0] = m0.x; Tinv[1] = m0.y; Tinv[2] = m0.z; Tinv[3] = m0.w; Tinv[4] = m1.x; Tinv[5] = m1.y;
                    Tinv[6] = m1.z; Tinv[7] = m1.w; Tinv[8] = m2.x; Tinv[9] = m2.y; Tinv[10] = m2.z; Tinv[11] = m2.w;
                }
                if (NESTED && (e & RT_LEAF_INSTANCE)) {
                    // Transform{BVHTree}: src/transform.jl:104-116 around src/BVH.jl:22-53.  The flagged entry is
                    // the last one a lane queues, so the queue is empty now.
                    uint32_t field = hd.x >> 8;
                    if (!ref_reaches<true>(S, prim, 0u, o, d, t_min, INF)) continue; // the reference culls the whole nested tree
                    so = o; sd = d;
                    ret = node; ret_end = end;
                    inst_rank = hd.z;
                    inst_xf = field;
                    if (field) {
                        const float* T = S.xforms + 24u * (field - 1u) + 12;
                        f3 no = apply_point(T, o), nd = apply_vec(T, d);
                        o = no; d = nd;
                    }
                    box_frame(S, o, d, R);
                    node = __float_as_uint(G.g[0]) + layout_base(S, d) + set;
                    end = node + __float_as_uint(G.g[1]);
                } else if (SPREAD) {
                    // surface_leaf() with the ray taken into the primitive's frame once, for boxes that are spread and
                    // for everything else alike
                    const uint32_t kind = hd.x & 15u;
                    uint32_t side = 0u, xf = inst_xf;
                    float t;
                    if ((CL & RT_CL_TRI) && kind == DP_TRIANGLE) {
                        f3 nrm;
                        float u, v;
                        t = triangle_t(G.g, o, d, t_min, t_max, nrm, u, v);
                    } else {
                        f3 ro = o, rd = d;
                        const uint32_t field = hd.x >> 8;
                        if ((CL & RT_CL_XFORM) && field) { // Transform: src/transform.jl:104-105
                            ro = apply_point(Tinv, o);
                            rd = apply_vec(Tinv, d);
                            xf = field;
                        }
                        if (box_lane) {
                            // the ray in the box's frame, the box, the bound and what a hit will be recorded with,
                            // where any lane of the wave can read them
                            float4* q = reinterpret_cast<float4*>(box_data[wave][my_box]);
                            q[0] = make_float4(ro.x, ro.y, ro.z, rd.x);
                            q[1] = make_float4(rd.y, rd.z, G.g[0], G.g[1]);
                            q[2] = make_float4(G.g[2], G.g[3], G.g[4], G.g[5]);
                            q[3] = make_float4(t_max, __uint_as_float(prim), __uint_as_float((xf << 3) | ((hd.x & DP_TYPE_MASK) << 24)),
                                               __uint_as_float(hd.z + inst_rank)); // (SPREAD implies PACKED: the third word is not used)
                            t = INF; // (the spread stage below considers the box)
                        } else {
                            t = framed_t<CL>(kind, G.g, ro, rd, time, t_min, t_max, side);
                        }
                    }
                    consider_packed(S, held, t_max, t, prim, side, hd.z);
                } else if (PACKED) {
                    const uint32_t kind = hd.x & 15u;
                    uint32_t side = 0u, xf = 0u;
                    const float t = surface_t<CL>(Tinv, kind, kind == DP_TRIANGLE ? 0u : hd.x >> 8, G.g, o, d, time, t_min, t_max, side, xf);
                    consider_packed(S, held, t_max, t, prim, side, hd.z);
                } else {
                    surface_leaf<false, CL>(Tinv, prim, hd, G, o, d, time, t_min, t_max, inst_rank, inst_xf, best, 0.0f, 0u);
                }
            }
            if (SPREAD && spread) {
                const uint32_t r = (lane * 43u) >> 8, f = lane - 6u * r; // lane / 6, lane % 6
                const float4* q = reinterpret_cast<const float4*>(box_data[wave][r < n_box ? r : 0u]);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                const float tm = box_data[wave][r < n_box ? r : 0u][12];
                const float v[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
                // side f of src/box.jl:13-22: 0, 1 = the planes z = max.z, min.z over (x, y); 2, 3 = y = max.y, min.y over
                // (x, z); 4, 5 = x = max.x, min.x over (y, z) -- the arguments box_t gives rect_t for that side
                const bool pz = f < 2u, px = f >= 4u, lo = (f & 1u) != 0u;
                const float ok = pz ? v[2] : (px ? v[0] : v[1]), dk = pz ? v[5] : (px ? v[3] : v[4]);
                const float oa = px ? v[1] : v[0], da = px ? v[4] : v[3];
                const float ob = pz ? v[1] : v[2], db = pz ? v[4] : v[5];
                const float a0 = px ? v[7] : v[6], a1 = px ? v[10] : v[9];
                const float b0 = pz ? v[7] : v[8], b1 = pz ? v[10] : v[11];
                const float kk = lo ? (pz ? v[8] : (px ? v[6] : v[7])) : (pz ? v[11] : (px ? v[9] : v[10]));
                float t = rect_t(ok, dk, oa, da, ob, db, a0, a1, b0, b1, kk, t_min, tm);
                if (r >= n_box) t = INF;
                float tf[6];
                for (uint32_t k = 0; k < 6u; ++k) tf[k] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)((6u * my_box + k) << 2), __float_as_int(t)));
                if (box_lane) {
                    float bt = INF;
                    uint32_t side = 0u;
                    for (uint32_t k = 0; k < 6u; ++k) if (tf[k] < bt) { bt = tf[k]; side = k; }
                    const float4 w = reinterpret_cast<const float4*>(box_data[wave][my_box])[3];
                    consider_packed(S, held, t_max, bt, __float_as_uint(w.y), side, __float_as_uint(w.w));
                }
            }
            continue;
        }

        // ---- hot loop: step through nodes until `step_dropout` more lanes have left it
        {
            const uint32_t floor_lanes = ns > P.step_dropout ? ns - P.step_dropout : 1u;
            // t_max does not change in this loop: widened once like the far distance of a box (node_test), and held
            // against the entry distance by a comparison of its own
            const float t_far = __builtin_fmaf(fabsf(t_max), 4.8e-7f, t_max);
            uint32_t n_step;
            bool stepping = ws;
            do {
                const unsigned long long m_idle = COUNT ? __builtin_amdgcn_ballot_w64(slot == NO_SLOT) : 0ull;
                if (COUNT && PAIRS && P.count_mode == 4u && lane == 0u) {
                    // lanes without a ray per step iteration, before and after the wave has run out of slots to fetch
                    const uint32_t idle = (uint32_t)__popcll(m_idle);
                    if (exhausted) { cnt.nodes += idle; cnt.prim_waves++; } else { cnt.prims += idle; cnt.node_waves++; }
                }
                if (PAIRS && stepping) {
                    if (COUNT && P.count_mode < 3u) { cnt.nodes += 2; cnt.node_waves += first_active_lane(); }
                    const uint4* rec = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(S.pairs) + (node << 5)); // (32-bit offset: fewer than 2^27 records)
                    const uint4 qa = rec[0], qb = rec[1];
                    const uint32_t top = pair_stack[sp ? sp - 1u : 0u][threadIdx.x]; // (wanted only if this step ends a subtree)
                    bool hit2[2];
                    for (int c = 0; c < 2; ++c) { // node_test() with t_max taken out of the min, for either child
                        const uint4 q = c ? qb : qa;
                        const v2h hx = __builtin_bit_cast(v2h, q.x), hy = __builtin_bit_cast(v2h, q.y), hz = __builtin_bit_cast(v2h, q.z);
                        const float ax = __builtin_fmaf((float)hx.x, R.i.x, R.cn.x), fx = __builtin_fmaf((float)hx.y, R.i.x, R.cf.x);
                        const float ay = __builtin_fmaf((float)hy.x, R.i.y, R.cn.y), fy = __builtin_fmaf((float)hy.y, R.i.y, R.cf.y);
                        const float az = __builtin_fmaf((float)hz.x, R.i.z, R.cn.z), fz = __builtin_fmaf((float)hz.y, R.i.z, R.cf.z);
                        const float tnear = fmaxf(fmaxf(ax, ay), fmaxf(az, t_min));
                        float tfar = fminf(fminf(fx, fy), fz);
                        tfar = __builtin_fmaf(fabsf(tfar), 4.8e-7f, tfar);
                        hit2[c] = tnear <= tfar && tnear <= t_far;
                    }
                    const bool leaf_a = (int32_t)qa.w < 0, leaf_b = (int32_t)qb.w < 0;
                    leaf_stack[nq][threadIdx.x] = qb.w; // the far leaf below the near one: the near one is tested first
                    nq += (hit2[1] && leaf_b) ? 1u : 0u;
                    leaf_stack[nq][threadIdx.x] = qa.w;
                    nq += (hit2[0] && leaf_a) ? 1u : 0u;
                    const bool go_a = hit2[0] && !leaf_a, go_b = hit2[1] && !leaf_b;
                    pair_stack[sp][threadIdx.x] = qb.w; // (sp <= depth of the record read < RT_PAIR_STACK)
                    {   // next record: the near child's, else the far child's, else the top of the stack, else none (all ones).
                        // Written on lane masks so that it stays straight-line code.
                        const uint32_t ga = go_a ? 0xFFFFFFFFu : 0u, gb = go_b ? 0xFFFFFFFFu : 0u, has = sp ? 0xFFFFFFFFu : 0u;
                        uint32_t nxt = (has & top) | ~has;
                        nxt = (gb & qb.w) | (~gb & nxt);
                        node = (ga & qa.w) | (~ga & nxt);
                        sp = sp + (ga & gb & 1u) - (~(ga | gb) & has & 1u);
                    }
                }
                if (!PAIRS && stepping) {
                    if (COUNT) { cnt.nodes++; cnt.node_waves += first_active_lane(); }
                    // node_test() with t_max taken out of the min
                    const uint4 q = S.nodes[node];
                    const v2h hx = __builtin_bit_cast(v2h, q.x), hy = __builtin_bit_cast(v2h, q.y), hz = __builtin_bit_cast(v2h, q.z);
                    const float ax = __builtin_fmaf((float)hx.x, R.i.x, R.cn.x), fx = __builtin_fmaf((float)hx.y, R.i.x, R.cf.x);
                    const float ay = __builtin_fmaf((float)hy.x, R.i.y, R.cn.y), fy = __builtin_fmaf((float)hy.y, R.i.y, R.cf.y);
                    const float az = __builtin_fmaf((float)hz.x, R.i.z, R.cn.z), fz = __builtin_fmaf((float)hz.y, R.i.z, R.cf.z);
                    const float tnear = fmaxf(fmaxf(ax, ay), fmaxf(az, t_min));
                    float tfar = fminf(fminf(fx, fy), fz);
                    tfar = __builtin_fmaf(fabsf(tfar), 4.8e-7f, tfar);
                    const bool hit = tnear <= tfar && tnear <= t_far;
                    const uint32_t link = q.w;
                    // branch-free bookkeeping: where the walk goes on, and the leaf queue (a shift register)
                    const bool is_leaf = (int32_t)link < 0;
                    const bool push = hit && is_leaf;
                    node = (hit || is_leaf) ? node + 1u : link;
                    q2 = push ? q1 : q2;
                    q1 = push ? q0 : q1;
                    q0 = push ? (link & ~RT_NODE_LEAF) : q0;
                    nq += push ? 1u : 0u;
                }
                // from the lane's state (the same test as `ws` above), each comparison voted on its own: the votes are
                // then the comparisons' lane masks themselves, combined by scalar instructions
                const bool more = node != end, room = PAIRS ? nq <= PAIR_LEAVES - 2u : nq < LEAF_QUEUE;
                const bool enter = NESTED && nq != 0u && (q0 & RT_LEAF_INSTANCE);
                stepping = more && room && !enter;
                n_step = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(more) & __builtin_amdgcn_ballot_w64(room) &
                                            ~(NESTED ? __builtin_amdgcn_ballot_w64(enter) : 0ull));
            } while (n_step >= floor_lanes);
        }
    }
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)cnt.rays);
        atomicAdd(&counters[1], (unsigned long long)cnt.nodes);
        atomicAdd(&counters[2], (unsigned long long)cnt.prims);
        atomicAdd(&counters[3], (unsigned long long)cnt.node_waves);
        atomicAdd(&counters[4], (unsigned long long)cnt.prim_waves);
        atomicAdd(&counters[5], (unsigned long long)cnt.bounce_waves);
    }
}

// The slots that rt_shade_kernel put on hold: the closest surface rt_extend_kernel found for their ray does not stand
// (hit_stands: the reference does not reach it, or it lies in front of its own box).  settle_hit finds the hit the
// reference ends up with; the hit record is marked settled, and the next rt_shade_kernel shades it.
// Runs once per batch of bounces: a slot may wait a few launches, which changes nothing for its path.
// sync[RT_REDO_COUNT]: number of entries in pool.redo, reset by the last workgroup to finish (sync[RT_REDO_COUNT + 1]
// counts the finished ones).
__global__ void __launch_bounds__(256) rt_redo_kernel(DevScene S, RtRenderParams P, PathPool pool, uint32_t* __restrict__ sync) {
    __shared__ Closest settle_list[256][SETTLE_LIST]; // per lane: the list front_sequence keeps
    const uint32_t n = __hip_atomic_load(sync + RT_REDO_COUNT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The slots are dealt to the waves first and to the lanes of a wave last: every slot takes its own way through the
    // loops below, and the lanes of one wave would run those ways one after the other.
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (uint32_t i = (threadIdx.x & 63u) * n_waves + wave; i < n; i += 64u * n_waves) {
        const uint32_t slot = pool.redo[i];
        const float4 p0 = QP(pool, 0, slot), p1 = QP(pool, 1, slot), h = QP(pool, 3, slot);
        const f3 o = mk3(p0.x, p0.y, p0.z), d = mk3(p1.x, p1.y, p1.z);
        Closest best;
        best.t = h.x; best.prim = __float_as_uint(h.y); best.sx = __float_as_uint(h.z); best.rank = __float_as_uint(h.w);
        // the hit written back is the final one, media included
        RngKey key;
        path_key(P, (__float_as_uint(p1.w) >> DEPTH_COLUMN_SHIFT) & DEPTH_COLUMN_MASK, __float_as_uint(QP(pool, 2, slot).w), key);
        key.bounce = (uint32_t)P.max_depth - (__float_as_uint(p1.w) & DEPTH_MASK);
        settle_hit(S, o, d, p0.w, 0.001f, INF, key, best, settle_list[threadIdx.x]);
        QP(pool, 3, slot) = make_float4(best.t, __uint_as_float(best.prim), __uint_as_float(best.sx), __uint_as_float(best.rank | RANK_SETTLED));
    }
    __syncthreads();
    if (threadIdx.x == 0u && n != 0u) {
        __threadfence();
        if (atomicAdd(sync + RT_REDO_COUNT + 1u, 1u) == gridDim.x - 1u) {
            sync[RT_REDO_COUNT] = 0u;
            sync[RT_REDO_COUNT + 1u] = 0u;
        }
    }
}

#ifndef SHADE_BLOCK
#define SHADE_BLOCK 256
#endif
// sync[RT_SLOT_DISP_BASE + c * DISP_STRIDE]: slot dispenser c of rt_extend_kernel (reset here), sync[DISP_BASE + c * DISP_STRIDE]: path dispenser c,
// active: set to 1 if any slot holds a ray after this bounce
template <bool TEXTURED>
__global__ void __launch_bounds__(SHADE_BLOCK) rt_shade_kernel(DevScene S, RtCameraDev cam, const float* __restrict__ pattern,
                                                               RtRenderParams P, PathPool pool, float4* __restrict__ radiance,
                                                               uint32_t* __restrict__ sync, uint32_t* __restrict__ active) {
    __shared__ uint32_t hist[8], cursor[8];
    __shared__ uint16_t order[SHADE_BLOCK];
    __shared__ float4 hits[SHADE_BLOCK];  // per slot of the block: {t, prim, side|xform, type} after the media
    __shared__ float4 ray0[SHADE_BLOCK], ray1[SHADE_BLOCK], ray2[SHADE_BLOCK]; // planes 0-2 of the block's slots, for phase 2
    __shared__ uint32_t block_alive, block_need, block_filled, block_nseg;
    __shared__ uint32_t hist2[66], cursor2[66], block_pinned;
    __shared__ uint32_t seg_first[RT_DISP_MAX + 1], seg_base[RT_DISP_MAX + 1], seg_disp[RT_DISP_MAX + 1]; // the paths the block's finished slots take
    const uint32_t t = threadIdx.x;
    const uint32_t base = blockIdx.x * SHADE_BLOCK;
    if (t < 8u) { hist[t] = 0u; }
    if (t == 0u) { block_alive = 0u; block_need = 0u; block_pinned = 0u; }
    if (t < 66u) hist2[t] = 0u;
    if (blockIdx.x == 0 && t < RT_SLOT_DISP) sync[RT_SLOT_DISP_BASE + t * RT_DISP_STRIDE] = 0u; // re-arm the slot dispensers
    __syncthreads();
    // Phase 1, thread t on slot base + t: the constant media of the scene against the slot's ray (every lane of the
    // wave has a ray, so the medium test runs converged), then the material type of the final hit; 7 = nothing to shade
    uint32_t my_key = 7u;
    if (base + t < pool.n_slots) {
        float4 q1 = QP(pool, 1, base + t), q2 = QP(pool, 2, base + t), q4 = QP(pool, 3, base + t);
        int depth0 = (int)(__float_as_uint(q1.w) & DEPTH_MASK);
        ray1[t] = q1; ray2[t] = q2;
        if (depth0 > 0) {
            const float4 q0 = QP(pool, 0, base + t);
            ray0[t] = q0;
            const f3 o = mk3(q0.x, q0.y, q0.z), d = mk3(q1.x, q1.y, q1.z);
            // rt_extend_kernel found the closest surface of the device tree.  Here, with every lane holding a ray, that
            // hit is held against the boxes of the reference's trees (hit_stands).  In the rare case that it does not
            // stand the slot is put on hold: rt_redo_kernel finds the hit that does, and the slot is shaded a few
            // launches later.
            Closest best;
            best.t = q4.x; best.prim = __float_as_uint(q4.y); best.sx = __float_as_uint(q4.z);
            best.rank = __float_as_uint(q4.w) & ~RANK_SETTLED;
            const bool held = (__float_as_uint(q1.w) & DEPTH_HOLD) != 0u;
            bool hold = false, waiting = false;
            if (held) {
                // settled by rt_redo_kernel (its hit is the final one, media included), or still waiting for it
                waiting = !(__float_as_uint(q4.w) & RANK_SETTLED);
            } else if (best.t != INF && !hit_stands(S, best.prim, (best.sx >> 3) & 0x3FFFFFFu, o, d, 0.001f, INF, best.t)) {
                hold = true;
            } else if (S.n_media != 0u) {
                RngKey key;
                path_key(P, (__float_as_uint(q1.w) >> DEPTH_COLUMN_SHIFT) & DEPTH_COLUMN_MASK, __float_as_uint(q2.w), key);
                key.bounce = (uint32_t)(P.max_depth - depth0);
                hold = !media_closest(S, o, d, q0.w, 0.001f, INF, key, best); // (a medium hit in front of its own box)
            }
            if (hold) {
                // plane 3 keeps {t, prim, sx, rank} of the closest surface of the device tree: where the redo starts from
                reinterpret_cast<uint32_t*>(&QP(pool, 1, base + t))[3] = __float_as_uint(q1.w) | DEPTH_HOLD;
                pool.redo[atomicAdd(sync + RT_REDO_COUNT, 1u)] = base + t;
            }
            if (hold || waiting) q4.w = __uint_as_float(HIT_WAIT);
            else q4 = make_float4(best.t, __uint_as_float(best.prim), __uint_as_float(best.sx),
                                  __uint_as_float(best.t != INF ? best.sx >> 29 : HIT_MISS));
            hits[t] = q4;
            my_key = __float_as_uint(q4.w) & 7u;
        }
    }
    uint32_t mine = t;
    if (P.regroup & 1u) {
        // counting sort of the block's slots by material type: afterwards thread t shades slot order[t], and
        // the threads of a wave (almost always) hold the same material
        atomicAdd(&hist[my_key], 1u);
        __syncthreads();
        if (t == 0u) {
            uint32_t acc = 0u;
            for (int q = 0; q < 8; ++q) { cursor[q] = acc; acc += hist[q]; }
        }
        __syncthreads();
        uint32_t pos = atomicAdd(&cursor[my_key], 1u);
        order[pos] = (uint16_t)t;
    }
    __syncthreads();
    if (P.regroup & 1u) mine = order[t];
    // Phase 2, thread t on slot base + mine: the body of trace()'s loop after the hit
    uint32_t slot = base + mine;
    bool valid = slot < pool.n_slots;
    float4 q1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q0 = q1, q2 = q1;
    if (valid) { q1 = ray1[mine]; q0 = ray0[mine]; q2 = ray2[mine]; } // ray0 is only defined (and only used) for slots that hold a ray
    const uint32_t depth_word = __float_as_uint(q1.w);
    int depth = (int)(depth_word & DEPTH_MASK);
    bool done = false, cont = false, keep = false, wcolor = false;
    float4 w0 = q1, w1 = q1, w2 = q1, w4 = q1; // (q1 is zero here for an invalid slot; overwritten below otherwise)
    w1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (valid && depth > 0 && (__float_as_uint(hits[mine].w) & 7u) == HIT_WAIT) {
        cont = true; // on hold: the slot keeps its ray
        keep = true;
        block_pinned = 1u;
    } else if (valid && depth > 0) {
        float4 q4 = hits[mine];
        const uint32_t kk = (depth_word >> DEPTH_COLUMN_SHIFT) & DEPTH_COLUMN_MASK, r = __float_as_uint(q2.w);
        RngKey key;
        path_key(P, kk, r, key);
        key.bounce = (uint32_t)(P.max_depth - depth);
        PathState ps;
        ps.acc = mk3(q2.x, q2.y, q2.z);
        ps.color = mk3(0.0f, 0.0f, 0.0f);
        if (depth_word & DEPTH_HAS_COLOR) {
            float4 q3 = QP(pool, 4, slot);
            ps.color = mk3(q3.x, q3.y, q3.z);
        }
        ps.depth = depth;
        Closest best;
        best.t = q4.x; best.prim = __float_as_uint(q4.y); best.sx = __float_as_uint(q4.z); best.rank = 0u;
        f3 result, no, nd;
        done = shade<TEXTURED>(S, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), q0.w, best, ps,
                     mk3(P.background[0], P.background[1], P.background[2]), key, result, no, nd);
        if (done) {
            {   // written once, read once at the end of the batch: kept out of the caches the path pool lives in
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f rv = {result.x, result.y, result.z, 0.0f};
                __builtin_nontemporal_store(rv, reinterpret_cast<v4f*>(&radiance[(size_t)kk * P.n_pixels + r]));
            }
        } else {
            cont = true;
            // the colour is +0 until a path ends in every scene the reference can describe; it only takes memory
            // traffic when it is not (non-finite throughput)
            bool has_color = (__float_as_uint(ps.color.x) | __float_as_uint(ps.color.y) | __float_as_uint(ps.color.z)) != 0u;
            w0 = make_float4(no.x, no.y, no.z, q0.w);
            w1 = make_float4(nd.x, nd.y, nd.z, __uint_as_float((uint32_t)ps.depth | (kk << DEPTH_COLUMN_SHIFT) | (has_color ? DEPTH_HAS_COLOR : 0u)));
            w2 = make_float4(ps.acc.x, ps.acc.y, ps.acc.z, q2.w);
            wcolor = has_color;
            if (has_color) w4 = make_float4(ps.color.x, ps.color.y, ps.color.z, 0.0f);
        }
    }
    // Where the slot's new contents go (P.regroup & 2): the rays that go on change places inside the workgroup's 256
    // slots, ordered by the octant of their direction and then by the octant of their origin about the scene's middle,
    // 64 bins.  The waves of rt_extend_kernel take consecutive slots: theirs then walk one or two of the eight layouts
    // of the tree from nearby points, and fetch the same nodes (-5 % on that kernel; a wider window than 256 slots
    // adds nothing).  The slots whose path ended form a bin of their own (the paths they start are camera rays).  A
    // workgroup with a slot on hold (rt_redo_kernel knows it by its number) or with fewer than 256 slots leaves every
    // ray where it is.  Which slot a ray lives in has no bearing on its path.
    uint32_t key2 = 65u;
    if (P.regroup & 2u) {
        if (done) key2 = 64u;
        else if (cont) {
            const uint32_t oct = (__float_as_uint(w1.x) >> 31) | ((__float_as_uint(w1.y) >> 31) << 1) | ((__float_as_uint(w1.z) >> 31) << 2);
            const uint32_t cell = (w0.x > S.middle[0] ? 1u : 0u) | (w0.y > S.middle[1] ? 2u : 0u) | (w0.z > S.middle[2] ? 4u : 0u);
            key2 = (oct << 3) | cell;
        }
        atomicAdd(&hist2[key2], 1u);
    }
    // slots whose path ended take the next unstarted paths: one atomic per workgroup
    unsigned long long m_done = __ballot(done);
    uint32_t wave_pos = 0u;
    if (m_done) {
        if ((t & 63u) == 0u) wave_pos = atomicAdd(&block_need, (uint32_t)__popcll(m_done));
        wave_pos = (uint32_t)__shfl((int)wave_pos, 0);
    }
    __syncthreads();
    if (t < 64u) {
        // The unstarted paths are handed out by P.n_disp counters, each 128 bytes from the next, so that the requests of
        // the workgroups do not queue up on one address.  A workgroup asks its own counter (one lane, one atomic).  Once
        // that one cannot serve the whole request the first wave looks at every counter, a lane each, and the request is
        // served from as many of them as it takes (segments): a slot is left without a ray only when EVERY counter has
        // run dry, so no path can stay behind however the slots are spread over the workgroups (the end-of-frame
        // compaction moves them).  A counter is left alone once it has passed its limit, so it cannot wrap.
        const uint32_t need = block_need;
        uint32_t filled = 0u, nseg = 0u;
        if (need) {
            const uint32_t c = (blockIdx.x + t) % P.n_disp;
            const uint32_t lim = dispenser_limit(P, c);
            uint32_t* ctr = sync + DISP_BASE + c * DISP_STRIDE;
            unsigned long long open = 1ull; // the workgroup's own counter first, unseen
            for (bool first = true; open != 0ull && filled < need; first = false) {
                const uint32_t l = (uint32_t)__ffsll((long long)open) - 1u;
                open &= open - 1ull;
                uint32_t v = 0xFFFFFFFFu;
                if (t == l && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lim) v = atomicAdd(ctr, need - filled);
                v = (uint32_t)__shfl((int)v, (int)l);
                const uint32_t vl = (uint32_t)__shfl((int)lim, (int)l), vc = (uint32_t)__shfl((int)c, (int)l);
                if (v < vl) {
                    const uint32_t got = min(need - filled, vl - v);
                    if (t == 0u) { seg_first[nseg] = filled; seg_base[nseg] = v; seg_disp[nseg] = vc; }
                    nseg += 1u;
                    filled += got;
                }
                if (first && filled < need) // the other counters that still have something, by the look of them
                    open = __ballot(t != 0u && t < P.n_disp && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lim);
            }
            if (t == 0u && filled < need) sync[RT_TAIL_FLAG] = 1u; // every counter is dry now; every writer stores the same value
        }
        if (t == 0u) { block_filled = filled; block_nseg = nseg; }
        if (P.regroup & 2u) { // first slot of every bin: a prefix sum over the wave, then the two bins without a ray
            const uint32_t h = hist2[t];
            uint32_t inc = h;
            for (uint32_t k = 1u; k < 64u; k <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, k);
                if (t >= k) inc += up;
            }
            cursor2[t] = inc - h;
            if (t == 63u) { cursor2[64] = inc; cursor2[65] = inc + hist2[64]; }
        }
    }
    __syncthreads();
    if (done) {
        bool started = false;
        const uint32_t idx = wave_pos + (uint32_t)__popcll(m_done & ((1ull << (t & 63u)) - 1ull));
        if (idx < block_filled) {
            uint32_t sg = 0u;
            while (sg + 1u < block_nseg && seg_first[sg + 1u] <= idx) ++sg;
            const uint32_t p = dispensed_path(P, seg_disp[sg], seg_base[sg] + (idx - seg_first[sg]));
            if (p < P.n_paths) { // (always: the counters hand out valid indices only)
                start_path_regs(cam, pattern, P, p, w0, w1, w2);
                cont = true;
                started = true;
            }
        }
        (void)started; // (w1 stays zero: no ray left for this slot)
    }
    const bool bin = (P.regroup & 2u) != 0u && base + SHADE_BLOCK <= pool.n_slots && block_pinned == 0u;
    uint32_t target = slot;
    if (bin) target = base + atomicAdd(&cursor2[key2], 1u);
    if (valid && !keep) {
        if (cont) {
            QP(pool, 0, target) = w0;
            QP(pool, 1, target) = w1;
            QP(pool, 2, target) = w2;
            if (wcolor) QP(pool, 4, target) = w4;
        } else if (bin || done) {
            QP(pool, 1, target) = w1; // (zero) no ray in this slot
        }
    }
    // number of slots that still hold a ray: the host stops launching bounces when it reaches zero
    unsigned long long m = __ballot(cont);
    if ((t & 63u) == 0u && m) atomicAdd(&block_alive, (uint32_t)__popcll(m));
    __syncthreads();
    if (t == 0u && block_alive) *active = 1u; // every writer stores the same value: no atomic needed
}

// ---- compaction of the path pool at the end of a frame: count the slots that hold a ray per 256, prefix sums, move
__global__ void __launch_bounds__(256) rt_compact_count_kernel(PathPool pool, uint32_t* __restrict__ counts) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const bool holds = s < pool.n_slots && (__float_as_uint(QP(pool, 1, s).w) & DEPTH_MASK) != 0u;
    __shared__ uint32_t n;
    if (threadIdx.x == 0u) n = 0u;
    __syncthreads();
    const unsigned long long m = __ballot(holds);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&n, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0u) counts[blockIdx.x] = n;
}
// one workgroup: counts[0 .. n_blocks) -> exclusive prefix sums in place, the total to counts[n_blocks] and the sync word;
// the slots of `to` from the total up to the next multiple of 1024 are marked empty
__global__ void __launch_bounds__(1024) rt_compact_scan_kernel(uint32_t* __restrict__ counts, uint32_t n_blocks, float4* __restrict__ to,
                                                               uint32_t n_slots, uint32_t* __restrict__ sync) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t lo = t * per, hi = min(lo + per, n_blocks);
    uint32_t sum = 0u;
    for (uint32_t k = lo; k < hi; ++k) sum += counts[k];
    part[t] = sum;
    __syncthreads();
    if (t == 0u) {
        uint32_t acc = 0u;
        for (uint32_t k = 0; k < 1024u; ++k) { uint32_t v = part[k]; part[k] = acc; acc += v; }
        counts[n_blocks] = acc;
        sync[RT_LIVE_SLOTS] = acc;
    }
    __syncthreads();
    uint32_t acc = part[t];
    for (uint32_t k = lo; k < hi; ++k) { uint32_t v = counts[k]; counts[k] = acc; acc += v; }
    __syncthreads();
    const uint32_t total = counts[n_blocks];
    const uint32_t n2 = RT_COMPACT_SLOTS(total, n_slots);
    for (uint32_t s = total + t; s < n2; s += 1024u) to[QPI(n2, 1, s)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // plane 1
}
__global__ void __launch_bounds__(256) rt_compact_move_kernel(PathPool pool, const uint32_t* __restrict__ offsets, float4* __restrict__ to) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    float4 p1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (s < pool.n_slots) p1 = QP(pool, 1, s);
    const bool holds = (__float_as_uint(p1.w) & DEPTH_MASK) != 0u;
    __shared__ uint32_t wave_n[4];
    const unsigned long long m = __ballot(holds);
    if ((threadIdx.x & 63u) == 0u) wave_n[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!holds) return;
    uint32_t rank = (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) rank += wave_n[w];
    const size_t d = (size_t)offsets[blockIdx.x] + rank, n = RT_COMPACT_SLOTS(offsets[gridDim.x], pool.n_slots);
    to[QPI(n, 0, d)] = QP(pool, 0, s);
    to[QPI(n, 1, d)] = p1;
    to[QPI(n, 2, d)] = QP(pool, 2, s);
    to[QPI(n, 3, d)] = QP(pool, 3, s); // a slot on hold keeps its settled hit
    if (__float_as_uint(p1.w) & DEPTH_HAS_COLOR) to[QPI(n, 4, d)] = QP(pool, 4, s);
}

// accum[pixel] (+)= radiance of the pixel's paths of this batch, added in column order like the sample loop of draw!
// (src/Ray_Tracing.jl:82-89); accum is in the Julia image layout img[height-j, i+1]
__global__ void __launch_bounds__(256) rt_sum_kernel(RtRenderParams P, const float4* __restrict__ radiance,
                                                     float* __restrict__ accum, int first_batch) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.n_pixels) return;
    int j = (int)(r / (uint32_t)P.width);
    int i = (int)(r - (uint32_t)j * (uint32_t)P.width);
    size_t idx = ((size_t)i * (size_t)P.height + (size_t)(P.height - 1 - j)) * 3u;
    f3 col = first_batch ? mk3(0.0f, 0.0f, 0.0f) : mk3(accum[idx], accum[idx + 1], accum[idx + 2]);
    uint32_t nk = P.k_end - P.k_begin;
    for (uint32_t kk = 0; kk < nk; ++kk) {
        float4 v = radiance[(size_t)kk * P.n_pixels + r];
        col = col + mk3(v.x, v.y, v.z);
    }
    accum[idx] = col.x; accum[idx + 1] = col.y; accum[idx + 2] = col.z;
}

// dst += src (the accumulator of another GPU, copied over, into the first one's)
__global__ void __launch_bounds__(256) rt_add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

// col /= N; sqrt per channel (src/Ray_Tracing.jl:90-91,152-153; the division is a multiply by Float32(1/N))
__global__ void __launch_bounds__(256) rt_finalize_kernel(const float* __restrict__ accum, float inv_n,
                                                          size_t image_floats, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= image_floats) return;
    out[i] = sqrtf(accum[i] * inv_n);
}

// rt_finalize_kernel + clamp01 + N0f8 (round to nearest even of x * 255); out is H x W x 3 bytes, row-major, row 0 = top
__global__ void __launch_bounds__(256) rt_finalize_rgb8_kernel(const float* __restrict__ accum, float inv_n, int width, int height,
                                                               uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * (size_t)height) return;
    int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * (size_t)width);
    const float* a = accum + ((size_t)x * (size_t)height + (size_t)y) * 3u;
    for (int c = 0; c < 3; ++c) {
        float v = sqrtf(a[c] * inv_n);
        if (!(v == v)) v = 0.0f;
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        out[i * 3u + (size_t)c] = (uint8_t)__builtin_rintf(v * 255.0f);
    }
}

__global__ void __launch_bounds__(256) rt_trace_rays_kernel(DevScene S, const float* __restrict__ rays, int n, float t_min,
                                                            float t_max, uint32_t seed32, int32_t* prim_id,
                                                            int32_t* sub_id, float* t_out, float* normal, float* uv,
                                                            uint8_t* front) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = k < n;
    if (!active) k = 0;
    const float* q = rays + 7 * (size_t)k;
    f3 o = mk3(q[0], q[1], q[2]), d = mk3(q[3], q[4], q[5]);
    float time = q[6];
    RngKey key = {(uint32_t)k, 0u, seed32, 0u};
    Counters cnt = {0, 0, 0, 0, 0, 0};
    Closest best;
    __shared__ Closest settle_list[256][SETTLE_LIST];
    closest_hit<false>(S, o, d, time, t_min, t_max, key, best, cnt, active, settle_list[threadIdx.x]);
    if (!active) return;
    HitRec h;
    h.p = mk3(INF, INF, INF); h.n = mk3(0, 0, 0); h.u = 0; h.v = 0; h.front = false; h.material = 0;
    int pid = -1, sid = -1;
    if (best.t != INF) {
        complete_hit<true>(S, o, d, time, best, h);
        pid = S.prim_top[best.prim];
        uint4 head = S.prims[3u * best.prim];
        sid = S.prim_sub[best.prim] + (((head.x & 15u) == DP_BOX) ? (int)(best.sx & 7u) : 0);
    }
    prim_id[k] = pid;
    sub_id[k] = sid;
    t_out[k] = best.t;
    normal[3 * k] = h.n.x; normal[3 * k + 1] = h.n.y; normal[3 * k + 2] = h.n.z;
    uv[2 * k] = h.u; uv[2 * k + 1] = h.v;
    front[k] = h.front ? 1 : 0;
}

__global__ void __launch_bounds__(256) rt_trace_paths_kernel(DevScene S, RtCameraDev cam, const float* __restrict__ pattern,
                                                             RtRenderParams P, const int32_t* __restrict__ ids,
                                                             int n_paths, float* __restrict__ radiance) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = k < n_paths;
    if (!alive) k = 0;
    int i = ids[3 * k], j = ids[3 * k + 1], n = ids[3 * k + 2];
    RngKey key;
    key.pixel = (uint32_t)j * (uint32_t)P.width + (uint32_t)i;
    key.sample = (uint32_t)n;
    key.seed32 = P.seed32;
    Counters cnt = {0, 0, 0, 0, 0, 0};
    f3 o, d;
    float time;
    camera_ray(cam, pattern, i, j, n, P.width, P.height, key, o, d, time);
    PathState ps;
    ps.acc = mk3(1.0f, 1.0f, 1.0f);
    ps.color = mk3(0.0f, 0.0f, 0.0f);
    ps.depth = P.max_depth;
    key.bounce = 0;
    f3 c = mk3(0.0f, 0.0f, 0.0f);
    f3 bgc = mk3(P.background[0], P.background[1], P.background[2]);
    Walker W;
    __shared__ Closest settle_list[256][SETTLE_LIST];
    walker_start(W, S, o, d, time, 0.001f, INF);
    Vote V = {P.leaf_threshold, 16u};
    for (;;) {
        bool ws = alive && walker_wants_step(W), wl = alive && walker_wants_leaf(W), wh = alive && walker_finished(W);
        int block = wave_vote(ws, wl, wh, V);
        if (block == 3) break;
        if (block == 0) { if (ws) walker_step<false>(W, S, cnt); }
        else if (block == 1) { if (wl) walker_leaf<false>(W, S, cnt); }
        else if (wh) {
            bool done = ps.depth <= 0;
            f3 no, nd;
            if (done) c = ps.color;
            else {
                settle_hit(S, W.o, W.d, W.time, 0.001f, INF, key, W.best, settle_list[threadIdx.x]);
                done = shade<true>(S, W.o, W.d, W.time, W.best, ps, bgc, key, c, no, nd);
            }
            if (done) alive = false;
            else walker_start(W, S, no, nd, W.time, 0.001f, INF);
        }
    }
    if (k >= n_paths || blockIdx.x * blockDim.x + threadIdx.x >= (unsigned)n_paths) return;
    radiance[3 * k] = c.x; radiance[3 * k + 1] = c.y; radiance[3 * k + 2] = c.z;
}

// ------------------------------------------------------------------ host-callable launchers
extern "C" void rt_launch_init(const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P, const PathPool* pool,
                               hipStream_t stream) {
    hipLaunchKernelGGL(rt_init_kernel, dim3((pool->n_slots + 255u) / 256u), dim3(256), 0, stream, *cam, d_pattern, *P, *pool);
}
extern "C" void rt_launch_bounce(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                                 const PathPool* pool, uint32_t extend_blocks, float4* d_radiance, uint32_t* d_sync,
                                 uint32_t* d_active, unsigned long long* d_counters, bool count, hipEvent_t* timing,
                                 hipStream_t stream, hipEvent_t before_shade) {
    // timing (optional): three events, before extend / between / after shade
    uint32_t shade_blocks = (pool->n_slots + SHADE_BLOCK - 1u) / SHADE_BLOCK;
    if (timing) (void)hipEventRecord(timing[0], stream);
    const bool nested = S->nested != 0u, pairs = !nested && S->use_pairs != 0u;
    static const bool extend_blocks_per_cu = std::getenv("RT_AMD_EXTEND_BLOCKS_PER_CU") != nullptr; // the caller's grid as it is
    static const bool low_occupancy = std::getenv("RT_AMD_EXTEND_SIX") != nullptr; // six workgroups per CU whatever the variant could hold (measurements)
    if (count && nested) hipLaunchKernelGGL((rt_extend_kernel<true, true, false>), dim3(extend_blocks), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    else if (count && pairs && S->pair_depth > RT_PAIR_STACK)
        hipLaunchKernelGGL((rt_extend_kernel<true, false, true, RT_CL_ALL, 4, RT_PAIR_STACK_DEEP>), dim3(extend_blocks / 6u * 4u), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    else if (count && pairs) hipLaunchKernelGGL((rt_extend_kernel<true, false, true>), dim3(extend_blocks), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    else if (count) hipLaunchKernelGGL((rt_extend_kernel<true, false, false>), dim3(extend_blocks), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    else if (nested) hipLaunchKernelGGL((rt_extend_kernel<false, true, false>), dim3(extend_blocks), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    else if (pairs) {
        // the variant compiled for the scene's classes of primitives: the first of the list that covers them
        const uint32_t have = S->classes;
#define EXTEND_VARIANT(CL, BLOCKS)                                                                                               \
        if ((have & ~(uint32_t)(CL)) == 0u) {                                                                                        \
            if (S->pair_depth <= 16u && !low_occupancy) {                                                                            \
                const uint32_t blocks = extend_blocks_per_cu ? extend_blocks : extend_blocks / 6u * (uint32_t)(BLOCKS);              \
                hipLaunchKernelGGL((rt_extend_kernel<false, false, true, (CL), (BLOCKS), 16u>), dim3(blocks), dim3(256), 0, stream,  \
                                   *S, *P, *pool, d_sync, d_counters);                                                               \
            } else if (S->pair_depth <= RT_PAIR_STACK) {                                                                             \
                hipLaunchKernelGGL((rt_extend_kernel<false, false, true, (CL), 6, RT_PAIR_STACK>), dim3(extend_blocks), dim3(256),   \
                                   0, stream, *S, *P, *pool, d_sync, d_counters);                                                    \
            } else { /* a deep tree (a large mesh): 35 KB of stacks a workgroup, four workgroups per CU */                           \
                const uint32_t blocks = extend_blocks_per_cu ? extend_blocks : extend_blocks / 6u * 4u;                              \
                hipLaunchKernelGGL((rt_extend_kernel<false, false, true, RT_CL_ALL, 4, RT_PAIR_STACK_DEEP>), dim3(blocks),           \
                                   dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);                                         \
            }                                                                                                                        \
        } else
        EXTEND_VARIANT(RT_CL_SPHERE, 8)
        EXTEND_VARIANT(RT_CL_TRI, 8)
        EXTEND_VARIANT(RT_CL_RECT | RT_CL_BOX | RT_CL_XFORM, 7)
        EXTEND_VARIANT(RT_CL_SPHERE | RT_CL_MOVING | RT_CL_RECT | RT_CL_BOX | RT_CL_XFORM, 7)
        EXTEND_VARIANT(RT_CL_ALL, 7)
        {}
#undef EXTEND_VARIANT
    }
    else hipLaunchKernelGGL((rt_extend_kernel<false, false, false>), dim3(extend_blocks), dim3(256), 0, stream, *S, *P, *pool, d_sync, d_counters);
    if (before_shade) (void)hipStreamWaitEvent(stream, before_shade, 0);
    if (timing) (void)hipEventRecord(timing[1], stream);
    if (S->textured) hipLaunchKernelGGL(rt_shade_kernel<true>, dim3(shade_blocks), dim3(SHADE_BLOCK), 0, stream, *S, *cam, d_pattern, *P, *pool,
                                        d_radiance, d_sync, d_active);
    else hipLaunchKernelGGL(rt_shade_kernel<false>, dim3(shade_blocks), dim3(SHADE_BLOCK), 0, stream, *S, *cam, d_pattern, *P, *pool,
                            d_radiance, d_sync, d_active);
    if (timing) (void)hipEventRecord(timing[2], stream);
}
extern "C" void rt_launch_redo(const DevScene* S, const RtRenderParams* P, const PathPool* pool, uint32_t* d_sync, hipStream_t stream) {
    hipLaunchKernelGGL(rt_redo_kernel, dim3(256), dim3(256), 0, stream, *S, *P, *pool, d_sync);
}
extern "C" void rt_launch_compact(const PathPool* from, float4* to, uint32_t* d_counts, uint32_t* d_sync, hipStream_t stream) {
    const uint32_t n_blocks = (from->n_slots + 255u) / 256u;
    hipLaunchKernelGGL(rt_compact_count_kernel, dim3(n_blocks), dim3(256), 0, stream, *from, d_counts);
    hipLaunchKernelGGL(rt_compact_scan_kernel, dim3(1), dim3(1024), 0, stream, d_counts, n_blocks, to, from->n_slots, d_sync);
    hipLaunchKernelGGL(rt_compact_move_kernel, dim3(n_blocks), dim3(256), 0, stream, *from, d_counts, to);
}
extern "C" void rt_launch_sum(const RtRenderParams* P, const float4* d_radiance, float* d_accum, int first_batch, hipStream_t stream) {
    hipLaunchKernelGGL(rt_sum_kernel, dim3((P->n_pixels + 255u) / 256u), dim3(256), 0, stream, *P, d_radiance, d_accum, first_batch);
}
extern "C" void rt_launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(rt_add_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_dst, d_src, n);
}
extern "C" void rt_launch_finalize(const float* d_accum, float inv_n, size_t image_floats, float* d_out, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((image_floats + 255) / 256);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(blocks), dim3(256), 0, stream, d_accum, inv_n, image_floats, d_out);
}
extern "C" void rt_launch_finalize_rgb8(const float* d_accum, float inv_n, int width, int height, uint8_t* d_out, hipStream_t stream) {
    uint32_t blocks = (uint32_t)(((size_t)width * (size_t)height + 255) / 256);
    hipLaunchKernelGGL(rt_finalize_rgb8_kernel, dim3(blocks), dim3(256), 0, stream, d_accum, inv_n, width, height, d_out);
}
extern "C" void rt_launch_trace_rays(const DevScene* S, const float* d_rays, int n, float t_min, float t_max, uint32_t seed32,
                                     int32_t* prim_id, int32_t* sub_id, float* t, float* normal, float* uv, uint8_t* front,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(rt_trace_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *S, d_rays, n, t_min, t_max, seed32,
                       prim_id, sub_id, t, normal, uv, front);
}
extern "C" void rt_launch_trace_paths(const DevScene* S, const RtCameraDev* cam, const float* d_pattern, const RtRenderParams* P,
                                      const int32_t* d_ids, int n_paths, float* d_radiance, hipStream_t stream) {
    hipLaunchKernelGGL(rt_trace_paths_kernel, dim3((n_paths + 255) / 256), dim3(256), 0, stream, *S, *cam, d_pattern, *P, d_ids,
                       n_paths, d_radiance);
}
