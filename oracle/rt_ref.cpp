/*
 * rt_ref.cpp -- CPU ORACLE (test infrastructure only, see rt_ref.h).
 *
 * Restates the CPU path of rafaellg79/raytracing in plain C++: same formulas, same
 * comparison operators, same tie-breaks, same evaluation order of every real
 * expression (Julia evaluates a*b + c*d + e*f left to right, no FMA contraction;
 * build with -ffp-contract=off).  Citations are file:line in the reference tree.
 *
 * Parity status: UNPINNED against an execution of the reference (no Julia here, the
 * reference has no tests or golden vectors and never seeds its RNG); pinned only by
 * hand-derived known-answer tests and the statistics of the deterministic gallery
 * images.
 */
#include "rt_ref.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef ref_real real;      /* float, or double in the Float64 build (-DREF_F64) */
#define R(x) ((real)(x))    /* a literal of the element type: F(x) in the reference */

const real INF = std::numeric_limits<real>::infinity();
thread_local std::string g_err;

/* ------------------------------------------------------------------ vec3.jl */
struct V3 {
    real x, y, z;
};
inline V3 v3(real x, real y, real z) { return V3{x, y, z}; }
inline V3 v3(real s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }    /* vec3.jl:18 */
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }    /* vec3.jl:19 */
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }    /* vec3.jl:20 */
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                          /* vec3.jl:25 */
inline V3 operator+(V3 a, real b) { return v3(a.x + b, a.y + b, a.z + b); }        /* vec3.jl:28 */
inline V3 operator-(V3 a, real b) { return v3(a.x - b, a.y - b, a.z - b); }        /* vec3.jl:29 */
inline V3 operator*(V3 a, real b) { return v3(a.x * b, a.y * b, a.z * b); }        /* vec3.jl:30 */
inline V3 operator*(real a, V3 b) { return v3(a * b.x, a * b.y, a * b.z); }        /* vec3.jl:35 */
inline V3 div_real(V3 a, real b) { return a * (R(1.0) / b); }                        /* vec3.jl:31  a * F(1/b) */
inline real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* vec3.jl:46 */
inline V3 cross(V3 a, V3 b) {                                                       /* vec3.jl:49 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline real length(V3 a) { return std::sqrt(dot(a, a)); }                              /* vec3.jl:52 */
inline V3 normalize(V3 a) { return div_real(a, length(a)); }                        /* vec3.jl:57 */
/* vec3.jl:62-64: isapprox(x, 0) with default tolerances holds only for x == 0 */
inline bool almost_zero(V3 a) { return a.x == R(0.0) && a.y == R(0.0) && a.z == R(0.0); }
inline V3 vmin(V3 a, V3 b) { return v3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
inline real comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Ray { /* ray.jl:3 */
    V3 o, d;
    real time;
};
inline V3 at(const Ray& r, real t) { return r.o + r.d * t; } /* ray.jl:15 */

struct AABB { /* bounding_box.jl:4 */
    V3 mn, mx;
};

/* --------------------------------------------------- shared deterministic math */
inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* natural log for x in (0,1] as produced by rand01 (normal floats); x == 0 -> -Inf.
 * Classic argument reduction x = 2^k * (1+f), log(1+f) by the atanh series. */
inline float det_logf(float x) {
    if (x == 0.0f) return -INF;
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = float_to_bits(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    int k = (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float m = bits_to_float(ix);
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

/* sine: quadrant reduction in double, polynomial in float */
inline float det_sinf(float x) {
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

/* atan(z) for z in [0,1], double: two half-angle reductions then the odd series */
inline double det_atan01(double z) {
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
inline double det_atan2d(double y, double x) {
    const double PI = 3.14159265358979323846;
    double ax = fabs(x), ay = fabs(y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax >= ay) a = det_atan01(ay / ax);
    else a = 0.5 * PI - det_atan01(ax / ay);
    if (std::signbit(x)) a = PI - a;
    if (std::signbit(y)) a = -a;
    return a;
}
inline float det_atan2f(float y, float x) { return (float)det_atan2d((double)y, (double)x); }
inline float det_acosf(float c) {
    double cd = (double)c;
    double s = sqrt((1.0 - cd) * (1.0 + cd));
    return (float)det_atan2d(s, cd);
}

/* pcg4d, Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 9(3) 2020 */
inline void pcg4d(uint32_t v[4]) {
    for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}


/* ---- the same four routines at double accuracy, for the Float64 build (render(Float64, ...), main.jl:21,46):
 * IEEE basic operations only, so that the kernels' copies give the same bits. */
inline double det_logd(double x) { /* x in (0,1): argument reduction x = 2^k * m, m in [sqrt(1/2), sqrt(2)); atanh series */
    if (x == 0.0) return -std::numeric_limits<double>::infinity();
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits; memcpy(&bits, &x, 8);
    uint32_t hx = (uint32_t)(bits >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    bits = ((uint64_t)hx << 32) | (bits & 0xffffffffu);
    double m; memcpy(&m, &bits, 8);
    double f = m - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return (((s * (hfsq + R) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}
inline double det_sind(double x) { /* quadrant by a two-part pi/2 (exact product for |x| < 1.6e6), then the two kernels */
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    double n = floor(x * TWO_OVER_PI + 0.5);
    double r = (x - n * PIO2_1) - n * PIO2_1T;
    long long q = (long long)n;
    double z = r * r;
    double sn, cs;
    {
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        sn = r + (r * z) * (S1 + z * p);
    }
    {
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        cs = 1.0 - (0.5 * z - z * p);
    }
    switch ((int)(q & 3)) {
        case 0: return sn;
        case 1: return cs;
        case 2: return -sn;
        default: return -cs;
    }
}
inline double det_atan01_full(double z) { /* atan(z), z in [0,1]: three half-angle reductions, then the odd series */
    z = z / (1.0 + sqrt(1.0 + z * z));
    z = z / (1.0 + sqrt(1.0 + z * z));
    z = z / (1.0 + sqrt(1.0 + z * z));
    double w = z * z;
    double p = 1.0 / 19.0;
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
inline double det_atan2_full(double y, double x) {
    const double PI = 3.14159265358979323846;
    double ax = fabs(x), ay = fabs(y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax >= ay) a = det_atan01_full(ay / ax);
    else a = 0.5 * PI - det_atan01_full(ax / ay);
    if (std::signbit(x)) a = PI - a;
    if (std::signbit(y)) a = -a;
    return a;
}
inline double det_acosd(double c) {
    double s = sqrt((1.0 - c) * (1.0 + c));
    return det_atan2_full(s, c);
}

/* the element type of the build picks its routines and the width of its random numbers */
#ifdef REF_F64
inline double det_log(double x) { return det_logd(x); }
inline double det_sin(double x) { return det_sind(x); }
inline double det_acos(double x) { return det_acosd(x); }
inline double det_atan2(double y, double x) { return det_atan2_full(y, x); }
#else
inline float det_log(float x) { return det_logf(x); }
inline float det_sin(float x) { return det_sinf(x); }
inline float det_acos(float x) { return det_acosf(x); }
inline float det_atan2(float y, float x) { return det_atan2f(y, x); }
#endif
inline float u32_to_01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
/* 53 bits from two words: 27 from the first, 26 from the second */
inline double u64_to_01(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

enum { PURPOSE_TRIAL = 0, PURPOSE_SCALAR = 1, PURPOSE_MEDIUM = 2 };
enum { CAMERA_BOUNCE = 63 };
inline uint32_t rng_counter(uint32_t bounce, uint32_t purpose, uint32_t k) {
    return (bounce << 26) | (purpose << 24) | (k & 0xFFFFFFu);
}

struct Stats {
    uint64_t rays = 0, node_visits = 0, prim_tests[12] = {0}, scatters[6] = {0};
};

struct Ctx { /* identifies the random stream of the path being traced */
    uint32_t pixel = 0, sample = 0, seed32 = 0, bounce = 0;
    Stats* st = nullptr;
    std::vector<Ray>* ray_log = nullptr; /* when set, trace() records every ray it casts */
    void draw(uint32_t purpose, uint32_t k, real out[4]) const {
#ifdef REF_F64
        /* rand(Float64): 53 bits each, two words of the generator per number, two hashes per draw */
        uint32_t a[4] = {pixel, sample, rng_counter(bounce, purpose, 2 * k), seed32};
        uint32_t b[4] = {pixel, sample, rng_counter(bounce, purpose, 2 * k + 1), seed32};
        pcg4d(a);
        pcg4d(b);
        out[0] = u64_to_01(a[0], a[1]); out[1] = u64_to_01(a[2], a[3]);
        out[2] = u64_to_01(b[0], b[1]); out[3] = u64_to_01(b[2], b[3]);
#else
        uint32_t v[4] = {pixel, sample, rng_counter(bounce, purpose, k), seed32};
        pcg4d(v);
        for (int i = 0; i < 4; ++i) out[i] = u32_to_01(v[i]);
#endif
    }
};
inline uint32_t fold_seed(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u); }

/* vec3.jl:87-94: rejection sampling, 3 rands per trial in x,y,z order, rand*(max-min)+min */
V3 random_in_unit_sphere(const Ctx& c) {
    for (uint32_t k = 0;; ++k) {
        real r[4];
        c.draw(PURPOSE_TRIAL, k, r);
        V3 p = v3(r[0] * R(2.0) + -R(1.0), r[1] * R(2.0) + -R(1.0), r[2] * R(2.0) + -R(1.0));
        if (!(dot(p, p) > R(1.0))) return p;
    }
}
/* vec3.jl:96-105 */
V3 random_in_unit_disk(const Ctx& c) {
    for (uint32_t k = 0;; ++k) {
        real r[4];
        c.draw(PURPOSE_TRIAL, k, r);
        V3 p = v3(r[0] * R(2.0) - R(1.0), r[1] * R(2.0) - R(1.0), R(0.0));
        if (!(dot(p, p) > R(1.0))) return p;
    }
}
real rand_scalar(const Ctx& c) {
    real r[4];
    c.draw(PURPOSE_SCALAR, 0, r);
    return r[0];
}

/* ------------------------------------------------------------ hit_record.jl */
struct Hit {
    real t = INF;
    V3 p = {INF, INF, INF};
    V3 n = {0, 0, 0};
    int material = -1;
    bool front = false;
    real u = 0, v = 0;
    int prim = -1; /* position in the world vector */
    int sub = 0;   /* box side / triangle / nested object */
};

/* ------------------------------------------------------------------- scene */
struct Node { /* BVH.jl:5-10 */
    AABB box;
    int left = 0, right = 0; /* 1-based like the reference */
    bool leaf = false;
};
struct Tree { /* BVH.jl:12-16 */
    std::vector<Node> nodes;   /* slot i-1 holds node id i */
    std::vector<uint8_t> valid;
    std::vector<int> objects;  /* object indices, permuted in place by the build */
};
struct MeshData {
    std::vector<real> vertices;
    std::vector<int> indices; /* 4 x n, permuted by the build (mesh.jl:237) */
    std::vector<int> order;   /* original triangle id of each column */
    std::vector<int> materials;
    int ntri = 0, nvert = 0;
};
struct Bucket {
    std::string key;
    bool is_mesh = false;
    Tree tree;
};

} // namespace

struct ref_scene {
    std::vector<refObject> objects;
    std::vector<int> top;
    std::vector<int> top_pos; /* object index -> position in top (or -1) */
    std::vector<refTransform> transforms;
    std::vector<refMaterial> materials;
    std::vector<refTexture> textures;
    std::vector<refImage> images;
    std::vector<std::vector<uint8_t>> image_data;
    std::vector<refPerlin> perlins;
    std::vector<MeshData> meshes;
    std::vector<Tree> groups; /* nested BVHTree objects */
    std::vector<Bucket> buckets;
};

namespace {

/* ------------------------------------------------------------ bounding boxes */
AABB bounding_box(const ref_scene& S, int oi);

AABB mesh_bbox(const MeshData& m, const int* vids, int count) { /* mesh.jl:45-49 */
    V3 mn = v3(INF), mx = v3(-INF);
    for (int k = 0; k < count; ++k) {
        const real* v = &m.vertices[8 * vids[k]];
        mn = vmin(mn, v3(v[0], v[1], v[2]));
        mx = vmax(mx, v3(v[0], v[1], v[2]));
    }
    /* Vec3{Float32} - 0.0001 (Float64 literal): computed in double, narrowed on construction */
    AABB b;
    b.mn = v3((real)((double)mn.x - 0.0001), (real)((double)mn.y - 0.0001), (real)((double)mn.z - 0.0001));
    b.mx = v3((real)((double)mx.x + 0.0001), (real)((double)mx.y + 0.0001), (real)((double)mx.z + 0.0001));
    return b;
}
AABB mesh_bbox_all(const MeshData& m) {
    std::vector<int> ids(m.nvert);
    for (int i = 0; i < m.nvert; ++i) ids[i] = i;
    return mesh_bbox(m, ids.data(), m.nvert);
}
AABB mesh_bbox_range(const MeshData& m, int left, int right) { /* indices[1:3, left:right], 1-based inclusive */
    std::vector<int> ids;
    for (int c = left; c <= right; ++c)
        for (int r = 0; r < 3; ++r) ids.push_back(m.indices[4 * (c - 1) + r]);
    return mesh_bbox(m, ids.data(), (int)ids.size());
}

AABB bounding_box(const ref_scene& S, int oi) {
    const refObject& o = S.objects[oi];
    const real* p = o.p;
    switch (o.kind) {
        case REF_SPHERE: { /* sphere.jl:32-34 */
            V3 c = v3(p[0], p[1], p[2]);
            real r = std::fabs(p[3]);
            return AABB{c - r, c + r};
        }
        case REF_MOVING_SPHERE: { /* sphere.jl:36-38 */
            V3 c0 = v3(p[0], p[1], p[2]), c1 = v3(p[4], p[5], p[6]);
            real r = std::fabs(p[8]);
            return AABB{vmin(c0 - r, c1 - r), vmax(c0 + r, c1 + r)};
        }
        case REF_XY_RECT: /* aarect.jl:37-39 */
            return AABB{v3(p[0], p[2], p[4] - R(0.0001)), v3(p[1], p[3], p[4] + R(0.0001))};
        case REF_XZ_RECT: /* aarect.jl:41-43 */
            return AABB{v3(p[0], p[4] - R(0.0001), p[2]), v3(p[1], p[4] + R(0.0001), p[3])};
        case REF_YZ_RECT: /* aarect.jl:45-47 */
            return AABB{v3(p[4] - R(0.0001), p[0], p[2]), v3(p[4] + R(0.0001), p[1], p[3])};
        case REF_BOX: /* box.jl:85-87 */
            return AABB{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])};
        case REF_TRANSFORM: { /* transform.jl:100-102 (cached at construction, :47-63) */
            const real* b = S.transforms[o.aux].bbox;
            return AABB{v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5])};
        }
        case REF_CONSTANT_MEDIUM: /* constant_medium.jl:25-27 */
            return bounding_box(S, o.child);
        case REF_BVH: /* BVH.jl:61-63 */
            return S.groups[o.aux].nodes[0].box;
        case REF_MESH:
            return mesh_bbox_all(S.meshes[o.aux]);
    }
    return AABB{v3(INF), v3(-INF)};
}

/* hittable.jl:17-19 / bounding_box.jl:9-13 */
AABB surrounding_box(const ref_scene& S, const std::vector<int>& objs, int left, int right) {
    AABB b = bounding_box(S, objs[left - 1]);
    for (int i = left + 1; i <= right; ++i) {
        AABB c = bounding_box(S, objs[i - 1]);
        b = AABB{vmin(b.mn, c.mn), vmax(b.mx, c.mx)};
    }
    return b;
}

int longest_axis(const AABB& box) { /* BVH.jl:76-77: findmax returns the first maximum */
    V3 e = box.mx - box.mn;
    int axis = 0;
    real best = e.x;
    if (e.y > best) { best = e.y; axis = 1; }
    if (e.z > best) { best = e.z; axis = 2; }
    return axis;
}

/* --------------------------------------------------------------- BVH.jl:65-96 */
int build_bvh(const ref_scene& S, Tree& tree, int node_id, int left, int right) {
    if (left == right) {
        Node n;
        n.box = surrounding_box(S, tree.objects, left, right);
        n.left = left; n.right = right; n.leaf = true;
        tree.nodes[node_id - 1] = n; tree.valid[node_id - 1] = 1;
        return node_id;
    }
    AABB box = surrounding_box(S, tree.objects, left, right);
    int axis = longest_axis(box);
    /* sort!(view; by=bounding_box, lt = a.min[axis] < b.min[axis]) -- stable merge sort */
    std::vector<std::pair<real, int>> keyed;
    for (int i = left; i <= right; ++i)
        keyed.push_back({comp(bounding_box(S, tree.objects[i - 1]).mn, axis), tree.objects[i - 1]});
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const std::pair<real, int>& a, const std::pair<real, int>& b) { return a.first < b.first; });
    for (int i = left; i <= right; ++i) tree.objects[i - 1] = keyed[i - left].second;

    int mid = (left + right) / 2;
    int ln = build_bvh(S, tree, node_id + 1, left, mid);
    int rn = build_bvh(S, tree, node_id + 2 * (mid + 1 - left), mid + 1, right);
    Node n;
    n.box = box; n.left = ln; n.right = rn; n.leaf = false;
    tree.nodes[node_id - 1] = n; tree.valid[node_id - 1] = 1;
    return node_id;
}
void build_tree(const ref_scene& S, Tree& tree) { /* BVH.jl:98-109 */
    size_t n = tree.objects.size();
    tree.nodes.assign(2 * n - 1, Node());
    tree.valid.assign(2 * n - 1, 0);
    build_bvh(S, tree, 1, 1, (int)n);
}

/* ------------------------------------------------------- mesh.jl:217-248 */
int build_mesh_tris(ref_scene& S, Tree& tree, MeshData& m, int node_id, int left, int right, bool left_child) {
    const int max_triangles = 1;
    auto put = [&](int id, const Node& n) {
        if ((size_t)id > tree.nodes.size()) { tree.nodes.resize(id); tree.valid.resize(id, 0); }
        tree.nodes[id - 1] = n; tree.valid[id - 1] = 1;
    };
    if (right - left < max_triangles) {
        Node n;
        n.box = mesh_bbox_range(m, left, right);
        n.left = left; n.right = right; n.leaf = true;
        put(node_id, n);
        return node_id;
    }
    AABB box = mesh_bbox_range(m, left, right);
    int axis = longest_axis(box);
    /* sortslices(view indices[:, left:right]; dims=2, by=bbox of the column, lt=min[axis] <) -- stable */
    struct Col { real key; int idx[4]; int order; };
    std::vector<Col> cols;
    for (int c = left; c <= right; ++c) {
        Col col;
        for (int r = 0; r < 4; ++r) col.idx[r] = m.indices[4 * (c - 1) + r];
        col.order = m.order[c - 1];
        col.key = comp(mesh_bbox(m, col.idx, 3).mn, axis);
        cols.push_back(col);
    }
    std::stable_sort(cols.begin(), cols.end(), [](const Col& a, const Col& b) { return a.key < b.key; });
    for (int c = left; c <= right; ++c) {
        for (int r = 0; r < 4; ++r) m.indices[4 * (c - 1) + r] = cols[c - left].idx[r];
        m.order[c - 1] = cols[c - left].order;
    }
    int mid = (left + right) / 2;
    int ln = build_mesh_tris(S, tree, m, node_id * 2, left, mid, left_child);
    int rn = build_mesh_tris(S, tree, m, node_id * (2 - (left_child ? 1 : 0)) + 1, mid + 1, right, false);
    Node n;
    n.box = box; n.left = ln; n.right = rn; n.leaf = false;
    put(node_id, n);
    return node_id;
}
/* ------------------------------------------------------- mesh.jl:250-283 */
int build_mesh_level(ref_scene& S, Tree& tree, int node_id, int left, int right) {
    auto put = [&](int id, const Node& n) {
        if ((size_t)id > tree.nodes.size()) { tree.nodes.resize(id); tree.valid.resize(id, 0); }
        tree.nodes[id - 1] = n; tree.valid[id - 1] = 1;
    };
    if (left == right) {
        MeshData& m = S.meshes[S.objects[tree.objects[right - 1]].aux];
        int ln = build_mesh_tris(S, tree, m, node_id * 2, 1, m.ntri, true);
        Node n;
        n.box = bounding_box(S, tree.objects[left - 1]);
        n.left = ln; n.right = right; n.leaf = true;
        put(node_id, n);
        return node_id;
    }
    AABB box = surrounding_box(S, tree.objects, left, right);
    int axis = longest_axis(box);
    std::vector<std::pair<real, int>> keyed;
    for (int i = left; i <= right; ++i)
        keyed.push_back({comp(bounding_box(S, tree.objects[i - 1]).mn, axis), tree.objects[i - 1]});
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const std::pair<real, int>& a, const std::pair<real, int>& b) { return a.first < b.first; });
    for (int i = left; i <= right; ++i) tree.objects[i - 1] = keyed[i - left].second;
    int mid = (left + right) / 2;
    int ln = build_mesh_level(S, tree, node_id * 2, left, mid);
    int rn = build_mesh_level(S, tree, node_id * 2 + 1, mid + 1, right);
    Node n;
    n.box = box; n.left = ln; n.right = rn; n.leaf = false;
    put(node_id, n);
    return node_id;
}
void build_mesh_tree(ref_scene& S, Tree& tree) { /* mesh.jl:294-305 */
    size_t total = 0;
    for (int oi : tree.objects) total += S.meshes[S.objects[oi].aux].ntri;
    tree.nodes.assign(4 * total - 1, Node());
    tree.valid.assign(4 * total - 1, 0);
    build_mesh_level(S, tree, 1, 1, (int)tree.objects.size());
}

/* ------------------------------------------- bounding_box.jl:20-67 (CPU slab) */
bool hit_aabb(const AABB& b, const Ray& r, real t_min, real t_max) {
    real tmin = (b.mn.x - r.o.x) / r.d.x;
    real tmax = (b.mx.x - r.o.x) / r.d.x;
    if (tmin > tmax) std::swap(tmin, tmax);
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;

    tmin = (b.mn.y - r.o.y) / r.d.y;
    tmax = (b.mx.y - r.o.y) / r.d.y;
    if (tmin > tmax) std::swap(tmin, tmax);
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;

    tmin = (b.mn.z - r.o.z) / r.d.z;
    tmax = (b.mx.z - r.o.z) / r.d.z;
    if (tmin > tmax) std::swap(tmin, tmax);
    if (tmax < t_max) t_max = tmax;
    if (tmin > t_min) t_min = tmin;
    if (t_max <= t_min) return false;
    return true;
}

/* -------------------------------------------------------------- primitives */
Hit hit_object(const ref_scene& S, int oi, const Ray& r, real t_min, real t_max, const Ctx& c);
Hit hit_tree(const ref_scene& S, const Tree& tree, const Ray& r, real t_min, real t_max, const Ctx& c, bool nested);

/* sphere.jl:104-108 */
inline void get_sphere_uv(V3 p, real& u, real& v) {
    const real PI_F = R(3.14159265358979323846);
    const real TWO_PI_F = (real)(2.0 * 3.14159265358979323846);
    real cy = -p.y;
    cy = cy < -R(1.0) ? -R(1.0) : (cy > R(1.0) ? R(1.0) : cy);
    real theta = det_acos(cy);
    real phi = det_atan2(-p.z, p.x) + PI_F;
    u = phi / TWO_PI_F;
    v = theta / PI_F;
}

/* sphere.jl:45-72 and :74-102 (centre passed in) */
Hit hit_sphere(V3 center, real radius, int material, const Ray& r, real t_min, real t_max) {
    V3 oc = r.o - center;
    real a = dot(r.d, r.d);
    real half_b = dot(oc, r.d);
    real c = dot(oc, oc) - radius * radius;
    real discriminant = half_b * half_b - a * c;
    if (discriminant < R(0.0)) return Hit();
    real sqrtd = std::sqrt(discriminant);
    real root = (-half_b - sqrtd);
    if (root < t_min * a || t_max * a < root) {
        root = (-half_b + sqrtd);
        if (root < t_min * a || t_max * a < root) return Hit();
    }
    root /= a;
    Hit h;
    h.t = root;
    h.p = at(r, root);
    V3 normal = div_real(h.p - center, radius);
    h.front = dot(r.d, normal) < R(0.0);
    get_sphere_uv(normal, h.u, h.v);
    h.n = (real)(-1 + ((int)h.front << 1)) * normal;
    h.material = material;
    return h;
}

/* aarect.jl:49-119; axis = the constant axis (2: xy_rect, 1: xz_rect, 0: yz_rect) */
Hit hit_rect(int axis, real a0, real a1, real b0, real b1, real k, int material, const Ray& r, real t_min,
             real t_max) {
    real ok, dk, oa, da, ob, db;
    if (axis == 2) { ok = r.o.z; dk = r.d.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
    else if (axis == 1) { ok = r.o.y; dk = r.d.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
    else { ok = r.o.x; dk = r.d.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
    real t = (k - ok) / dk;
    if (t < t_min || t_max < t) return Hit();
    real a = oa + t * da;
    real b = ob + t * db;
    if (a < a0 || a1 < a || b < b0 || b1 < b) return Hit();
    Hit h;
    h.t = t;
    if (axis == 2) h.p = v3(a, b, k);
    else if (axis == 1) h.p = v3(a, k, b);
    else h.p = v3(k, a, b);
    /* aarect.jl:97-119: dot(direction, +axis) < 0 */
    V3 normal = axis == 2 ? v3(0, 0, 1) : (axis == 1 ? v3(0, 1, 0) : v3(1, 0, 0));
    h.front = dot(r.d, normal) < R(0.0);
    h.u = (a - a0) / (a1 - a0);
    h.v = (b - b0) / (b1 - b0);
    h.n = (real)(-1 + 2 * (int)h.front) * normal;
    h.material = material;
    return h;
}

/* box.jl:9-24 (side order) and :43-83 */
Hit hit_box(const real* p, int material, const Ray& r, real t_min, real t_max) {
    V3 p0 = v3(p[0], p[1], p[2]), p1 = v3(p[3], p[4], p[5]);
    Hit closest;
    for (int s = 0; s < 6; ++s) {
        Hit h;
        switch (s) {
            case 0: h = hit_rect(2, p0.x, p1.x, p0.y, p1.y, p1.z, material, r, t_min, t_max); break;
            case 1: h = hit_rect(2, p0.x, p1.x, p0.y, p1.y, p0.z, material, r, t_min, t_max); break;
            case 2: h = hit_rect(1, p0.x, p1.x, p0.z, p1.z, p1.y, material, r, t_min, t_max); break;
            case 3: h = hit_rect(1, p0.x, p1.x, p0.z, p1.z, p0.y, material, r, t_min, t_max); break;
            case 4: h = hit_rect(0, p0.y, p1.y, p0.z, p1.z, p1.x, material, r, t_min, t_max); break;
            default: h = hit_rect(0, p0.y, p1.y, p0.z, p1.z, p0.x, material, r, t_min, t_max); break;
        }
        if (h.t < closest.t) {
            closest = h;
            closest.sub = s;
            t_max = h.t;
        }
    }
    return closest;
}

/* transform.jl:76-98 */
inline V3 apply_point(const real* T, V3 v) {
    return v3(T[0] * v.x + T[3] * v.y + T[6] * v.z + T[9], T[1] * v.x + T[4] * v.y + T[7] * v.z + T[10],
              T[2] * v.x + T[5] * v.y + T[8] * v.z + T[11]);
}
inline V3 apply_vec(const real* T, V3 v) {
    return v3(T[0] * v.x + T[3] * v.y + T[6] * v.z, T[1] * v.x + T[4] * v.y + T[7] * v.z,
              T[2] * v.x + T[5] * v.y + T[8] * v.z);
}

/* triangle.jl:73-91 */
real hit_triangle_geom(V3 v1, V3 v2, V3 v3_, const Ray& r, real t_min, real t_max, V3& normal, real& u, real& v) {
    V3 e1 = v2 - v1;
    V3 e2 = v3_ - v1;
    V3 origin_v1 = v1 - r.o;
    normal = cross(e1, e2);
    V3 s = cross(origin_v1, r.d);
    real c = R(1.0) / dot(normal, r.d);
    u = dot(s, e2) * c;
    v = dot(s, e1) * -c;
    real t = dot(normal, origin_v1) * c;
    if (u < R(0.0) || v < R(0.0) || u + v > R(1.0) || t < t_min || t_max < t) {
        u = -R(1.0); v = -R(1.0);
        return INF;
    }
    return t;
}
/* mesh.jl:53-59 + triangle.jl:95-121 */
Hit hit_mesh_triangle(const MeshData& m, int col /*1-based*/, const Ray& r, real t_min, real t_max) {
    const int* id = &m.indices[4 * (col - 1)];
    const real* a = &m.vertices[8 * id[0]];
    const real* b = &m.vertices[8 * id[1]];
    const real* cc = &m.vertices[8 * id[2]];
    V3 normal;
    real u, v;
    real t = hit_triangle_geom(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]), v3(cc[0], cc[1], cc[2]), r, t_min, t_max,
                                normal, u, v);
    if (t == INF) return Hit();
    /* get_triangle_vertex_at_uv: v1*w + v2*u + v3*v on every Vertex field (vertex.jl:20-25) */
    real w = R(1.0) - u - v;
    V3 vn = (v3(a[3], a[4], a[5]) * w + v3(b[3], b[4], b[5]) * u) + v3(cc[3], cc[4], cc[5]) * v;
    real vu = (a[6] * w + b[6] * u) + cc[6] * v;
    real vv = (a[7] * w + b[7] * u) + cc[7] * v;
    /* triangle.jl:110-115: the flag is true when the interpolated normal is exactly zero, so meshes that
     * carry normals are shaded with the geometric normal and meshes without get a zero (-> NaN) normal */
    bool normal_flag = almost_zero(vn);
    bool uv_flag = vu >= R(0.0) && vv >= R(0.0);
    if (normal_flag) normal = vn;
    if (uv_flag) { u = vu; v = vv; }
    Hit h;
    h.t = t;
    h.p = at(r, t);
    h.front = dot(r.d, normal) < R(0.0);
    h.n = (real)(-1 + ((int)h.front << 1)) * normalize(normal);
    h.material = m.materials[id[3]];
    h.u = u; h.v = v;
    h.sub = m.order[col - 1];
    return h;
}

/* constant_medium.jl:29-65 */
Hit hit_medium(const ref_scene& S, int oi, const Ray& r, real t_min, real t_max, const Ctx& c) {
    const refObject& o = S.objects[oi];
    real t1 = hit_object(S, o.child, r, -INF, INF, c).t;
    if (t1 == INF) return Hit();
    real t2 = hit_object(S, o.child, r, t1 + R(0.0001), INF, c).t;
    if (t2 == INF) return Hit();
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return Hit();
    if (t1 < R(0.0)) t1 = R(0.0);
    real ray_length = length(r.d);
    real distance_inside_boundary = (t2 - t1) * ray_length;
    real rnd[4];
    c.draw(PURPOSE_MEDIUM, (uint32_t)oi, rnd);
    real hit_distance = o.p[0] * det_log(rnd[0]);
    if (hit_distance > distance_inside_boundary) return Hit();
    Hit h;
    h.t = t1 + hit_distance / ray_length;
    h.p = at(r, h.t);
    h.n = v3(1, 0, 0);
    h.material = o.material;
    h.front = false;
    h.u = 0; h.v = 0;
    return h;
}

Hit hit_object(const ref_scene& S, int oi, const Ray& r, real t_min, real t_max, const Ctx& c) {
    const refObject& o = S.objects[oi];
    if (c.st) c.st->prim_tests[o.kind]++;
    const real* p = o.p;
    switch (o.kind) {
        case REF_SPHERE:
            return hit_sphere(v3(p[0], p[1], p[2]), p[3], o.material, r, t_min, t_max);
        case REF_MOVING_SPHERE: { /* sphere.jl:24-26 center_at */
            V3 c0 = v3(p[0], p[1], p[2]), c1 = v3(p[4], p[5], p[6]);
            V3 center = c0 + ((r.time - p[3]) / (p[7] - p[3])) * (c1 - c0);
            return hit_sphere(center, p[8], o.material, r, t_min, t_max);
        }
        case REF_XY_RECT: return hit_rect(2, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max);
        case REF_XZ_RECT: return hit_rect(1, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max);
        case REF_YZ_RECT: return hit_rect(0, p[0], p[1], p[2], p[3], p[4], o.material, r, t_min, t_max);
        case REF_BOX: return hit_box(p, o.material, r, t_min, t_max);
        case REF_TRANSFORM: { /* transform.jl:104-116 */
            const refTransform& T = S.transforms[o.aux];
            Ray rr{apply_point(T.inv, r.o), apply_vec(T.inv, r.d), r.time};
            Hit h = hit_object(S, o.child, rr, t_min, t_max, c);
            if (h.t == INF) return h;
            h.p = at(r, h.t);
            h.n = apply_vec(T.xform, h.n);
            return h;
        }
        case REF_CONSTANT_MEDIUM: return hit_medium(S, oi, r, t_min, t_max, c);
        case REF_BVH: return hit_tree(S, S.groups[o.aux], r, t_min, t_max, c, true);
        case REF_MESH: { /* mesh.jl:73-86 linear scan (a mesh reached outside a mesh bucket) */
            const MeshData& m = S.meshes[o.aux];
            Hit closest;
            for (int i = 1; i <= m.ntri; ++i) {
                Hit h = hit_mesh_triangle(m, i, r, t_min, t_max);
                if (h.t < closest.t) { closest = h; t_max = closest.t; }
            }
            return closest;
        }
    }
    return Hit();
}

/* BVH.jl:22-53 */
Hit hit_tree(const ref_scene& S, const Tree& tree, const Ray& r, real t_min, real t_max, const Ctx& c, bool nested) {
    if (tree.nodes.empty()) return Hit();
    std::vector<int> next(64);
    int n = 1;
    next[0] = 1;
    Hit closest;
    while (n > 0) {
        const Node& node = tree.nodes[next[n - 1] - 1];
        n -= 1;
        if (c.st) c.st->node_visits++;
        if (!hit_aabb(node.box, r, t_min, t_max)) continue;
        if (node.leaf) {
            /* hittable.jl:21-31 list hit over objects[left:right] (a single object) */
            Hit lh;
            real tm = t_max;
            for (int i = node.left; i <= node.right; ++i) {
                int oi = tree.objects[i - 1];
                Hit h = hit_object(S, oi, r, t_min, tm, c);
                if (h.t < lh.t) {
                    lh = h;
                    if (nested) lh.sub = oi; else lh.prim = S.top_pos[oi];
                    tm = lh.t;
                }
            }
            if (lh.t < closest.t) { t_max = lh.t; closest = lh; }
        } else {
            if ((size_t)n + 2 > next.size()) next.resize(next.size() * 2);
            next[n++] = node.left;
            next[n++] = node.right;
        }
    }
    return closest;
}

/* mesh.jl:115-163 (CPU: t_max is NOT narrowed) */
Hit hit_mesh_tree(const ref_scene& S, const Tree& tree, const Ray& r, real t_min, real t_max, const Ctx& c) {
    if (tree.nodes.empty()) return Hit();
    std::vector<int> next(64);
    int n = 1;
    next[0] = 1;
    int active_mesh_index = 0;
    Hit closest;
    while (n > 0) {
        if (active_mesh_index == n) {
            active_mesh_index = 0;
            n -= 1;
            continue;
        }
        const Node& node = tree.nodes[next[n - 1] - 1];
        n -= 1;
        if (c.st) c.st->node_visits++;
        if (!hit_aabb(node.box, r, t_min, t_max)) continue;
        if ((size_t)n + 2 > next.size()) next.resize(next.size() * 2);
        if (node.leaf) {
            if (active_mesh_index == 0) {
                next[n++] = node.right;
                active_mesh_index = n;
                next[n++] = node.left;
            } else {
                int oi = tree.objects[next[active_mesh_index - 1] - 1];
                const MeshData& m = S.meshes[S.objects[oi].aux];
                Hit lh; /* mesh.jl:61-71 */
                real tm = t_max;
                for (int i = node.left; i <= node.right; ++i) {
                    if (c.st) c.st->prim_tests[REF_MESH]++;
                    Hit h = hit_mesh_triangle(m, i, r, t_min, tm);
                    if (h.t < lh.t) { lh = h; tm = lh.t; }
                }
                if (lh.t < closest.t) { closest = lh; closest.prim = S.top_pos[oi]; }
            }
        } else {
            next[n++] = node.left;
            next[n++] = node.right;
        }
    }
    return closest;
}

/* scene_manager.jl:40-53 */
Hit hit_scene(const ref_scene& S, const Ray& r, real t_min, real t_max, const Ctx& c) {
    if (c.st) c.st->rays++;
    Hit a;
    real closest_t = t_max;
    bool first = true;
    for (const Bucket& b : S.buckets) {
        Hit h = b.is_mesh ? hit_mesh_tree(S, b.tree, r, t_min, closest_t, c) : hit_tree(S, b.tree, r, t_min, closest_t, c, false);
        if (first) { a = h; first = false; continue; }
        if (a.t <= h.t) { /* keep a */ }
        else { closest_t = h.t; a = h; }
    }
    return a;
}

/* ------------------------------------------------------ perlin.jl / texture.jl */
real perlin_noise(const refPerlin& P, V3 p) { /* perlin.jl:48-85 */
    real fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    V3 c[2][2][2];
    for (int dk = 1; dk <= 2; ++dk)
        for (int dj = 1; dj <= 2; ++dj)
            for (int di = 1; di <= 2; ++di) {
                int idx = P.perm_x[(i + di) & 255] ^ P.perm_y[(j + dj) & 255] ^ P.perm_z[(k + dk) & 255];
                c[di - 1][dj - 1][dk - 1] = v3(P.ranvec[3 * idx], P.ranvec[3 * idx + 1], P.ranvec[3 * idx + 2]);
            }
    real uu = u * u * (R(3.0) - R(2.0) * u);
    real vv = v * v * (R(3.0) - R(2.0) * v);
    real ww = w * w * (R(3.0) - R(2.0) * w);
    real accum = R(0.0);
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                V3 weight_v = v3(u - (real)ii, v - (real)jj, w - (real)kk);
                real fi = (real)ii, fj = (real)jj, fk = (real)kk;
                accum += (fi * uu + (R(1.0) - fi) * (R(1.0) - uu)) * (fj * vv + (R(1.0) - fj) * (R(1.0) - vv)) *
                         (fk * ww + (R(1.0) - fk) * (R(1.0) - ww)) * dot(c[ii][jj][kk], weight_v);
            }
    return accum;
}
real perlin_noise_scalar(const refPerlin& P, V3 p) { /* perlin.jl:48-68,87-97 with Perlin3{Float32}: ranvec[3*i] holds the scalar */
    real fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    real accum = R(0.0);
    for (int kk = 0; kk <= 1; ++kk)
        for (int jj = 0; jj <= 1; ++jj)
            for (int ii = 0; ii <= 1; ++ii) {
                int idx = P.perm_x[(i + ii + 1) & 255] ^ P.perm_y[(j + jj + 1) & 255] ^ P.perm_z[(k + kk + 1) & 255];
                real fi = (real)ii, fj = (real)jj, fk = (real)kk;
                accum += (fi * u + (R(1.0) - fi) * (R(1.0) - u)) * (fj * v + (R(1.0) - fj) * (R(1.0) - v)) *
                         (fk * w + (R(1.0) - fk) * (R(1.0) - w)) * P.ranvec[3 * idx];
            }
    return accum;
}
real turbulence(const refPerlin& P, V3 p) { /* perlin.jl:35-46 depth 7 */
    real accum = R(0.0), weight = R(1.0);
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(P, p);
        weight *= R(0.5);
        p = p * R(2.0);
    }
    return std::fabs(accum);
}

V3 texture_value(const ref_scene& S, int ti, real u, real v, V3 p) { /* texture.jl:42-100 */
    const refTexture& T = S.textures[ti];
    switch (T.kind) {
        case REF_TEX_SOLID: return v3(T.color[0], T.color[1], T.color[2]);
        case REF_TEX_CHECKER: {
            real sines = det_sin(R(10.0) * p.x) * det_sin(R(10.0) * p.y) * det_sin(R(10.0) * p.z);
            return texture_value(S, sines < R(0.0) ? T.a : T.b, u, v, p);
        }
        case REF_TEX_IMAGE: {
            const refImage& im = S.images[T.a];
            if (im.width == 0 || im.height == 0) return v3(0, 1, 1);
            u = u < R(0.0) ? R(0.0) : (u > R(1.0) ? R(1.0) : u);
            v = v < R(0.0) ? R(0.0) : (v > R(1.0) ? R(1.0) : v);
            v = R(1.0) - v;
            int i = (int)(v * (real)im.height); /* trunc */
            int j = (int)(u * (real)im.width);
            if (i == 0) i = 1;
            if (j == 0) j = 1;
            const uint8_t* px = &S.image_data[T.a][3 * ((size_t)(i - 1) * im.width + (j - 1))];
            return v3((real)px[0] / R(255.0), (real)px[1] / R(255.0), (real)px[2] / R(255.0));
        }
        case REF_TEX_NOISE_MARBLE: {
            /* texture.jl:98-100; the reference evaluates the sine in Float64 (scale is Float64); the oracle and
             * the kernels use the shared real sine, a difference of a few 1e-7 in the returned value */
            real arg = T.scale * p.z + R(10.0) * turbulence(S.perlins[T.a], p);
            real val = R(0.5) * (R(1.0) + det_sin(arg));
            return v3(val);
        }
        case REF_TEX_NOISE_SCALAR: /* texture.jl:79-81 */
            return v3(R(1.0)) * perlin_noise_scalar(S.perlins[T.a], T.scale * p);
    }
    return v3(0);
}

/* ----------------------------------------------------- material.jl:23-38 */
inline V3 reflect(V3 v, V3 n) { return v - (R(2.0) * dot(v, n)) * n; }
inline V3 refract(V3 direction, V3 normal, real etai_over_etat) {
    real cos_theta = dot(-direction, normal);
    V3 perp = etai_over_etat * (direction + cos_theta * normal);
    V3 parallel = (-std::sqrt(std::fabs(R(1.0) - dot(perp, perp)))) * normal;
    return perp + parallel;
}
inline real schlick(real cos_theta, real eta) {
    real k = (R(1.0) - eta) / (R(1.0) + eta);
    real r0 = k * k;
    real x = R(1.0) - cos_theta;
    real x2 = x * x;
    real x5 = (x2 * x2) * x;
    return r0 + (R(1.0) - r0) * x5;
}

/* hittable.jl:36-75 */
void scatter(const ref_scene& S, const Ray& ray, const Hit& h, const Ctx& c, Ray& scattered, V3& attenuation) {
    const refMaterial& m = S.materials[h.material];
    if (c.st) c.st->scatters[m.type]++;
    if (m.type == REF_LAMBERTIAN) {
        V3 dir = h.n + normalize(random_in_unit_sphere(c));
        if (almost_zero(dir)) dir = h.n;
        scattered = Ray{h.p, dir, ray.time};
        attenuation = texture_value(S, m.texture, h.u, h.v, h.p);
    } else if (m.type == REF_METAL) {
        V3 reflected = reflect(normalize(ray.d), h.n);
        scattered = Ray{h.p, reflected + m.fuzz * random_in_unit_sphere(c), ray.time};
        real keep = dot(scattered.d, h.n) > R(0.0) ? R(1.0) : R(0.0);
        V3 a = texture_value(S, m.texture, h.u, h.v, h.p);
        attenuation = keep != R(0.0) ? a : v3(0); /* Bool * Vec3: false is a strong zero */
    } else if (m.type == REF_DIELECTRIC) {
        real ratio = h.front ? (R(1.0) / m.eta) : m.eta;
        V3 unit_dir = normalize(ray.d);
        real cos_theta = std::fmin(dot(-unit_dir, h.n), R(1.0));
        real sin_theta = std::sqrt(R(1.0) - cos_theta * cos_theta);
        V3 dir;
        /* || short-circuits: the random number is only drawn when total reflection does not apply */
        if ((ratio * sin_theta > R(1.0)) || (schlick(cos_theta, m.eta) > rand_scalar(c))) dir = reflect(unit_dir, h.n);
        else dir = refract(unit_dir, h.n, ratio);
        scattered = Ray{h.p, dir, ray.time};
        attenuation = v3(R(1.0));
    } else if (m.type == REF_ISOTROPIC) {
        scattered = Ray{h.p, random_in_unit_sphere(c), ray.time};
        attenuation = texture_value(S, m.texture, h.u, h.v, h.p);
    } else {
        scattered = Ray{v3(0), v3(0), R(0.0)};
        attenuation = v3(0);
    }
}
/* hittable.jl:77-83 */
V3 emit(const ref_scene& S, const Hit& h) {
    const refMaterial& m = S.materials[h.material];
    if (m.type == REF_EMISSIVE) return texture_value(S, m.texture, h.u, h.v, h.p);
    return v3(0);
}

/* Ray_Tracing.jl:11-29 */
V3 trace(const ref_scene& S, Ray ray, V3 background, int depth, Ctx& c) {
    V3 acc = v3(R(1.0));
    V3 color = v3(R(0.0));
    c.bounce = 0;
    while (depth > 0) {
        if (c.ray_log) c.ray_log->push_back(ray);
        Hit h = hit_scene(S, ray, R(0.001), INF, c);
        if (h.t == INF) return color + acc * background;
        Ray scattered;
        V3 attenuation;
        scatter(S, ray, h, c, scattered, attenuation);
        ray = scattered;
        color = color + acc * emit(S, h);
        if (almost_zero(attenuation)) return color;
        acc = acc * attenuation;
        depth -= 1;
        c.bounce += 1;
    }
    return color;
}

/* camera.jl:38-42 */
Ray get_ray(const refCamera& cam, real s, real t, Ctx& c) {
    c.bounce = CAMERA_BOUNCE;
    V3 u = v3(cam.u[0], cam.u[1], cam.u[2]), v = v3(cam.v[0], cam.v[1], cam.v[2]);
    V3 origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    V3 llc = v3(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    V3 rd = cam.lens_radius * random_in_unit_disk(c);
    V3 offset_origin = (u * rd.x + v * rd.y) + origin;
    real time = rand_scalar(c) * (cam.t1 - cam.t0) + cam.t0;
    return Ray{offset_origin, ((llc + s * hor) + t * ver) - offset_origin, time};
}

V3 path_radiance(const ref_scene& S, const refCamera& cam, const real* pattern, int i, int j, int n, int width,
                 int height, V3 background, int max_depth, uint32_t seed32, Stats* st,
                 std::vector<Ray>* ray_log = nullptr) {
    /* Ray_Tracing.jl:83-88 */
    real dx = pattern[2 * n], dy = pattern[2 * n + 1];
    real s = ((real)i + dx) / (real)(width - 1);
    real t = ((real)j + dy) / (real)(height - 1);
    Ctx c;
    c.pixel = (uint32_t)j * (uint32_t)width + (uint32_t)i;
    c.sample = (uint32_t)n;
    c.seed32 = seed32;
    c.st = st;
    c.ray_log = ray_log;
    Ray r = get_ray(cam, s, t, c);
    return trace(S, r, background, max_depth, c);
}

std::string type_key(const ref_scene& S, int oi);
std::string tex_key(const ref_scene& S, int ti) {
    if (ti < 0) return "none";
    const refTexture& T = S.textures[ti];
    switch (T.kind) {
        case REF_TEX_SOLID: return "solid";
        case REF_TEX_CHECKER: return "checker<" + tex_key(S, T.a) + "," + tex_key(S, T.b) + ">";
        case REF_TEX_IMAGE: return "image";
        case REF_TEX_NOISE_MARBLE: return "noisev";
        default: return "noisef";
    }
}
std::string mat_key(const ref_scene& S, int mi) { return mi < 0 ? "none" : tex_key(S, S.materials[mi].texture); }
/* the concrete Julia type of an object, which is what SceneManager groups by (scene_manager.jl:9-19) */
std::string type_key(const ref_scene& S, int oi) {
    const refObject& o = S.objects[oi];
    switch (o.kind) {
        case REF_TRANSFORM: return "Transform<" + type_key(S, o.child) + ">";
        case REF_CONSTANT_MEDIUM: return "ConstantMedium<" + mat_key(S, o.material) + "," + type_key(S, o.child) + ">";
        case REF_BVH: {
            const Tree& g = S.groups[o.aux];
            std::string k = type_key(S, g.objects[0]);
            for (int x : g.objects) if (type_key(S, x) != k) k = "Hittable";
            return "BVHTree<" + k + ">";
        }
        case REF_MESH: {
            const MeshData& m = S.meshes[o.aux];
            return "Mesh<" + (m.materials.empty() ? std::string("none") : mat_key(S, m.materials[0])) + ">";
        }
        default: return "K" + std::to_string(o.kind) + "<" + mat_key(S, o.material) + ">";
    }
}

} // namespace

/* ------------------------------------------------------------------ C API */
extern "C" {

const char* ref_last_error(void) { return g_err.c_str(); }

int ref_scene_create(const refWorld* w, ref_scene** out) {
    if (!w || !out) { g_err = "null argument"; return -1; }
    ref_scene* S = new ref_scene();
    S->objects.assign(w->objects, w->objects + w->n_objects);
    S->top.assign(w->top, w->top + w->n_top);
    S->top_pos.assign(w->n_objects, -1);
    for (int i = 0; i < w->n_top; ++i) S->top_pos[S->top[i]] = i;
    if (w->n_transforms) S->transforms.assign(w->transforms, w->transforms + w->n_transforms);
    if (w->n_materials) S->materials.assign(w->materials, w->materials + w->n_materials);
    if (w->n_textures) S->textures.assign(w->textures, w->textures + w->n_textures);
    for (int i = 0; i < w->n_images; ++i) {
        refImage im = w->images[i];
        S->image_data.emplace_back(im.rgb, im.rgb + (size_t)3 * im.width * im.height);
        S->images.push_back(im);
    }
    if (w->n_perlins) S->perlins.assign(w->perlins, w->perlins + w->n_perlins);
    for (int i = 0; i < w->n_meshes; ++i) {
        const refMesh& m = w->meshes[i];
        MeshData d;
        d.nvert = m.n_vertices; d.ntri = m.n_triangles;
        d.vertices.assign(m.vertices, m.vertices + (size_t)8 * m.n_vertices);
        d.indices.assign(m.indices, m.indices + (size_t)4 * m.n_triangles);
        d.materials.assign(m.materials, m.materials + m.n_materials);
        d.order.resize(m.n_triangles);
        for (int k = 0; k < m.n_triangles; ++k) d.order[k] = k;
        S->meshes.push_back(std::move(d));
    }
    /* nested BVHTree objects are built where the scene constructs them (cover_the_next_week.jl:75) */
    S->groups.resize(w->n_groups);
    for (int g = 0; g < w->n_groups; ++g) {
        Tree& t = S->groups[g];
        t.objects.assign(w->group_items + w->groups[g].first, w->group_items + w->groups[g].first + w->groups[g].count);
    }
    for (int g = 0; g < w->n_groups; ++g) build_tree(*S, S->groups[g]);
    /* scene_manager.jl:9-19: one vector per concrete type, in order of first appearance */
    for (int oi : S->top) {
        std::string key = type_key(*S, oi);
        Bucket* b = nullptr;
        for (Bucket& x : S->buckets) if (x.key == key) b = &x;
        if (!b) {
            S->buckets.emplace_back();
            b = &S->buckets.back();
            b->key = key;
            b->is_mesh = S->objects[oi].kind == REF_MESH;
        }
        b->tree.objects.push_back(oi);
    }
    /* scene_manager.jl:25-38 */
    for (Bucket& b : S->buckets) {
        if (b.is_mesh) build_mesh_tree(*S, b.tree);
        else build_tree(*S, b.tree);
    }
    *out = S;
    return 0;
}

void ref_scene_destroy(ref_scene* s) { delete s; }

int ref_render(ref_scene* S, const refCamera* cam, const real* pattern, int n, const real background[3], int width,
               int height, int max_depth, uint64_t seed, int threads, real* out, refStats* stats) {
    if (!S || !cam || !pattern || !out || n <= 0 || width <= 1 || height <= 1) { g_err = "bad argument"; return -1; }
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    V3 bg = v3(background[0], background[1], background[2]);
    uint32_t seed32 = fold_seed(seed);
    std::atomic<int> next_col(0);
    std::vector<Stats> tstats(threads);
    auto t0 = std::chrono::steady_clock::now();
    /* Ray_Tracing.jl:78: Threads.@threads over columns */
    auto worker = [&](int tid) {
        Stats* st = stats ? &tstats[tid] : nullptr;
        for (;;) {
            int i = next_col.fetch_add(1);
            if (i >= width) break;
            for (int j = height - 1; j >= 0; --j) {
                V3 col = v3(0, 0, 0);
                for (int k = 0; k < n; ++k)
                    col = col + path_radiance(*S, *cam, pattern, i, j, k, width, height, bg, max_depth, seed32, st);
                col = col * (real)(1.0 / (double)n); /* vec3.jl:31 with an Int divisor */
                col = v3(std::sqrt(col.x), std::sqrt(col.y), std::sqrt(col.z));
                size_t idx = ((size_t)i * height + (size_t)(height - 1 - j)) * 3; /* img[height-j, i+1] */
                out[idx] = col.x; out[idx + 1] = col.y; out[idx + 2] = col.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->paths = (uint64_t)width * height * n;
        for (const Stats& s : tstats) {
            stats->rays += s.rays;
            stats->node_visits += s.node_visits;
            for (int k = 0; k < 12; ++k) stats->prim_tests[k] += s.prim_tests[k];
            for (int k = 0; k < 6; ++k) stats->scatters[k] += s.scatters[k];
        }
    }
    return 0;
}

int ref_trace_paths(ref_scene* S, const refCamera* cam, const real* pattern, int n, const real background[3],
                    int width, int height, int max_depth, uint64_t seed, const int32_t* ids, int n_paths,
                    real* radiance) {
    if (!S || !cam || !pattern || !ids || !radiance) { g_err = "bad argument"; return -1; }
    V3 bg = v3(background[0], background[1], background[2]);
    uint32_t seed32 = fold_seed(seed);
    for (int k = 0; k < n_paths; ++k) {
        int i = ids[3 * k], j = ids[3 * k + 1], s = ids[3 * k + 2];
        if (i < 0 || i >= width || j < 0 || j >= height || s < 0 || s >= n) { g_err = "path id out of range"; return -1; }
        V3 c = path_radiance(*S, *cam, pattern, i, j, s, width, height, bg, max_depth, seed32, nullptr);
        radiance[3 * k] = c.x; radiance[3 * k + 1] = c.y; radiance[3 * k + 2] = c.z;
    }
    return 0;
}

/* The rays (origin, direction, time) that one path casts, bounce by bounce: for finding the bounce at which
 * another implementation of the same path parts from this one. */
int ref_path_rays(ref_scene* S, const refCamera* cam, const real* pattern, int n, const real background[3], int width,
                  int height, int max_depth, uint64_t seed, int i, int j, int s, real* rays, int max_rays) {
    if (!S || !cam || !pattern || !rays) { g_err = "bad argument"; return -1; }
    if (i < 0 || i >= width || j < 0 || j >= height || s < 0 || s >= n) { g_err = "path id out of range"; return -1; }
    std::vector<Ray> log;
    path_radiance(*S, *cam, pattern, i, j, s, width, height, v3(background[0], background[1], background[2]),
                  max_depth, fold_seed(seed), nullptr, &log);
    int m = (int)log.size() < max_rays ? (int)log.size() : max_rays;
    for (int k = 0; k < m; ++k) {
        const Ray& r = log[k];
        real* q = &rays[7 * k];
        q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = r.d.x; q[4] = r.d.y; q[5] = r.d.z; q[6] = r.time;
    }
    return m;
}

int ref_trace_rays(ref_scene* S, const real* rays, int n, real tmin, real tmax, uint64_t seed, int32_t* prim_id,
                   int32_t* sub_id, real* t, real* normal, real* uv, uint8_t* front) {
    if (!S || !rays) { g_err = "bad argument"; return -1; }
    uint32_t seed32 = fold_seed(seed);
    for (int k = 0; k < n; ++k) {
        const real* q = &rays[7 * k];
        Ray r{v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Ctx c;
        c.pixel = (uint32_t)k; c.sample = 0; c.seed32 = seed32; c.bounce = 0;
        Hit h = hit_scene(*S, r, tmin, tmax, c);
        if (prim_id) prim_id[k] = h.t == INF ? -1 : h.prim;
        if (sub_id) sub_id[k] = h.t == INF ? -1 : h.sub;
        if (t) t[k] = h.t;
        if (normal) { normal[3 * k] = h.n.x; normal[3 * k + 1] = h.n.y; normal[3 * k + 2] = h.n.z; }
        if (uv) { uv[2 * k] = h.u; uv[2 * k + 1] = h.v; }
        if (front) front[k] = h.front ? 1 : 0;
    }
    return 0;
}

int ref_bucket_count(ref_scene* S) { return S ? (int)S->buckets.size() : -1; }

int ref_bucket_info(ref_scene* S, int b, int32_t* n_objects, int32_t* n_node_slots, int32_t* is_mesh) {
    if (!S || b < 0 || b >= (int)S->buckets.size()) return -1;
    if (n_objects) *n_objects = (int32_t)S->buckets[b].tree.objects.size();
    if (n_node_slots) *n_node_slots = (int32_t)S->buckets[b].tree.nodes.size();
    if (is_mesh) *is_mesh = S->buckets[b].is_mesh ? 1 : 0;
    return 0;
}

int ref_bucket_bvh(ref_scene* S, int b, int32_t* sorted_objects, real* bounds, int32_t* left, int32_t* right,
                   uint8_t* leaf, uint8_t* valid) {
    if (!S || b < 0 || b >= (int)S->buckets.size()) return -1;
    const Tree& t = S->buckets[b].tree;
    for (size_t i = 0; i < t.objects.size(); ++i)
        if (sorted_objects) sorted_objects[i] = S->top_pos[t.objects[i]];
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        const Node& n = t.nodes[i];
        if (bounds) {
            bounds[6 * i] = n.box.mn.x; bounds[6 * i + 1] = n.box.mn.y; bounds[6 * i + 2] = n.box.mn.z;
            bounds[6 * i + 3] = n.box.mx.x; bounds[6 * i + 4] = n.box.mx.y; bounds[6 * i + 5] = n.box.mx.z;
        }
        if (left) left[i] = n.left;
        if (right) right[i] = n.right;
        if (leaf) leaf[i] = n.leaf ? 1 : 0;
        if (valid) valid[i] = t.valid[i];
    }
    return 0;
}

int ref_mesh_order(ref_scene* S, int mesh, int32_t* tri_order) {
    if (!S || mesh < 0 || mesh >= (int)S->meshes.size()) return -1;
    const MeshData& m = S->meshes[mesh];
    for (int k = 0; k < m.ntri; ++k) tri_order[k] = m.order[k];
    return 0;
}

/* MSAA_patterns.jl:25-43 */
static void halton(int N, int b, real* seq, int stride) {
    long long n = 0, d = 1;
    for (int i = 0; i < N; ++i) {
        long long x = d - n;
        if (x == 1) {
            n = 1;
            d *= b;
        } else {
            long long y = d / b;
            while (x <= y) y /= b;
            n = (b + 1) * y - x;
        }
        seq[(size_t)i * stride] = (real)n / (real)d; /* T(n) / d with T = Float32 */
    }
}
/* MSAA_patterns.jl:50-58 */
void ref_halton_sequence32(int n, real* out) {
    halton(n, 2, out, 2);
    halton(n, 3, out + 1, 2);
    for (int i = 0; i < 2 * n; ++i) out[i] -= R(0.5);
}

void ref_pcg4d(uint32_t v[4]) { pcg4d(v); }
real ref_rand01(uint32_t pixel, uint32_t sample, uint32_t counter, uint32_t seed32, int lane) {
    uint32_t v[4] = {pixel, sample, counter, seed32};
    pcg4d(v);
#ifdef REF_F64
    return u64_to_01(v[(2 * lane) & 3], v[(2 * lane + 1) & 3]); /* lane 0 or 1: the two numbers of one hash */
#else
    return u32_to_01(v[lane & 3]);
#endif
}
real ref_logf(real x) { return det_log(x); }
real ref_sinf(real x) { return det_sin(x); }
real ref_acosf(real x) { return det_acos(x); }
real ref_atan2f(real y, real x) { return det_atan2(y, x); }

} // extern "C"
