/*
 * rt_ref.h -- CPU ORACLE for the path-tracing hot path of rafaellg79/raytracing.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain C++ restatement of
 * the reference's *CPU* algorithm (render -> draw! -> trace -> SceneManager/BVH hit
 * -> primitive hit -> scatter/emit), written from the Julia sources, one function
 * per reference function, each citing the file:line it follows.  Only tests/, the
 * smoke check and the cpu_baseline leg of bench.py may load it.  The shipped HIP
 * library never links, includes or calls anything in this directory.
 *
 * Pinning status: the reference has no tests, golden vectors or fixtures and its
 * toolchain (Julia + CUDA.jl) is absent, so the oracle cannot be checked against an
 * execution of the reference: sample-level parity is UNPINNED by construction (the
 * reference never seeds its RNG).  What pins it: hand-derived known-answer tests for
 * every formula (tests/test_oracle_kat.py) and the statistics of the two
 * deterministic gallery images (outputs/cornell_box_*.png, cornell_smoke_box_*.png)
 * committed as numbers under tests/golden/.
 *
 * Randomness: the reference draws from an unseeded task-local RNG.  The oracle and
 * the HIP kernels both draw from the same stateless counter-based generator
 * (pcg4d of Jarzynski & Olano, JCGT 2020) keyed by (pixel, sample, bounce, purpose,
 * trial, seed), so that a path's random numbers do not depend on thread, GPU or
 * traversal order.
 *
 * All structs below are plain C, shared layout with include/rt_amd.h (the product
 * ABI) so one host-side flattener can feed both.
 */
#ifndef RT_REF_H
#define RT_REF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The element type F of render(F, ...) (main.jl:21,46).  The file is compiled twice: as it stands (Float32, the
 * reference's default, _build/librt_ref.so) and with -DREF_F64 (Float64, _build/librt_ref64.so: same names, every
 * ref_real below a double, random numbers of 53 bits). */
#ifdef REF_F64
typedef double ref_real;
#else
typedef float ref_real;
#endif

/* object kinds: one per concrete Hittable of the reference */
enum {
    REF_SPHERE = 1,          /* src/sphere.jl:3   p = cx,cy,cz,r */
    REF_MOVING_SPHERE = 2,   /* src/sphere.jl:12  p = c0(3),t0,c1(3),t1,r */
    REF_XY_RECT = 3,         /* src/aarect.jl:3   p = x0,x1,y0,y1,k */
    REF_XZ_RECT = 4,         /* src/aarect.jl:13  p = x0,x1,z0,z1,k */
    REF_YZ_RECT = 5,         /* src/aarect.jl:23  p = y0,y1,z0,z1,k */
    REF_BOX = 6,             /* src/box.jl:3      p = p0(3),p1(3) */
    REF_TRANSFORM = 7,       /* src/transform.jl:3  child = object, aux = transform index */
    REF_CONSTANT_MEDIUM = 8, /* src/constant_medium.jl:3  child = boundary, material = phase fn, p[0] = neg_inv_density */
    REF_BVH = 9,             /* src/BVH.jl:12 nested tree (samples/cover_the_next_week.jl:75) aux = group */
    REF_MESH = 10            /* src/mesh.jl:5     aux = mesh index */
};

/* material types, src/material.jl:3-9 */
enum { REF_LAMBERTIAN = 1, REF_METAL = 2, REF_DIELECTRIC = 3, REF_EMISSIVE = 4, REF_ISOTROPIC = 5 };

/* texture kinds, src/texture.jl:8-37 */
enum { REF_TEX_SOLID = 0, REF_TEX_CHECKER = 1, REF_TEX_IMAGE = 2, REF_TEX_NOISE_MARBLE = 3, REF_TEX_NOISE_SCALAR = 4 };

typedef struct refObject {
    int32_t kind;
    int32_t material; /* index into materials, -1 if none */
    int32_t child;    /* index into objects (Transform.object / ConstantMedium.boundary), -1 if none */
    int32_t aux;      /* transform / group / mesh index */
    ref_real p[12];
} refObject;

typedef struct refTransform {
    ref_real xform[12]; /* 3x4 column-major, src/transform.jl:65 */
    ref_real inv[12];
    ref_real bbox[6];   /* cached world AABB min(3), max(3), src/transform.jl:47-63 */
} refTransform;

typedef struct refMaterial {
    int32_t type;
    int32_t texture;
    ref_real fuzz;
    ref_real eta;
} refMaterial;

typedef struct refTexture {
    int32_t kind;
    int32_t a; /* checker: odd texture; image: image index; noise: perlin index */
    int32_t b; /* checker: even texture */
    int32_t pad;
    ref_real color[3];
    ref_real scale;
} refTexture;

typedef struct refImage {
    int32_t width, height;
    const uint8_t* rgb; /* row-major, row 0 = top, 3 bytes per texel */
} refImage;

typedef struct refPerlin {
    ref_real ranvec[256 * 3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
} refPerlin;

typedef struct refMesh {
    int32_t n_vertices, n_triangles, n_materials, pad;
    const ref_real* vertices;    /* 8 floats per vertex: position, normal, u, v  (src/vertex.jl:3-8) */
    const int32_t* indices;   /* 4 x n_triangles column-major: 3 vertex ids (0-based) + local material id */
    const int32_t* materials; /* local material id -> global material index */
    const int32_t* unused;    /* keeps the layout equal to the product ABI's mesh record */
} refMesh;

typedef struct refGroup {
    int32_t first, count; /* slice of group_items */
} refGroup;

typedef struct refWorld {
    int32_t n_objects;    const refObject* objects;
    int32_t n_top;        const int32_t* top; /* the `world` vector of render(): object indices in insertion order */
    int32_t n_transforms; const refTransform* transforms;
    int32_t n_materials;  const refMaterial* materials;
    int32_t n_textures;   const refTexture* textures;
    int32_t n_images;     const refImage* images;
    int32_t n_perlins;    const refPerlin* perlins;
    int32_t n_meshes;     const refMesh* meshes;
    int32_t n_groups;     const refGroup* groups;
    const int32_t* group_items;
} refWorld;

typedef struct refCamera { /* src/camera.jl:3-14 */
    ref_real origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    ref_real lens_radius, t0, t1;
} refCamera;

typedef struct refStats {
    double seconds;
    uint64_t paths, rays, node_visits;
    uint64_t prim_tests[12]; /* by object kind */
    uint64_t scatters[6];    /* by material type */
} refStats;

typedef struct ref_scene ref_scene;

int ref_scene_create(const refWorld* world, ref_scene** out);
void ref_scene_destroy(ref_scene*);
const char* ref_last_error(void);

/* src/Ray_Tracing.jl:71-100: out is 3*H*W floats in the memory order of a Julia
 * Array{RGB{Float32}}(H,W): index ((col*H) + row)*3 + c, row 0 = top. */
int ref_render(ref_scene*, const refCamera*, const ref_real* pattern /*2 x n col-major*/, int n,
               const ref_real background[3], int width, int height, int max_depth, uint64_t seed,
               int threads, ref_real* out, refStats* stats);

/* radiance of single paths: ids = n_paths triples (i = column, j = row from bottom, n = sample, 0-based) */
int ref_trace_paths(ref_scene*, const refCamera*, const ref_real* pattern, int n,
                    const ref_real background[3], int width, int height, int max_depth, uint64_t seed,
                    const int32_t* ids, int n_paths, ref_real* radiance /*3 x n_paths*/);

/* the rays one path casts, bounce by bounce (7 floats each: origin, direction, time); returns how many were
 * written (at most max_rays) or -1.  For finding the bounce at which another implementation parts from this one. */
int ref_path_rays(ref_scene*, const refCamera*, const ref_real* pattern, int n, const ref_real background[3],
                  int width, int height, int max_depth, uint64_t seed, int i, int j, int s, ref_real* rays,
                  int max_rays);

/* closest hit of fixed rays against the whole SceneManager (src/scene_manager.jl:40-53).
 * rays: 7 floats each (origin, direction, time).  prim_id = position in `top`, -1 on miss. */
int ref_trace_rays(ref_scene*, const ref_real* rays, int n, ref_real tmin, ref_real tmax, uint64_t seed,
                   int32_t* prim_id, int32_t* sub_id, ref_real* t, ref_real* normal /*3n*/, ref_real* uv /*2n*/,
                   uint8_t* front);

/* bucket / BVH introspection (src/scene_manager.jl:9-19, src/BVH.jl:65-109, src/mesh.jl:217-305) */
int ref_bucket_count(ref_scene*);
int ref_bucket_info(ref_scene*, int bucket, int32_t* n_objects, int32_t* n_node_slots, int32_t* is_mesh);
int ref_bucket_bvh(ref_scene*, int bucket, int32_t* sorted_objects, ref_real* bounds /*6 per slot*/,
                   int32_t* left, int32_t* right, uint8_t* leaf, uint8_t* valid);
/* triangle order of a mesh after the build permuted its index matrix (src/mesh.jl:237):
 * tri_order[k] = original 0-based triangle id now stored in column k */
int ref_mesh_order(ref_scene*, int mesh, int32_t* tri_order);

/* src/MSAA_patterns.jl:25-58 */
void ref_halton_sequence32(int n, ref_real* out /*2 x n col-major*/);

/* the shared deterministic generator and math kernels */
void ref_pcg4d(uint32_t v[4]);
ref_real ref_rand01(uint32_t pixel, uint32_t sample, uint32_t counter, uint32_t seed32, int lane);
ref_real ref_logf(ref_real x);
ref_real ref_sinf(ref_real x);
ref_real ref_acosf(ref_real x);
ref_real ref_atan2f(ref_real y, ref_real x);

#ifdef __cplusplus
}
#endif
#endif
