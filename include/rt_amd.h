/*
 * rt_amd.h -- C ABI of librt_amd.so, the MI355X (gfx950) path-tracing backend that
 * replaces the CUDA.jl path of rafaellg79/raytracing at its single device seam:
 * the `gpu=true` branch of render() (main.jl:34-38) -> draw!(::CuArray, ...)
 * (src/Ray_Tracing.jl:141-157).  Everything below that seam (draw_kernel, cu_trace,
 * cu_hit, scatter, emit; src/Ray_Tracing.jl:51-69,102-139, src/cuscene_manager.jl,
 * src/cuBVH.jl, src/mesh.jl:165-215) is implemented by hand-written HIP kernels.
 *
 * Conventions: every entry point is extern "C", takes plain pointers and sizes,
 * returns 0 (RT_OK) or a negative rtStatus, never throws, never calls back into the
 * host language and never exits the process.  rt_last_error() gives the message of
 * the last failure on the calling thread.  The library copies every input buffer
 * during rt_scene_create(); the caller owns all inputs and may free them on return.
 * One rtScene lives on one GPU (one process per GPU; spp shards are summed by the
 * host with an RCCL all-reduce between rt_render_accumulate and rt_finalize).
 * A handle is not re-entrant; distinct handles may be used concurrently.
 *
 * Pixel layout of every image buffer: the memory order of a Julia
 * Array{RGB{Float32},2}(H, W): float index ((col*H) + row)*3 + c, row 0 = top
 * (src/Ray_Tracing.jl:92,130-132).
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rtStatus {
    RT_OK = 0,
    RT_EINVAL = -1,
    RT_ENODEV = -2,
    RT_ENOMEM = -3,
    RT_EHIP = -4,
    RT_EUNSUPPORTED = -5,
    RT_ERCCL = -6
} rtStatus;

/* object kinds: one per concrete Hittable of the reference */
enum {
    RT_SPHERE = 1,          /* src/sphere.jl:3   p = cx,cy,cz,r */
    RT_MOVING_SPHERE = 2,   /* src/sphere.jl:12  p = c0(3),t0,c1(3),t1,r */
    RT_XY_RECT = 3,         /* src/aarect.jl:3   p = x0,x1,y0,y1,k */
    RT_XZ_RECT = 4,         /* src/aarect.jl:13  p = x0,x1,z0,z1,k */
    RT_YZ_RECT = 5,         /* src/aarect.jl:23  p = y0,y1,z0,z1,k */
    RT_BOX = 6,             /* src/box.jl:3      p = p0(3),p1(3) */
    RT_TRANSFORM = 7,       /* src/transform.jl:3  child = wrapped object, aux = transform index */
    RT_CONSTANT_MEDIUM = 8, /* src/constant_medium.jl:3  child = boundary, material = phase function, p[0] = neg_inv_density */
    RT_BVH = 9,             /* src/BVH.jl:12 used as an object (samples/cover_the_next_week.jl:75,82)  aux = group index */
    RT_MESH = 10            /* src/mesh.jl:5     aux = mesh index */
};

/* src/material.jl:3-9 */
enum { RT_LAMBERTIAN = 1, RT_METAL = 2, RT_DIELECTRIC = 3, RT_EMISSIVE = 4, RT_ISOTROPIC = 5 };

/* src/texture.jl:8-37 */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2, RT_TEX_NOISE_MARBLE = 3, RT_TEX_NOISE_SCALAR = 4 };

typedef struct rtObject {
    int32_t kind;
    int32_t material; /* index into materials, -1 if none */
    int32_t child;    /* index into objects, -1 if none */
    int32_t aux;
    float p[12];
} rtObject;

typedef struct rtTransform { /* src/transform.jl:3-8 */
    float xform[12];         /* 3x4 column-major 12-tuple, src/transform.jl:65 */
    float inv_xform[12];
    float bbox[6];           /* min(3), max(3) */
} rtTransform;

typedef struct rtMaterial { /* src/material.jl:11-16 */
    int32_t type;
    int32_t texture;
    float fuzz;
    float eta;
} rtMaterial;

typedef struct rtTexture {
    int32_t kind;
    int32_t a; /* checker: odd texture; image: image index; noise: perlin index */
    int32_t b; /* checker: even texture */
    int32_t pad;
    float color[3];
    float scale;
} rtTexture;

typedef struct rtImage { /* src/texture.jl:23-25, decoded by the host */
    int32_t width, height;
    const uint8_t* rgb; /* row-major, row 0 = top, 3 bytes per texel */
} rtImage;

typedef struct rtPerlin { /* src/perlin.jl:5-10 with point_count = 256 */
    float ranvec[256 * 3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
} rtPerlin;

typedef struct rtMesh { /* src/mesh.jl:5-9 */
    int32_t n_vertices, n_triangles, n_materials, pad;
    const float* vertices;    /* 8 floats per vertex: position, normal, u, v (src/vertex.jl:3-8) */
    const int32_t* indices;   /* 4 x n_triangles column-major, 0-based, in the order the BVH build left them (src/mesh.jl:237) */
    const int32_t* materials; /* local material id -> index into materials */
    const int32_t* tri_ids;   /* optional: original triangle id of each column (reported as sub_id); NULL = identity */
} rtMesh;

/* One BVHTree as the host built it (src/BVH.jl:65-109, src/mesh.jl:217-305): node ids are
 * 1-based, slot i-1 describes node i, unused slots are never reached from node 1.
 * kind 0: leaf.left == leaf.right == 1-based position in `objects`.
 * kind 1 (vector of meshes): a mesh-level leaf has left = id of the mesh's triangle root and
 * right = 1-based position of the mesh in `objects`; a triangle-level leaf has left..right =
 * 1-based inclusive range of columns of that mesh's index matrix. */
typedef struct rtBvh {
    int32_t kind;
    int32_t n_objects;
    const int32_t* objects;   /* object indices in tree order (tree.objects after the build permuted it) */
    int32_t n_nodes;          /* number of node slots */
    int32_t pad;
    const float* node_bounds; /* 6 per slot: min(3), max(3) */
    const int32_t* node_left;
    const int32_t* node_right;
    const uint8_t* node_leaf;
} rtBvh;

typedef struct rtSceneDesc {
    int32_t n_objects;    const rtObject* objects;
    int32_t n_top;        const int32_t* top;      /* the `world` vector of render(): prim_id = position here */
    int32_t n_transforms; const rtTransform* transforms;
    int32_t n_materials;  const rtMaterial* materials;
    int32_t n_textures;   const rtTexture* textures;
    int32_t n_images;     const rtImage* images;
    int32_t n_perlins;    const rtPerlin* perlins;
    int32_t n_meshes;     const rtMesh* meshes;
    int32_t n_groups;     const rtBvh* groups;     /* BVHTree objects referenced by RT_BVH.aux */
    int32_t n_buckets;    const rtBvh* buckets;    /* SceneManager tuple, in order (src/scene_manager.jl:9-19,36-38) */
} rtSceneDesc;

typedef struct rtCamera { /* src/camera.jl:3-14, passed by value to draw_kernel in the reference */
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius, t0, t1;
} rtCamera;

typedef struct rtStats {
    double seconds;       /* device time of the accumulate + finalize kernels */
    uint64_t paths;       /* trace() calls = W*H*samples */
    uint64_t rays;        /* hit(scene, ray) calls; 0 unless RT_FLAG_COUNTERS */
    uint64_t node_visits; /* BVH nodes fetched and slab-tested; 0 unless RT_FLAG_COUNTERS */
    uint64_t prim_tests;  /* leaf primitive tests; 0 unless RT_FLAG_COUNTERS */
    /* wave-level executions (64 lanes each) of the three blocks of the extend kernel: node step, leaf test,
     * retire/fetch.  Lane utilisation of a block = per-lane events / (64 * wave steps): node_visits, prim_tests
     * and rays respectively.  0 unless RT_FLAG_COUNTERS */
    uint64_t node_wave_steps, prim_wave_steps, bounce_wave_steps;
    /* device time (HIP events on the launch stream) and launch counts of the two kernels of a bounce;
     * 0 unless RT_FLAG_TIMING */
    double extend_seconds, shade_seconds;
    uint64_t extend_launches, shade_launches;
} rtStats;

#define RT_FLAG_COUNTERS 1u /* run the counting variant of the extend kernel (slower) */
#define RT_FLAG_TIMING 2u   /* bracket every extend / shade launch with HIP events (adds a little launch overhead) */

typedef struct rtScene rtScene;

int rt_abi_version(void);
const char* rt_last_error(void);
int rt_device_count(int* count);

/* replaces CuSceneManager(bvh) / cu(::BVHTree) (src/cuscene_manager.jl:16-19, src/cuBVH.jl:20-22,
 * src/mesh.jl:24-43): flattens the host trees into the device layout and uploads it to `device`. */
int rt_scene_create(const rtSceneDesc* desc, int device, rtScene** out);
int rt_scene_destroy(rtScene* scene);

/* replaces draw!(img::CuArray, world, camera; background, pattern) (src/Ray_Tracing.jl:141-157) on one GPU:
 * renders all n pattern columns, divides by n, takes the per-channel sqrt and copies the image to host
 * memory `out` (3*H*W floats).  Blocking.  flags: RT_FLAG_*. */
int rt_render(rtScene* scene, const rtCamera* camera, const float* pattern /* 2 x n column-major, host */, int n,
              const float background[3], int width, int height, int max_depth, uint64_t seed, uint32_t flags,
              float* out, rtStats* stats);

/* One host process, several GPUs.  The reference's render(...; gpu=true) is one process on one device (main.jl:34-38,
 * src/Ray_Tracing.jl:141-157); a group is what lets that one call use every GPU of a node: the scene is copied to each
 * device of the list (n_devices = 0: every visible device), rt_group_render deals the pattern columns to them (member g
 * of G traces columns g, g + G, ...; one host thread per member), sums the members' FP32 accumulators on the first device
 * -- with RCCL (ncclReduce over xGMI; librccl.so.1 is loaded when a group of more than one device first renders, and
 * RT_ERCCL reports its absence or failure), or with peer copies and adds in member order when the environment says
 * RT_AMD_REDUCE=peer or the list names a device twice (several shards on one GPU, for tests) -- finalises there and
 * copies the image to `out` like rt_render.  A group of one device renders exactly what rt_render renders, bit for bit;
 * with more members only the order of the FP32 sums differs.  Blocking; a group is not re-entrant. */
typedef struct rtGroup rtGroup;
int rt_group_create(const rtSceneDesc* desc, const int* devices, int n_devices, rtGroup** out);
int rt_group_size(const rtGroup* group, int* n_devices);
int rt_group_render(rtGroup* group, const rtCamera* camera, const float* pattern, int n, const float background[3],
                    int width, int height, int max_depth, uint64_t seed, uint32_t flags, float* out, rtStats* stats);
int rt_group_destroy(rtGroup* group);
/* No counterpart in the reference.  RT_OK if the RCCL library a group of several devices sums through can be loaded
 * (librccl.so.1 from the loader's path, or the file RT_AMD_RCCL_LIB names), RT_ERCCL with the loader's message if not:
 * what rt_group_render would answer on its first frame, without needing a GPU to ask.  The answer is kept for the life
 * of the process. */
int rt_rccl_available(void);

/* The two halves of rt_render for spp sharding across GPUs (one process per GPU):
 * rt_render_accumulate replaces the @cuda draw_kernel launch (src/Ray_Tracing.jl:151): it WRITES to the device
 * buffer d_accum (3*H*W floats) the un-normalised sum of trace() over pattern columns sample_first,
 * sample_first + sample_stride, ... < n (none if sample_first >= n: the buffer is zeroed).  rt_finalize replaces `img ./= N; sqrt` (src/Ray_Tracing.jl:152-153):
 * d_out[i] = sqrt(d_accum[i] * Float32(1/n_total)).  Both enqueue on `stream` (a hipStream_t, NULL = default
 * stream) and return without synchronising.  d_accum and d_out may alias. */
int rt_render_accumulate(rtScene* scene, const rtCamera* camera, const float* pattern, int n, int sample_first,
                         int sample_stride, const float background[3], int width, int height, int max_depth,
                         uint64_t seed, uint32_t flags, float* d_accum, void* stream);
int rt_finalize(rtScene* scene, const float* d_accum, int n_total, int width, int height, float* d_out, void* stream);
/* No counterpart in the reference.  rt_render_accumulate for callers that cannot allocate device memory: tight
 * serving loops, harnesses that forbid hipMalloc, code that must leave the device's memory map untouched (the call is
 * NOT capturable into a HIP graph: the frame loop is driven by the host, which reads a counter back and waits for the
 * stream after every batch of bounces).  Same arguments (no flags), same result
 * bit for bit, but the path pool, the list of slots on hold, the dispenser words, the pattern copy and the per-path
 * radiance buffer live in a fixed scratch array inside the library's code object (53 MiB, held by
 * every GPU the library is loaded on until the process ends), so the call makes no hipMalloc / hipFree and creates no
 * stream or event.  Limits (RT_EINVAL beyond them): n <= RT_STATIC_MAX_COLUMNS pattern columns, width * height <=
 * RT_STATIC_MAX_PIXELS; the pool has RT_STATIC_SLOTS slots, so large frames run slower than through
 * rt_render_accumulate.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two
 * threads, two scenes) race; calls on different GPUs do not.  The scene must have been created by rt_scene_create
 * (which allocates), and a camera far outside the scene makes the call rebuild the node boxes on the host first, like
 * rt_render_accumulate (a synchronous copy into memory the handle already owns). */
#define RT_STATIC_SLOTS (256u * 1024u)
#define RT_STATIC_MAX_COLUMNS 16384u
#define RT_STATIC_MAX_PIXELS (2u * 1024u * 1024u)
int rt_render_accumulate_static(rtScene* scene, const rtCamera* camera, const float* pattern, int n, int sample_first,
                                int sample_stride, const float background[3], int width, int height, int max_depth,
                                uint64_t seed, float* d_accum, void* stream);
/* counters of the last rt_render_accumulate with RT_FLAG_COUNTERS (synchronises the stream) */
int rt_get_counters(rtScene* scene, rtStats* stats);

/* Parity hooks.  rt_trace_rays: closest hit of fixed rays against the whole scene = hit(::SceneManager, ray,
 * t_min, t_max) (src/scene_manager.jl:40-53 / src/cuscene_manager.jl:48-50).  rays: 7 floats each (origin,
 * direction, time), host memory.  Outputs (host, any may be NULL): prim_id = position in `top` (-1 = miss),
 * sub_id = box side 0..5 / triangle id / nested object index, t, normal(3), uv(2), front_face. */
int rt_trace_rays(rtScene* scene, const float* rays, int n, float t_min, float t_max, uint64_t seed, int32_t* prim_id,
                  int32_t* sub_id, float* t, float* normal, float* uv, uint8_t* front);
/* radiance of single paths = trace(get_ray(camera, s, t), scene) (src/Ray_Tracing.jl:83-88) for n_paths
 * triples (i = column, j = row counted from the bottom, sample) -> 3 floats each, host memory. */
int rt_trace_paths(rtScene* scene, const rtCamera* camera, const float* pattern, int n, const float background[3],
                   int width, int height, int max_depth, uint64_t seed, const int32_t* ids, int n_paths,
                   float* radiance);

/* Host-side BVH builders for hosts that do not carry the reference's Julia builder.
 * rt_bvh_build restates build_bvh(objects) (src/BVH.jl:65-109) from the objects' bounding boxes
 * (6 floats each): order[k] = input index stored at sorted position k; node arrays have 2n-1 slots. */
int rt_bvh_build(int n, const float* object_bounds, int32_t* order, float* node_bounds, int32_t* node_left,
                 int32_t* node_right, uint8_t* node_leaf);
/* rt_bvh_build_meshes restates build_bvh(meshes::Vector{<:Mesh}) (src/mesh.jl:217-305).  indices[m] (4 x n_tri,
 * 0-based) is permuted in place like mesh.indices; tri_ids[m] receives the original triangle id of each column;
 * mesh_order[k] = input mesh stored at tree position k; node arrays have 4*sum(n_tri)-1 slots (unused slots
 * keep leaf = 255). */
int rt_bvh_build_meshes(int n_meshes, const int32_t* n_vertices, const float* const* vertices,
                        const int32_t* n_triangles, int32_t* const* indices, int32_t* const* tri_ids,
                        int32_t* mesh_order, float* node_bounds, int32_t* node_left, int32_t* node_right,
                        uint8_t* node_leaf);


/* The step after the path, which the reference leaves to Images.jl: `save(path, clamp01.(img))` (README.md:50-67).
 * rt_finalize_rgb8 = rt_finalize followed by clamp to [0, 1] (NaN -> 0) and N0f8 quantisation (round to nearest even of
 * x * 255) on the device; d_rgb8 is H x W x 3 bytes, row-major, row 0 = top (the order PNG wants).
 * rt_image_to_rgb8 does the same on the host for an image in rt_render's layout. */
int rt_finalize_rgb8(rtScene* scene, const float* d_accum, int n_total, int width, int height, uint8_t* d_rgb8, void* stream);
int rt_image_to_rgb8(const float* img, int width, int height, uint8_t* rgb8);
int rt_write_png(const char* path, const uint8_t* rgb8, int width, int height); /* 8-bit RGB, deflate */
int rt_write_ppm(const char* path, const uint8_t* rgb8, int width, int height); /* binary P6 */

/* The step before the path for render(filename, material) (main.jl:53-81): a Wavefront OBJ file as the vertex and
 * index arrays of an rtMesh (vertices: 8 floats each = position, normal, u, v; indices: 4 int32 per triangle = three
 * 0-based vertex numbers + material slot 0).  One vertex per distinct v/vt/vn triple in order of first appearance,
 * polygons fan-triangulated, missing normals (0,0,0), missing texture coordinates (-1,-1) (main.jl:62-67).  The
 * arrays are allocated by the library; release them with rt_obj_free. */
typedef struct rtObjMesh {
    int32_t n_vertices, n_triangles;
    float* vertices;
    int32_t* indices;
} rtObjMesh;
int rt_obj_load(const char* path, rtObjMesh* out);
void rt_obj_free(rtObjMesh* mesh);

/* Test hook without a counterpart in the reference: the acceleration structure the library builds for a scene
 * (its own SAH trees in eight ray-octant layouts, see raytracing_amd/csrc/rt_device.h), computed on the host without
 * a GPU, so that its invariants (every primitive once per layout, links in range, boxes that enclose the reference's
 * boxes, conservative culling) can be checked anywhere.  `nodes` receives 4 words per node for n_sets * 8 * nodes_per_layout
 * nodes and `prim_top` the position in the world vector of every primitive; either may be NULL, and a call with too
 * little capacity only fills `info` (and returns RT_OK), so a first call sizes the buffers. */
typedef struct rtLayout {
    uint32_t nodes_per_layout, n_prims;
    uint32_t surface_root, surface_size; /* layout-relative root of the tree of surfaces, its node count */
    uint32_t medium_root, medium_size;   /* medium_size = 0: the constant media are kept as a list, not in a tree */
    uint32_t nested;    /* 1 if nested trees follow the two main trees */
    float node_scale;   /* power of two the stored boxes were multiplied by */
    uint32_t n_sets;    /* node sets (8 layouts each): 1, or 2 = {tight boxes, the reference's boxes} */
    float centre[3], radius; /* rays that start inside this ball walk set 0, all others the last set */
} rtLayout;
int rt_scene_layout(const rtSceneDesc* desc, rtLayout* info, uint32_t* nodes, size_t node_capacity,
                    int32_t* prim_top, size_t prim_capacity);
/* The same of a scene that is on a device: the node sets as the kernels walk them, read back.  rt_scene_create builds the
 * trees on the device itself (raytracing_amd/csrc/accel_device.hip: the surface-area heuristic top down, level by
 * level, then a layout kernel; RT_AMD_HOST_BUILD=1 in the environment: the host builder, as rt_scene_layout,
 * which gives the same trees), where the
 * reference builds its trees on the host (src/BVH.jl:65-96, src/mesh.jl:217-283).  `build`, if not NULL: the number of
 * trees the device built (node sets x trees), their leaf depth and the time on the device (all 0 for a host build). */
typedef struct rtBuildInfo {
    uint32_t on_device, n_trees, max_depth;
    float build_ms;
} rtBuildInfo;
int rt_scene_nodes(rtScene* scene, rtLayout* info, uint32_t* nodes, size_t node_capacity, int32_t* prim_top,
                   size_t prim_capacity, rtBuildInfo* build);
/* The tree of surfaces once more, as the pair nodes a walk with a stack reads (one record of eight words per interior
 * node: {box of the near child, word}, {box of the far child, word}; n_sets * 8 layouts of *pairs_per_layout records;
 * raytracing_amd/csrc/accel_build.h): as the host builder makes them / as they are on the device.  *depth: the deepest
 * leaf; *in_use: 1 if rt_render walks them (no nested trees, tree no deeper than the stack).  `pairs` may be NULL; with
 * too little capacity (in records) only the counts are filled. */
int rt_scene_layout_pairs(const rtSceneDesc* desc, uint32_t* pairs_per_layout, uint32_t* depth, uint32_t* pairs, size_t capacity);
int rt_scene_pairs(rtScene* scene, uint32_t* pairs_per_layout, uint32_t* in_use, uint32_t* pairs, size_t capacity);

/* A scene description in one binary file.  The reference has no scene files -- its scenes are Julia functions
 * (the files of samples/, registered in scenes.jl:8-11) -- this is the data they hand to draw!: the tables and the trees of an
 * rtSceneDesc, so that a host other than the reference's Julia code, or a later run, can pass a scene to
 * rt_scene_create without rebuilding it.  rt_scene_desc_load allocates the description and everything it points to;
 * release it with rt_scene_desc_free (after rt_scene_create, which copies what it needs). */
int rt_scene_desc_save(const rtSceneDesc* desc, const char* path);
int rt_scene_desc_load(const char* path, rtSceneDesc** out);
void rt_scene_desc_free(rtSceneDesc* desc);

/* The tables of a NoiseTexture (Perlin3 constructors, src/perlin.jl:17-33): 256 unit vectors drawn as
 * normalize(rand(Vec3, -1, 1)) (vector != 0) or 256 scalars in [0, 1) (vector == 0), then perm_y, perm_z and perm_x as
 * three shuffles of 0..255, in that order.  The reference draws from Julia's unseeded global rand(); this draws from the
 * stream the library's sample scenes use: draw k is pcg4d(k, 0x5CE7E5, seed), with k = *counter on entry; on return
 * *counter is the index of the next unused draw (768 or 256 values + 765 shuffle draws later). */
int rt_perlin_generate(uint32_t seed, uint64_t* counter, int32_t vector, rtPerlin* out);

/* ---------------------------------------------------------------------------------------------------------------
 * F = Float64: render(Float64, ...) (main.jl:21,46).  The reference is generic in its element type F (Vec3{F},
 * Ray{F}, Camera{F}, BVHNode{F}, ...; main.jl:21 takes it as its first argument); everything above is F = Float32,
 * its default.  For Float64 the same records exist with double fields, and a second set of entry points takes
 * them.  The device then walks the reference's own trees (the ones in the description, one per type group) with the
 * reference's own box test in double, so results equal the CPU path's for Float64 bit for bit (random numbers: 53
 * bits from two words of the same counter-based generator).  Image buffers are double: the memory order of a Julia
 * Array{RGB{Float64},2}(H, W).  One process can drive several GPUs (rt_group_*_f64).  Unsupported nestings are the Float32 path's (RT_EUNSUPPORTED); in addition the trees
 * may need at most 40 stack entries per walk (a tree of meshes: its two levels together), which holds for any tree the
 * reference's median split builds over fewer than 2^30 objects. */
typedef struct rtObject64 { int32_t kind, material, child, aux; double p[12]; } rtObject64;     /* as rtObject */
typedef struct rtTransform64 { double xform[12], inv_xform[12], bbox[6]; } rtTransform64;       /* as rtTransform */
typedef struct rtMaterial64 { int32_t type, texture; double fuzz, eta; } rtMaterial64;          /* as rtMaterial */
typedef struct rtTexture64 { int32_t kind, a, b, pad; double color[3]; double scale; } rtTexture64; /* as rtTexture; a
                                                                  checker's two textures must precede it in the table */
typedef struct rtPerlin64 { double ranvec[256 * 3]; int32_t perm_x[256], perm_y[256], perm_z[256]; } rtPerlin64;
typedef struct rtMesh64 {
    int32_t n_vertices, n_triangles, n_materials, pad;
    const double* vertices; /* 8 per vertex */
    const int32_t* indices;
    const int32_t* materials;
    const int32_t* tri_ids;
} rtMesh64;
typedef struct rtBvh64 {
    int32_t kind, n_objects;
    const int32_t* objects;
    int32_t n_nodes, pad;
    const double* node_bounds;
    const int32_t* node_left;
    const int32_t* node_right;
    const uint8_t* node_leaf;
} rtBvh64;
typedef struct rtSceneDesc64 {
    int32_t n_objects;    const rtObject64* objects;
    int32_t n_top;        const int32_t* top;
    int32_t n_transforms; const rtTransform64* transforms;
    int32_t n_materials;  const rtMaterial64* materials;
    int32_t n_textures;   const rtTexture64* textures;
    int32_t n_images;     const rtImage* images;
    int32_t n_perlins;    const rtPerlin64* perlins;
    int32_t n_meshes;     const rtMesh64* meshes;
    int32_t n_groups;     const rtBvh64* groups;
    int32_t n_buckets;    const rtBvh64* buckets;
} rtSceneDesc64;
typedef struct rtCamera64 { /* src/camera.jl:3-14 with F = Float64 */
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius, t0, t1;
} rtCamera64;
typedef struct rtScene64 rtScene64;

/* rt_scene_create / rt_scene_destroy for a Float64 scene: validates and copies the description's tables and trees as
 * they are (no device-side tree build: the walk is the reference's, see above). */
int rt_scene_create_f64(const rtSceneDesc64* desc, int device, rtScene64** out);
int rt_scene_destroy_f64(rtScene64* scene);
/* draw!(img::CuArray{RGB{Float64}}, ...) (src/Ray_Tracing.jl:141-157): as rt_render, `out` receives 3*H*W doubles.
 * stats: seconds and paths are filled, the counters stay 0.  flags are ignored. */
int rt_render_f64(rtScene64* scene, const rtCamera64* camera, const double* pattern, int n, const double background[3],
                  int width, int height, int max_depth, uint64_t seed, uint32_t flags, double* out, rtStats* stats);
/* the two halves, as rt_render_accumulate / rt_finalize (d_accum, d_out: 3*H*W doubles in device memory) */
int rt_render_accumulate_f64(rtScene64* scene, const rtCamera64* camera, const double* pattern, int n, int sample_first,
                             int sample_stride, const double background[3], int width, int height, int max_depth,
                             uint64_t seed, double* d_accum, void* stream);
int rt_finalize_f64(rtScene64* scene, const double* d_accum, int n_total, int width, int height, double* d_out, void* stream);
/* the step after the path (README.md:50-67: save(path, clamp01.(img))) for an RGB{Float64} image, as rt_finalize_rgb8 /
 * rt_image_to_rgb8: N0f8(x::Float64) rounds x * 255 to the nearest even in double.  The bytes go to rt_write_png / _ppm. */
int rt_finalize_rgb8_f64(rtScene64* scene, const double* d_accum, int n_total, int width, int height, uint8_t* d_rgb8, void* stream);
int rt_image_to_rgb8_f64(const double* img, int width, int height, uint8_t* rgb8);
/* parity hooks, as rt_trace_rays / rt_trace_paths */
int rt_trace_rays_f64(rtScene64* scene, const double* rays, int n, double t_min, double t_max, uint64_t seed,
                      int32_t* prim_id, int32_t* sub_id, double* t, double* normal, double* uv, uint8_t* front);
int rt_trace_paths_f64(rtScene64* scene, const rtCamera64* camera, const double* pattern, int n, const double background[3],
                       int width, int height, int max_depth, uint64_t seed, const int32_t* ids, int n_paths, double* radiance);
/* rt_group_* for Float64 scenes: one host process, several GPUs, the pattern columns dealt to them, the double
 * accumulators summed on the first device (ncclReduce of ncclDouble, or peer copies and adds: RT_AMD_REDUCE=peer or a
 * device listed twice).  A group of one device renders exactly what rt_render_f64 renders. */
typedef struct rtGroup64 rtGroup64;
int rt_group_create_f64(const rtSceneDesc64* desc, const int* devices, int n_devices, rtGroup64** out);
int rt_group_size_f64(const rtGroup64* group, int* n_devices);
int rt_group_render_f64(rtGroup64* group, const rtCamera64* camera, const double* pattern, int n, const double background[3],
                        int width, int height, int max_depth, uint64_t seed, uint32_t flags, double* out, rtStats* stats);
int rt_group_destroy_f64(rtGroup64* group);
/* build_bvh (src/BVH.jl:65-109, src/mesh.jl:217-305) over Float64 boxes and vertices, as rt_bvh_build / rt_bvh_build_meshes */
/* scene descriptions in a file, as rt_scene_desc_save / _load / _free; the file carries its element type and each loader
 * refuses the other's files (RT_EINVAL) */
int rt_scene_desc_save_f64(const rtSceneDesc64* desc, const char* path);
int rt_scene_desc_load_f64(const char* path, rtSceneDesc64** out);
void rt_scene_desc_free_f64(rtSceneDesc64* desc);

int rt_bvh_build_f64(int n, const double* object_bounds, int32_t* order, double* node_bounds, int32_t* node_left,
                     int32_t* node_right, uint8_t* node_leaf);
int rt_bvh_build_meshes_f64(int n_meshes, const int32_t* n_vertices, const double* const* vertices,
                            const int32_t* n_triangles, int32_t* const* indices, int32_t* const* tri_ids,
                            int32_t* mesh_order, double* node_bounds, int32_t* node_left, int32_t* node_right,
                            uint8_t* node_leaf);

#ifdef __cplusplus
}
#endif
#endif
