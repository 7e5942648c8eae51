// rt_device.h -- device-resident scene layout shared by the host flattener (rt_scene.cpp)
// and the gfx950 kernels (rt_kernels.hip).
//
// Layout in HBM (scenes of the size the reference ships stay L2-resident):
//   nodes   : 16 B per node = one uint4: the box as six IEEE halves, one word per axis (near | far << 16: the plane a
//             ray of the layout's octant meets first, then the other one; layout 0 = min | max << 16) of the bounds
//             times node_scale, rounded strictly outward, and a link word w:
//             leaves carry RT_NODE_LEAF | (primitive index + 1) | RT_LEAF_INSTANCE / RT_LEAF_XFORM / RT_LEAF_BOX flags, interior nodes
//             the skip link.  Nodes are stored in depth-first pre-order.  Traversal is stackless: an interior node
//             whose box is hit continues at node+1 (its first child), one that is missed at its skip link (the next
//             node in pre-order that is not a descendant); a leaf always continues at node+1.  The walk of a tree is
//             over when it reaches the index one past the tree's last node.  One tree covers every surface of the
//             SceneManager, and the nested trees kept as instances (Transform{BVHTree}) follow; the constant media
//             are a list (`media`), not a tree.  The whole array exists eight times, once per octant of ray
//             directions, with the children of every node ordered near side first for that octant: layout o occupies
//             nodes [o * n_nodes, (o + 1) * n_nodes) and its skip links stay inside it; roots are given relative to
//             the layout.  A second set of eight layouts (same trees, looser leaf boxes) may follow, see far_set.
//   prim_box: the boxes of the reference's own trees per primitive, for the test the reference makes before it
//             reaches a primitive (see RT_BOX_OUTER / RT_BOX_INNER).
//   prims   : 48 B per primitive = 3 x uint4: {head, material, rank, g0}, {g1..g4}, {g5..g8}
//             head = kind (bits 0-3) | medium flag (bit 4) | index field (bits 8-31):
//             transform index + 1 (0 = none), or for triangles the attribute record index.
//   xforms  : 24 floats per transform: forward 3x4 column-major, inverse 3x4 column-major.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H
#include <stdint.h>

#define RT_NODE_NONE 0xFFFFFFFFu    // "no node" (e.g. no outer tree to return to)
#define RT_PAIR_STACK 20u          // entries of the LDS stack of a lane that walks the pair nodes at six workgroups per CU
#define RT_PAIR_STACK_DEEP 32u     // ... at four workgroups per CU: the deepest tree walked through pair nodes (beyond: the stackless walk)
#define RT_NODE_LEAF 0x80000000u    // node link word: the node is a leaf
#define RT_LEAF_INSTANCE 0x40000000u // leaf word flag: the primitive is a nested tree (Transform{BVHTree})
#define RT_LEAF_XFORM 0x20000000u    // leaf word flag: the primitive sits in a Transform (prim_inv holds its inverse matrix)
// prim_box[4 * prim + 0..1] = {min.xyz, max.x}, {max.yz, flags, 0}: the leaf box of the primitive in its top-level
// tree (for an object of a nested tree: the box of that tree's leaf), tested with the ray as cast;
// prim_box[4 * prim + 2..3] = {min.xyz, max.x}, {max.yz, 0, 0}: the leaf box inside a nested tree, tested with the
// ray taken into the frame of the Transform around that tree
#define RT_BOX_OUTER 1u // flags: the first box is present
#define RT_BOX_INNER 2u // flags: the second box is present
#define RT_LEAF_BOX 0x10000000u      // leaf word flag: the primitive is a Box (rt_extend_kernel spreads its six sides over lanes)
#define RT_LEAF_PRIM_MASK 0x0FFFFFFFu // leaf word: primitive index + 1

enum DevPrimKind {
    DP_SPHERE = 1,
    DP_MOVING_SPHERE = 2,
    DP_XY_RECT = 3,
    DP_XZ_RECT = 4,
    DP_YZ_RECT = 5,
    DP_BOX = 6,
    DP_TRIANGLE = 7,
    DP_INSTANCE = 8
};
// Classes of surface primitives a scene holds (DevScene::classes): rt_extend_kernel is compiled once per set of classes
// that occurs in practice, so that a scene of spheres carries no rectangle or triangle code (registers, instruction
// cache) through its walk.
#define RT_CL_SPHERE 1u  // spheres
#define RT_CL_MOVING 2u  // moving spheres (centre from the ray's time)
#define RT_CL_RECT 4u    // the three rectangle kinds
#define RT_CL_BOX 8u     // boxes (six rectangle tests)
#define RT_CL_TRI 16u    // triangles
#define RT_CL_XFORM 32u  // a primitive sits in a Transform (ray taken into its frame first)
#define RT_CL_ALL 63u
#define DP_MEDIUM_FLAG 16u
#define DP_TYPE_MASK 0xE0u // primitive head bits 5-7: material type of the primitive (1..5)

#define DM_NEEDS_UV 0x100 // material type word: texture lookup reads u,v (image textures)

struct DevImage {
    uint32_t offset, width, height, pad;
};

struct DevScene {
    const uint4* nodes;
    const uint4* pairs;          // the tree of surfaces as pair nodes: two uint4 per record, {near child's box, word}, {far child's box, word}
    uint32_t n_pairs, use_pairs; // records per octant layout (second set: 8 * n_pairs further on); 1: rt_extend_kernel walks them
    const uint4* prims;
    const float4* prim_box;      // 4 per primitive: the boxes the reference walks for it (RT_BOX_* below)
    const float4* prim_inv;      // 3 per primitive: inverse 3x4 matrix (column-major) of its Transform, zeros without one
    const float* xforms;
    const uint4* materials;      // {type | flags, texture, bits(fuzz), bits(eta)}
    const uint4* textures;       // 2 per texture: {kind, a, b, 0}, {bits(color.xyz), bits(scale)}
    const float* tri_attr;       // 16 floats per triangle: n1(3) n2(3) n3(3) u1 v1 u2 v2 u3 v3 pad
    const DevImage* images;
    const uint8_t* image_data;
    const float* perlin_vec;     // 768 floats per perlin
    const int32_t* perlin_perm;  // 768 ints per perlin (x, y, z)
    const uint32_t* media;       // the constant media (primitive indices) in descending rank
    const uint32_t* media_group; // ... and the type group of each (position in the SceneManager tuple)
    const uint32_t* bucket_base; // lowest rank of every type group of the SceneManager, in its order (descending ranks)
    const int32_t* prim_top;     // per primitive: position in the world vector (reported prim_id)
    const int32_t* prim_sub;     // per primitive: reported sub id base (triangle id / nested object index)
    uint32_t n_nodes, n_prims; // n_nodes: nodes per octant layout
    // Two sets of eight layouts each: the first with tight boxes, for rays that start inside the ball (centre, radius2 =
    // radius squared), the second -- far_set nodes further on -- with the reference's own boxes, for all other rays.
    // far_set = 0: one set for every ray.
    uint32_t far_set;
    float centre[3], radius2;
    float middle[3];             // a point with half of the scene's surface primitives on either side, axis by axis
    float node_scale, inv_node_scale; // power of two applied to the node boxes (and to the ray for the box test)
    uint32_t surface_root, surface_size; // root of the surface tree (layout-relative) and its node count (0: no surfaces)
    uint32_t nested;       // 1 if the scene holds nested trees (Transform{BVHTree})
    uint32_t textured;     // 1 if any texture is not a SolidColor
    uint32_t n_media;      // entries in `media`
    uint32_t n_buckets;    // entries in `bucket_base`
    uint32_t first_group_base; // bucket_base[0]: a rank at or above it belongs to the first type group
    uint32_t pair_depth;   // deepest leaf of the tree of surfaces (entries a lane's stack needs in the walk over pair nodes)
    uint32_t classes;      // RT_CL_* of the surface primitives (the trees' leaves; media boundaries are not among them)
};

#endif
