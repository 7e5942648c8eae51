// rt_device.h -- device-resident scene layout shared by the host flattener (rt_scene.cpp)
// and the gfx950 kernels (rt_kernels.hip).
//
// Layout in HBM (everything is small enough to stay L2-resident):
//   nodes   : 32 B per node = 2 x float4:  {min.xyz, bits(skip)}, {max.xyz, bits(leaf)}
//             Nodes are stored in depth-first pre-order.  Traversal is stackless: a node whose box is
//             hit continues at node+1 (its first child) and every other case continues at `skip`
//             (the next node in pre-order that is not a descendant).  leaf = primitive index + 1,
//             0 for interior nodes.  The per-type trees of the host are laid out one after another,
//             so one walk covers the whole SceneManager: first all trees of surfaces (ending in
//             RT_NODE_END), then all trees of constant media (a second chain ending in RT_NODE_END).
//   prims   : 48 B per primitive = 3 x uint4: {head, material, rank, g0}, {g1..g4}, {g5..g8}
//             head = kind (bits 0-3) | medium flag (bit 4) | index field (bits 8-31):
//             transform index + 1 (0 = none), or for triangles the attribute record index.
//   xforms  : 24 floats per transform: forward 3x4 column-major, inverse 3x4 column-major.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H
#include <stdint.h>

#define RT_NODE_END 0xFFFFFFFFu     // traversal finished
#define RT_NODE_END_SUB 0xFFFFFFFEu // end of a nested tree: restore the outer ray

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
#define DP_MEDIUM_FLAG 16u

#define DM_NEEDS_UV 0x100 // material type word: texture lookup reads u,v (image textures)

struct DevImage {
    uint32_t offset, width, height, pad;
};

struct DevScene {
    const float4* nodes;
    const uint4* prims;
    const float* xforms;
    const uint4* materials;      // {type | flags, texture, bits(fuzz), bits(eta)}
    const uint4* textures;       // 2 per texture: {kind, a, b, 0}, {bits(color.xyz), bits(scale)}
    const float* tri_attr;       // 16 floats per triangle: n1(3) n2(3) n3(3) u1 v1 u2 v2 u3 v3 pad
    const DevImage* images;
    const uint8_t* image_data;
    const float* perlin_vec;     // 768 floats per perlin
    const int32_t* perlin_perm;  // 768 ints per perlin (x, y, z)
    const int32_t* prim_top;     // per primitive: position in the world vector (reported prim_id)
    const int32_t* prim_sub;     // per primitive: reported sub id base (triangle id / nested object index)
    uint32_t n_nodes, n_prims;
    uint32_t surface_root; // first node of the surface trees, RT_NODE_END if there are none
    uint32_t medium_root;  // first node of the constant-medium trees, RT_NODE_END if there are none
};

#endif
