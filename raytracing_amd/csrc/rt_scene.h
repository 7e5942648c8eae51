// rt_scene.h -- host-side staging of the device scene layout (see rt_device.h)
#ifndef RT_SCENE_H
#define RT_SCENE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/rt_amd.h"
#include "accel_device.h"
#include "rt_device.h"

// One leaf of a device tree, as the scene description gives it: what rt_build_nodes turns into boxes.
struct HostItem {
    uint32_t tree;       // 0: surfaces, 1: constant media, 2 + k: k-th nested tree kept as an instance
    uint32_t leaf;       // leaf word: (primitive index + 1) | RT_LEAF_* flags
    float ref[6];        // the box the reference walks for it: its leaf's box -- for an object taken out of a nested tree
                         // the box of that tree's leaf in the enclosing tree (world frame)
    bool inlined;        // object of a nested tree (Transform{BVHTree}) taken into the enclosing tree
    double wlo[3], whi[3]; // inlined: world-frame bounds of the object's leaf box in the nested tree, unpadded
};

struct HostScene {
    // Nodes: n_sets * 8 layouts of nodes_per_layout nodes each (one layout per ray octant).  Set 0 holds tight boxes,
    // valid for rays that start inside the ball (centre, radius); set 1 (present if n_sets == 2) the reference's own
    // boxes, valid for every ray that starts within `reach` of the origin.
    std::vector<uint4> nodes;
    // The tree of surfaces once more as pair nodes (accel_emit_pairs), n_sets * 8 layouts of pairs_per_layout records of
    // two uint4 (0: no surfaces, or nested trees present: those scenes are walked through `nodes` only); tree_depth: its
    // deepest leaf (host_trees only; the device build reports its own).
    std::vector<uint4> pairs;
    uint32_t pairs_per_layout = 0;
    uint32_t tree_depth = 0;
    // host_trees == false: rt_build_nodes stops at the leaves' boxes (set_items, the items of a tree contiguous, trees
    // as listed in device_trees) and leaves `nodes` empty: the trees are then built on the device (accel_device_build)
    bool host_trees = true;
    std::vector<AccelItem> set_items[2];
    std::vector<AccelDeviceTree> device_trees;
    uint32_t nodes_per_layout = 0;
    uint32_t n_sets = 1;
    float centre[3] = {0.0f, 0.0f, 0.0f};
    float middle[3] = {0.0f, 0.0f, 0.0f}; // median of the surface leaves' box centres, axis by axis
    float radius = __builtin_inff();
    double reach = 0.0;        // largest |coordinate| of a ray origin the node boxes were built for
    double scene_reach = 0.0;  // largest finite |coordinate| of the scene's own boxes
    std::vector<HostItem> items;
    std::vector<uint32_t> tree_first; // first node (layout-relative) of every tree in `items`' numbering
    float node_scale = 1.0f; // power of two the node boxes are multiplied by
    bool nested = false; // nested trees (Transform{BVHTree}) present
    bool textured = false; // a texture other than SolidColor present
    std::vector<uint4> prims;  // 3 per primitive
    std::vector<float4> prim_inv; // 3 per primitive
    std::vector<float4> prim_box; // 4 per primitive
    std::vector<float> xforms; // 24 per transform
    std::vector<uint4> materials;
    std::vector<uint4> textures; // 2 per texture
    std::vector<float> tri_attr; // 16 per triangle
    std::vector<DevImage> images;
    std::vector<uint8_t> image_data;
    std::vector<float> perlin_vec;
    std::vector<int32_t> perlin_perm;
    std::vector<int32_t> prim_top, prim_sub;
    std::vector<uint32_t> media; // the constant media (primitive indices) in descending rank
    std::vector<uint32_t> media_group; // ... and the type group (position in the SceneManager tuple) of each
    std::vector<uint32_t> bucket_base; // lowest rank of every type group of the SceneManager, in its order (descending)
    uint32_t surface_root = 0, surface_size = 0, medium_root = 0, medium_size = 0; // layout-relative roots, node counts
};

// validates the description and fills `out` (nodes built for rays that start inside the scene); returns an rtStatus
// and sets rt_last_error()
int rt_flatten_scene(const rtSceneDesc* desc, HostScene* out);
// (re)builds out->nodes and the ball of set 0 for ray origins with |coordinate| <= reach
void rt_build_nodes(HostScene* out, double reach);

#endif
