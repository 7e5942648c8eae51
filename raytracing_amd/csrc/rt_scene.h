// rt_scene.h -- host-side staging of the device scene layout (see rt_device.h)
#ifndef RT_SCENE_H
#define RT_SCENE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/rt_amd.h"
#include "rt_device.h"

struct HostScene {
    std::vector<uint4> nodes;  // 1 per node; eight layouts of nodes_per_layout nodes each, one per ray octant
    uint32_t nodes_per_layout = 0;
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
    uint32_t surface_root = 0, surface_size = 0, medium_root = 0, medium_size = 0; // layout-relative roots, node counts
};

// validates the description and fills `out`; returns an rtStatus and sets rt_last_error()
int rt_flatten_scene(const rtSceneDesc* desc, HostScene* out);

#endif
