// bvh_build.cpp -- host-side BVH builders exported through the C ABI for hosts that do not carry the
// reference's Julia builder.  They reproduce the trees of build_bvh (src/BVH.jl:65-109) and of the two-level
// mesh build (src/mesh.jl:217-305): same split rule (median of the objects sorted by box minimum along the
// first longest axis of the enclosing box, stable sort), same node numbering, same leaf encoding.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace f32 {
typedef float real;
#include "bvh_build_impl.h"
} // namespace f32
namespace f64 { // render(Float64, ...): boxes and vertices in double, everything else the same
typedef double real;
#include "bvh_build_impl.h"
} // namespace f64


extern "C" int rt_bvh_build(int n, const float* object_bounds, int32_t* order, float* node_bounds, int32_t* node_left,
                            int32_t* node_right, uint8_t* node_leaf) {
    if (n <= 0 || !object_bounds || !order || !node_bounds || !node_left || !node_right || !node_leaf)
        return rt_fail(RT_EINVAL, "rt_bvh_build: null argument or n <= 0");
    for (int i = 0; i < n; ++i) order[i] = i;
    f32::GenericBuild g{object_bounds, order, f32::NodeOut{node_bounds, node_left, node_right, node_leaf}};
    g.build(1, 1, n);
    return RT_OK;
}

extern "C" int rt_bvh_build_meshes(int n_meshes, const int32_t* n_vertices, const float* const* vertices,
                                   const int32_t* n_triangles, int32_t* const* indices, int32_t* const* tri_ids,
                                   int32_t* mesh_order, float* node_bounds, int32_t* node_left, int32_t* node_right,
                                   uint8_t* node_leaf) {
    if (n_meshes <= 0 || !n_vertices || !vertices || !n_triangles || !indices || !tri_ids || !mesh_order ||
        !node_bounds || !node_left || !node_right || !node_leaf)
        return rt_fail(RT_EINVAL, "rt_bvh_build_meshes: null argument or n_meshes <= 0");
    f32::MeshBuild b;
    size_t total = 0;
    for (int m = 0; m < n_meshes; ++m) {
        if (n_triangles[m] <= 0 || n_vertices[m] <= 0) return rt_fail(RT_EINVAL, "rt_bvh_build_meshes: empty mesh");
        for (int k = 0; k < 4 * n_triangles[m]; ++k)
            if ((k & 3) != 3 && (indices[m][k] < 0 || indices[m][k] >= n_vertices[m]))
                return rt_fail(RT_EINVAL, "rt_bvh_build_meshes: vertex index out of range");
        f32::MeshRef r{n_vertices[m], n_triangles[m], vertices[m], indices[m], tri_ids[m]};
        for (int t = 0; t < r.nt; ++t) r.tri_ids[t] = t;
        b.meshes.push_back(r);
        b.mesh_boxes.push_back(f32::mesh_box_all(r));
        mesh_order[m] = m;
        total += (size_t)n_triangles[m];
    }
    b.slots = 4 * total - 1;
    std::memset(node_leaf, 255, b.slots);
    b.order = mesh_order;
    b.out = f32::NodeOut{node_bounds, node_left, node_right, node_leaf};
    b.build_level(1, 1, n_meshes);
    if (b.overflow) return rt_fail(RT_EUNSUPPORTED, "rt_bvh_build_meshes: node id exceeds the 4*triangles-1 slots of the reference layout");
    return RT_OK;
}


extern "C" int rt_bvh_build_f64(int n, const double* object_bounds, int32_t* order, double* node_bounds, int32_t* node_left,
                            int32_t* node_right, uint8_t* node_leaf) {
    if (n <= 0 || !object_bounds || !order || !node_bounds || !node_left || !node_right || !node_leaf)
        return rt_fail(RT_EINVAL, "rt_bvh_build_f64: null argument or n <= 0");
    for (int i = 0; i < n; ++i) order[i] = i;
    f64::GenericBuild g{object_bounds, order, f64::NodeOut{node_bounds, node_left, node_right, node_leaf}};
    g.build(1, 1, n);
    return RT_OK;
}

extern "C" int rt_bvh_build_meshes_f64(int n_meshes, const int32_t* n_vertices, const double* const* vertices,
                                   const int32_t* n_triangles, int32_t* const* indices, int32_t* const* tri_ids,
                                   int32_t* mesh_order, double* node_bounds, int32_t* node_left, int32_t* node_right,
                                   uint8_t* node_leaf) {
    if (n_meshes <= 0 || !n_vertices || !vertices || !n_triangles || !indices || !tri_ids || !mesh_order ||
        !node_bounds || !node_left || !node_right || !node_leaf)
        return rt_fail(RT_EINVAL, "rt_bvh_build_meshes_f64: null argument or n_meshes <= 0");
    f64::MeshBuild b;
    size_t total = 0;
    for (int m = 0; m < n_meshes; ++m) {
        if (n_triangles[m] <= 0 || n_vertices[m] <= 0) return rt_fail(RT_EINVAL, "rt_bvh_build_meshes_f64: empty mesh");
        for (int k = 0; k < 4 * n_triangles[m]; ++k)
            if ((k & 3) != 3 && (indices[m][k] < 0 || indices[m][k] >= n_vertices[m]))
                return rt_fail(RT_EINVAL, "rt_bvh_build_meshes_f64: vertex index out of range");
        f64::MeshRef r{n_vertices[m], n_triangles[m], vertices[m], indices[m], tri_ids[m]};
        for (int t = 0; t < r.nt; ++t) r.tri_ids[t] = t;
        b.meshes.push_back(r);
        b.mesh_boxes.push_back(f64::mesh_box_all(r));
        mesh_order[m] = m;
        total += (size_t)n_triangles[m];
    }
    b.slots = 4 * total - 1;
    std::memset(node_leaf, 255, b.slots);
    b.order = mesh_order;
    b.out = f64::NodeOut{node_bounds, node_left, node_right, node_leaf};
    b.build_level(1, 1, n_meshes);
    if (b.overflow) return rt_fail(RT_EUNSUPPORTED, "rt_bvh_build_meshes_f64: node id exceeds the 4*triangles-1 slots of the reference layout");
    return RT_OK;
}
