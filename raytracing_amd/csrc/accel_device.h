// accel_device.h -- the device trees built on the GPU (accel_device.hip): the counterpart of accel_build / accel_emit
// for a scene that is being put on a device.  The reference builds its trees on the host (src/BVH.jl:65-96,
// src/mesh.jl:217-283); which boxes are walked does not change the closest hit, so the device builds its own.
#ifndef ACCEL_DEVICE_H
#define ACCEL_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "accel_build.h"

// One tree of one node set: `n_items` leaves, taken from the set's item array at `item_first`, whose 2 n - 1 nodes go to
// [layout_first, layout_first + 2 n - 1) of each of the set's eight layouts.
struct AccelDeviceTree {
    uint32_t item_first, n_items, layout_first;
    bool pairs; // also written as pair nodes (accel_emit_pairs), at the start of each pair layout
};

struct AccelDeviceStats {
    uint32_t n_trees;    // trees built (over all node sets)
    uint32_t max_depth;  // deepest leaf over all trees
    float build_ms;      // device time of the build and layout kernels
};

// The device memory a build works in, for scenes of up to n_items leaves (per node set) in n_trees trees: made once, so
// that builds (the first, and later ones for another reach of the rays) allocate nothing.
struct AccelDeviceWorkspace;
int accel_device_workspace_create(size_t n_items, size_t n_trees, AccelDeviceWorkspace** out);
void accel_device_workspace_destroy(AccelDeviceWorkspace* ws);

// Builds every tree of `trees` for each of the `n_sets` node sets (items[s]: the leaves of set s, the trees' items
// contiguous) and writes the n_sets * 8 layouts of `nodes_per_layout` nodes to d_nodes (device memory of the current
// device), boxes multiplied by `scale`; a tree marked `pairs` also goes to d_pairs, n_sets * 8 layouts of
// `pairs_per_layout` records.  Blocks until the nodes are there.  Returns an rtStatus.
int accel_device_build(AccelDeviceWorkspace* ws, const std::vector<AccelItem>* items, uint32_t n_sets,
                       const std::vector<AccelDeviceTree>& trees, uint32_t nodes_per_layout, float scale, uint4* d_nodes,
                       uint32_t pairs_per_layout, uint4* d_pairs, AccelDeviceStats* stats);

#endif
