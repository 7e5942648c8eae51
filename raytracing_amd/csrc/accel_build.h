// accel_build.h -- the device-side acceleration structure: one binary BVH over the surfaces and one per nested
// tree kept as an instance, built with the surface-area heuristic over the primitives' bounding boxes and laid out once per ray-direction
// octant so that the stackless walk of rt_kernels.hip meets the near child of every node first.
#ifndef ACCEL_BUILD_H
#define ACCEL_BUILD_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "rt_device.h"

struct AccelItem {
    float b[6];    // min.xyz, max.xyz
    uint32_t leaf; // leaf word: (primitive index + 1) | flags
};

struct AccelNode {
    float b[6];
    int32_t child[2]; // child[0] holds the lower centroids along `axis`; -1 in leaves
    int32_t axis;
    uint32_t leaf;    // 0 in interior nodes
    uint32_t size;    // nodes in the subtree, this one included
};

struct AccelTree {
    std::vector<AccelNode> nodes; // root = 0; empty for no items
};

void accel_build(std::vector<AccelItem>& items, AccelTree& tree);

// Appends the tree to `out`, 16 bytes per node: the box as six IEEE halves, one word per axis {near | far << 16} where
// near is the plane of that axis a ray of `octant` meets first (min for a positive direction component, max for a
// negative one), of the bounds multiplied by `scale` (a power of two chosen by the caller so that the scene fits the range of a half),
// each rounded strictly outward, and one word: RT_NODE_LEAF | leaf word for leaves, else the skip link (index into `out` of the
// next node in depth-first order that is not a descendant; one past the tree's last node for the links that leave it).
// Nodes are in depth-first order; at every interior node the child on the near side for rays of `octant` (bit a set:
// direction component a is negative) comes first.
void accel_emit(const AccelTree& tree, int octant, float scale, std::vector<uint4>& out);

// The same tree as pair nodes, for a walk that keeps a stack: one record of two uint4 per interior node, {box of the
// child on the near side for rays of `octant` (three words as above), word} and the same for the far child; a word is
// RT_NODE_LEAF | leaf word for a leaf, else the index of the child's own record, counted from `base` on (records are
// appended in depth-first order, near child first: the first one is the root's).  A tree of one leaf gives one record
// that holds the leaf twice.  Returns the number of records appended.
uint32_t accel_emit_pairs(const AccelTree& tree, int octant, float scale, uint32_t base, std::vector<uint4>& out);
// deepest leaf of the tree (root = 0)
uint32_t accel_depth(const AccelTree& tree);

// float -> IEEE half bits strictly below (up = false) or strictly above (up = true) x (infinities stay)
uint16_t accel_half(float x, bool up);

#endif
