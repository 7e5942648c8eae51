// accel_build.cpp -- SAH builder and per-octant depth-first layout of the device BVH (see accel_build.h).
//
// The reference walks one median-split tree per object type (src/BVH.jl:56-100, src/scene_manager.jl:36-53).  Which
// boxes are walked does not change which primitive is the closest hit, so the device is free to walk a better tree:
// one tree over all surfaces, split by the surface-area heuristic, children ordered front to back for the ray.
#include "accel_build.h"
#include "accel_half.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const float* b) {
        for (int a = 0; a < 3; ++a) {
            if (b[a] < lo[a]) lo[a] = b[a];
            if (b[3 + a] > hi[a]) hi[a] = b[3 + a];
        }
    }
    void grow(const Box& o) {
        for (int a = 0; a < 3; ++a) {
            if (o.lo[a] < lo[a]) lo[a] = o.lo[a];
            if (o.hi[a] > hi[a]) hi[a] = o.hi[a];
        }
    }
    double half_area() const {
        double e[3];
        for (int a = 0; a < 3; ++a) {
            e[a] = (double)hi[a] - (double)lo[a];
            if (!(e[a] >= 0.0)) return 0.0; // empty or NaN
            if (e[a] > 1e30) e[a] = 1e30;   // unbounded boxes still compare
        }
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
};

inline double centroid(const AccelItem& it, int a) {
    double c = 0.5 * ((double)it.b[a] + (double)it.b[3 + a]);
    if (!(c == c)) c = 0.0;
    if (c > 1e30) c = 1e30;
    if (c < -1e30) c = -1e30;
    return c;
}

constexpr int BINS = 32;
constexpr size_t SWEEP_LIMIT = 64; // ranges up to this size try every split position

struct Split {
    int axis = -1;
    size_t mid = 0; // items [lo, mid) go left
};

// full sweep: sorts the range along each axis and evaluates every split; leaves the range sorted along the best axis
Split sweep_split(std::vector<AccelItem>& items, size_t lo, size_t hi) {
    const size_t n = hi - lo;
    double best = std::numeric_limits<double>::infinity();
    Split s;
    std::vector<double> right_area(n);
    for (int a = 0; a < 3; ++a) {
        std::stable_sort(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi,
                         [a](const AccelItem& x, const AccelItem& y) { return centroid(x, a) < centroid(y, a); });
        Box r;
        r.reset();
        for (size_t k = n; k-- > 1;) {
            r.grow(items[lo + k].b);
            right_area[k] = r.half_area();
        }
        Box l;
        l.reset();
        for (size_t k = 1; k < n; ++k) {
            l.grow(items[lo + k - 1].b);
            double cost = l.half_area() * (double)k + right_area[k] * (double)(n - k);
            if (cost < best) { best = cost; s.axis = a; s.mid = lo + k; }
        }
    }
    if (s.axis < 0) { s.axis = 0; s.mid = lo + n / 2; }
    if (s.axis != 2)
        std::stable_sort(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi,
                         [&s](const AccelItem& x, const AccelItem& y) { return centroid(x, s.axis) < centroid(y, s.axis); });
    return s;
}

// binned SAH over the centroid bounds; partitions the range in place
Split binned_split(std::vector<AccelItem>& items, size_t lo, size_t hi) {
    const size_t n = hi - lo;
    double cmin[3], cmax[3];
    for (int a = 0; a < 3; ++a) { cmin[a] = std::numeric_limits<double>::infinity(); cmax[a] = -cmin[a]; }
    for (size_t k = lo; k < hi; ++k)
        for (int a = 0; a < 3; ++a) {
            double c = centroid(items[k], a);
            cmin[a] = std::min(cmin[a], c);
            cmax[a] = std::max(cmax[a], c);
        }
    double best = std::numeric_limits<double>::infinity();
    Split s;
    int best_bin = 0;
    for (int a = 0; a < 3; ++a) {
        double ext = cmax[a] - cmin[a];
        if (!(ext > 0.0)) continue;
        double scale = (double)BINS / ext;
        Box bb[BINS];
        size_t cnt[BINS];
        for (int b = 0; b < BINS; ++b) { bb[b].reset(); cnt[b] = 0; }
        for (size_t k = lo; k < hi; ++k) {
            int b = (int)((centroid(items[k], a) - cmin[a]) * scale);
            b = std::min(std::max(b, 0), BINS - 1);
            bb[b].grow(items[k].b);
            cnt[b]++;
        }
        double ra[BINS];
        size_t rc[BINS];
        Box r;
        r.reset();
        size_t c = 0;
        for (int b = BINS - 1; b >= 1; --b) {
            r.grow(bb[b]);
            c += cnt[b];
            ra[b] = r.half_area();
            rc[b] = c;
        }
        Box l;
        l.reset();
        size_t lc = 0;
        for (int b = 1; b < BINS; ++b) {
            l.grow(bb[b - 1]);
            lc += cnt[b - 1];
            if (lc == 0 || rc[b] == 0) continue;
            double cost = l.half_area() * (double)lc + ra[b] * (double)rc[b];
            if (cost < best) { best = cost; s.axis = a; best_bin = b; }
        }
    }
    if (s.axis < 0) { // every centroid coincides: halve the range
        s.axis = 0;
        s.mid = lo + n / 2;
        return s;
    }
    const int a = s.axis;
    const double scale = (double)BINS / (cmax[a] - cmin[a]);
    const double base = cmin[a];
    auto it = std::stable_partition(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi, [=](const AccelItem& x) {
        int b = (int)((centroid(x, a) - base) * scale);
        b = std::min(std::max(b, 0), BINS - 1);
        return b < best_bin;
    });
    s.mid = (size_t)(it - items.begin());
    if (s.mid == lo || s.mid == hi) s.mid = lo + n / 2;
    return s;
}

} // namespace

void accel_build(std::vector<AccelItem>& items, AccelTree& tree) {
    tree.nodes.clear();
    if (items.empty()) return;
    tree.nodes.reserve(2 * items.size());
    struct Job { size_t lo, hi; int32_t node; };
    std::vector<Job> jobs;
    tree.nodes.push_back(AccelNode{});
    jobs.push_back({0, items.size(), 0});
    while (!jobs.empty()) {
        Job j = jobs.back();
        jobs.pop_back();
        Box b;
        b.reset();
        for (size_t k = j.lo; k < j.hi; ++k) b.grow(items[k].b);
        AccelNode nd{};
        std::memcpy(nd.b, b.lo, sizeof(b.lo));
        std::memcpy(nd.b + 3, b.hi, sizeof(b.hi));
        nd.child[0] = nd.child[1] = -1;
        nd.axis = 0;
        nd.leaf = 0;
        nd.size = 1;
        if (j.hi - j.lo == 1) {
            std::memcpy(nd.b, items[j.lo].b, sizeof(nd.b)); // keep NaN / infinite bounds exactly as given
            nd.leaf = items[j.lo].leaf;
            tree.nodes[(size_t)j.node] = nd;
            continue;
        }
        Split s = (j.hi - j.lo) <= SWEEP_LIMIT ? sweep_split(items, j.lo, j.hi) : binned_split(items, j.lo, j.hi);
        nd.axis = s.axis;
        nd.child[0] = (int32_t)tree.nodes.size();
        nd.child[1] = nd.child[0] + 1;
        tree.nodes[(size_t)j.node] = nd;
        tree.nodes.push_back(AccelNode{});
        tree.nodes.push_back(AccelNode{});
        jobs.push_back({j.lo, s.mid, nd.child[0]});
        jobs.push_back({s.mid, j.hi, nd.child[1]});
    }
    // subtree sizes: children always have larger indices than their parent
    for (size_t k = tree.nodes.size(); k-- > 0;) {
        AccelNode& nd = tree.nodes[k];
        if (nd.leaf == 0) nd.size = 1 + tree.nodes[(size_t)nd.child[0]].size + tree.nodes[(size_t)nd.child[1]].size;
    }
}

uint16_t accel_half(float x, bool up) { return accel_half_bits(x, up); }

uint32_t accel_emit_pairs(const AccelTree& tree, int octant, float scale, uint32_t base, std::vector<uint4>& out) {
    if (tree.nodes.empty()) return 0;
    auto word = [](const AccelNode& nd, uint32_t index) { return nd.leaf ? (RT_NODE_LEAF | nd.leaf) : index; };
    if (tree.nodes[0].leaf) { // a single leaf: one record that holds it twice
        const uint4 rec = accel_pack_node(tree.nodes[0].b, octant, scale, word(tree.nodes[0], 0));
        out.push_back(rec);
        out.push_back(rec);
        return 1;
    }
    struct Visit { int32_t node; uint32_t index; };
    std::vector<Visit> stack;
    stack.push_back({0, base});
    const size_t first = out.size();
    const uint32_t n_records = (tree.nodes[0].size - 1u) / 2u;
    out.resize(first + 2u * (size_t)n_records);
    while (!stack.empty()) {
        const Visit v = stack.back();
        stack.pop_back();
        const AccelNode& nd = tree.nodes[(size_t)v.node];
        const int near_side = (octant >> nd.axis) & 1; // negative direction: the upper child is nearer
        const AccelNode& near = tree.nodes[(size_t)nd.child[near_side]];
        const AccelNode& far = tree.nodes[(size_t)nd.child[1 - near_side]];
        // depth-first, near child first: the near child's record follows its parent's, the far child's comes after the
        // near subtree's (size - 1) / 2 records
        const uint32_t near_index = v.index + 1u, far_index = v.index + 1u + (near.size - 1u) / 2u;
        out[first + 2u * (size_t)(v.index - base)] = accel_pack_node(near.b, octant, scale, word(near, near_index));
        out[first + 2u * (size_t)(v.index - base) + 1u] = accel_pack_node(far.b, octant, scale, word(far, far_index));
        if (!far.leaf) stack.push_back({nd.child[1 - near_side], far_index});
        if (!near.leaf) stack.push_back({nd.child[near_side], near_index});
    }
    return n_records;
}

uint32_t accel_depth(const AccelTree& tree) {
    if (tree.nodes.empty()) return 0;
    std::vector<uint32_t> depth(tree.nodes.size(), 0u); // children have larger indices than their parent
    uint32_t deepest = 0;
    for (size_t k = 0; k < tree.nodes.size(); ++k) {
        const AccelNode& nd = tree.nodes[k];
        if (nd.leaf) { deepest = std::max(deepest, depth[k]); continue; }
        depth[(size_t)nd.child[0]] = depth[(size_t)nd.child[1]] = depth[k] + 1u;
    }
    return deepest;
}

void accel_emit(const AccelTree& tree, int octant, float scale, std::vector<uint4>& out) {
    if (tree.nodes.empty()) return;
    std::vector<int32_t> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const AccelNode& nd = tree.nodes[(size_t)stack.back()];
        stack.pop_back();
        uint32_t idx = (uint32_t)out.size();
        out.push_back(accel_pack_node(nd.b, octant, scale, nd.leaf ? (RT_NODE_LEAF | nd.leaf) : idx + nd.size));
        if (nd.leaf == 0) {
            int first = (octant >> nd.axis) & 1; // negative direction: the upper child is nearer
            stack.push_back(nd.child[1 - first]);
            stack.push_back(nd.child[first]);
        }
    }
}
