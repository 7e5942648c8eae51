// bvh_build_impl.h -- the body of the host-side BVH builders, included once per element type by bvh_build.cpp
// (inside a namespace, with `real` defined as float or double).  No include guard on purpose.


struct Box6 {
    real v[6];
};

inline void grow(Box6& b, const real* o) {
    for (int a = 0; a < 3; ++a) {
        b.v[a] = std::fmin(b.v[a], o[a]);
        b.v[3 + a] = std::fmax(b.v[3 + a], o[3 + a]);
    }
}

// findmax over (max - min) returns the first maximum (src/BVH.jl:76-77)
inline int split_axis(const Box6& b) {
    real ex = b.v[3] - b.v[0], ey = b.v[4] - b.v[1], ez = b.v[5] - b.v[2];
    int axis = 0;
    real m = ex;
    if (ey > m) { m = ey; axis = 1; }
    if (ez > m) { axis = 2; }
    return axis;
}

struct NodeOut {
    real* bounds;
    int32_t* left;
    int32_t* right;
    uint8_t* leaf;
    void set(int id, const Box6& b, int l, int r, bool is_leaf) const {
        std::memcpy(bounds + 6 * (size_t)(id - 1), b.v, sizeof(b.v));
        left[id - 1] = l;
        right[id - 1] = r;
        leaf[id - 1] = is_leaf ? 1 : 0;
    }
};

// ---- generic objects: src/BVH.jl:65-96.  order[] is tree.objects, permuted in place.
struct GenericBuild {
    const real* bounds;
    int32_t* order;
    NodeOut out;

    Box6 enclose(int l, int r) const {
        Box6 b;
        std::memcpy(b.v, bounds + 6 * (size_t)order[l - 1], sizeof(b.v));
        for (int i = l + 1; i <= r; ++i) grow(b, bounds + 6 * (size_t)order[i - 1]);
        return b;
    }
    int build(int id, int l, int r) const {
        Box6 b = enclose(l, r);
        if (l == r) {
            out.set(id, b, l, r, true);
            return id;
        }
        int axis = split_axis(b);
        const real* bb = bounds;
        std::stable_sort(order + (l - 1), order + r,
                         [bb, axis](int32_t x, int32_t y) { return bb[6 * (size_t)x + axis] < bb[6 * (size_t)y + axis]; });
        int mid = (l + r) / 2;
        int ln = build(id + 1, l, mid);
        int rn = build(id + 2 * (mid + 1 - l), mid + 1, r);
        out.set(id, b, ln, rn, false);
        return id;
    }
};

// ---- meshes: src/mesh.jl:217-283
struct MeshRef {
    int nv, nt;
    const real* vertices; // 8 floats per vertex
    int32_t* indices;      // 4 per triangle, permuted in place
    int32_t* tri_ids;
};

// src/mesh.jl:45-49: minimum/maximum of the positions, padded by the Float64 literal 0.0001, narrowed to Float32
inline Box6 padded(const real mn[3], const real mx[3]) {
    Box6 b;
    for (int a = 0; a < 3; ++a) {
        b.v[a] = (real)((double)mn[a] - 0.0001);
        b.v[3 + a] = (real)((double)mx[a] + 0.0001);
    }
    return b;
}
inline void span_vertex(const MeshRef& m, int vid, real mn[3], real mx[3]) {
    const real* p = m.vertices + 8 * (size_t)vid;
    for (int a = 0; a < 3; ++a) {
        mn[a] = std::fmin(mn[a], p[a]);
        mx[a] = std::fmax(mx[a], p[a]);
    }
}
Box6 mesh_box_all(const MeshRef& m) {
    real mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = 0; v < m.nv; ++v) span_vertex(m, v, mn, mx);
    return padded(mn, mx);
}
Box6 mesh_box_cols(const MeshRef& m, int l, int r) { // columns l..r, 1-based inclusive
    real mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = l; c <= r; ++c)
        for (int k = 0; k < 3; ++k) span_vertex(m, m.indices[4 * (size_t)(c - 1) + k], mn, mx);
    return padded(mn, mx);
}

struct MeshBuild {
    std::vector<MeshRef> meshes;
    std::vector<Box6> mesh_boxes;
    int32_t* order; // tree.objects (mesh ids), permuted in place
    NodeOut out;
    size_t slots;
    bool overflow = false;

    void set(int id, const Box6& b, int l, int r, bool leaf) {
        if (id < 1 || (size_t)id > slots) { overflow = true; return; }
        out.set(id, b, l, r, leaf);
    }
    // src/mesh.jl:217-248
    int build_tris(MeshRef& m, int id, int l, int r, bool left_child) {
        const int max_triangles = 1;
        Box6 b = mesh_box_cols(m, l, r);
        if (r - l < max_triangles) {
            set(id, b, l, r, true);
            return id;
        }
        int axis = split_axis(b);
        struct Col { real key; int32_t idx[4]; int32_t tri; };
        std::vector<Col> cols((size_t)(r - l + 1));
        for (int c = l; c <= r; ++c) {
            Col& col = cols[(size_t)(c - l)];
            std::memcpy(col.idx, m.indices + 4 * (size_t)(c - 1), sizeof(col.idx));
            col.tri = m.tri_ids[c - 1];
            real mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = 0; k < 3; ++k) span_vertex(m, col.idx[k], mn, mx);
            col.key = padded(mn, mx).v[axis];
        }
        std::stable_sort(cols.begin(), cols.end(), [](const Col& x, const Col& y) { return x.key < y.key; });
        for (int c = l; c <= r; ++c) {
            std::memcpy(m.indices + 4 * (size_t)(c - 1), cols[(size_t)(c - l)].idx, sizeof(cols[0].idx));
            m.tri_ids[c - 1] = cols[(size_t)(c - l)].tri;
        }
        int mid = (l + r) / 2;
        int ln = build_tris(m, id * 2, l, mid, left_child);
        int rn = build_tris(m, id * (2 - (left_child ? 1 : 0)) + 1, mid + 1, r, false);
        set(id, b, ln, rn, false);
        return id;
    }
    // src/mesh.jl:250-283
    int build_level(int id, int l, int r) {
        if (l == r) {
            MeshRef& m = meshes[(size_t)order[r - 1]];
            int ln = build_tris(m, id * 2, 1, m.nt, true);
            set(id, mesh_boxes[(size_t)order[l - 1]], ln, r, true);
            return id;
        }
        Box6 b = mesh_boxes[(size_t)order[l - 1]];
        for (int i = l + 1; i <= r; ++i) grow(b, mesh_boxes[(size_t)order[i - 1]].v);
        int axis = split_axis(b);
        const std::vector<Box6>& mb = mesh_boxes;
        std::stable_sort(order + (l - 1), order + r,
                         [&mb, axis](int32_t x, int32_t y) { return mb[(size_t)x].v[axis] < mb[(size_t)y].v[axis]; });
        int mid = (l + r) / 2;
        int ln = build_level(id * 2, l, mid);
        int rn = build_level(id * 2 + 1, mid + 1, r);
        set(id, b, ln, rn, false);
        return id;
    }
};

