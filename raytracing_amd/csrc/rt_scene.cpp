// rt_scene.cpp -- turns the host's scene description (include/rt_amd.h: per-type trees as build_bvh left them,
// object / transform / material / texture tables) into the device layout of rt_device.h.  This is the job that
// cu(::BVHTree), CuSceneManager and the Adapt rules do in the reference (src/cuBVH.jl:20-22,
// src/cuscene_manager.jl:16-19, src/mesh.jl:24-43, src/transform.jl:12-15, src/constant_medium.jl:12-16).
#include "rt_scene.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "accel_build.h"
#include "rt_internal.h"

namespace {

inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bitsf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct Builder {
    const rtSceneDesc& D;
    HostScene& H;
    std::string err;
    int status = RT_OK;

    std::vector<int64_t> group_root;          // per group: root of its tree relative to the first nested tree, -1 = not collected
    std::vector<uint32_t> group_size;         // per group: nodes in its tree
    uint32_t n_group_trees = 0;               // nested trees collected so far (tree ids 2, 3, ...)
    uint32_t group_nodes = 0;                 // ... and their nodes
    std::vector<uint32_t> instance_prims;     // prims whose g0 holds a relative sub-tree root
    std::map<std::vector<uint32_t>, uint32_t> xform_index; // identical matrices share one device record
    std::vector<int> top_pos;                 // object index -> position in `top`
    size_t visited = 0;

    explicit Builder(const rtSceneDesc& d, HostScene& h) : D(d), H(h) {}

    bool fail(int st, const std::string& m) {
        if (status == RT_OK) { status = st; err = m; }
        return false;
    }

    // ---------------------------------------------------------------- tables
    bool needs_uv(int ti, int depth = 0) {
        if (ti < 0 || ti >= D.n_textures || depth > 8) return false;
        const rtTexture& t = D.textures[ti];
        if (t.kind == RT_TEX_IMAGE) return true;
        if (t.kind == RT_TEX_CHECKER) return needs_uv(t.a, depth + 1) || needs_uv(t.b, depth + 1);
        return false;
    }

    bool tables() {
        for (int i = 0; i < D.n_textures; ++i) {
            const rtTexture& t = D.textures[i];
            if (t.kind != RT_TEX_SOLID) H.textured = true;
            switch (t.kind) {
                case RT_TEX_SOLID: break;
                case RT_TEX_CHECKER:
                    if (t.a < 0 || t.a >= D.n_textures || t.b < 0 || t.b >= D.n_textures) return fail(RT_EINVAL, "checker texture index out of range");
                    break;
                case RT_TEX_IMAGE:
                    if (t.a < 0 || t.a >= D.n_images) return fail(RT_EINVAL, "image index out of range");
                    break;
                case RT_TEX_NOISE_MARBLE:
                case RT_TEX_NOISE_SCALAR:
                    if (t.a < 0 || t.a >= D.n_perlins) return fail(RT_EINVAL, "perlin index out of range");
                    break;
                default: return fail(RT_EUNSUPPORTED, "unknown texture kind");
            }
            H.textures.push_back(make_uint4((uint32_t)t.kind, (uint32_t)t.a, (uint32_t)t.b, 0u));
            H.textures.push_back(make_uint4(fbits(t.color[0]), fbits(t.color[1]), fbits(t.color[2]), fbits(t.scale)));
        }
        for (int i = 0; i < D.n_materials; ++i) {
            const rtMaterial& m = D.materials[i];
            if (m.type < RT_LAMBERTIAN || m.type > RT_ISOTROPIC) return fail(RT_EUNSUPPORTED, "unknown material type");
            if (m.texture < 0 || m.texture >= D.n_textures) return fail(RT_EINVAL, "material texture index out of range");
            uint32_t flags = needs_uv(m.texture) ? DM_NEEDS_UV : 0u;
            H.materials.push_back(make_uint4((uint32_t)m.type | flags, (uint32_t)m.texture, fbits(m.fuzz), fbits(m.eta)));
        }
        for (int i = 0; i < D.n_images; ++i) {
            const rtImage& im = D.images[i];
            if (im.width < 0 || im.height < 0 || (im.width > 0 && im.height > 0 && !im.rgb)) return fail(RT_EINVAL, "bad image");
            DevImage d{(uint32_t)H.image_data.size(), (uint32_t)im.width, (uint32_t)im.height, 0u};
            size_t bytes = (size_t)3 * im.width * im.height;
            if (bytes) H.image_data.insert(H.image_data.end(), im.rgb, im.rgb + bytes);
            while (H.image_data.size() & 3u) H.image_data.push_back(0);
            H.images.push_back(d);
        }
        for (int i = 0; i < D.n_perlins; ++i) {
            const rtPerlin& p = D.perlins[i];
            H.perlin_vec.insert(H.perlin_vec.end(), p.ranvec, p.ranvec + 768);
            for (int k = 0; k < 256; ++k) if (p.perm_x[k] < 0 || p.perm_x[k] > 255 || p.perm_y[k] < 0 || p.perm_y[k] > 255 || p.perm_z[k] < 0 || p.perm_z[k] > 255)
                return fail(RT_EINVAL, "perlin permutation entry outside 0..255");
            H.perlin_perm.insert(H.perlin_perm.end(), p.perm_x, p.perm_x + 256);
            H.perlin_perm.insert(H.perlin_perm.end(), p.perm_y, p.perm_y + 256);
            H.perlin_perm.insert(H.perlin_perm.end(), p.perm_z, p.perm_z + 256);
        }
        return true;
    }

    uint32_t xform(int ti) {
        const rtTransform& t = D.transforms[ti];
        std::vector<uint32_t> key(24);
        for (int k = 0; k < 12; ++k) { key[k] = fbits(t.xform[k]); key[12 + k] = fbits(t.inv_xform[k]); }
        auto it = xform_index.find(key);
        if (it != xform_index.end()) return it->second;
        uint32_t idx = (uint32_t)(H.xforms.size() / 24);
        H.xforms.insert(H.xforms.end(), t.xform, t.xform + 12);
        H.xforms.insert(H.xforms.end(), t.inv_xform, t.inv_xform + 12);
        xform_index[key] = idx;
        return idx;
    }

    // ---------------------------------------------------------------- primitives
    uint32_t push_prim(uint32_t head, uint32_t material, uint32_t rank, const float g[9], int top, int sub) {
        uint32_t idx = (uint32_t)(H.prims.size() / 3);
        H.prims.push_back(make_uint4(head, material, rank, fbits(g[0])));
        H.prims.push_back(make_uint4(fbits(g[1]), fbits(g[2]), fbits(g[3]), fbits(g[4])));
        H.prims.push_back(make_uint4(fbits(g[5]), fbits(g[6]), fbits(g[7]), fbits(g[8])));
        H.prim_top.push_back(top);
        H.prim_sub.push_back(sub);
        // inverse matrix of the Transform around the primitive, next to it so that a leaf test needs one round of loads
        uint32_t kind = head & 15u, field = head >> 8;
        float m[12] = {0};
        if (field && kind != DP_TRIANGLE) std::memcpy(m, &H.xforms[24 * (size_t)(field - 1) + 12], sizeof(m));
        for (int k = 0; k < 3; ++k) H.prim_inv.push_back(make_float4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]));
        for (int k = 0; k < 4; ++k) H.prim_box.push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        return idx;
    }
    // The boxes the reference tests before it reaches the primitive (src/BVH.jl:22-53, src/mesh.jl:115-163): only the
    // leaf's own box matters, every box above it contains it and is tested with the same ray and the same arithmetic.
    // outer: leaf box in the top-level tree, ray as cast; inner: leaf box inside a nested tree, ray in that tree's frame.
    void set_ref_boxes(uint32_t prim, const float* outer, const float* inner) {
        float4* b = &H.prim_box[4 * (size_t)prim];
        uint32_t flags = 0;
        if (outer) {
            b[0] = make_float4(outer[0], outer[1], outer[2], outer[3]);
            b[1].x = outer[4]; b[1].y = outer[5];
            flags |= RT_BOX_OUTER;
        }
        if (inner) {
            b[2] = make_float4(inner[0], inner[1], inner[2], inner[3]);
            b[3] = make_float4(inner[4], inner[5], 0.0f, 0.0f);
            flags |= RT_BOX_INNER;
        }
        b[1].z = bitsf(flags);
    }

    bool valid_obj(int oi) { return oi >= 0 && oi < D.n_objects; }

    // number of tie-break ranks an object occupies
    uint32_t flat_count(int oi, int depth = 0) {
        if (!valid_obj(oi) || depth > 4) return 1;
        const rtObject* o = &D.objects[oi];
        if (o->kind == RT_TRANSFORM && valid_obj(o->child)) o = &D.objects[o->child];
        if (o->kind == RT_BVH && o->aux >= 0 && o->aux < D.n_groups) {
            const rtBvh& G = D.groups[o->aux];
            uint32_t acc = 0;
            for (int k = 0; k < G.n_objects && G.objects; ++k) acc += flat_count(G.objects[k], depth + 1);
            return acc;
        }
        if (o->kind == RT_MESH && o->aux >= 0 && o->aux < D.n_meshes) return (uint32_t)D.meshes[o->aux].n_triangles;
        return 1;
    }

    // Object (possibly wrapped: ConstantMedium{Transform{shape}}) -> one device primitive.  Returns false on error.
    // outer_xf: transform index + 1 of the Transform around the nested tree the object is taken out of (0: none)
    bool make_prim(int oi, uint32_t rank, int top, int sub, bool nested, uint32_t outer_xf, uint32_t& out) {
        if (!valid_obj(oi)) return fail(RT_EINVAL, "object index out of range");
        const rtObject* o = &D.objects[oi];
        bool medium = false;
        float neg_inv_density = 0.0f;
        int material = -1, medium_oi = -1;
        uint32_t xf = outer_xf;
        if (o->kind == RT_CONSTANT_MEDIUM) {
            if (nested) return fail(RT_EUNSUPPORTED, "ConstantMedium inside a nested BVHTree is not supported");
            medium = true;
            neg_inv_density = o->p[0];
            material = o->material;
            medium_oi = oi;
            if (!valid_obj(o->child)) return fail(RT_EINVAL, "ConstantMedium boundary index out of range");
            o = &D.objects[o->child];
        }
        if (o->kind == RT_TRANSFORM) {
            if (o->aux < 0 || o->aux >= D.n_transforms) return fail(RT_EINVAL, "transform index out of range");
            if (nested) return fail(RT_EUNSUPPORTED, "Transform inside a nested BVHTree is not supported");
            xf = xform(o->aux) + 1u;
            if (!valid_obj(o->child)) return fail(RT_EINVAL, "Transform child index out of range");
            o = &D.objects[o->child];
        }
        float g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t kind;
        switch (o->kind) {
            case RT_SPHERE: kind = DP_SPHERE; std::memcpy(g, o->p, 4 * sizeof(float)); break;
            case RT_MOVING_SPHERE: kind = DP_MOVING_SPHERE; std::memcpy(g, o->p, 9 * sizeof(float)); break;
            case RT_XY_RECT: kind = DP_XY_RECT; std::memcpy(g, o->p, 5 * sizeof(float)); break;
            case RT_XZ_RECT: kind = DP_XZ_RECT; std::memcpy(g, o->p, 5 * sizeof(float)); break;
            case RT_YZ_RECT: kind = DP_YZ_RECT; std::memcpy(g, o->p, 5 * sizeof(float)); break;
            case RT_BOX: kind = DP_BOX; std::memcpy(g, o->p, 6 * sizeof(float)); break;
            case RT_BVH: {
                if (medium) return fail(RT_EUNSUPPORTED, "ConstantMedium with a BVHTree boundary is not supported");
                if (nested) return fail(RT_EUNSUPPORTED, "BVHTree inside a nested BVHTree is not supported");
                if (o->aux < 0 || o->aux >= D.n_groups) return fail(RT_EINVAL, "group index out of range");
                int64_t root;
                if (!emit_group(o->aux, top, root)) return false;
                g[0] = bitsf((uint32_t)root);
                g[1] = bitsf(group_size[(size_t)o->aux]);
                out = push_prim(DP_INSTANCE | (xf << 8), 0u, rank, g, top, 0);
                instance_prims.push_back(out);
                return true;
            }
            case RT_MESH: return fail(RT_EUNSUPPORTED, "a Mesh is only supported as a top-level object");
            default: return fail(RT_EUNSUPPORTED, "unsupported nesting of Transform / ConstantMedium");
        }
        if (medium) {
            if (kind != DP_SPHERE && kind != DP_BOX) return fail(RT_EUNSUPPORTED, "ConstantMedium boundary must be a Sphere or a Box (optionally inside a Transform)");
            g[6] = neg_inv_density;
            g[7] = bitsf((uint32_t)medium_oi);
        } else {
            material = o->material;
        }
        if (material < 0 || material >= D.n_materials) return fail(RT_EINVAL, "material index out of range");
        uint32_t type = ((uint32_t)D.materials[material].type << 5) & DP_TYPE_MASK;
        out = push_prim(kind | (medium ? DP_MEDIUM_FLAG : 0u) | type | (xf << 8), (uint32_t)material, rank, g, top, sub);
        return true;
    }

    // ---------------------------------------------------------------- trees
    // The host trees (src/BVH.jl, src/mesh.jl) are walked once, depth first and left child first, to create the
    // primitives with the tie-break ranks their tree positions imply and to pick up the reference's bounding box of
    // every primitive.  The boxes then go to the device tree builder (accel_build.cpp).
    bool check_node(const rtBvh& T, int id, int depth) {
        if (id < 1 || id > T.n_nodes) return fail(RT_EINVAL, "BVH node id out of range");
        if (depth > 512) return fail(RT_EINVAL, "BVH deeper than 512 levels");
        if (++visited > (size_t)4 * (size_t)(T.n_nodes + 1) + 1024) return fail(RT_EINVAL, "BVH links form a cycle");
        return true;
    }
    // ref: the box the reference walks for the primitive; world_lo / world_hi: for an object taken out of a nested tree,
    // the world-frame bounds of its leaf box in that tree
    void add_item(uint32_t tree, const float* ref, uint32_t prim, uint32_t flags, const double* world_lo = nullptr,
                  const double* world_hi = nullptr) {
        HostItem it;
        it.tree = tree;
        it.leaf = (prim + 1u) | flags;
        std::memcpy(it.ref, ref, sizeof(it.ref));
        it.inlined = world_lo != nullptr;
        for (int i = 0; i < 3; ++i) { it.wlo[i] = world_lo ? world_lo[i] : 0.0; it.whi[i] = world_hi ? world_hi[i] : 0.0; }
        H.items.push_back(it);
    }

    // A nested tree taken apart: its objects become primitives of the enclosing tree, each carrying the Transform of the
    // nested tree.  hit(::Transform{BVHTree}) (src/transform.jl:104-116) takes the ray into the object frame once and
    // keeps the closest object; testing every object with the same object-frame ray gives the same distances, the same
    // winner (ranks = rank of the nested tree + position inside it) and the same hit record.
    struct Inlined {
        uint32_t xf;           // transform index + 1, 0 = none
        const float* clip_box; // the box the reference walks for the whole nested tree (world frame)
    };
    bool inline_groups() const {
        const char* e = std::getenv("RT_AMD_INLINE_GROUPS");
        return !(e && e[0] == '0');
    }

    // generic tree (src/BVH.jl)
    bool collect_generic(const rtBvh& T, int id, int depth, uint32_t tree, uint32_t rank_base,
                         const std::vector<uint32_t>* rank_offset, bool nested, int nested_top, const Inlined* inl = nullptr) {
        if (!check_node(T, id, depth)) return false;
        if (T.node_leaf[id - 1]) {
            int l = T.node_left[id - 1], r = T.node_right[id - 1];
            if (l != r || l < 1 || l > T.n_objects) return fail(RT_EINVAL, "BVH leaf must reference exactly one object");
            int oi = T.objects[l - 1];
            uint32_t rank = rank_base + (rank_offset ? (*rank_offset)[(size_t)(l - 1)] : (uint32_t)(l - 1));
            int top = nested ? nested_top : (valid_obj(oi) ? top_pos[(size_t)oi] : -1);
            const float* leaf_box = T.node_bounds + 6 * (size_t)(id - 1);
            if (!nested && valid_obj(oi) && inline_groups()) {
                const rtObject* o = &D.objects[oi];
                uint32_t xf = 0;
                if (o->kind == RT_TRANSFORM && o->aux >= 0 && o->aux < D.n_transforms && valid_obj(o->child) &&
                    D.objects[o->child].kind == RT_BVH) {
                    xf = xform(o->aux) + 1u;
                    o = &D.objects[o->child];
                }
                if (o->kind == RT_BVH && o->aux >= 0 && o->aux < D.n_groups) {
                    const rtBvh& G = D.groups[o->aux];
                    if (G.kind == 0 && G.n_objects > 0 && G.n_nodes > 0 && G.objects && G.node_bounds && G.node_left &&
                        G.node_right && G.node_leaf) {
                        std::vector<uint32_t> offsets;
                        uint32_t acc = 0;
                        for (int k = 0; k < G.n_objects; ++k) {
                            offsets.push_back(acc);
                            acc += flat_count(G.objects[k]);
                        }
                        Inlined ctx{xf, leaf_box};
                        size_t saved = visited;
                        visited = 0;
                        bool ok = collect_generic(G, 1, 0, tree, rank, &offsets, true, top, &ctx);
                        visited = saved;
                        return ok;
                    }
                }
            }
            uint32_t prim;
            if (!make_prim(oi, rank, top, nested ? oi : 0, nested, inl ? inl->xf : 0u, prim)) return false;
            uint32_t head = H.prims[3 * (size_t)prim].x;
            bool instance = (head & 15u) == DP_INSTANCE;
            set_ref_boxes(prim, nested ? (inl ? inl->clip_box : nullptr) : leaf_box, nested ? leaf_box : nullptr);
            const uint32_t flags = instance ? RT_LEAF_INSTANCE
                                            : (((head & 15u) != DP_TRIANGLE && (head >> 8)) ? RT_LEAF_XFORM : 0u) | ((head & 15u) == DP_BOX ? RT_LEAF_BOX : 0u);
            if (inl) { // the object-frame box of the leaf in the world frame; the reference's box is that of the whole nested tree
                double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                double M[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
                if (inl->xf) for (int k = 0; k < 12; ++k) M[k] = (double)H.xforms[24 * (size_t)(inl->xf - 1) + k];
                for (int corner = 0; corner < 8; ++corner) {
                    double p[3];
                    for (int a = 0; a < 3; ++a) p[a] = leaf_box[((corner >> a) & 1) ? 3 + a : a];
                    for (int i = 0; i < 3; ++i) {
                        double w = M[i] * p[0] + M[3 + i] * p[1] + M[6 + i] * p[2] + M[9 + i];
                        lo[i] = std::min(lo[i], w); hi[i] = std::max(hi[i], w);
                    }
                }
                add_item(tree, inl->clip_box, prim, flags, lo, hi);
            } else {
                add_item(tree, leaf_box, prim, flags);
            }
            return true;
        }
        return collect_generic(T, T.node_left[id - 1], depth + 1, tree, rank_base, rank_offset, nested, nested_top, inl) &&
               collect_generic(T, T.node_right[id - 1], depth + 1, tree, rank_base, rank_offset, nested, nested_top, inl);
    }

    // A nested tree kept as an instance: its objects are collected into a tree of their own (a binary tree over n leaves
    // has 2n - 1 nodes, so root and size are known before any box is).
    bool emit_group(int gi, int top, int64_t& root) {
        if (group_root[(size_t)gi] >= 0) { root = group_root[(size_t)gi]; return true; }
        const rtBvh& T = D.groups[gi];
        if (T.kind != 0) return fail(RT_EUNSUPPORTED, "nested tree of meshes is not supported");
        if (T.n_objects <= 0 || T.n_nodes <= 0 || !T.objects || !T.node_bounds || !T.node_left || !T.node_right || !T.node_leaf)
            return fail(RT_EINVAL, "empty or incomplete nested tree");
        size_t saved = visited;
        visited = 0;
        std::vector<uint32_t> offsets;
        uint32_t acc = 0;
        for (int k = 0; k < T.n_objects; ++k) {
            offsets.push_back(acc);
            acc += flat_count(T.objects[k]);
        }
        const uint32_t tree = 2u + n_group_trees++;
        const size_t first_item = H.items.size();
        bool ok = collect_generic(T, 1, 0, tree, 0u, &offsets, true, top);
        visited = saved;
        if (!ok) return false;
        const uint32_t n_items = (uint32_t)(H.items.size() - first_item);
        group_root[(size_t)gi] = (int64_t)group_nodes;
        group_size[(size_t)gi] = n_items ? 2u * n_items - 1u : 0u;
        group_nodes += group_size[(size_t)gi];
        root = group_root[(size_t)gi];
        return true;
    }

    // power of two that brings the largest finite box coordinate of the scene just below 2^15, well inside the range
    // of the half-precision node boxes
    void choose_node_scale() {
        float m = 0.0f;
        auto scan = [&m](const rtBvh& T) {
            if (T.n_nodes <= 0 || !T.node_bounds) return;
            for (int k = 0; k < 6; ++k) {
                float v = std::fabs(T.node_bounds[k]);
                if (std::isfinite(v) && v > m) m = v;
            }
        };
        for (int b = 0; b < D.n_buckets; ++b) scan(D.buckets[b]);
        for (int g = 0; g < D.n_groups; ++g) scan(D.groups[g]);
        int k = 0;
        if (m > 0.0f) {
            int e;
            std::frexp(m, &e); // m = f * 2^e, f in [0.5, 1)
            k = 15 - e;
        }
        k = std::max(-100, std::min(100, k));
        H.node_scale = std::ldexp(1.0f, k);
    }

    // tree over meshes (src/mesh.jl:250-283): every triangle becomes a primitive with its own bounding box
    bool collect_tris(const rtBvh& T, const rtMesh& M, int top, int id, int depth, uint32_t rank_base, uint32_t tree) {
        if (!check_node(T, id, depth)) return false;
        if (T.node_leaf[id - 1] == 1) {
            int l = T.node_left[id - 1], r = T.node_right[id - 1];
            if (l < 1 || r > M.n_triangles || l > r) return fail(RT_EINVAL, "triangle range out of bounds");
            for (int c = l; c <= r; ++c) {
                const int32_t* ix = M.indices + 4 * (size_t)(c - 1);
                for (int k = 0; k < 3; ++k) if (ix[k] < 0 || ix[k] >= M.n_vertices) return fail(RT_EINVAL, "vertex index out of range");
                if (ix[3] < 0 || ix[3] >= M.n_materials) return fail(RT_EINVAL, "mesh material id out of range");
                int material = M.materials[ix[3]];
                if (material < 0 || material >= D.n_materials) return fail(RT_EINVAL, "material index out of range");
                float g[9];
                uint32_t attr = (uint32_t)(H.tri_attr.size() / 16);
                float a[16] = {0};
                for (int k = 0; k < 3; ++k) {
                    const float* v = M.vertices + 8 * (size_t)ix[k];
                    g[3 * k] = v[0]; g[3 * k + 1] = v[1]; g[3 * k + 2] = v[2];
                    a[3 * k] = v[3]; a[3 * k + 1] = v[4]; a[3 * k + 2] = v[5];
                    a[9 + 2 * k] = v[6]; a[10 + 2 * k] = v[7];
                }
                H.tri_attr.insert(H.tri_attr.end(), a, a + 16);
                uint32_t type = ((uint32_t)D.materials[material].type << 5) & DP_TYPE_MASK;
                uint32_t prim = push_prim(DP_TRIANGLE | type | (attr << 8), (uint32_t)material, rank_base + (uint32_t)(c - 1), g, top,
                                          M.tri_ids ? M.tri_ids[c - 1] : c - 1);
                const float* lb = T.node_bounds + 6 * (size_t)(id - 1);
                set_ref_boxes(prim, lb, nullptr);
                add_item(tree, lb, prim, 0u);
            }
            return true;
        }
        if (T.node_leaf[id - 1] != 0) return fail(RT_EINVAL, "reached an unused BVH node slot");
        return collect_tris(T, M, top, T.node_left[id - 1], depth + 1, rank_base, tree) &&
               collect_tris(T, M, top, T.node_right[id - 1], depth + 1, rank_base, tree);
    }
    bool collect_mesh_level(const rtBvh& T, int id, int depth, uint32_t rank_base, const std::vector<uint32_t>& rank_offset,
                            uint32_t tree) {
        if (!check_node(T, id, depth)) return false;
        if (T.node_leaf[id - 1] > 1) return fail(RT_EINVAL, "reached an unused BVH node slot");
        if (T.node_leaf[id - 1]) {
            int pos = T.node_right[id - 1];
            if (pos < 1 || pos > T.n_objects) return fail(RT_EINVAL, "mesh position out of range");
            int oi = T.objects[pos - 1];
            if (!valid_obj(oi) || D.objects[oi].kind != RT_MESH) return fail(RT_EINVAL, "mesh tree over a non-mesh object");
            int mi = D.objects[oi].aux;
            if (mi < 0 || mi >= D.n_meshes) return fail(RT_EINVAL, "mesh index out of range");
            const rtMesh& M = D.meshes[mi];
            if (M.n_triangles < 0 || M.n_vertices < 0 || M.n_materials < 0) return fail(RT_EINVAL, "mesh with a negative count");
            if ((M.n_triangles > 0 && (!M.indices || !M.vertices)) || (M.n_materials > 0 && !M.materials) || (M.n_triangles > 0 && M.n_materials == 0))
                return fail(RT_EINVAL, "mesh has a null table with a non-zero count");
            return collect_tris(T, M, top_pos[(size_t)oi], T.node_left[id - 1], depth + 1,
                                rank_base + rank_offset[(size_t)(pos - 1)], tree);
        }
        return collect_mesh_level(T, T.node_left[id - 1], depth + 1, rank_base, rank_offset, tree) &&
               collect_mesh_level(T, T.node_right[id - 1], depth + 1, rank_base, rank_offset, tree);
    }

    bool run() {
        if (D.n_objects < 0 || D.n_top < 0 || D.n_buckets < 0) return fail(RT_EINVAL, "negative count");
        if (D.n_buckets == 0) return fail(RT_EINVAL, "scene has no objects");
        if (!tables()) return false;
        top_pos.assign((size_t)D.n_objects, -1);
        for (int i = 0; i < D.n_top; ++i) {
            if (!valid_obj(D.top[i])) return fail(RT_EINVAL, "top-level object index out of range");
            top_pos[(size_t)D.top[i]] = i;
        }
        group_root.assign((size_t)D.n_groups, -1);
        group_size.assign((size_t)D.n_groups, 0u);
        // tie-break ranks: earlier type group wins, inside a group the later tree position wins
        std::vector<std::vector<uint32_t>> offsets((size_t)D.n_buckets);
        std::vector<uint32_t> totals((size_t)D.n_buckets, 0);
        for (int b = 0; b < D.n_buckets; ++b) {
            const rtBvh& T = D.buckets[b];
            if (T.n_objects <= 0 || T.n_nodes <= 0 || !T.objects || !T.node_bounds || !T.node_left || !T.node_right || !T.node_leaf)
                return fail(RT_EINVAL, "empty or incomplete tree record");
            uint32_t acc = 0;
            for (int k = 0; k < T.n_objects; ++k) {
                offsets[(size_t)b].push_back(acc);
                acc += flat_count(T.objects[k]);
            }
            totals[(size_t)b] = acc;
        }
        std::vector<uint32_t> base((size_t)D.n_buckets, 0);
        for (int b = D.n_buckets - 2; b >= 0; --b) base[(size_t)b] = base[(size_t)b + 1] + totals[(size_t)b + 1];
        H.bucket_base = base;
        // tree 0: every surface; "tree" 1: the constant media, which the kernels take as a list after the surfaces (they are
        // few) -- no nodes are built for them; trees 2..: nested trees kept as instances.  Collected here as leaves with the
        // reference's boxes; rt_build_nodes makes the device trees.
        choose_node_scale();
        uint32_t n_main[2] = {0, 0};
        for (uint32_t region = 0; region < 2; ++region) {
            const size_t before = H.items.size();
            for (int b = 0; b < D.n_buckets; ++b) {
                const rtBvh& T = D.buckets[b];
                int first = T.objects[0];
                bool is_medium = valid_obj(first) && D.objects[first].kind == RT_CONSTANT_MEDIUM;
                if (is_medium != (region == 1)) continue;
                visited = 0;
                bool ok = T.kind == 1 ? collect_mesh_level(T, 1, 0, base[(size_t)b], offsets[(size_t)b], region)
                                      : collect_generic(T, 1, 0, region, base[(size_t)b], &offsets[(size_t)b], false, -1);
                if (!ok) return false;
            }
            for (size_t k = before; k < H.items.size(); ++k) n_main[region] += H.items[k].tree == region ? 1u : 0u;
        }
        // layout of one octant: surfaces, media, then the nested trees in the order they were collected
        H.surface_root = 0;
        H.surface_size = n_main[0] ? 2u * n_main[0] - 1u : 0u;
        H.medium_root = H.surface_size;
        H.medium_size = 0u;
        const uint64_t main_count = (uint64_t)H.surface_size + H.medium_size;
        const uint64_t per_layout = main_count + group_nodes;
        if (16 * per_layout >= (uint64_t)RT_NODE_LEAF - 1u) return fail(RT_EUNSUPPORTED, "too many BVH nodes");
        if (H.prims.size() / 3 >= (size_t)RT_LEAF_PRIM_MASK) return fail(RT_EUNSUPPORTED, "too many primitives");
        H.nodes_per_layout = (uint32_t)per_layout;
        H.nested = group_nodes != 0;
        H.tree_first.assign(2u + n_group_trees, 0u);
        H.tree_first[1] = H.medium_root;
        {
            std::vector<uint32_t> count(2u + n_group_trees, 0u);
            for (const HostItem& it : H.items) count[it.tree]++;
            uint32_t at = (uint32_t)main_count;
            for (uint32_t t = 2; t < 2u + n_group_trees; ++t) { H.tree_first[t] = at; at += count[t] ? 2u * count[t] - 1u : 0u; }
            if (at != per_layout) return fail(RT_EINVAL, "internal: node count of the nested trees does not add up");
        }
        for (uint32_t p : instance_prims) H.prims[3 * (size_t)p].w += (uint32_t)main_count; // g0 holds the root as raw bits
        for (const HostItem& it : H.items)
            if (it.tree == 1u) H.media.push_back((it.leaf & RT_LEAF_PRIM_MASK) - 1u);
        std::sort(H.media.begin(), H.media.end(), [this](uint32_t a, uint32_t b) {
            return H.prims[3 * (size_t)a].z > H.prims[3 * (size_t)b].z; // rank: higher = met earlier by the reference
        });
        for (uint32_t m : H.media) {
            const uint32_t rank = H.prims[3 * (size_t)m].z;
            uint32_t g = 0;
            while (g + 1u < (uint32_t)D.n_buckets && rank < base[g]) ++g;
            H.media_group.push_back(g);
        }
        H.scene_reach = 0.0;
        for (const HostItem& it : H.items)
            for (int k = 0; k < 6; ++k)
                if (std::isfinite(it.ref[k])) H.scene_reach = std::max(H.scene_reach, (double)std::fabs(it.ref[k]));
        rt_build_nodes(&H, 1.05 * H.scene_reach); // rays scattered inside the scene start inside its boxes
        return true;
    }
};

// Bounds of primitive `prim` for the tight node set: never larger than `loose` (the box the reference walks for it,
// so nothing is found that the reference culls), but tight around spheres and rectangles where the reference's box is
// loose (the box of a rotated Transform is the box of eight corners).  The primitive tests accept points slightly outside
// the exact shape, the further the ray has come the more: the sphere test (src/sphere.jl:46-63) loses about
// 15 ulp * |o - c|^2 of its discriminant to cancellation, the rectangle test (src/aarect.jl:49-95) about 3 ulp * |o - p|
// of the in-plane coordinates.  The padding covers that for rays that start within *valid_for of the primitive;
// returns false (out = loose) if the primitive is not tightened.
bool tight_box(const HostScene& H, uint32_t prim, const float* loose, float* out, double* valid_for) {
    std::memcpy(out, loose, 6 * sizeof(float));
    const uint4 h = H.prims[3 * (size_t)prim], a = H.prims[3 * (size_t)prim + 1];
    const uint32_t kind = h.x & 15u, field = h.x >> 8;
    if (h.x & DP_MEDIUM_FLAG) return false;
    if (kind != DP_SPHERE && kind != DP_XY_RECT && kind != DP_XZ_RECT && kind != DP_YZ_RECT) return false;
    const float g[5] = {bitsf(h.w), bitsf(a.x), bitsf(a.y), bitsf(a.z), bitsf(a.w)};
    double M[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; // forward matrix, column-major 3x4
    if (field) for (int k = 0; k < 12; ++k) M[k] = (double)H.xforms[24 * (size_t)(field - 1) + k];
    auto point = [&M](double x, double y, double z, double* w) {
        for (int i = 0; i < 3; ++i) w[i] = M[i] * x + M[3 + i] * y + M[6 + i] * z + M[9 + i];
    };
    double lo[3], hi[3];
    if (kind == DP_SPHERE) {
        double c[3];
        point(g[0], g[1], g[2], c);
        const double r = std::fabs((double)g[3]);
        double shrink = INFINITY; // smallest stretch of the matrix along an axis (1 for rotations)
        for (int i = 0; i < 3; ++i) {
            const double n = std::sqrt(M[i] * M[i] + M[3 + i] * M[3 + i] + M[6 + i] * M[6 + i]);
            shrink = std::min(shrink, n);
            const double e = r * n * 1.02 + 4e-6 * std::fabs(c[i]);
            lo[i] = c[i] - e; hi[i] = c[i] + e;
        }
        // 1.02^2 - 1 = 0.0404 of r^2 against 15 * 2^-24 * |o - c|^2: |o - c| < 212 r
        *valid_for = 200.0 * r * shrink;
    } else {
        const int ka = kind == DP_XY_RECT ? 2 : (kind == DP_XZ_RECT ? 1 : 0); // constant axis
        const int aa = kind == DP_YZ_RECT ? 1 : 0, ba = kind == DP_XY_RECT ? 1 : 2;
        for (int i = 0; i < 3; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
        for (int corner = 0; corner < 8; ++corner) {
            double p[3], w[3];
            p[aa] = (corner & 1) ? g[1] : g[0];
            p[ba] = (corner & 2) ? g[3] : g[2];
            p[ka] = (double)g[4] + ((corner & 4) ? 0.0001 : -0.0001); // the padding of src/aarect.jl:37-47
            point(p[0], p[1], p[2], w);
            for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], w[i]); hi[i] = std::max(hi[i], w[i]); }
        }
        const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
        for (int i = 0; i < 3; ++i) {
            const double e = 1e-4 * ext + 4e-6 * std::max(std::fabs(lo[i]), std::fabs(hi[i]));
            lo[i] -= e; hi[i] += e;
        }
        *valid_for = 250.0 * ext; // 1e-4 ext against 3 * 2^-24 * |o - p| (and as much again for the transform of the ray)
    }
    float box[6];
    bool tighter = false;
    for (int i = 0; i < 3; ++i) {
        if (!std::isfinite(lo[i]) || !std::isfinite(hi[i])) return false;
        box[i] = std::max(std::nextafter((float)lo[i], -INFINITY), loose[i]);
        box[3 + i] = std::min(std::nextafter((float)hi[i], INFINITY), loose[3 + i]);
        if (!(box[i] <= box[3 + i])) return false; // NaN or empty: keep the reference's box
        tighter = tighter || box[i] != loose[i] || box[3 + i] != loose[3 + i];
    }
    if (!tighter) return false;
    std::memcpy(out, box, sizeof(box));
    return true;
}

} // namespace

// Node sets.  Every leaf gets two boxes:
//   loose: the box the reference walks for it.  For an object taken out of a nested tree that is the box of its leaf in
//          that tree, carried into the world frame and padded by 2e-6 * reach -- the reference tests that box with the
//          ray taken into the tree's frame, which moves the ray by up to ~1e-6 of its origin's coordinates and of the
//          distance it has come -- and cut to the box of the whole nested tree (which the reference tests with the ray as cast).
//   tight: tight_box() of that, valid for rays that start close enough to the primitive.
// Set 0 is built from the tight boxes and serves the rays that start inside a ball in which all of them are valid; set 1,
// from the loose boxes, serves the others (rays scattered far out in a scene-wide medium, a camera far away).  Without a
// tightened primitive (or without a ball) there is one set.  A leaf holds one primitive, so both sets have the same
// trees at the same places.
void rt_build_nodes(HostScene* out, double reach) {
    HostScene& H = *out;
    H.reach = reach;
    const auto t_start = std::chrono::steady_clock::now();
    const size_t n = H.items.size();
    std::vector<AccelItem> loose(n), tight(n);
    double csum[3] = {0, 0, 0};
    std::vector<double> valid(n, INFINITY);
    size_t n_tight = 0;
    for (size_t k = 0; k < n; ++k) {
        const HostItem& it = H.items[k];
        loose[k].leaf = tight[k].leaf = it.leaf;
        std::memcpy(loose[k].b, it.ref, sizeof(it.ref));
        if (it.inlined) {
            const double pad = 2e-6 * reach;
            float box[6];
            bool ok = true;
            for (int i = 0; i < 3; ++i) {
                box[i] = std::max(std::nextafter((float)(it.wlo[i] - pad), -INFINITY), it.ref[i]);
                box[3 + i] = std::min(std::nextafter((float)(it.whi[i] + pad), INFINITY), it.ref[3 + i]);
                if (!(box[i] <= box[3 + i])) ok = false;
            }
            if (ok) std::memcpy(loose[k].b, box, sizeof(box)); // NaN / infinite: the box of the whole nested tree
        }
        const uint32_t prim = (it.leaf & RT_LEAF_PRIM_MASK) - 1u;
        if (tight_box(H, prim, loose[k].b, tight[k].b, &valid[k])) {
            n_tight++;
            for (int i = 0; i < 3; ++i) csum[i] += 0.5 * ((double)tight[k].b[i] + (double)tight[k].b[3 + i]);
        } else {
            valid[k] = INFINITY;
        }
    }
    double radius = INFINITY;
    H.centre[0] = H.centre[1] = H.centre[2] = 0.0f;
    if (n_tight) {
        for (int i = 0; i < 3; ++i) H.centre[i] = (float)(csum[i] / (double)n_tight);
        for (size_t k = 0; k < n; ++k) {
            if (valid[k] == INFINITY) continue;
            double d2 = 0;
            for (int i = 0; i < 3; ++i) {
                const double c = 0.5 * ((double)tight[k].b[i] + (double)tight[k].b[3 + i]) - (double)H.centre[i];
                d2 += c * c;
            }
            radius = std::min(radius, valid[k] - std::sqrt(d2));
        }
    }
    // Spheres that are not tightened limit the ball as well: their leaves must grow to hold every point the sphere test can
    // report (below), by sqrt(r^2 + 1.4e-6 D^2) - r for rays that have come at most D, and the ball is kept small enough
    // for that to stay within a tenth of the radius (1.4e-6 D^2 <= 0.21 r^2: D <= 387 r).  A scene of small spheres on a
    // huge one (the ground of samples/cover_in_one_weekend.jl is a sphere of radius 1000 around 488 of radius 0.2) would
    // otherwise carry boxes twenty radii wide for the sake of rays that start at its far end.
    auto world_radius = [&H](uint32_t prim) { // radius of a sphere primitive as the world sees it, at least; 0: not a sphere
        const uint4 h = H.prims[3 * (size_t)prim], g1 = H.prims[3 * (size_t)prim + 1], g2 = H.prims[3 * (size_t)prim + 2];
        const uint32_t kind = h.x & 15u;
        double r = 0.0;
        if (kind == DP_SPHERE) r = std::fabs((double)bitsf(g1.z));
        if (kind == DP_MOVING_SPHERE) r = std::fabs((double)bitsf(g2.w));
        if ((kind == DP_SPHERE || kind == DP_MOVING_SPHERE) && (h.x >> 8) != 0u) {
            const float* M = &H.xforms[24 * (size_t)((h.x >> 8) - 1u)];
            double shrink = INFINITY;
            for (int i = 0; i < 3; ++i) shrink = std::min(shrink, std::sqrt((double)M[i] * M[i] + (double)M[3 + i] * M[3 + i] + (double)M[6 + i] * M[6 + i]));
            r *= std::min(1.0, shrink);
        }
        return std::isfinite(r) ? r : 0.0;
    };
    // How far a leaf can be from a point of the ball, beyond the ball's radius.  The box of a leaf inside a nested tree is in
    // that tree's own frame: its place in the world is only known to lie within the scene.
    auto offset_from_centre = [&](size_t k) {
        double off = 0.0, c2 = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double a = std::max(std::fabs((double)loose[k].b[i] - H.centre[i]), std::fabs((double)loose[k].b[3 + i] - H.centre[i]));
            if (std::isfinite(a)) off += a * a;
            c2 += (double)H.centre[i] * H.centre[i];
        }
        return H.items[k].tree >= 2u ? std::sqrt(c2) + std::sqrt(3.0) * H.scene_reach : std::sqrt(off);
    };
    {
        std::vector<size_t> plain; // surface leaves that hold a sphere and keep the reference's box
        for (size_t k = 0; k < n; ++k)
            if (H.items[k].tree != 1u && valid[k] == INFINITY && world_radius((H.items[k].leaf & RT_LEAF_PRIM_MASK) - 1u) > 0.0) plain.push_back(k);
        if (!plain.empty()) {
            if (!n_tight) { // no tightened primitive to centre the ball on: the median of those spheres, axis by axis
                for (int i = 0; i < 3; ++i) {
                    std::vector<double> c;
                    for (size_t k : plain) c.push_back(0.5 * ((double)loose[k].b[i] + (double)loose[k].b[3 + i]));
                    std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
                    H.centre[i] = std::isfinite(c[c.size() / 2]) ? (float)c[c.size() / 2] : 0.0f;
                }
            }
            for (size_t k : plain)
                radius = std::min(radius, 387.0 * world_radius((H.items[k].leaf & RT_LEAF_PRIM_MASK) - 1u) - offset_from_centre(k));
        }
    }
    for (int i = 0; i < 3; ++i) { // (used to sort rays by where they start: any point does, a central one sorts best)
        std::vector<double> c;
        for (size_t k = 0; k < n; ++k) {
            const double m = 0.5 * ((double)loose[k].b[i] + (double)loose[k].b[3 + i]);
            if (H.items[k].tree != 1u && std::isfinite(m)) c.push_back(m);
        }
        H.middle[i] = 0.0f;
        if (!c.empty()) {
            std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
            H.middle[i] = (float)c[c.size() / 2];
        }
    }
    const bool two_sets = radius > 0.0 && std::isfinite(radius);
    H.n_sets = two_sets ? 2u : 1u;
    H.radius = two_sets ? (float)(radius * 0.999) : INFINITY;
    // The walk drops a box that begins behind the closest hit so far, so a box must hold every point at which its
    // primitive can report a hit.  The sphere test (src/sphere.jl:45-65) reports points with
    // |p - c|^2 <= r^2 + 1.4e-6 |o - c|^2 (the rounding of its discriminant, see tight_box): sphere leaves, and the
    // leaves of nested trees (which may hold spheres), grow by sqrt(r^2 + 1.4e-6 D^2) - r, D the largest distance a ray
    // of the set can have come: sqrt(3) reach + |c| for set 1, the ball's radius + |c - centre| for set 0 (whose
    // tightened spheres already carry their own allowance inside the ball).
    for (size_t k = 0; k < n; ++k) {
        const uint32_t prim = (H.items[k].leaf & RT_LEAF_PRIM_MASK) - 1u;
        const uint4 h = H.prims[3 * (size_t)prim], g1 = H.prims[3 * (size_t)prim + 1], g2 = H.prims[3 * (size_t)prim + 2];
        const uint32_t kind = h.x & 15u;
        if (kind != DP_SPHERE && kind != DP_MOVING_SPHERE && kind != DP_INSTANCE) continue;
        double r = 0.0; // nested trees: the largest allowance
        if (kind == DP_SPHERE) r = std::fabs((double)bitsf(g1.z));
        if (kind == DP_MOVING_SPHERE) r = std::fabs((double)bitsf(g2.w));
        if (kind != DP_INSTANCE && (h.x >> 8) != 0u) { // inside a Transform: the radius as the world sees it, at least
            const float* M = &H.xforms[24 * (size_t)((h.x >> 8) - 1u)];
            double shrink = INFINITY;
            for (int i = 0; i < 3; ++i) shrink = std::min(shrink, std::sqrt((double)M[i] * M[i] + (double)M[3 + i] * M[3 + i] + (double)M[6 + i] * M[6 + i]));
            r *= std::min(1.0, shrink);
        }
        if (!std::isfinite(r)) r = 0.0;
        auto grow = [&](float* b, double D) {
            const double e = std::sqrt(r * r + 1.4e-6 * D * D) - r;
            for (int i = 0; i < 3; ++i) {
                if (!(b[i] <= b[3 + i])) return; // NaN: left alone
                b[i] = std::nextafter((float)((double)b[i] - e), -INFINITY);
                b[3 + i] = std::nextafter((float)((double)b[3 + i] + e), INFINITY);
            }
        };
        double far_c = 0.0; // farthest corner of the box from the origin
        for (int i = 0; i < 3; ++i) {
            const double a0 = std::max(std::fabs((double)loose[k].b[i]), std::fabs((double)loose[k].b[3 + i]));
            if (std::isfinite(a0)) far_c += a0 * a0;
        }
        const bool tightened = valid[k] != INFINITY;
        grow(loose[k].b, std::sqrt(3.0) * reach + std::sqrt(far_c) + (H.items[k].tree >= 2u ? H.scene_reach : 0.0)); // nested: box in the tree's own frame
        if (two_sets && !tightened) grow(tight[k].b, (double)H.radius + offset_from_centre(k));
    }
    if (!two_sets) tight = loose; // no ball in which every tight box is valid: one set of loose boxes
    if (std::getenv("RT_AMD_BUILD_LOG")) // (measurements: what the leaves' boxes cost on the host, before any tree is built)
        std::fprintf(stderr, "rt_build_nodes: the boxes of %zu leaves took %.3f ms on the host\n", n,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    const uint32_t n_trees = (uint32_t)H.tree_first.size();
    const uint32_t per_layout = H.nodes_per_layout;
    H.nodes.clear();
    H.pairs.clear();
    H.tree_depth = 0;
    {   // the tree of surfaces is also kept as pair nodes unless nested trees are walked (those enter and leave trees mid-walk)
        size_t n_surfaces = 0;
        for (size_t k = 0; k < n; ++k) n_surfaces += H.items[k].tree == 0u ? 1u : 0u;
        H.pairs_per_layout = (H.nested || n_surfaces == 0) ? 0u : (uint32_t)std::max<size_t>(n_surfaces - 1u, 1u);
    }
    H.set_items[0].clear();
    H.set_items[1].clear();
    H.device_trees.clear();
    if (!H.host_trees) {
        for (uint32_t t = 0; t < n_trees; ++t) {
            if (t == 1u) continue; // the constant media are a list, not a tree
            AccelDeviceTree dt{(uint32_t)H.set_items[0].size(), 0u, H.tree_first[t], t == 0u && H.pairs_per_layout != 0u};
            for (size_t k = 0; k < n; ++k)
                if (H.items[k].tree == t) {
                    H.set_items[0].push_back(tight[k]);
                    if (two_sets) H.set_items[1].push_back(loose[k]);
                    dt.n_items++;
                }
            if (dt.n_items) H.device_trees.push_back(dt);
        }
        return;
    }
    H.nodes.reserve((size_t)H.n_sets * 8u * per_layout);
    H.pairs.reserve((size_t)H.n_sets * 16u * H.pairs_per_layout);
    for (uint32_t set = 0; set < H.n_sets; ++set) {
        const std::vector<AccelItem>& boxes = set == 0 ? tight : loose;
        std::vector<uint4> layout[8];
        for (uint32_t t = 0; t < n_trees; ++t) {
            if (t == 1u) continue; // the constant media are a list, not a tree
            std::vector<AccelItem> items;
            for (size_t k = 0; k < n; ++k) if (H.items[k].tree == t) items.push_back(boxes[k]);
            if (items.empty()) continue;
            AccelTree tree;
            accel_build(items, tree);
            for (int o = 0; o < 8; ++o) accel_emit(tree, o, H.node_scale, layout[o]);
            if (t == 0u && H.pairs_per_layout != 0u) {
                for (int o = 0; o < 8; ++o) accel_emit_pairs(tree, o, H.node_scale, (set * 8u + (uint32_t)o) * H.pairs_per_layout, H.pairs);
                H.tree_depth = std::max(H.tree_depth, accel_depth(tree));
            }
        }
        for (uint32_t o = 0; o < 8; ++o) {
            // skip links (interior nodes) become absolute; leaf words stay
            const uint32_t shift = (set * 8u + o) * per_layout;
            for (uint4& nd : layout[o])
                if (!(nd.w & RT_NODE_LEAF)) nd.w += shift;
            H.nodes.insert(H.nodes.end(), layout[o].begin(), layout[o].end());
        }
    }
}

int rt_flatten_scene(const rtSceneDesc* desc, HostScene* out) {
    if (!desc || !out) return rt_fail(RT_EINVAL, "rt_flatten_scene: null argument");
    if ((desc->n_objects && !desc->objects) || (desc->n_top && !desc->top) || (desc->n_buckets && !desc->buckets) ||
        (desc->n_materials && !desc->materials) || (desc->n_textures && !desc->textures) ||
        (desc->n_transforms && !desc->transforms) || (desc->n_images && !desc->images) ||
        (desc->n_perlins && !desc->perlins) || (desc->n_meshes && !desc->meshes) || (desc->n_groups && !desc->groups))
        return rt_fail(RT_EINVAL, "scene description has a null table with a non-zero count");
    Builder b(*desc, *out);
    if (!b.run()) return rt_fail(b.status, "%s", b.err.c_str());
    return RT_OK;
}
