// rt_scene_file.cpp -- a scene description (rtSceneDesc: the object / transform / material / texture tables and the
// trees build_bvh left, include/rt_amd.h) written to and read from one binary file, so that a host other than the
// reference's Julia code -- or a later run -- can hand a scene to rt_scene_create without rebuilding it.  The reference
// has no scene files: its scenes are Julia functions (samples/*.jl, scenes.jl:8-11); this is the data they produce.
//
// File: "RTSCENE1", then the ten tables in the order of the struct, each as a 64-bit element count followed by the
// elements; variable-length members (image texels, mesh arrays, tree arrays) follow their record as counted arrays.
// Little-endian, 4-byte floats, as in memory.  A Float64 description (rtSceneDesc64) is written the same way under the
// tag "RTSCEN64" with 8-byte floats; each loader refuses the other's files.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace {

// the record types of one element type
struct Float32Records {
    typedef rtSceneDesc Desc; typedef rtObject Object; typedef rtTransform Transform; typedef rtMaterial Material;
    typedef rtTexture Texture; typedef rtPerlin Perlin; typedef rtMesh Mesh; typedef rtBvh Bvh; typedef float real;
    static const char* magic() { return "RTSCENE1"; }
    static const char* suffix() { return ""; }
};
struct Float64Records {
    typedef rtSceneDesc64 Desc; typedef rtObject64 Object; typedef rtTransform64 Transform; typedef rtMaterial64 Material;
    typedef rtTexture64 Texture; typedef rtPerlin64 Perlin; typedef rtMesh64 Mesh; typedef rtBvh64 Bvh; typedef double real;
    static const char* magic() { return "RTSCEN64"; }
    static const char* suffix() { return "_f64"; }
};
const size_t MAGIC_BYTES = 8;

struct Writer {
    FILE* f;
    bool ok = true;
    void raw(const void* p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    void count(uint64_t n) { raw(&n, sizeof(n)); }
    template <typename T> void array(const T* p, size_t n) {
        count(p ? n : 0);
        if (p) raw(p, n * sizeof(T));
    }
    template <typename Bvh> void tree(const Bvh& t) {
        int32_t head[2] = {t.kind, t.n_nodes};
        raw(head, sizeof(head));
        array(t.objects, (size_t)(t.n_objects > 0 ? t.n_objects : 0));
        const size_t n = (size_t)(t.n_nodes > 0 ? t.n_nodes : 0);
        array(t.node_bounds, 6 * n);
        array(t.node_left, n);
        array(t.node_right, n);
        array(t.node_leaf, n);
    }
};

// everything the pointers of a loaded description point into
template <class R> struct Loaded {
    std::vector<typename R::Object> objects;
    std::vector<int32_t> top;
    std::vector<typename R::Transform> transforms;
    std::vector<typename R::Material> materials;
    std::vector<typename R::Texture> textures;
    std::vector<rtImage> images;
    std::vector<typename R::Perlin> perlins;
    std::vector<typename R::Mesh> meshes;
    std::vector<typename R::Bvh> groups, buckets;
    std::vector<std::vector<uint8_t>> bytes;            // image texels, leaf flags
    std::vector<std::vector<int32_t>> ints;             // mesh indices / materials / ids, tree arrays
    std::vector<std::vector<typename R::real>> reals;   // mesh vertices, tree bounds
};

// what the loader hands out: the description first (plain data, so a pointer to it is a pointer to the handle)
template <class R> struct Handle {
    typename R::Desc desc;
    Loaded<R>* tables;
};

struct Reader {
    const uint8_t* p;
    size_t left;
    bool ok = true;
    bool raw(void* out, size_t n) {
        if (!ok || n > left) { ok = false; return false; }
        std::memcpy(out, p, n);
        p += n; left -= n;
        return true;
    }
    uint64_t count(size_t elem) {
        uint64_t n = 0;
        if (!raw(&n, sizeof(n))) return 0;
        if (elem && n > left / elem) { ok = false; return 0; } // more elements than the file has bytes left
        return n;
    }
    template <typename T> bool table(std::vector<T>& v) {
        const uint64_t n = count(sizeof(T));
        if (!ok || n > 0x7FFFFFFFull) { ok = false; return false; }
        v.resize((size_t)n);
        return n == 0 || raw(v.data(), (size_t)n * sizeof(T));
    }
    template <typename T> const T* array(std::vector<std::vector<T>>& store, uint64_t* n_out = nullptr) {
        std::vector<T> v;
        if (!table(v)) return nullptr;
        if (n_out) *n_out = v.size();
        if (v.empty()) return nullptr;
        store.push_back(std::move(v));
        return store.back().data();
    }
    template <class R> bool tree(Loaded<R>& L, typename R::Bvh& t) {
        int32_t head[2];
        if (!raw(head, sizeof(head))) return false;
        std::memset(&t, 0, sizeof(t));
        t.kind = head[0];
        t.n_nodes = head[1];
        uint64_t n_obj = 0, nb = 0, nl = 0, nr = 0, nf = 0;
        t.objects = array(L.ints, &n_obj);
        t.node_bounds = array(L.reals, &nb);
        t.node_left = array(L.ints, &nl);
        t.node_right = array(L.ints, &nr);
        t.node_leaf = array(L.bytes, &nf);
        t.n_objects = (int32_t)n_obj;
        const uint64_t n = t.n_nodes > 0 ? (uint64_t)t.n_nodes : 0;
        if (ok && (t.n_nodes < 0 || nb != 6 * n || nl != n || nr != n || nf != n)) ok = false; // arrays must match the node count
        return ok;
    }
};

template <class R> int desc_save(const typename R::Desc* d, const char* path) {
    if (!d || !path) return rt_fail(RT_EINVAL, "rt_scene_desc_save%s: null argument", R::suffix());
    if (d->n_objects < 0 || d->n_top < 0 || d->n_transforms < 0 || d->n_materials < 0 || d->n_textures < 0 || d->n_images < 0 ||
        d->n_perlins < 0 || d->n_meshes < 0 || d->n_groups < 0 || d->n_buckets < 0)
        return rt_fail(RT_EINVAL, "rt_scene_desc_save%s: negative count", R::suffix());
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_fail(RT_EINVAL, "rt_scene_desc_save%s: cannot open %s for writing", R::suffix(), path);
    Writer w{f};
    w.raw(R::magic(), MAGIC_BYTES);
    w.array(d->objects, (size_t)d->n_objects);
    w.array(d->top, (size_t)d->n_top);
    w.array(d->transforms, (size_t)d->n_transforms);
    w.array(d->materials, (size_t)d->n_materials);
    w.array(d->textures, (size_t)d->n_textures);
    w.count((uint64_t)d->n_images);
    for (int i = 0; i < d->n_images; ++i) {
        const rtImage& im = d->images[i];
        int32_t size[2] = {im.width, im.height};
        w.raw(size, sizeof(size));
        w.array(im.rgb, im.width > 0 && im.height > 0 ? (size_t)3 * (size_t)im.width * (size_t)im.height : 0);
    }
    w.array(d->perlins, (size_t)d->n_perlins);
    w.count((uint64_t)d->n_meshes);
    for (int i = 0; i < d->n_meshes; ++i) {
        const typename R::Mesh& m = d->meshes[i];
        w.array(m.vertices, (size_t)8 * (size_t)(m.n_vertices > 0 ? m.n_vertices : 0));
        w.array(m.indices, (size_t)4 * (size_t)(m.n_triangles > 0 ? m.n_triangles : 0));
        w.array(m.materials, (size_t)(m.n_materials > 0 ? m.n_materials : 0));
        w.array(m.tri_ids, (size_t)(m.n_triangles > 0 ? m.n_triangles : 0));
    }
    w.count((uint64_t)d->n_groups);
    for (int i = 0; i < d->n_groups; ++i) w.tree(d->groups[i]);
    w.count((uint64_t)d->n_buckets);
    for (int i = 0; i < d->n_buckets; ++i) w.tree(d->buckets[i]);
    const bool closed = std::fclose(f) == 0;
    if (!w.ok || !closed) return rt_fail(RT_EINVAL, "rt_scene_desc_save%s: write to %s failed", R::suffix(), path);
    return RT_OK;
}

template <class R> int desc_load(const char* path, typename R::Desc** out) {
    if (!path || !out) return rt_fail(RT_EINVAL, "rt_scene_desc_load%s: null argument", R::suffix());
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return rt_fail(RT_EINVAL, "rt_scene_desc_load%s: cannot open %s", R::suffix(), path);
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    std::fclose(f);
    if (buf.size() < MAGIC_BYTES || std::memcmp(buf.data(), R::magic(), MAGIC_BYTES) != 0)
        return rt_fail(RT_EINVAL, "rt_scene_desc_load%s: %s is not a scene file of this element type", R::suffix(), path);
    Loaded<R>* L = new (std::nothrow) Loaded<R>();
    if (!L) return rt_fail(RT_ENOMEM, "out of host memory");
    Reader r{buf.data() + MAGIC_BYTES, buf.size() - MAGIC_BYTES};
    r.table(L->objects);
    r.table(L->top);
    r.table(L->transforms);
    r.table(L->materials);
    r.table(L->textures);
    uint64_t n = r.count(8);
    for (uint64_t i = 0; r.ok && i < n; ++i) {
        rtImage im;
        int32_t size[2];
        if (!r.raw(size, sizeof(size))) break;
        uint64_t texels = 0;
        im.width = size[0]; im.height = size[1];
        im.rgb = r.array(L->bytes, &texels);
        if (r.ok && (im.width < 0 || im.height < 0 || texels != (uint64_t)3 * (uint64_t)im.width * (uint64_t)im.height)) r.ok = false;
        L->images.push_back(im);
    }
    r.table(L->perlins);
    n = r.count(32);
    for (uint64_t i = 0; r.ok && i < n; ++i) {
        typename R::Mesh m;
        std::memset(&m, 0, sizeof(m));
        uint64_t nv = 0, ni = 0, nm = 0, nt = 0;
        m.vertices = r.array(L->reals, &nv);
        m.indices = r.array(L->ints, &ni);
        m.materials = r.array(L->ints, &nm);
        m.tri_ids = r.array(L->ints, &nt);
        if (r.ok && (nv % 8 || ni % 4 || (nt != 0 && nt != ni / 4))) r.ok = false;
        m.n_vertices = (int32_t)(nv / 8); m.n_triangles = (int32_t)(ni / 4); m.n_materials = (int32_t)nm;
        L->meshes.push_back(m);
    }
    n = r.count(8);
    for (uint64_t i = 0; r.ok && i < n; ++i) { typename R::Bvh t; if (r.tree<R>(*L, t)) L->groups.push_back(t); }
    n = r.count(8);
    for (uint64_t i = 0; r.ok && i < n; ++i) { typename R::Bvh t; if (r.tree<R>(*L, t)) L->buckets.push_back(t); }
    if (!r.ok || r.left != 0) {
        delete L;
        return rt_fail(RT_EINVAL, "rt_scene_desc_load%s: %s is truncated or damaged", R::suffix(), path);
    }
    Handle<R>* h = new (std::nothrow) Handle<R>();
    if (!h) { delete L; return rt_fail(RT_ENOMEM, "out of host memory"); }
    h->tables = L;
    typename R::Desc& d = h->desc;
    std::memset(&d, 0, sizeof(d));
    d.n_objects = (int32_t)L->objects.size();       d.objects = L->objects.data();
    d.n_top = (int32_t)L->top.size();               d.top = L->top.data();
    d.n_transforms = (int32_t)L->transforms.size(); d.transforms = L->transforms.data();
    d.n_materials = (int32_t)L->materials.size();   d.materials = L->materials.data();
    d.n_textures = (int32_t)L->textures.size();     d.textures = L->textures.data();
    d.n_images = (int32_t)L->images.size();         d.images = L->images.data();
    d.n_perlins = (int32_t)L->perlins.size();       d.perlins = L->perlins.data();
    d.n_meshes = (int32_t)L->meshes.size();         d.meshes = L->meshes.data();
    d.n_groups = (int32_t)L->groups.size();         d.groups = L->groups.data();
    d.n_buckets = (int32_t)L->buckets.size();       d.buckets = L->buckets.data();
    *out = &h->desc;
    return RT_OK;
}

template <class R> void desc_free(typename R::Desc* d) {
    if (!d) return;
    Handle<R>* h = reinterpret_cast<Handle<R>*>(d); // the description is the first member of the handle
    delete h->tables;
    delete h;
}

} // namespace

extern "C" {

int rt_scene_desc_save(const rtSceneDesc* d, const char* path) { return desc_save<Float32Records>(d, path); }
int rt_scene_desc_load(const char* path, rtSceneDesc** out) { return desc_load<Float32Records>(path, out); }
void rt_scene_desc_free(rtSceneDesc* d) { desc_free<Float32Records>(d); }

int rt_scene_desc_save_f64(const rtSceneDesc64* d, const char* path) { return desc_save<Float64Records>(d, path); }
int rt_scene_desc_load_f64(const char* path, rtSceneDesc64** out) { return desc_load<Float64Records>(path, out); }
void rt_scene_desc_free_f64(rtSceneDesc64* d) { desc_free<Float64Records>(d); }

} // extern "C"
