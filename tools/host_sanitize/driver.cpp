// Host-side robustness run of the library's CPU entry points under AddressSanitizer / UndefinedBehaviorSanitizer (no GPU
// call is made): scene files and OBJ files, whole and damaged (cut short, bytes flipped, words replaced by extreme values),
// are loaded; whatever loads is laid out by the host builder (rt_scene_layout, rt_scene_layout_pairs), which validates a
// description the way rt_scene_create does; the tree builders, the Perlin tables and the image writers get random input.
// Every call must return (RT_OK or an error code): a crash or a sanitizer report fails the run.
//   usage: driver <dir with *.rtscene / *.rtscene64 / *.obj> <mutations per file> [seed]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <dirent.h>
#include <random>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

static std::vector<uint8_t> slurp(const std::string& path) {
    std::vector<uint8_t> b;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return b;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) b.insert(b.end(), chunk, chunk + got);
    std::fclose(f);
    return b;
}
static void spill(const std::string& path, const std::vector<uint8_t>& b) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
    std::fclose(f);
}
static bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

static std::mt19937_64 rng;
static std::vector<uint8_t> damaged(const std::vector<uint8_t>& in) {
    std::vector<uint8_t> b = in;
    if (b.empty()) return b;
    switch (rng() % 5) {
    case 0: b.resize(rng() % b.size()); break;                                       // cut short
    case 1: for (int k = 0, n = 1 + (int)(rng() % 8); k < n; ++k) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8)); break;
    case 2: {                                                                          // a word replaced by an extreme value
        static const uint32_t extreme[] = {0u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 0x7F800000u, 0x7FC00000u, 0xFF800000u, 1u << 30};
        if (b.size() >= 4) { const size_t at = (rng() % (b.size() / 4)) * 4; std::memcpy(&b[at], &extreme[rng() % 8], 4); }
        break;
    }
    case 3: { const size_t at = rng() % b.size(), n = 1 + rng() % 64; b.insert(b.begin() + (long)at, n, (uint8_t)rng()); break; } // bytes inserted
    default: {                                                                         // a word early in the file (the counts live there)
        const size_t span = b.size() < 4096 ? b.size() : 4096;
        if (span >= 4) { const size_t at = (rng() % (span / 4)) * 4; const uint32_t v = (uint32_t)rng() >> (rng() % 32); std::memcpy(&b[at], &v, 4); }
    }
    }
    return b;
}

static long loaded = 0, refused = 0, laid_out = 0, layout_refused = 0;

static void lay_out(const rtSceneDesc* d) {
    rtLayout info;
    // counts first (no room), then the nodes themselves if the description stood
    int st = rt_scene_layout(d, &info, nullptr, 0, nullptr, 0);
    if (st != RT_OK) { ++layout_refused; return; }
    const size_t n_nodes = (size_t)info.nodes_per_layout * 8u * info.n_sets;
    if (n_nodes > (64u << 20)) { ++layout_refused; return; }
    std::vector<uint32_t> nodes(4 * n_nodes + 4);
    std::vector<int32_t> prim_top(info.n_prims + 1);
    st = rt_scene_layout(d, &info, nodes.data(), n_nodes, prim_top.data(), prim_top.size());
    uint32_t per = 0, depth = 0;
    if (rt_scene_layout_pairs(d, &per, &depth, nullptr, 0) == RT_OK && per != 0 && per < (8u << 20)) {
        std::vector<uint32_t> pairs((size_t)per * 8u * 16u + 8);
        rt_scene_layout_pairs(d, &per, &depth, pairs.data(), (size_t)per * 16u);
    }
    if (st == RT_OK) ++laid_out; else ++layout_refused;
}

static void scene_file(const std::string& path, bool f64, int mutations, const std::string& tmp) {
    const std::vector<uint8_t> whole = slurp(path);
    for (int k = -1; k < mutations; ++k) {
        spill(tmp, k < 0 ? whole : damaged(whole));
        if (f64) {
            rtSceneDesc64* d = nullptr;
            if (rt_scene_desc_load_f64(tmp.c_str(), &d) == RT_OK) {
                ++loaded;
                const std::string again = tmp + ".again";
                rt_scene_desc_save_f64(d, again.c_str());
                rt_scene_desc_free_f64(d);
                std::remove(again.c_str());
            } else ++refused;
            rtSceneDesc* wrong = nullptr; // the other loader must refuse the file
            if (rt_scene_desc_load(tmp.c_str(), &wrong) == RT_OK) rt_scene_desc_free(wrong);
        } else {
            rtSceneDesc* d = nullptr;
            if (rt_scene_desc_load(tmp.c_str(), &d) == RT_OK) {
                ++loaded;
                lay_out(d);
                rt_scene_desc_free(d);
            } else ++refused;
        }
        if (k < 0 && loaded == 0) { std::fprintf(stderr, "%s: the undamaged file did not load: %s\n", path.c_str(), rt_last_error()); std::exit(1); }
    }
}

static void obj_file(const std::string& path, int mutations, const std::string& tmp) {
    const std::vector<uint8_t> whole = slurp(path);
    for (int k = -1; k < mutations; ++k) {
        spill(tmp, k < 0 ? whole : damaged(whole));
        rtObjMesh m;
        std::memset(&m, 0, sizeof(m));
        if (rt_obj_load(tmp.c_str(), &m) == RT_OK) { ++loaded; rt_obj_free(&m); } else ++refused;
    }
}

static void builders() {
    std::uniform_real_distribution<float> u(-100.0f, 100.0f);
    for (int round = 0; round < 40; ++round) {
        const int n = round < 4 ? round : 1 + (int)(rng() % 600);
        std::vector<float> bounds(6 * (size_t)n + 6);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                float lo = u(rng), hi = lo + std::fabs(u(rng)) * 0.1f;
                if (round % 7 == 5) hi = lo;                        // flat boxes
                if (round % 11 == 7 && i % 5 == 0) lo = hi = 0.0f;  // many equal boxes
                bounds[6 * i + a] = lo; bounds[6 * i + 3 + a] = hi;
            }
        const int cap = 2 * n + 2;
        std::vector<int32_t> order(n + 1), left(cap), right(cap);
        std::vector<uint8_t> leaf(cap);
        std::vector<float> nb(6 * (size_t)cap);
        rt_bvh_build(n, bounds.data(), order.data(), nb.data(), left.data(), right.data(), leaf.data());
        // the same boxes as triangles of one mesh, then of two (three vertices a box: min, max and a mix)
        if (n > 0 && n <= 200) {
            for (int n_meshes = 1; n_meshes <= 2; ++n_meshes) {
                std::vector<float> verts(8 * 3 * (size_t)n);
                std::vector<int32_t> idx(4 * (size_t)n), ids(n);
                for (int i = 0; i < n; ++i) {
                    const float* b = &bounds[6 * i];
                    const float v[3][3] = {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}, {b[0], b[4], b[2]}};
                    for (int c = 0; c < 3; ++c) { float* o = &verts[8 * (3 * i + c)]; o[0] = v[c][0]; o[1] = v[c][1]; o[2] = v[c][2]; o[3] = 0; o[4] = 1; o[5] = 0; o[6] = 0; o[7] = 0; }
                    idx[4 * i] = 3 * i; idx[4 * i + 1] = 3 * i + 1; idx[4 * i + 2] = 3 * i + 2; idx[4 * i + 3] = 0;
                }
                const int half = n / 2;
                const int32_t nv[2] = {3 * n, 3 * n}, nt1[1] = {n}, nt2[2] = {half, n - half};
                const float* vp[2] = {verts.data(), verts.data()};
                int32_t* ip[2] = {idx.data(), idx.data() + 4 * (size_t)half};
                int32_t* tp[2] = {ids.data(), ids.data() + half};
                if (n_meshes == 2 && (half == 0)) continue;
                const size_t slots = 4 * (size_t)n + 4;
                std::vector<int32_t> ml(slots), mr(slots), mo(2);
                std::vector<uint8_t> mf(slots);
                std::vector<float> mb(6 * slots);
                rt_bvh_build_meshes(n_meshes, nv, vp, n_meshes == 1 ? nt1 : nt2, ip, tp, mo.data(), mb.data(), ml.data(), mr.data(), mf.data());
            }
        }
    }
    for (int k = 0; k < 8; ++k) {
        rtPerlin p;
        uint64_t counter = rng() % 1000;
        rt_perlin_generate((uint32_t)rng(), &counter, k & 1, &p);
    }
    for (int k = 0; k < 6; ++k) {
        const int w = 1 + (int)(rng() % 97), h = 1 + (int)(rng() % 61);
        std::vector<float> img(3 * (size_t)w * h);
        for (float& x : img) x = u(rng) * 0.02f;
        if (k == 3) { img[0] = __builtin_nanf(""); img[1] = __builtin_inff(); img[2] = -__builtin_inff(); }
        std::vector<uint8_t> rgb(3 * (size_t)w * h);
        rt_image_to_rgb8(img.data(), w, h, rgb.data());
        rt_write_png("/tmp/rt_sanitize_out.png", rgb.data(), w, h);
        rt_write_ppm("/tmp/rt_sanitize_out.ppm", rgb.data(), w, h);
    }
    rt_write_png("/nonexistent-dir/x.png", nullptr, 4, 4);
    rt_write_png("/tmp/rt_sanitize_out.png", nullptr, 0, 0);
    std::remove("/tmp/rt_sanitize_out.png");
    std::remove("/tmp/rt_sanitize_out.ppm");
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <dir> <mutations per file> [seed]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const int mutations = std::atoi(argv[2]);
    rng.seed(argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1);
    const std::string tmp = "/tmp/rt_sanitize_case.bin";
    DIR* dp = opendir(dir.c_str());
    if (!dp) { std::perror(dir.c_str()); return 2; }
    std::vector<std::string> names;
    while (dirent* e = readdir(dp)) names.push_back(e->d_name);
    closedir(dp);
    for (const std::string& name : names) {
        const std::string path = dir + "/" + name;
        if (ends_with(name, ".rtscene")) scene_file(path, false, mutations, tmp);
        else if (ends_with(name, ".rtscene64")) scene_file(path, true, mutations, tmp);
        else if (ends_with(name, ".obj")) obj_file(path, mutations, tmp);
        else continue;
        std::printf("%-28s files loaded %ld refused %ld | laid out %ld refused %ld\n", name.c_str(), loaded, refused, laid_out, layout_refused);
        std::fflush(stdout);
    }
    builders();
    std::remove(tmp.c_str());
    std::printf("done: no crash, no sanitizer report\n");
    return 0;
}
