// rt_obj.cpp -- Wavefront OBJ -> the vertex / index arrays of a Mesh: the host-side step of render(filename, material)
// (main.jl:53-81), which the reference delegates to MeshIO / GeometryBasics.  What the reference keeps of the file is
// one Vertex per distinct (position, texcoord, normal) combination and one index column per triangle; this loader
// defines: vertices = distinct v/vt/vn triples in order of first appearance, polygons fan-triangulated (1,2,3),(1,3,4),...
// Missing normals become zero vectors and missing texture coordinates (-1, -1) (main.jl:62-67).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace {

// whitespace-separated tokens of a line (like str.split() without arguments)
void split(const std::string& line, std::vector<std::string>& out) {
    out.clear();
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && (unsigned char)line[i] <= ' ') ++i;
        size_t j = i;
        while (j < n && (unsigned char)line[j] > ' ') ++j;
        if (j > i) out.emplace_back(line, i, j - i);
        i = j;
    }
}

bool parse_float(const std::string& s, float& v) { // decimal -> double -> float, the way float() then float32 rounds
    char* end = nullptr;
    double d = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') return false;
    v = (float)d;
    return true;
}

bool parse_int(const std::string& s, long& v) {
    if (s.empty()) { v = 0; return true; }
    char* end = nullptr;
    v = std::strtol(s.c_str(), &end, 10);
    return end != s.c_str() && *end == '\0';
}

} // namespace

int rt_obj_load(const char* path, rtObjMesh* out) {
    if (!path || !out) return rt_fail(RT_EINVAL, "rt_obj_load: null argument");
    std::memset(out, 0, sizeof(*out));
    FILE* f = std::fopen(path, "rb");
    if (!f) return rt_fail(RT_EINVAL, "rt_obj_load: cannot open %s", path);
    std::vector<float> pos, tex, nrm;
    std::vector<std::tuple<long, long, long>> verts;
    std::map<std::tuple<long, long, long>, int32_t> index;
    std::vector<int32_t> tris;
    std::vector<std::string> parts;
    std::string line;
    char buf[4096];
    long line_no = 0;
    int status = RT_OK;
    auto bad = [&](const char* what) { status = rt_fail(RT_EINVAL, "rt_obj_load: %s:%ld: %s", path, line_no, what); };
    bool more = true;
    while (more && status == RT_OK) {
        line.clear();
        for (;;) { // one line of any length
            if (!std::fgets(buf, sizeof(buf), f)) { more = false; break; }
            line += buf;
            if (!line.empty() && line.back() == '\n') break;
        }
        if (line.empty()) break;
        ++line_no;
        split(line, parts);
        if (parts.empty()) continue;
        const std::string& tag = parts[0];
        if (tag == "v" || tag == "vn" || tag == "vt") {
            const size_t want = tag == "vt" ? 2 : 3;
            if (parts.size() < 1 + want) { bad("too few coordinates"); break; }
            std::vector<float>& dst = tag == "v" ? pos : (tag == "vn" ? nrm : tex);
            for (size_t k = 0; k < want; ++k) {
                float v;
                if (!parse_float(parts[1 + k], v)) { bad("malformed number"); break; }
                dst.push_back(v);
            }
        } else if (tag == "f") {
            std::vector<int32_t> ids;
            for (size_t t = 1; t < parts.size() && status == RT_OK; ++t) {
                std::string fields[3];
                size_t start = 0;
                for (int k = 0; k < 3; ++k) {
                    size_t slash = parts[t].find('/', start);
                    fields[k] = parts[t].substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                long key[3];
                const long counts[3] = {(long)(pos.size() / 3), (long)(tex.size() / 2), (long)(nrm.size() / 3)};
                for (int k = 0; k < 3; ++k) {
                    if (!parse_int(fields[k], key[k])) { bad("malformed face index"); break; }
                    if (key[k] < 0) key[k] += counts[k] + 1; // relative to the elements read so far
                    if (key[k] < 0 || key[k] > counts[k]) { bad("face index out of range"); break; }
                }
                if (status != RT_OK) break;
                if (key[0] == 0) { bad("face without a position index"); break; }
                auto tup = std::make_tuple(key[0], key[1], key[2]);
                auto it = index.find(tup);
                if (it == index.end()) {
                    it = index.emplace(tup, (int32_t)verts.size()).first;
                    verts.push_back(tup);
                }
                ids.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < ids.size(); ++k) {
                tris.push_back(ids[0]); tris.push_back(ids[k]); tris.push_back(ids[k + 1]); tris.push_back(0);
            }
        }
    }
    std::fclose(f);
    if (status != RT_OK) return status;
    const size_t nv = verts.size(), nt = tris.size() / 4;
    float* vertices = (float*)std::malloc(std::max<size_t>(1, nv * 8) * sizeof(float));
    int32_t* indices = (int32_t*)std::malloc(std::max<size_t>(1, nt * 4) * sizeof(int32_t));
    if (!vertices || !indices) {
        std::free(vertices); std::free(indices);
        return rt_fail(RT_ENOMEM, "rt_obj_load: out of host memory");
    }
    for (size_t i = 0; i < nv; ++i) {
        long vi, ti, ni;
        std::tie(vi, ti, ni) = verts[i];
        float* v = vertices + 8 * i;
        for (int k = 0; k < 3; ++k) v[k] = pos[3 * (size_t)(vi - 1) + k];
        for (int k = 0; k < 3; ++k) v[3 + k] = ni ? nrm[3 * (size_t)(ni - 1) + k] : 0.0f;
        for (int k = 0; k < 2; ++k) v[6 + k] = ti ? tex[2 * (size_t)(ti - 1) + k] : -1.0f;
    }
    if (nt) std::memcpy(indices, tris.data(), nt * 4 * sizeof(int32_t));
    out->n_vertices = (int32_t)nv;
    out->n_triangles = (int32_t)nt;
    out->vertices = vertices;
    out->indices = indices;
    return RT_OK;
}

void rt_obj_free(rtObjMesh* mesh) {
    if (!mesh) return;
    std::free(mesh->vertices);
    std::free(mesh->indices);
    std::memset(mesh, 0, sizeof(*mesh));
}
