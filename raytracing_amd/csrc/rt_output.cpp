// rt_output.cpp -- the step after the path: what the reference leaves to `save(path, clamp01.(img))` (README.md:50-67,
// Images.jl): clamp to [0, 1], quantise to 8 bits like N0f8 (round to nearest even of x * 255) and write a PNG / PPM.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace {

inline uint8_t quantise(float x) {
    if (!(x == x)) x = 0.0f; // clamp01nan
    x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
    return (uint8_t)std::nearbyintf(x * 255.0f);
}

inline uint8_t quantise(double x) { // N0f8(x::Float64): the product and the rounding in double
    if (!(x == x)) x = 0.0;
    x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
    return (uint8_t)std::nearbyint(x * 255.0);
}

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    put32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

} // namespace

int rt_image_to_rgb8(const float* img, int width, int height, uint8_t* rgb8) {
    if (!img || !rgb8) return rt_fail(RT_EINVAL, "rt_image_to_rgb8: null argument");
    if (width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_image_to_rgb8: bad size");
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c)
                rgb8[((size_t)y * width + x) * 3 + c] = quantise(img[((size_t)x * height + y) * 3 + c]);
    return RT_OK;
}

int rt_image_to_rgb8_f64(const double* img, int width, int height, uint8_t* rgb8) {
    if (!img || !rgb8) return rt_fail(RT_EINVAL, "rt_image_to_rgb8_f64: null argument");
    if (width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_image_to_rgb8_f64: bad size");
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c)
                rgb8[((size_t)y * width + x) * 3 + c] = quantise(img[((size_t)x * height + y) * 3 + c]);
    return RT_OK;
}

int rt_write_png(const char* path, const uint8_t* rgb8, int width, int height) {
    if (!path || !rgb8) return rt_fail(RT_EINVAL, "rt_write_png: null argument");
    if (width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_write_png: bad size");
    const size_t row = (size_t)width * 3;
    std::vector<uint8_t> raw((row + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0; // filter: none
        std::memcpy(&raw[(row + 1) * y + 1], rgb8 + row * y, row);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return rt_fail(RT_ENOMEM, "rt_write_png: deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0}; // 8 bits, RGB, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zlen);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_fail(RT_EINVAL, "rt_write_png: cannot open %s", path);
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    if (std::fclose(f) != 0 || w != out.size()) return rt_fail(RT_EINVAL, "rt_write_png: short write to %s", path);
    return RT_OK;
}

int rt_write_ppm(const char* path, const uint8_t* rgb8, int width, int height) {
    if (!path || !rgb8) return rt_fail(RT_EINVAL, "rt_write_ppm: null argument");
    if (width < 1 || height < 1) return rt_fail(RT_EINVAL, "rt_write_ppm: bad size");
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_fail(RT_EINVAL, "rt_write_ppm: cannot open %s", path);
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    size_t n = (size_t)width * height * 3, w = std::fwrite(rgb8, 1, n, f);
    if (std::fclose(f) != 0 || w != n) return rt_fail(RT_EINVAL, "rt_write_ppm: short write to %s", path);
    return RT_OK;
}
