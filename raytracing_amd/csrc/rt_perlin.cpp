// rt_perlin.cpp -- the tables of a Perlin noise texture (src/perlin.jl:17-33): 256 random unit vectors (or 256 random
// scalars for NoiseTexture{F}) and three shuffles of 0..255, made in native code so that a host without the Python
// mirror builds the same marble as the library's own sample scenes.  The reference draws them from Julia's unseeded
// global rand(); here every draw is pcg4d(counter, 0x5CE7E5, seed) of a running 64-bit counter, the stream the sample
// scenes of this library use for all their random choices, so (seed, counter) names a position in a scene's stream.
#include <cmath>
#include <cstdint>

#include "../../include/rt_amd.h"
#include "rt_internal.h"

namespace {

struct Stream {
    uint32_t seed;
    uint64_t ctr;
    uint32_t u32() {
        uint32_t v[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x5CE7E5u, seed};
        ++ctr;
        for (uint32_t& x : v) x = x * 1664525u + 1013904223u;
        for (int round = 0; round < 2; ++round) {
            v[0] += v[1] * v[3];
            v[1] += v[2] * v[0];
            v[2] += v[0] * v[1];
            v[3] += v[1] * v[2];
            if (round == 0) for (uint32_t& x : v) x ^= x >> 16;
        }
        return v[0];
    }
    float rand32() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); } // rand(Float32): 24 bits in [0, 1)
    void shuffle(int32_t* p, int n) {                                       // Fisher-Yates from the top (shuffle, src/perlin.jl:19-22)
        for (int i = 0; i < n; ++i) p[i] = i;
        for (int i = n - 1; i > 0; --i) {
            const int j = (int)(u32() % (uint32_t)(i + 1));
            const int32_t t = p[i]; p[i] = p[j]; p[j] = t;
        }
    }
};

} // namespace

extern "C" int rt_perlin_generate(uint32_t seed, uint64_t* counter, int32_t vector, rtPerlin* out) {
    if (!counter || !out) return rt_fail(RT_EINVAL, "rt_perlin_generate: null argument");
    Stream s{seed, *counter};
    for (int i = 0; i < 256; ++i) {
        float* r = out->ranvec + 3 * i;
        if (vector) { // normalize(rand(Vec3{F}, -1, 1)), src/perlin.jl:18 with src/vec3.jl:31,52,57,85
            const float x = s.rand32() * 2.0f + -1.0f, y = s.rand32() * 2.0f + -1.0f, z = s.rand32() * 2.0f + -1.0f;
            const float inv = 1.0f / std::sqrt((x * x + y * y) + z * z);
            r[0] = x * inv; r[1] = y * inv; r[2] = z * inv;
        } else {      // rand(F), src/perlin.jl:18 for a scalar table
            r[0] = s.rand32(); r[1] = 0.0f; r[2] = 0.0f;
        }
    }
    s.shuffle(out->perm_y, 256);
    s.shuffle(out->perm_z, 256);
    s.shuffle(out->perm_x, 256);
    *counter = s.ctr;
    return RT_OK;
}
