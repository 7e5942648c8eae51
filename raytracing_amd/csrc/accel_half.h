// accel_half.h -- float -> IEEE half rounded strictly outward, for the boxes of the device trees (host and device code)
#ifndef ACCEL_HALF_H
#define ACCEL_HALF_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// float -> IEEE half strictly below (up = false) or strictly above (up = true) x; +-infinity stay; NaN gives the
// infinity of the rounding direction, so a box with a NaN bound stays unconstrained on that side.  Strictly: a ray that
// runs inside a face of the reference's box (direction component 0, origin exactly on the plane: 0 * Inf = NaN in the
// slab test, which the reference's comparisons then ignore) is strictly inside the stored box and never meets that NaN.
__host__ __device__ inline uint16_t accel_half_bits(float x, bool up) {
    if (!(x == x)) return up ? 0x7C00u : 0xFC00u;
    const _Float16 hf = (_Float16)x; // round to nearest even
    uint16_t h = __builtin_bit_cast(uint16_t, hf);
    const float back = (float)hf;
    if (back == x && (h & 0x7FFFu) == 0x7C00u) return h; // infinite
    if (up && back <= x) {        // one step toward +infinity
        if (h == 0x8000u) h = 0x0001u;
        else if (h & 0x8000u) h -= 1u;
        else h += 1u;
    } else if (!up && back >= x) { // one step toward -infinity
        if (h == 0x0000u) h = 0x8001u;
        else if (h & 0x8000u) h += 1u;
        else h -= 1u;
    }
    return h;
}

// The 16-byte node: one word per axis {near | far << 16}, near = the plane of that axis a ray of `octant` meets first
// (min for a positive direction component, max for a negative one), of the bounds multiplied by `scale`, and `word`.
__host__ __device__ inline uint4 accel_pack_node(const float* b, int octant, float scale, uint32_t word) {
    uint32_t w[3];
    for (int a = 0; a < 3; ++a) {
        const uint32_t lo = accel_half_bits(b[a] * scale, false), hi = accel_half_bits(b[3 + a] * scale, true);
        w[a] = ((octant >> a) & 1) ? (hi | (lo << 16)) : (lo | (hi << 16));
    }
    return make_uint4(w[0], w[1], w[2], word);
}

#endif
