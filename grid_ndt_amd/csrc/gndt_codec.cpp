// gndt_codec.cpp — host-side key codec of the C ABI (no HIP).
// The reference builds its map keys through binary-digit strings (include/Stopwatch.h:39-47,
// 102-110, 116-147, 171-189).  Here the same VALUES are produced with integer bit operations:
//   countMorton(a,b) = decimal print of int32( low 32 bits of interleave(|a|,|b|) ), a on the odd bits
//   mortonToXY(m)    = (odd bits of |m|, even bits of |m|), each parsed into 32 bits
// including the wrap for indices >= 65536 and the |.| that the digit loop applies to negative input
// (C++ '%' and '/' truncate toward zero, so the digits of -n are the digits of n).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gndt.h"
#include "gndt_math.hpp"

namespace {

inline uint64_t magnitude(int32_t v) { return v < 0 ? (uint64_t)(-(int64_t)v) : (uint64_t)v; }

// spread the low 32 bits of x over the even bit positions of a 64-bit word
inline uint64_t spread_bits(uint64_t x) {
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
inline uint32_t gather_even_bits(uint64_t x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}

int32_t morton_value(int32_t a, int32_t b) {
    uint64_t inter = (spread_bits(magnitude(a)) << 1) | spread_bits(magnitude(b));
    return (int32_t)(uint32_t)(inter & 0xFFFFFFFFull);  // the unsigned accumulator keeps 32 bits
}

}  // namespace

extern "C" {

int gndt_count_morton(int32_t a, int32_t b, char out[16]) {
    if (!out) return GNDT_ERR_INVALID;
    snprintf(out, 16, "%d", morton_value(a, b));
    return GNDT_OK;
}

int gndt_morton_to_xy(int32_t morton, int32_t* a, int32_t* b) {
    if (!a || !b) return GNDT_ERR_INVALID;
    uint64_t m = magnitude(morton);
    *a = (int32_t)gather_even_bits(m >> 1);
    *b = (int32_t)gather_even_bits(m);
    return GNDT_OK;
}

uint64_t gndt_pack_key(int32_t sx, int32_t sy, int32_t sz) { return gndt::pack_key(sx, sy, sz); }

void gndt_unpack_key(uint64_t key, int32_t* sx, int32_t* sy, int32_t* sz) {
    int x, y, z;
    gndt::unpack_key(key, x, y, z);
    if (sx) *sx = x;
    if (sy) *sy = y;
    if (sz) *sz = z;
}

int gndt_trans_morton_xyz(const float origin[3], float grid_len, float z_len, const float p[3],
                          char* quadrant, int32_t* nx, int32_t* ny, int32_t* sz, char key_out[16]) {
    if (!origin || !p) return GNDT_ERR_INVALID;
    gndt::PointKey k = gndt::point_key(p[0], p[1], p[2], origin[0], origin[1], origin[2], grid_len, z_len);
    // quadrant letters, include/map2D.h:952-962: A (+,+)  B (+,-)  C (-,+)  D (-,-)
    char q = (k.sx > 0) ? ((k.sy > 0) ? 'A' : 'B') : ((k.sy > 0) ? 'C' : 'D');
    int ax = k.sx > 0 ? k.sx : -k.sx, ay = k.sy > 0 ? k.sy : -k.sy;
    if (quadrant) *quadrant = q;
    if (nx) *nx = ax;
    if (ny) *ny = ay;
    if (sz) *sz = k.sz;
    if (key_out) snprintf(key_out, 16, "%c%d", q, morton_value(ax, ay));
    return k.ok ? GNDT_OK : GNDT_ERR_KEY_RANGE;
}

}  // extern "C"
