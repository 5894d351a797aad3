#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

namespace w2x {

// IEEE binary16 <-> binary32 on the host (round to nearest even), used when packing weights.
inline uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}

inline float f16_to_f32(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1F, m = h & 0x3FF, o;
    if (e == 0) {
        if (m == 0) o = s << 31;
        else { e = 127 - 15 + 1; while (!(m & 0x400)) { m <<= 1; --e; } m &= 0x3FF; o = (s << 31) | (e << 23) | (m << 13); }
    } else if (e == 31) o = (s << 31) | 0x7F800000u | (m << 13);
    else o = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
    float f; memcpy(&f, &o, 4); return f;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace w2x
