// ORACLE — test infrastructure only (see vec.h).
// Restates cuda/random.h:31-67 (tea<N>, lcg, rnd).  NB the hot path includes
// cuda/random.h, not device_include/random.h.
// Pinned: oracle/_ref builds the reference's own cuda/random.h; tests compare.
#pragma once
#include <cstdint>

namespace orc {

template <unsigned N>
inline uint32_t tea(uint32_t val0, uint32_t val1) {  // random.h:31-46
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (unsigned n = 0; n < N; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

inline uint32_t lcg(uint32_t& prev) {  // random.h:49-55
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}

inline float rnd(uint32_t& prev) {  // random.h:64-67
    return (float)lcg(prev) / (float)0x01000000;
}

}  // namespace orc
