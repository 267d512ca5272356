// MurmurHash64A over a u32 sequence, seed 0.
//
// The reference's encoder matches dictionary entries by this hash ALONE, with
// no key comparison (include/dint/single_dictionary.hpp:167-175,
// include/dint/hash_utils.hpp:7-80), so an encoder that wants to choose the
// same codewords has to reproduce the function bit for bit. Since the keys are
// whole u32 words the tail switch of the published algorithm only ever sees
// len & 7 in {0, 4}.
#pragma once
#include <cstdint>
#include <cstring>

namespace dint {

inline uint64_t murmur64a(void const* key, size_t len, uint64_t seed) {
    constexpr uint64_t m = 0xc6a4a7935bd1e995ULL;
    constexpr int r = 47;
    uint64_t h = seed ^ (uint64_t(len) * m);
    auto p = static_cast<unsigned char const*>(key);
    size_t nblocks = len / 8;
    for (size_t i = 0; i != nblocks; ++i, p += 8) {
        uint64_t k;
        std::memcpy(&k, p, 8);
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    size_t tail = len & 7;
    if (tail) {
        uint64_t t = 0;
        for (size_t i = tail; i-- != 0;) t = (t << 8) | p[i];
        h ^= t;
        h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

inline uint64_t hash_u32s(uint32_t const* p, size_t n) {
    return murmur64a(p, n * sizeof(uint32_t), 0);
}

}  // namespace dint
