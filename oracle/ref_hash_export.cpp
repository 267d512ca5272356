// Exposes the REFERENCE's own murmur_hash64 / hash_bytes64
// (/root/reference/include/dint/hash_utils.hpp, compiled where it lies) with C
// linkage so tests can pin dint::murmur64a against it. This file contains no
// reference code: it only includes the header by name (-I points at it).
#include "hash_utils.hpp"

extern "C" unsigned long long ref_hash_u32s(const unsigned int* p, unsigned long n) {
    return ds2i::hash_bytes64(p, n);
}
extern "C" unsigned long long ref_murmur64(const void* key, unsigned long len, unsigned long long seed) {
    return ds2i::murmur_hash64(key, len, seed);
}
