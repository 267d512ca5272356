// Block statistics on the device: the counting half of dictionary construction —
// adjusted::collect (reference include/dint/statistics_collectors.hpp:90-118): every ALIGNED 16/8/4/2/1-gram of
// every sampled list (multi-dictionary flavour: of every whole 256-integer block, under the block's context,
// :21-40), keyed like the reference by the MurmurHash64A of its integers alone (block_statistics.hpp:82-106,
// hash_utils.hpp:7-71), counted. The selection (filter, frequency sort, DSF, packing) stays on the host: it is
// a sort over the distinct n-grams, a thousandth of the counting work.
//
// One wavefront per 256-integer chunk of a list (chunks are list-aligned, so every aligned n-gram lies inside
// one chunk): the chunk in LDS, 496 n-grams over 64 lanes, each hashed and counted in an open-addressing table in
// device memory (linear probing; key claimed with a 64-bit CAS, the count a 32-bit atomic add, the first
// occurrence an atomic minimum over `position << 8 | length code << 3 | context`).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dint_hip.h"

namespace dint_dev {

__device__ __forceinline__ uint64_t murmur64a_u32s(const uint32_t* p, uint32_t n) {  // seed 0 (dint/hash.hpp)
    constexpr uint64_t m = 0xc6a4a7935bd1e995ull;
    uint64_t h = uint64_t(4 * n) * m;
    for (uint32_t i = 0; i + 2 <= n; i += 2) {
        uint64_t k = uint64_t(p[i]) | (uint64_t(p[i + 1]) << 32);
        k *= m;
        k ^= k >> 47;
        k *= m;
        h ^= k;
        h *= m;
    }
    if (n & 1u) {  // a 4-byte tail
        h ^= uint64_t(p[n - 1]);
        h *= m;
    }
    h ^= h >> 47;
    h *= m;
    h ^= h >> 47;
    return h;
}

// context of a block = ceil_log2(ceil_log2(max + 1)), 0 when max <= 1 (statistics_collectors.hpp:21-40); the + 1 in 32 bits,
// as there (:23,36): a block holding 0xFFFFFFFF is context 0
__device__ __forceinline__ uint32_t ceil_log2_dev(uint64_t x) { return x <= 1 ? 0u : 64u - uint32_t(__builtin_clzll(x - 1)); }
__device__ __forceinline__ uint32_t block_context(uint32_t max_value) {
    return max_value > 1 ? ceil_log2_dev(ceil_log2_dev(uint32_t(max_value + 1u))) : 0u;
}

struct ngram_table {
    unsigned long long* keys;  // 0 = empty
    unsigned long long* info;  // min over occurrences of position << 8 | length code << 3 | context
    uint32_t* freq;
    uint64_t mask;             // slots - 1
    uint32_t* overflow;        // set when a probe sequence found no room
};

constexpr uint32_t kStatsWaves = 2;        // waves per workgroup, one chunk each
constexpr uint32_t kLocalSlots = 1024;     // per wave: the chunk's own table in LDS (496 n-grams at most)

__device__ __forceinline__ void table_add(const ngram_table& t, unsigned long long key, unsigned long long info, uint32_t count) {
    uint64_t slot = (key * 0xD6E8FEB86659FD93ull >> 17) & t.mask;
    for (uint32_t probe = 0; probe != 4096; ++probe) {
        unsigned long long cur = t.keys[slot];
        if (cur == 0) cur = atomicCAS(&t.keys[slot], 0ull, key);
        if (cur == 0 || cur == key) {
            atomicAdd(&t.freq[slot], count);
            atomicMin(&t.info[slot], info);
            return;
        }
        slot = (slot + 1) & t.mask;
    }
    *t.overflow = 1u;
}

// A chunk's n-grams are counted in LDS first and reach the device-wide table once each, with their count: the
// common n-grams (a lone 0, a run of zeros) would otherwise be millions of atomic adds to one address — 0.14 s for a
// 2e7-integer sample, all of it waiting for that address.
__global__ __launch_bounds__(64 * kStatsWaves) void count_ngrams_kernel(const uint32_t* gaps, const uint64_t* chunk_start,
                                                                       const uint32_t* chunk_n, uint64_t n_chunks, uint32_t multi,
                                                                       ngram_table t) {
    __shared__ uint32_t stage[kStatsWaves][256];
    __shared__ unsigned long long lkey[kStatsWaves][kLocalSlots], linfo[kStatsWaves][kLocalSlots];
    __shared__ uint32_t lcount[kStatsWaves][kLocalSlots];
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint64_t u = uint64_t(blockIdx.x) * kStatsWaves + wave;
    if (u >= n_chunks) return;
    const uint64_t start = chunk_start[u];
    const uint32_t n = chunk_n[u];
    if (n == 0 || n > 256 || (multi && n != 256)) return;  // (multi: whole blocks only, statistics_collectors.hpp:95-99)
    uint32_t* const s = stage[wave];
    unsigned long long* const keys = lkey[wave];
    unsigned long long* const infos = linfo[wave];
    uint32_t* const counts = lcount[wave];
    for (uint32_t i = lane; i < kLocalSlots; i += 64) keys[i] = 0, infos[i] = ~0ull, counts[i] = 0;
    uint32_t mx = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t v = gaps[start + i];
        s[i] = v;
        mx = v > mx ? v : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d);
        mx = o > mx ? o : mx;
    }
    const uint32_t ctx = multi ? block_context(mx) : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // n-gram ids of a chunk: [0, 16) length 16, [16, 48) 8, [48, 112) 4, [112, 240) 2, [240, 496) 1
    for (uint32_t id = lane; id < 496; id += 64) {
        uint32_t len, idx, code;
        if (id < 16) len = 16, idx = id, code = 4;
        else if (id < 48) len = 8, idx = id - 16, code = 3;
        else if (id < 112) len = 4, idx = id - 48, code = 2;
        else if (id < 240) len = 2, idx = id - 112, code = 1;
        else len = 1, idx = id - 240, code = 0;
        const uint32_t at = idx * len;
        if (at + len > n) continue;
        const uint64_t h = murmur64a_u32s(s + at, len);
        // the table's key: the hash with the context folded in (a bijection per context), never 0
        unsigned long long key = h ^ (uint64_t(ctx + 1) * 0x9E3779B97F4A7C15ull);
        if (key == 0) key = 1;
        const unsigned long long info = ((start + at) << 8) | (uint64_t(code) << 3) | ctx;
        uint32_t slot = uint32_t(key * 0xD6E8FEB86659FD93ull >> 40) & (kLocalSlots - 1);
        for (;;) {  // (at most 496 of the 1024 slots fill up)
            unsigned long long cur = keys[slot];
            if (cur == 0) cur = atomicCAS(&keys[slot], 0ull, key);
            if (cur == 0 || cur == key) {
                atomicAdd(&counts[slot], 1u);
                atomicMin(&infos[slot], info);
                break;
            }
            slot = (slot + 1) & (kLocalSlots - 1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t i = lane; i < kLocalSlots; i += 64)
        if (keys[i] != 0) table_add(t, keys[i], infos[i], counts[i]);
}

// The selection takes, per context, the first top_k n-grams by (occurrences, length, integers) among those the
// reference's filter keeps (decreasing_static_frequencies::filter, dictionary_builders.hpp:15-38: saving > 1e-7 or a
// single integer). Which occurrence count still makes the cut is found here — counting passes over the compacted
// entries, bisection on the host — and only the entries at or above it travel to the host (ties included).
__device__ __forceinline__ bool ngram_kept(const dint_ngram& e, double total_ints) {
    const double codeword_bits = 16.0, initial_bpi = 3 * codeword_bits;  // log2(65536)
    const double saving = double(e.freq) * (initial_bpi * double(e.len) - codeword_bits) / total_ints;
    return saving > 0.0001 / 1000 || e.len == 1;
}
__global__ void count_at_least_kernel(const dint_ngram* e, uint64_t n, double total_ints, const uint32_t* at_least /*[8]*/,
                                      unsigned long long* counts /*[8]*/) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dint_ngram x = e[i];
    if (x.freq >= at_least[x.ctx & 7] && ngram_kept(x, total_ints)) atomicAdd(&counts[x.ctx & 7], 1ull);
}
__global__ void keep_at_least_kernel(const dint_ngram* e, uint64_t n, const uint32_t* at_least, dint_ngram* out, unsigned long long* n_out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dint_ngram x = e[i];
    if (x.freq >= at_least[x.ctx & 7]) out[atomicAdd(n_out, 1ull)] = x;
}

// ---- the selection itself (decreasing_static_frequencies::build, dictionary_builders.hpp:55-75; the order:
// block_statistics.hpp:246-276 freq_length sorter, ties by the integers — the deterministic tie-break of
// dint/statistics.hpp): the kept n-grams of every context in dictionary order, the first top_k of each.
__global__ void keep_filtered_kernel(const dint_ngram* e, uint64_t n, double total_ints, dint_ngram* out, unsigned long long* n_out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dint_ngram x = e[i];
    if (ngram_kept(x, total_ints)) out[atomicAdd(n_out, 1ull)] = x;
}
struct ngram_dictionary_order {
    const uint32_t* gaps;
    __device__ bool operator()(const dint_ngram& a, const dint_ngram& b) const {
        if (a.ctx != b.ctx) return a.ctx < b.ctx;
        if (a.freq != b.freq) return a.freq > b.freq;  // most frequent first
        if (a.len != b.len) return a.len > b.len;      // then the longer
        for (uint32_t i = 0; i != a.len; ++i) {        // then by their integers
            const uint32_t x = gaps[a.pos + i], y = gaps[b.pos + i];
            if (x != y) return x < y;
        }
        return false;
    }
};
// first index of every context's run in the sorted entries (first[c] = n where the context has none: preset by the host)
__global__ void context_starts_kernel(const dint_ngram* e, uint64_t n, unsigned long long* first /*[8]*/) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || e[i - 1].ctx != e[i].ctx) first[e[i].ctx & 7] = i;
}
// the first top_k of every context, in order: entry i of context c goes to base[c] + (i - first[c])
__global__ void take_top_kernel(const dint_ngram* e, uint64_t n, const unsigned long long* first, const unsigned long long* base, uint32_t top_k,
                                dint_ngram* out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = e[i].ctx & 7;
    const uint64_t rank = i - first[c];
    if (rank < top_k) out[base[c] + rank] = e[i];
}

__global__ void collect_ngrams_kernel(ngram_table t, dint_ngram* out, unsigned long long* n_out, uint64_t capacity) {
    const uint64_t slot = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (slot > t.mask || t.keys[slot] == 0) return;
    const unsigned long long k = atomicAdd(n_out, 1ull);
    if (k >= capacity) return;
    const unsigned long long info = t.info[slot];
    dint_ngram e;
    e.pos = info >> 8;
    e.freq = t.freq[slot];
    e.len = uint8_t(1u << ((info >> 3) & 7u));
    e.ctx = uint8_t(info & 7u);
    e.pad = 0;
    out[k] = e;
}

}  // namespace dint_dev
