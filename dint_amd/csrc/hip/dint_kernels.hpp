// CDNA4 (gfx950) decode kernels for the DINT codeword streams.
//
// What is computed is the reference's single_dint::decode / multi_opt_dint::
// decode (vroom_env/dint_codecs.hpp:37-107, :521-619); how is unrelated to its
// one-codeword-at-a-time loop:
//
//  * one 64-lane wavefront walks one unit (include/dint_hip.h) 64 codeword
//    SLOTS at a time, lane l owning slot l;
//  * which slots are codeword headers and which are exception payloads is a
//    3-state machine over the slots; it is resolved on the scalar unit from two
//    `__ballot` masks (slot == 0, slot == 1) — one scalar iteration per
//    exception, none in the common all-dictionary chunk;
//  * header lanes look up (size, source offset) — LDS for the hot codewords,
//    L2 for the cold ones — and a DPP wave prefix sum turns sizes into output
//    offsets;
//  * expansion is OUTPUT-centric: each header lane drops a flag byte at its
//    first output position and its (source − position) delta into a compact
//    table; then for every 64 consecutive output integers the wave reads 64 flag
//    bytes, `__ballot`s them into a bitmap, ranks with mbcnt to find the owning
//    codeword, gathers the source word (LDS or L2) and issues ONE fully
//    coalesced 256-byte store. Zero runs are ordinary entries that point at a
//    256-word zero region, exceptions are entries that point at a per-wave
//    literal pool. Exactly n integers are written per unit, nothing past them
//    (the reference needs a pre-zeroed buffer and a 256-word overflow area,
//    include/dint/dint_codecs.hpp:11, dict_posting_list.hpp:296).
//
// LDS (160 KB/CU, one 1024-thread workgroup per CU):
//   [ hot meta | 256 zero words | hot payload ]  <= kHotImageWords, shared
//   16 x [ 1 KiB flag bytes | 64-word delta table | 64-word literal pool ]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dint_hip.h"

namespace dint_dev {

constexpr uint32_t kWave = 64;
constexpr uint32_t kBlockThreads = 1024;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;
constexpr uint32_t kLdsWords = 160 * 1024 / 4;
constexpr uint32_t kCap = 1024;                       // outputs per expansion batch (>= 256)
constexpr uint32_t kScratchWords = kCap / 4 + 64 + 64;  // flags + delta table + literal pool
constexpr uint32_t kHotImageWords = kLdsWords - kWavesPerBlock * kScratchWords;
constexpr uint32_t kZeroWords = 256;                  // longest run codeword
constexpr uint32_t kColdBase = 1u << 24;              // source offsets >= this live in global memory

// Device view of one dictionary (single kinds: num_dicts == 1).
struct dict_view {
    const uint32_t* gmeta;      // per codeword: (size-1) << 24 | word offset into gtable
    const uint32_t* gtable;     // [256 zeros][payload words...]
    const uint32_t* lds_image;  // kHot image, hot_words long
    uint32_t gtable_words;
    uint32_t hot_words;         // multiple of 4
    uint32_t hot_k;             // codewords < hot_k have their meta + payload in the LDS image
};

struct decode_args {
    dict_view dict;
    const uint8_t* enc;
    uint64_t enc_bytes;
    const dint_unit* units;
    uint64_t n_units;
    uint32_t* out;
    uint64_t out_capacity;
    uint64_t* end_off;  // nullable
};

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

__device__ __forceinline__ bool lane_bit(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// Inclusive prefix sum over the 64 lanes, in registers: four row_shr steps inside
// each row of 16, then row_bcast:15 / row_bcast:31 across rows (gfx9 DPP).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return x;
}

__device__ __forceinline__ uint32_t readlane(uint32_t x, uint32_t l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// Orders this wave's LDS traffic between phases that communicate across lanes.
// LDS operations of one wave execute in issue order, so no hardware barrier is
// needed; this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t load_slot16(const uint8_t* enc, uint64_t byte_off, uint64_t last_valid) {
    uint64_t o = byte_off < last_valid ? byte_off : last_valid;
    uint16_t v;
    __builtin_memcpy(&v, enc + o, 2);  // payloads start at arbitrary byte addresses
    return v;
}

// Scalar resolution of the header/payload state machine for 64 slots.
//   e0 / e1 : lanes whose slot value is 0 / 1
//   carry   : payload slots (0..2) the previous chunk's last exception still owns
// Returns the mask of exception HEADER lanes; `pay` receives the payload lanes,
// `carry_out` the payload slots spilling into the next chunk.
__device__ __forceinline__ uint64_t resolve_slots16(uint64_t e0, uint64_t e1, uint32_t carry, uint64_t& pay,
                                                     uint32_t& carry_out) {
    pay = (1ull << carry) - 1ull;
    carry_out = 0;
    uint64_t exc = 0;
    uint64_t cand = (e0 | e1) & ~pay;
    while (cand) {
        uint32_t p = uint32_t(__builtin_ctzll(cand));
        uint64_t bit = 1ull << p;
        exc |= bit;
        uint32_t len = (e1 & bit) ? 2u : 1u;
        uint64_t m = (bit << 1) | (len == 2 ? (bit << 2) : 0ull);  // bits shifted past 63 fall off
        pay |= m;
        uint32_t end = p + 1 + len;
        carry_out = end > 64 ? end - 64 : 0;
        cand &= ~(pay | bit);
    }
    return exc;
}

// One unit of a single-dictionary stream (rectangular or packed: the streams
// are byte-identical, only the dictionary source layout differed on the host).
__device__ __forceinline__ void decode_unit_single(const decode_args& a, uint32_t* lds, uint32_t* scratch,
                                                   uint64_t unit_index, uint32_t lane) {
    uint8_t* flags = reinterpret_cast<uint8_t*>(scratch);
    uint32_t* flag_words = scratch;
    uint32_t* delta = scratch + kCap / 4;
    uint32_t* lit = delta + 64;
    const uint32_t lit_base = uint32_t(lit - lds);

    const dint_unit* up = a.units + unit_index;
    const uint64_t in_off = up->in_off;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0) return;
    const uint64_t last_valid = a.enc_bytes >= 2 ? a.enc_bytes - 2 : 0;
    const uint32_t hot_k = a.dict.hot_k;

    uint32_t produced = 0;
    uint32_t carry = 0;
    // slot values of the next two chunks are kept in flight in registers
    uint64_t slot_byte = in_off + 2ull * lane;
    uint32_t v_next = load_slot16(a.enc, slot_byte, last_valid);
    slot_byte += 2 * kWave;
    uint32_t v_next2 = load_slot16(a.enc, slot_byte, last_valid);
    uint64_t chunk_base = in_off;  // byte offset of slot 0 of the current chunk
    uint32_t end_slot = 0;

    while (produced < n) {
        const uint32_t v = v_next;
        v_next = v_next2;
        slot_byte += 2 * kWave;
        v_next2 = load_slot16(a.enc, slot_byte, last_valid);  // prefetch two chunks ahead

        // ---- classify slots ------------------------------------------------
        const uint64_t e0 = __ballot(v == 0);
        const uint64_t e1 = __ballot(v == 1);
        uint64_t pay;
        uint32_t carry_out;
        const uint64_t exc = resolve_slots16(e0, e1, carry, pay, carry_out);
        const bool is_hdr = lane_bit(~pay);
        const bool is_exc = lane_bit(exc);

        // ---- size + source of every header lane -----------------------------
        uint32_t size = 0, src = 0;
        if (exc) {  // wave-uniform: rare
            uint32_t s1 = __shfl_down(v, 1);
            uint32_t s2 = __shfl_down(v, 2);
            const uint32_t nx0 = readlane(v_next, 0), nx1 = readlane(v_next, 1);
            if (lane == 63) s1 = nx0;
            if (lane == 62) s2 = nx0;
            if (lane == 63) s2 = nx1;
            if (is_exc) lit[lane] = (v == 1) ? (s1 | (s2 << 16)) : s1;
        }
        if (is_hdr) {
            if (is_exc) {
                size = 1;
                src = lit_base + lane;
            } else {
                // two address spaces, two instructions: an unconditional LDS read and a
                // global read under the cold lanes' exec mask (a pointer select would
                // turn both into one slow flat load)
                const bool hot = v < hot_k;
                uint32_t m = lds[hot ? v : 0u];
                asm volatile("" : "+v"(m));  // keep the DS read a DS read
                if (!hot) m = a.dict.gmeta[v];
                size = (m >> 24) + 1;
                src = (m & 0xFFFFFFu) + (hot ? 0u : kColdBase);
            }
        }

        // ---- output offsets ---------------------------------------------------
        const uint32_t incl = wave_inclusive_sum(size);
        const uint32_t excl = incl - size;
        const uint32_t remaining = n - produced;
        const bool act = is_hdr && excl < remaining;
        const uint32_t size_c = act ? (size < remaining - excl ? size : remaining - excl) : 0;
        const uint32_t endpos = excl + size_c;
        uint32_t total = readlane(incl, 63);
        total = total < remaining ? total : remaining;

        if (produced + total >= n) {  // last chunk of the unit: where does the stream end
            const uint64_t am = __ballot(act);
            const uint32_t last = 63u - uint32_t(__builtin_clzll(am | 1ull));
            const uint64_t lb = 1ull << last;
            end_slot = last + 1 + ((exc & lb) ? ((e1 & lb) ? 2u : 1u) : 0u);
        }

        // ---- expand, kCap outputs at a time ---------------------------------
        uint32_t done = 0;
        while (done < total) {
            const uint32_t rel = excl - done;
            const bool inb = act && excl >= done && (endpos - done) <= kCap;
            const uint64_t bm = __ballot(inb);
            const uint32_t last = 63u - uint32_t(__builtin_clzll(bm | 1ull));
            const uint32_t bend = readlane(endpos, last);
            const uint32_t bt = bend - done;  // outputs in this batch, 1..kCap

            for (uint32_t w = lane; w * 4 < bt; w += kWave) flag_words[w] = 0;
            wave_lds_fence();
            if (inb) {
                flags[rel] = 1;
                delta[mbcnt(bm)] = src - rel;
            }
            wave_lds_fence();

            uint32_t rank_base = 0;
            const uint64_t obase = out_off + produced + done;
            for (uint32_t q = 0; q < bt; q += kWave) {
                const uint32_t pos = q + lane;
                const bool ok = pos < bt;
                const uint32_t f = ok ? flags[pos] : 0u;
                const uint64_t fm = __ballot(f != 0);
                const uint32_t rank = rank_base + mbcnt(fm) + (f ? 1u : 0u) - 1u;
                rank_base += uint32_t(__builtin_popcountll(fm));
                if (ok) {
                    const uint32_t s = delta[rank] + pos;
                    uint32_t val = lds[s < kColdBase ? s : 0u];
                    asm volatile("" : "+v"(val));  // keep the DS read a DS read
                    if (s >= kColdBase) {
                        uint32_t g = s - kColdBase;
                        g = g < a.dict.gtable_words ? g : a.dict.gtable_words - 1;
                        val = a.dict.gtable[g];
                    }
                    const uint64_t o = obase + pos;
                    if (o < a.out_capacity) a.out[o] = val;
                }
            }
            wave_lds_fence();
            done = bend;
        }

        produced += total;
        carry = carry_out;
        if (produced < n) chunk_base += 2 * kWave;
    }
    if (a.end_off && lane == 0) a.end_off[unit_index] = chunk_base + 2ull * end_slot;
}

__global__ __launch_bounds__(kBlockThreads) void decode_single_kernel(decode_args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < a.dict.hot_words; i += kBlockThreads) lds[i] = a.dict.lds_image[i];
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x / kWave);
    uint32_t* scratch = lds + a.dict.hot_words + wave * kScratchWords;
    const uint64_t total_waves = uint64_t(gridDim.x) * kWavesPerBlock;
    for (uint64_t u = uint64_t(blockIdx.x) * kWavesPerBlock + wave; u < a.n_units; u += total_waves) {
        decode_unit_single(a, lds, scratch, u, lane);
    }
}

// test hook: out[i] = inclusive prefix sum of in[0..i] over one wave
__global__ void debug_wave_scan_kernel(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]);
}

}  // namespace dint_dev
