// CDNA4 (gfx950) decode kernels for the DINT codeword streams.
//
// What is computed is the reference's single_dint::decode
// (vroom_env/dint_codecs.hpp:37-107); how is unrelated to its one-codeword-at-a-
// time loop:
//
//  * one 64-lane wavefront walks one unit (include/dint_hip.h) in TILES of
//    64 * kSPL 16-bit slots; lane l owns the kSPL CONSECUTIVE slots kSPL*l ..
//    (one unaligned load), so (lane, k) order is stream order is output order;
//  * per slot one metadata word ((size-1) << 24 | payload offset): LDS for the hot
//    codewords, L2 for the cold ones, looked up one tile ahead of use;
//  * header/payload classification costs nothing unless a tile holds a 0 or 1
//    slot (or an exception straddles in); then a short per-lane state machine is
//    iterated until the lane-to-lane carries agree (one or two rounds);
//  * a local prefix plus ONE DPP wave scan gives every codeword its output offset;
//  * EXPANSION is output-centric: every codeword sets ONE bit at its first output
//    position in a per-wave flag bitmap (ds_or) and stores `source - position` in
//    a table indexed by its ordinal; a small scan over the bitmap's word
//    popcounts gives per-word rank bases. Then each lane takes 4 consecutive
//    output integers: flag word + rank base -> 4 ranks -> 4 table reads -> 4
//    gathers (LDS: hot payloads and the zero region of the runs; L2: cold
//    payloads; literal table: exceptions) -> one 16-byte store, so every global
//    store instruction covers 1 KB of consecutive output. Exactly n integers are
//    written per unit, nothing past them (the reference needs a pre-zeroed buffer
//    and a 256-word overflow area, include/dint/dint_codecs.hpp:11,
//    dict_posting_list.hpp:296).
//
// LDS (160 KB/CU, one 1024-thread workgroup per CU):
//   [ hot meta | 256 zero words | hot payloads ]  <= kHotImageWords, shared by 16 waves
//   16 x [ flag bitmap | rank bases | per-codeword delta table | literal table ]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dint_hip.h"

namespace dint_dev {

#ifndef DINT_BLOCK_THREADS
#define DINT_BLOCK_THREADS 1024
#endif
#ifndef DINT_BLOCKS_PER_CU
#define DINT_BLOCKS_PER_CU 1
#endif

constexpr uint32_t kWave = 64;
constexpr uint32_t kBlockThreads = DINT_BLOCK_THREADS;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;
constexpr uint32_t kBlocksPerCU = DINT_BLOCKS_PER_CU;
constexpr uint32_t kLdsWords = 160 * 1024 / 4 / kBlocksPerCU;
constexpr uint32_t kSPL = 4;                          // slots per lane per tile
constexpr uint32_t kTileSlots = kWave * kSPL;         // 256 slots per tile
constexpr uint32_t kCap = 2048;                       // outputs per expansion batch (>= kSPL * 256)
// per wave: flag bitmap, per-word rank bases, per-codeword delta and literal tables
constexpr uint32_t kScratchWords = kCap / 32 + kCap / 32 + kTileSlots + kTileSlots;
constexpr uint32_t kHotImageWords = kLdsWords - kWavesPerBlock * kScratchWords;
constexpr uint32_t kZeroWords = 256;                  // longest run codeword
constexpr uint32_t kColdBase = 1u << 24;              // source offsets >= this live in global memory
constexpr uint32_t kLitAddr = 0x70000000u;            // source "address" of an exception literal
constexpr uint32_t kQueueShards = 8;                  // dynamic unit queue: one counter per shard
constexpr uint32_t kQueueStride = 32;                 // words between counters (own 128-byte line each)

// One dictionary of the (possibly multi-) dictionary file.
struct dict_desc {
    uint32_t meta_base;  // first slot of this dictionary in gmeta
    uint32_t hot_base;   // LDS word offset of its hot meta table
    uint32_t hot_k;      // codewords < hot_k have meta + payload in the LDS image
    uint32_t pad;
};

// Device view of a dictionary file.
struct dict_view {
    const uint32_t* gmeta;      // per codeword slot: (size-1) << 24 | word offset into gtable
    const uint32_t* gtable;     // [256 zeros][payload words...]
    const uint32_t* lds_image;  // [256 zeros]{[hot meta of dictionary d]}[hot payloads], hot_words long
    const dict_desc* descs;     // one per dictionary (multi: 6)
    uint32_t gtable_words;
    uint32_t hot_words;         // multiple of 4
    dict_desc first;            // descs[0], for the single-dictionary kernel
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
    uint32_t* queue;    // kQueueShards counters, kQueueStride words apart, zero at launch
    uint32_t n_shards;  // counters in use
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) u32x4_a4 {
    u32x4 v;
};
struct __attribute__((packed, aligned(1))) u32x2_a1 {
    u32x2 v;
};
struct __attribute__((packed, aligned(1))) u32_a1 {
    uint32_t v;
};

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

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

// kSPL consecutive W-bit slots of one lane (W = 16: 8 bytes, W = 8: 4 bytes) from an
// arbitrary byte address. Never reads past the buffer: the tail
// lanes load the final bytes and shift (bytes past the end read as zero).
template <int W>
__device__ __forceinline__ uint64_t load_lane_slots(const uint8_t* enc, uint64_t byte_off, uint64_t enc_bytes) {
    constexpr uint32_t kBytes = kSPL * W / 8;
    const uint64_t last_valid = enc_bytes - kBytes;  // enc_bytes >= 8 is checked by the host
    const uint64_t o = byte_off < last_valid ? byte_off : last_valid;
    const uint64_t over = byte_off - o;  // 0 for all but the tail lanes
    uint64_t q;
    if (W == 16) {
        const u32x2 r = reinterpret_cast<const u32x2_a1*>(enc + o)->v;
        q = (uint64_t(r.y) << 32) | r.x;
    } else {
        q = reinterpret_cast<const u32_a1*>(enc + o)->v;
    }
    return over < kBytes ? q >> (8 * uint32_t(over)) : 0ull;
}

struct tile_regs {
    uint32_t s[kSPL];  // slot values
    uint32_t m[kSPL];  // metadata of each slot read as a codeword (garbage for payload slots)
};

template <int W>
__device__ __forceinline__ void unpack_slots(uint64_t raw, tile_regs& t) {
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) t.s[k] = uint32_t(raw >> (W * k)) & ((1u << W) - 1u);
}

// Metadata word of a codeword: LDS for the hot codewords, L2 for the cold ones.
// Two address spaces, two instructions: an unconditional DS read and a global
// read under the cold lanes' exec mask (a pointer select would turn both into
// one slow flat load).
__device__ __forceinline__ uint32_t lookup_meta(const dict_view& d, const uint32_t* lds, const dict_desc& dd,
                                                uint32_t v) {
#ifdef DINT_EXP_NOCOLD
    v = v < dd.hot_k ? v : 7 + v % (dd.hot_k - 7);
#endif
    const bool hot = v < dd.hot_k;
    uint32_t m = lds[hot ? dd.hot_base + v : 0u];
    asm volatile("" : "+v"(m));  // keep the DS read a DS read
    if (!hot) m = d.gmeta[dd.meta_base + v];
    return m;
}

// One SEGMENT: n integers from W-bit slots starting at byte in_off, all against one
// dictionary. A single-dictionary unit is one 16-bit segment; a multi-dictionary
// unit is a sequence of <= 256-integer segments (blocks), each 16- or 8-bit
// (vroom_env/dint_codecs.hpp:521-619). Exception payloads are 1 / 2 slots (W = 16)
// or 2 / 4 slots (W = 8). Returns the byte offset one past the last consumed slot.
template <int W>
__device__ __forceinline__ uint64_t decode_segment(const decode_args& a, const uint32_t* lds, uint32_t* scratch,
                                                   const dict_desc& dd, uint64_t in_off, uint32_t n,
                                                   uint32_t* const out, uint32_t lane) {
    constexpr uint32_t kSlotBytes = W / 8;
    constexpr uint32_t kTileBytes = kTileSlots * kSlotBytes;
    uint32_t* flagw = scratch;              // kCap / 32 words
    uint32_t* wbase = scratch + kCap / 32;  // per flag word: (#flags before it) - 1
    uint32_t* delta = wbase + kCap / 32;    // per codeword ordinal: source - position
    uint32_t* lit = delta + kTileSlots;     // per codeword ordinal: exception value

    const uint32_t hot_k = dd.hot_k;

    // pipeline: tile t in `cur` (slots + metadata), tile t+1 in `nxt`, tile t+2's slots in flight
    uint64_t slot_byte = in_off + uint64_t(kSlotBytes * kSPL) * lane;
    tile_regs cur, nxt;
    unpack_slots<W>(load_lane_slots<W>(a.enc, slot_byte, a.enc_bytes), cur);
    slot_byte += kTileBytes;
    uint64_t raw1 = load_lane_slots<W>(a.enc, slot_byte, a.enc_bytes);
    slot_byte += kTileBytes;
    uint64_t raw2 = load_lane_slots<W>(a.enc, slot_byte, a.enc_bytes);
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) cur.m[k] = lookup_meta(a.dict, lds, dd, cur.s[k]);

    uint32_t produced = 0;
    uint32_t carry = 0;            // payload slots an exception of the previous tile still owns
    uint64_t tile_base = in_off;   // byte offset of slot 0 of the current tile
    uint32_t end_slot = 0;

    while (produced < n) {
        // ---- stage tile t+1: unpack, start its metadata lookups ---------------------
        unpack_slots<W>(raw1, nxt);
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) nxt.m[k] = lookup_meta(a.dict, lds, dd, nxt.s[k]);
        const uint32_t next_lo = uint32_t(raw1);  // first slots of the next tile (exception spill)

        // ---- 1. classification ------------------------------------------------------------
        uint32_t smin = cur.s[0];
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) smin = smin < cur.s[k] ? smin : cur.s[k];
        const bool any_exc = __ballot(smin < 2) != 0 || carry != 0;
        uint32_t paybits = 0, excbits = 0;
        uint32_t carry_out = 0;
        uint32_t excval[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) excval[k] = 0;
        if (any_exc) {
            uint32_t st_in = lane == 0 ? carry : 0u;
            uint32_t st_out;
            for (;;) {
                uint32_t st = st_in;
                paybits = 0;
                excbits = 0;
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) {
                    const bool p = st != 0;
                    const bool e = !p && cur.s[k] < 2;
                    paybits |= uint32_t(p) << k;
                    excbits |= uint32_t(e) << k;
                    st = p ? st - 1 : (e ? (W == 16 ? cur.s[k] + 1 : 2 * cur.s[k] + 2) : 0u);
                }
                st_out = st;
                uint32_t prev = __shfl_up(st_out, 1);
                if (lane == 0) prev = carry;
                if (__ballot(prev != st_in) == 0) break;
                st_in = prev;
            }
            carry_out = readlane(st_out, 63);
            if (__ballot(excbits != 0)) {
                // slot values after this lane's: the next lane's first ones (lane 63: next tile's)
                if (W == 16) {
                    uint32_t nlo = __shfl_down((cur.s[1] << 16) | cur.s[0], 1);
                    if (lane == 63) nlo = readlane(next_lo, 0);
                    uint32_t e[kSPL + 2];
#pragma unroll
                    for (uint32_t k = 0; k != kSPL; ++k) e[k] = cur.s[k];
                    e[kSPL] = nlo & 0xFFFFu;
                    e[kSPL + 1] = nlo >> 16;
#pragma unroll
                    for (uint32_t k = 0; k != kSPL; ++k)
                        excval[k] = e[k] == 0 ? e[k + 1] : (e[k + 1] | (e[k + 2] << 16));
                } else {
                    uint32_t nlo = __shfl_down(cur.s[0] | (cur.s[1] << 8) | (cur.s[2] << 16) | (cur.s[3] << 24), 1);
                    if (lane == 63) nlo = readlane(next_lo, 0);
                    uint32_t e[kSPL + 4];
#pragma unroll
                    for (uint32_t k = 0; k != kSPL; ++k) e[k] = cur.s[k];
#pragma unroll
                    for (uint32_t k = 0; k != 4; ++k) e[kSPL + k] = (nlo >> (8 * k)) & 0xFFu;
#pragma unroll
                    for (uint32_t k = 0; k != kSPL; ++k) {
                        const uint32_t lo16 = e[k + 1] | (e[k + 2] << 8);
                        excval[k] = e[k] == 0 ? lo16 : (lo16 | (e[k + 3] << 16) | (e[k + 4] << 24));
                    }
                }
            }
        }

        // ---- 2. sizes, offsets, ordinals --------------------------------------------------
        uint32_t size[kSPL], src[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            const bool hdr = !((paybits >> k) & 1u);
            const bool exc = (excbits >> k) & 1u;
            size[k] = hdr ? (exc ? 1u : (cur.m[k] >> 24) + 1u) : 0u;
            src[k] = (cur.m[k] & 0xFFFFFFu) + (cur.s[k] < hot_k ? 0u : kColdBase);
#ifdef DINT_EXP_NOCOLD
            src[k] = cur.m[k] & 0xFFFFFFu;
#endif
        }
        const uint32_t hdrbits = ~paybits & 0xFu;
        uint32_t off[kSPL];
        off[0] = 0;
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) off[k] = off[k - 1] + size[k - 1];
        const uint32_t lsum = off[kSPL - 1] + size[kSPL - 1];
        const uint32_t packed = (uint32_t(__builtin_popcount(hdrbits)) << 24) | lsum;
        const uint32_t pincl = wave_inclusive_sum(packed);
        const uint32_t pexcl = pincl - packed;
        const uint32_t obase = pexcl & 0xFFFFFFu;  // first output of this lane's codewords
        const uint32_t rbase = pexcl >> 24;        // ordinal of this lane's first codeword
        const uint32_t remaining = n - produced;
        uint32_t total = readlane(pincl, 63) & 0xFFFFFFu;
        uint32_t actbits = hdrbits;
        uint32_t lsum_c = lsum;
        if (total >= remaining) {  // last tile of the unit: clamp, and find where the stream ends
            total = remaining;
            uint32_t cand = 0;
            actbits = 0;
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t pos = obase + off[k];
                const bool act = ((hdrbits >> k) & 1u) && pos < remaining;
                if (act) {
                    size[k] = size[k] < remaining - pos ? size[k] : remaining - pos;
                    actbits |= 1u << k;
                    const bool exc = (excbits >> k) & 1u;
                    cand = kSPL * lane + k + 1 + (exc ? (W == 16 ? cur.s[k] + 1 : 2 * cur.s[k] + 2) : 0u);
                } else {
                    size[k] = 0;
                }
            }
            lsum_c = obase < remaining ? (obase + lsum < remaining ? lsum : remaining - obase) : 0u;
            const uint64_t am = __ballot(actbits != 0);
            end_slot = readlane(cand, 63u - uint32_t(__builtin_clzll(am | 1ull)));
        }
        // ordinal of codeword k inside the lane
        uint32_t lord[kSPL];
        lord[0] = 0;
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) lord[k] = lord[k - 1] + ((hdrbits >> (k - 1)) & 1u);

        // ---- 3./4. batches of <= kCap outputs -------------------------------------------------
        uint32_t done = 0, rdone = 0;
        while (done < total) {
            const bool inb = lsum_c != 0 && obase >= done && (obase + lsum_c - done) <= kCap;
            const uint64_t bm = __ballot(inb);
            const uint32_t last = 63u - uint32_t(__builtin_clzll(bm | 1ull));
            const uint32_t bend = readlane(obase + lsum_c, last);
            const uint32_t rend = readlane(rbase + uint32_t(__builtin_popcount(actbits)), last);
            const uint32_t bt = bend - done;  // outputs in this batch, 1..kCap
            const uint32_t nwords = (bt + 31) >> 5;

            if (lane < nwords) flagw[lane] = 0;
            wave_lds_fence();
            bool any_lit = false;
            if (inb) {
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) {
                    if ((actbits >> k) & 1u) {
                        const uint32_t rel = obase + off[k] - done;
                        const uint32_t ord = rbase + lord[k] - rdone;
                        __hip_atomic_fetch_or(&flagw[rel >> 5], 1u << (rel & 31u), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if ((excbits >> k) & 1u) {
                            delta[ord] = kLitAddr - rel;  // position rel resolves to the literal marker
                            lit[ord] = excval[k];
                            any_lit = true;
                        } else {
                            delta[ord] = src[k] - rel;
                        }
                    }
                }
            }
            const bool batch_lit = __ballot(any_lit) != 0;
            wave_lds_fence();
            {
                const uint32_t wv = lane < nwords ? flagw[lane] : 0u;
                const uint32_t pc = uint32_t(__builtin_popcount(wv));
                const uint32_t pi = wave_inclusive_sum(pc);
                wbase[lane] = pi - pc - 1u;
            }
            wave_lds_fence();

            uint32_t* const obatch = out + (produced + done);
            for (uint32_t q0 = 0; q0 < bt; q0 += 4 * kWave) {
                const uint32_t p0 = q0 + 4 * lane;
                if (p0 < bt) {
                    const uint32_t widx = p0 >> 5, sh = p0 & 31u;
                    const uint32_t w = flagw[widx];
                    const uint32_t nib = (w >> sh) & 15u;
                    const uint32_t base = wbase[widx] + uint32_t(__builtin_popcount(w & ((1u << sh) - 1u)));
                    uint32_t r[4], d[4];
                    r[0] = base + (nib & 1u);
                    r[1] = base + uint32_t(__builtin_popcount(nib & 3u));
                    r[2] = base + uint32_t(__builtin_popcount(nib & 7u));
                    r[3] = base + uint32_t(__builtin_popcount(nib));
#pragma unroll
                    for (int k = 0; k != 4; ++k) d[k] = delta[r[k]];
                    uint32_t x[4], ad[4];
#pragma unroll
                    for (int k = 0; k != 4; ++k) {
                        ad[k] = d[k] + p0 + k;
                        x[k] = lds[ad[k] < kColdBase ? ad[k] : 0u];
                    }
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));  // keep DS reads DS reads
#ifndef DINT_EXP_NOCOLD
#pragma unroll
                    for (int k = 0; k != 4; ++k) {
                        if (ad[k] >= kColdBase && ad[k] != kLitAddr) {
                            uint32_t g = ad[k] - kColdBase;
                            g = g < a.dict.gtable_words ? g : a.dict.gtable_words - 1;
                            x[k] = a.dict.gtable[g];
                        }
                    }
#endif
                    if (batch_lit) {
#pragma unroll
                        for (int k = 0; k != 4; ++k) {
                            if (ad[k] == kLitAddr) x[k] = lit[r[k]];
                        }
                    }
#ifdef DINT_EXP_NOSTORE
                    if (x[0] == 0xDEADBEEFu && x[1] == 0x12345u) obatch[p0] = x[2] + x[3];
#else
                    if (p0 + 4 <= bt) {
                        u32x4 xv = {x[0], x[1], x[2], x[3]};
                        reinterpret_cast<u32x4_a4*>(obatch + p0)->v = xv;
                    } else {
                        obatch[p0] = x[0];
                        if (p0 + 1 < bt) obatch[p0 + 1] = x[1];
                        if (p0 + 2 < bt) obatch[p0 + 2] = x[2];
                    }
#endif
                }
            }
            wave_lds_fence();
            done = bend;
            rdone = rend;
        }

        produced += total;
        carry = carry_out;
        if (produced < n) tile_base += kTileBytes;

        // ---- rotate the pipeline ---------------------------------------------------
        cur = nxt;
        raw1 = raw2;
        slot_byte += kTileBytes;
        raw2 = load_lane_slots<W>(a.enc, slot_byte, a.enc_bytes);
    }
    return tile_base + uint64_t(kSlotBytes) * end_slot;
}

// A single-dictionary unit (rectangular or packed: the streams are byte-identical,
// only the dictionary source layout differed on the host) is one 16-bit segment.
__device__ __forceinline__ void decode_unit_single(const decode_args& a, const uint32_t* lds, uint32_t* scratch,
                                                   uint64_t unit_index, uint32_t lane) {
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || out_off + n > a.out_capacity) return;
    const uint64_t end = decode_segment<16>(a, lds, scratch, a.dict.first, up->in_off, n, a.out + out_off, lane);
    if (a.end_off && lane == 0) a.end_off[unit_index] = end;
}

// A multi-dictionary unit: blocks of 256 integers (the last one shorter), each opened
// by a selector byte: < 6 -> 16-bit codewords against dictionary `selector`, else 8-bit
// codewords against dictionary `selector - 6` (vroom_env/dint_codecs.hpp:521-619).
// Blocks carry no length, so they are decoded one after the other.
__device__ __forceinline__ void decode_unit_multi(const decode_args& a, const uint32_t* lds, uint32_t* scratch,
                                                  uint64_t unit_index, uint32_t lane) {
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || out_off + n > a.out_capacity) return;
    uint64_t pos = up->in_off;
    for (uint32_t done = 0; done < n;) {
        const uint32_t bsize = n - done < 256u ? n - done : 256u;
        const uint64_t sp = pos < a.enc_bytes ? pos : a.enc_bytes - 1;
        const uint32_t sel = uniform(a.enc[sp]);
        const bool narrow = sel >= 6;
        const uint32_t d = (narrow ? sel - 6 : sel) % 6;
        dict_desc dd;
        dd.meta_base = uniform(a.dict.descs[d].meta_base);
        dd.hot_base = uniform(a.dict.descs[d].hot_base);
        dd.hot_k = uniform(a.dict.descs[d].hot_k);
        dd.pad = 0;
        uint32_t* const out = a.out + out_off + done;
        if (narrow) pos = decode_segment<8>(a, lds, scratch, dd, pos + 1, bsize, out, lane);
        else pos = decode_segment<16>(a, lds, scratch, dd, pos + 1, bsize, out, lane);
        done += bsize;
    }
    if (a.end_off && lane == 0) a.end_off[unit_index] = pos;
}

// Units are handed out dynamically: their cost varies a lot (a sparse list full
// of exceptions takes several times longer than a dense one of the same length),
// so a static unit -> wave map leaves most of the chip idle behind the slowest
// waves. kQueueShards counters (one per group of workgroups, blockIdx % 8 — the
// workgroups that share an XCD under round-robin placement; a speed choice only)
// each serve the units u = shard + n_shards * j; a wave draws its next index
// while it is still decoding the current unit.
template <bool MULTI>
__device__ __forceinline__ void decode_kernel_body(const decode_args& a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < a.dict.hot_words; i += kBlockThreads) lds[i] = a.dict.lds_image[i];
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x / kWave);
    uint32_t* scratch = lds + a.dict.hot_words + wave * kScratchWords;
    const uint32_t shard = blockIdx.x % a.n_shards;  // n_shards = min(kQueueShards, gridDim.x)
    uint32_t* counter = a.queue + shard * kQueueStride;
    const uint64_t shard_units = (a.n_units + a.n_shards - 1 - shard) / a.n_shards;
    auto draw = [&]() -> uint32_t {
        uint32_t j = 0;
        if (lane == 0) j = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return uniform(j);
    };
    uint32_t j = draw();
    while (j < shard_units) {
        const uint32_t j_next = draw();
        const uint64_t u = uint64_t(shard) + uint64_t(a.n_shards) * j;
        if (MULTI) decode_unit_multi(a, lds, scratch, u, lane);
        else decode_unit_single(a, lds, scratch, u, lane);
        j = j_next;
    }
}

__global__ __launch_bounds__(kBlockThreads) void decode_single_kernel(decode_args a) { decode_kernel_body<false>(a); }
__global__ __launch_bounds__(kBlockThreads) void decode_multi_kernel(decode_args a) { decode_kernel_body<true>(a); }

// test hook: out[i] = inclusive prefix sum of in[0..i] over one wave
__global__ void debug_wave_scan_kernel(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]);
}

}  // namespace dint_dev
