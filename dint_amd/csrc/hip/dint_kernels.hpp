// CDNA4 (gfx950) decode kernels for the DINT codeword streams.
//
// What is computed is the reference's single_dint::decode
// (vroom_env/dint_codecs.hpp:37-107); how is unrelated to its one-codeword-at-a-
// time loop:
//
//  * one 64-lane wavefront walks one work item — a unit (include/dint_hip.h), or a bundle of
//    tiny units packed into one tile — in TILES of 64 * kSPL 16-bit slots; lane l owns the kSPL
//    CONSECUTIVE slots kSPL*l .. (one unaligned load), so (lane, k) order is stream order is
//    output order;
//  * per slot one metadata word ((size-1) << 24 | payload offset): LDS for the hot
//    codewords (a prefix of the dictionary), L2 for the cold ones, looked up one tile ahead;
//  * header/payload classification: a table-driven per-lane state machine, iterated until the
//    lane-to-lane carries agree (one or two rounds); nothing to do in tiles without 0 / 1 slots;
//  * a local prefix plus ONE DPP wave scan gives every codeword its output offset;
//  * EXPANSION is output-centric: every codeword sets ONE bit at its first output
//    position in a per-wave flag bitmap (ds_or) and stores `source - position` in
//    a table indexed by its ordinal; a small scan over the bitmap's word
//    popcounts gives per-word rank bases. Then each lane takes 4 consecutive
//    output integers: flag word + rank base -> 4 ranks -> 4 table reads -> 4 LDS
//    gathers -> one 16-byte non-temporal store, so every global store instruction covers
//    1 KB of consecutive output. Every source is in LDS by then: hot payloads and the zero
//    region of the runs live there; the COLD payloads of a batch are fetched once per
//    codeword (a compact worklist, one 16-byte load per lane and quad, while the batch
//    tables are being built) into per-wave staging cells, where the exception literals go
//    too. Exactly n integers are written per unit, nothing past them (the reference needs a
//    pre-zeroed buffer and a 256-word overflow area, include/dint/dint_codecs.hpp:11,
//    dict_posting_list.hpp:296);
//  * WAITS: gfx950 counts loads and stores in one in-order counter, so a tile has exactly one
//    wait point — after its cold fetch, before its stores — where everything prefetched is
//    consumed (read-write asm barriers, so that the compiler never adds a wait behind the
//    stores, which would be a wait for their acknowledgements).
//
// LDS (160 KB/CU, one 1024-thread workgroup per CU):
//   [ 256 zero words | hot meta | hot payloads ]  <= kHotImageWords, shared by 16 waves
//   [ slot classification table, 1.3 KB ]
//   16 x [ {flag word, rank base} pairs | per-codeword delta table | staging cells ]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dint_hip.h"

namespace dint_dev {

// Cache policy of the output stores (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1). The decoded integers are
// written once and never read by this kernel: non-temporal stores keep the 4 bytes/integer output stream
// from evicting the dictionary's cold part and the block directories out of L2 and from queueing
// behind write-back traffic — 0.97 -> 0.72 ms on the 4e8-posting run, the largest single gain measured.
#ifndef DINT_STORE_AUX
#define DINT_STORE_AUX 2
#endif
#ifndef DINT_STREAM_LOADS_NT
#define DINT_STREAM_LOADS_NT 0  // 1: the codeword stream is read with non-temporal loads as well
#endif
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
#ifndef DINT_GROUPS
#define DINT_GROUPS 2
#endif
#ifndef DINT_EARLY_METAS
#define DINT_EARLY_METAS 0
#endif
#ifndef DINT_UNIT_CHAIN
#define DINT_UNIT_CHAIN 0  // 1: single-dictionary units request their successor's first tiles (measured: no gain)
#endif
constexpr uint32_t kGroups = DINT_GROUPS;             // 256-output groups expanded together (one round)
constexpr uint32_t kRounds = 8 / kGroups;             // rounds per batch (single-dictionary segments)
constexpr uint32_t kMaxCap = 2048;                    // outputs per expansion batch at most: the flag bitmap's bits
// per wave: 64 {flag word, rank base} pairs (+ spare), per-codeword delta table (+ 4 dummy
// entries for codewords that are not live in a batch), staging cells
constexpr uint32_t kFwWords = 2 * 64 + 4;              // 64 pairs: flag positions are taken mod 2048
constexpr uint32_t kDeltaWords = kTileSlots + 4;
#ifndef DINT_STAGE_QUADS
#define DINT_STAGE_QUADS 128
#endif
constexpr uint32_t kStageQuads = DINT_STAGE_QUADS;     // 16-byte cells: cold payloads and exception literals of a batch
constexpr uint32_t kStageWords = 4 * kStageQuads;      // (its first half doubles as the fetch worklist)
constexpr uint32_t kScratchWords = kFwWords + kDeltaWords + kStageWords;
constexpr uint32_t kClassTableWords = 328;            // slot classification table: 648 u16 rows, padded
constexpr uint32_t kHotImageWords = kLdsWords - kClassTableWords - kWavesPerBlock * kScratchWords;
constexpr uint32_t kZeroWords = 256;                  // longest run codeword
constexpr uint32_t kColdBase = 1u << 23;              // source offsets >= this live in global memory
constexpr uint32_t kColdBase4 = 4 * kColdBase;        // the same in bytes
constexpr uint32_t kQueueShards = 8;                  // dynamic unit queue: one counter per shard
constexpr uint32_t kQueueStride = 32;                 // words between counters (own 128-byte line each)

// One dictionary of the (possibly multi-) dictionary file.
struct dict_desc {
    uint32_t meta_base;    // first slot of this dictionary in gmeta
    uint32_t hot_base;     // LDS word offset of its hot meta table
    uint32_t hot_k;        // codewords < hot_k have meta + payload in the LDS image
    uint32_t pad;
};

// Device view of a dictionary file.
struct dict_view {
    const uint32_t* gmeta;      // per codeword slot: (size-1) << 24 | kColdBase | word offset into gtable
    const uint32_t* gtable;     // [256 zeros][payload words...]
    const uint32_t* lds_image;  // [256 zeros]{[hot meta of dictionary d]}[hot payloads], hot_words long
    const dict_desc* descs;     // one per dictionary (multi: 6)
    uint32_t gmeta_words;
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
    uint32_t only_full; // in-index path: decode units of exactly 256 integers only (tails are interpolative)
    const uint8_t* sched;  // nullable; per unit: 0 = member of a bundle led by an earlier unit, 1 = on its own,
                           // c > 1 = leads a bundle of c consecutive tiny units (bundle_schedule_kernel)
    const uint32_t* items; // with sched: the units with sched != 0, in order — what the queue hands out
    const uint32_t* n_items;
    const uint32_t* spans; // nullable; per unit an upper bound of its stream bytes (else: up to the next unit's start)
    uint32_t plus_one;     // in-index freqs parts: every decoded integer + 1 (dict_posting_list.hpp:164-169)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) u32x4_a4 {
    u32x4 v;
};
struct __attribute__((packed, aligned(1))) u32x4_a1 {
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

// Inclusive prefix maximum over the 64 lanes (same DPP pattern; zero fill is neutral for unsigned max).
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false)));
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false)));
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false)));
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false)));
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false)));
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false)));
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
// arbitrary byte address. `tile_byte` is the (wave-uniform) offset of the tile's
// first slot: when the whole tile lies inside the buffer — every tile but the stream's last —
// this is one plain load whose result nothing touches until the tile is unpacked, two tiles
// later. Otherwise it never reads past the buffer: the tail lanes load the final bytes and
// shift (bytes past the end read as zero), which waits for the data on the spot.
template <int W>
__device__ __forceinline__ uint64_t load_lane_slots(const uint8_t* enc, uint64_t tile_byte, uint32_t lane,
                                                    uint64_t enc_bytes) {
    constexpr uint32_t kBytes = kSPL * W / 8;
    const uint64_t byte_off = tile_byte + uint64_t(kBytes) * lane;
    if (tile_byte + uint64_t(kBytes) * kWave <= enc_bytes) {  // wave-uniform
        if (W == 16) {
#if DINT_STREAM_LOADS_NT
            const u32x2 r = __builtin_nontemporal_load(&reinterpret_cast<const u32x2_a1*>(enc + byte_off)->v);
#else
            const u32x2 r = reinterpret_cast<const u32x2_a1*>(enc + byte_off)->v;
#endif
            return (uint64_t(r.y) << 32) | r.x;
        }
#if DINT_STREAM_LOADS_NT
        return __builtin_nontemporal_load(&reinterpret_cast<const u32_a1*>(enc + byte_off)->v);
#else
        return reinterpret_cast<const u32_a1*>(enc + byte_off)->v;
#endif
    }
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

// Slot classification table. Whether a slot is a codeword header or an exception
// payload depends on its predecessors; per lane (4 consecutive slots) the outcome is a
// function of how many payload slots the previous lane still owes (st_in) and of the
// digits d_k = 2 - min(slot_k, 2) (0: ordinary, 1: value 1, 2: value 0). One u16 row per
// (st_in, d3 d2 d1 d0 in base 3):
//   bits 0-3 payload slots | 4-7 exception headers | 8-10 st_out |
//   bit 11, bits 12-13, bits 14-15: headers before slot 1, 2, 3 (ordinal inside the lane)
// Rows [0, 243) are for 16-bit slots (payloads of 1 / 2 slots), rows [243, 648) for
// 8-bit slots (2 / 4 slots).
constexpr uint32_t kRows16 = 3 * 81, kRows8 = 5 * 81;
constexpr uint32_t kClassRows = kRows16 + kRows8;

__device__ __forceinline__ uint32_t class_row(bool w16, uint32_t st, uint32_t digits) {
    uint32_t pay = 0, exc = 0, hdr_before = 0, ords = 0;
    for (uint32_t k = 0; k != 4; ++k) {
        const uint32_t d = digits % 3;
        digits /= 3;
        if (k == 1) ords |= hdr_before << 11;
        if (k == 2) ords |= hdr_before << 12;
        if (k == 3) ords |= hdr_before << 14;
        const bool p = st != 0;
        const bool e = !p && d != 0;
        pay |= uint32_t(p) << k;
        exc |= uint32_t(e) << k;
        hdr_before += p ? 0u : 1u;
        st = p ? st - 1 : (e ? (w16 ? (d == 1 ? 2u : 1u) : (d == 1 ? 4u : 2u)) : 0u);
    }
    return pay | (exc << 4) | (st << 8) | ords;
}

__device__ __forceinline__ void build_class_table(uint16_t* table) {
    for (uint32_t i = threadIdx.x; i < kClassRows; i += kBlockThreads) {
        const bool w16 = i < kRows16;
        const uint32_t j = w16 ? i : i - kRows16;
        table[i] = uint16_t(class_row(w16, j / 81, j % 81));
    }
}

// Section marks for tools/isa_count.py (comments in the assembly under -DDINT_MARKS; nothing otherwise).
#ifdef DINT_MARKS
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name) do {} while (0)
#endif

// Segment chaining. Multi-dictionary units: a block's bytes are known only when the previous block
// has been parsed, so a block on its own pays the full memory latency of its selector and its slots
// before it can start. Single-dictionary units: nine in ten posting lists are shorter than one tile
// (Gov2-shaped lengths), each is its own unit, and a wave knows its next unit while it decodes the
// current one. Chained, the first kChainBytes of the next block (16 per lane, from the byte
// after its selector on) and its selector are requested as soon as the current block's end is known
// — after the scans, before its expansion and stores — and are re-laid-out lane to lane
// (ds_bpermute) into the first two tiles of the next segment.
constexpr uint32_t kChainBytes = 16 * kWave;
struct chain_io {
    u32x4 data;     // bytes [16 * lane, 16 * lane + 16) of the segment's slot stream
    uint32_t sel;   // the byte before them (the block's selector) in bits 0-7
    uint64_t next_off;  // in: where the next segment's selector byte is, or ~0: right after this segment
    bool more;      // in: a segment follows this one
    bool valid;     // out: data / sel hold the next segment's bytes
};

__device__ __forceinline__ void chain_request(const uint8_t* enc, uint64_t selector_byte, uint32_t lane, chain_io& ch) {
    ch.sel = enc[selector_byte];
    ch.data = reinterpret_cast<const u32x4_a1*>(enc + selector_byte + 1 + 16u * lane)->v;
    ch.valid = true;
}

// slots of tile t (0 or 1) of a chained segment, in the lane layout load_lane_slots produces
template <int W>
__device__ __forceinline__ uint64_t chain_tile(const u32x4& d, uint32_t t, uint32_t lane) {
    if (W == 16) {  // lane l: bytes [512 t + 8 l, + 8) = half (l & 1) of lane 32 t + l / 2
        const int src = int(32 * t + (lane >> 1));
        const uint32_t x = __shfl(d.x, src), y = __shfl(d.y, src), z = __shfl(d.z, src), w = __shfl(d.w, src);
        const bool hi = (lane & 1u) != 0;
        return (uint64_t(hi ? w : y) << 32) | (hi ? z : x);
    }
    // lane l: bytes [256 t + 4 l, + 4) = dword (l & 3) of lane 16 t + l / 4
    const int src = int(16 * t + (lane >> 2));
    const uint32_t x = __shfl(d.x, src), y = __shfl(d.y, src), z = __shfl(d.z, src), w = __shfl(d.w, src);
    const uint32_t c = lane & 3u;
    return c == 0 ? x : c == 1 ? y : c == 2 ? z : w;
}

// What a tile's front end hands to its expansion, per lane (slot k = 0..3 of the lane) and per wave.
struct tile_slots {
    uint32_t live[kSPL];    // all ones: a codeword header inside the segment
    uint32_t off4[kSPL];    // byte offset of its first output behind the lane's first
    uint32_t lord[kSPL];    // its ordinal behind the lane's first codeword
    uint32_t src4[kSPL];    // source byte address (LDS; cold: table offset + kColdBase4)
    uint32_t pk[kSPL];      // staging cells it needs | 1 << 16 if they are fetched from the table
    uint32_t cpre[kSPL];    // prefix of pk inside the lane
    uint32_t lit[kSPL];     // all ones: exception header (its value is in excval)
    uint32_t excval[kSPL];
    uint32_t lsum, obase;   // outputs of the lane's live codewords, position of the first
    uint32_t rbase, nlive;  // ordinal of its first live codeword, how many it has
    uint32_t cl, qb, wb;    // pk sum of the lane; first cell / first fetch of the lane
    uint32_t total;         // outputs of the tile (wave-uniform)
    uint32_t plus_one;      // add one to every output (wave-uniform; the in-index freqs parts)
    bool tile_exc, tile_staged;  // some lane holds an exception / needs staging (wave-uniform)
};

// Steps 3 and 4 of a tile: batches of at most ROUNDS x GROUPS x 256 outputs and kStageQuads staging
// cells (normally one: the whole tile) — cold fetch, flag/delta tables, staging, expansion, stores.
// `before_stores` runs once before the first store is issued: the caller parks there the waits for
// everything it has prefetched (see decode_segment).
template <uint32_t ROUNDS, uint32_t GROUPS, class BeforeStores>
__device__ __forceinline__ void expand_tile(const tile_slots& t, uint32_t out_int0, const uint32_t* lds, uint32_t* scratch,
                                            const __amdgpu_buffer_rsrc_t rs_table, const __amdgpu_buffer_rsrc_t rs_out,
                                            uint32_t* const out, uint32_t lane, BeforeStores&& before_stores) {
    constexpr uint32_t kCap = ROUNDS * GROUPS * 256;
    static_assert(kCap <= kMaxCap, "the flag bitmap holds 2048 positions");
    // per-wave scratch (byte offsets): {flag word, rank base} pairs | delta table | staging cells
    uint8_t* const fw = reinterpret_cast<uint8_t*>(scratch);                  // 64 pairs of 8 bytes (+1 spare)
    uint8_t* const delta = reinterpret_cast<uint8_t*>(scratch + kFwWords);    // 256 entries + 4 dummies
    uint8_t* const stage = reinterpret_cast<uint8_t*>(scratch + kFwWords + kDeltaWords);  // 128 cells of 16 bytes
    const uint8_t* const lds_bytes = reinterpret_cast<const uint8_t*>(lds);
    const uint32_t stage_off = uint32_t(stage - lds_bytes);                   // the cells as gather sources
    // lane constants of the expansion: this lane owns outputs 4*lane .. 4*lane+3 of every group
    const uint32_t sh = (4 * lane) & 31u;                 // bit position of its nibble in its flag word
    const uint32_t pair_byte = (lane >> 3) * 8;           // its {flag, base} pair inside a group's 8 pairs
    (void)out;
    MARK("4_batch_select");
    // ---- 3./4. batches of <= kCap outputs and <= kStageQuads cells (normally one: the whole tile)
    // (a do-while: the compiler must see that the wait inside precedes the register rotation
    // below on every path, or it waits again there — after the stores, for their acknowledgements)
    uint32_t done = 0, rdone = 0, qdone = 0, wdone = 0;
    do {
        const bool inb = t.lsum != 0 && t.obase >= done && (t.obase + t.lsum - done) <= kCap &&
                         (t.qb + (t.cl & 0xFFFFu) - qdone) <= kStageQuads &&
                         (kStageQuads <= 2 * kWave || (t.wb + (t.cl >> 16) - wdone) <= 2 * kWave);  // two fetches per lane
        const uint64_t bm = __ballot(inb);
        const uint32_t last = 63u - uint32_t(__builtin_clzll(bm | 1ull));
        const uint32_t bend = readlane(t.obase + t.lsum, last);
        const uint32_t rend = readlane(t.rbase + t.nlive, last);
        const uint32_t qend = readlane(t.qb + (t.cl & 0xFFFFu), last);
        const uint32_t wend = readlane(t.wb + (t.cl >> 16), last);
        if (bend <= done) {  // (malformed input: nothing decodable left in this tile)
            before_stores();  // every way out of the loop passes the caller's wait point
            break;
        }
        const uint32_t bt = bend - done;        // outputs in this batch, 1..kCap
        const uint32_t nfetch = wend - wdone;   // cold codewords to fetch, <= kStageQuads
        const uint32_t inbM = inb ? ~0u : 0u;

        MARK("5_worklist");
        // (a) worklist of the cold codewords {table byte offset, cell | quads << 16}, then each
        // lane takes up to two of them and fetches their first two quads (sizes 1..8); the
        // fetches fly while the batch tables are built
        uint32_t srcb[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) srcb[k] = t.src4[k];
        // (deliberately uninitialised: each is written and read under the same lane predicate)
        u32x2 e0, e1;
        u32x4 q00, q01, q10, q11;
        if (t.tile_staged) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t lv = t.live[k] & inbM;
                const uint32_t cell = t.qb - qdone + (t.cpre[k] & 0xFFFFu);
                if (lv != 0 && (t.pk[k] & 0xFFFFu) != 0) srcb[k] = stage_off + 16u * cell;
                // every slot writes an entry: the ones with nothing to fetch park it past the list
                const bool fetch = lv != 0 && (t.pk[k] >> 16) != 0;
                const u32x2 e = {t.src4[k] - kColdBase4, cell | (t.pk[k] << 16)};
                *reinterpret_cast<u32x2*>(stage + 8u * (fetch ? t.wb - wdone + (t.cpre[k] >> 16) : kStageQuads + k)) = e;
            }
            wave_lds_fence();
            if (lane < nfetch) e0 = *reinterpret_cast<const u32x2*>(stage + 8u * lane);
            if (nfetch > 64u && lane + 64u < nfetch) e1 = *reinterpret_cast<const u32x2*>(stage + 8u * (lane + 64u));
            wave_lds_fence();
#ifndef DINT_EXP_NOFETCH  // timing experiment: no cold payload reads (results are wrong)
            if (lane < nfetch) {
                q00 = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e0.x, 0, 0);
                if (((e0.y >> 16) & 7u) > 1u) q01 = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e0.x + 16u, 0, 0);
            }
            if (nfetch > 64u && lane + 64u < nfetch) {
                q10 = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e1.x, 0, 0);
                if (((e1.y >> 16) & 7u) > 1u) q11 = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e1.x + 16u, 0, 0);
            }
#endif
        }

        MARK("6_flags");
        // (b) flags and deltas. Every slot runs the same instructions: a codeword that is not
        // live in this batch ORs a zero into an in-range flag word and parks its delta in a dummy.
        *reinterpret_cast<uint32_t*>(fw + 8 * lane) = 0;  // clear all 64 flag words
        wave_lds_fence();
        const uint32_t rel0 = t.obase - done, ord0 = t.rbase - rdone;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            const uint32_t lv = t.live[k] & inbM;
            const uint32_t rel = rel0 + (t.off4[k] >> 2);
            uint32_t* const fword = reinterpret_cast<uint32_t*>(fw + ((rel >> 2) & 0x1F8u));
            __hip_atomic_fetch_or(fword, lv & (1u << (rel & 31u)), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint32_t ord = (lv & (ord0 + t.lord[k])) | (~lv & (kTileSlots + k));
            *reinterpret_cast<uint32_t*>(delta + 4 * ord) = srcb[k] - 4 * rel;
        }
        wave_lds_fence();
        {
            uint32_t* const pair = reinterpret_cast<uint32_t*>(fw + 8 * lane);
            const uint32_t pc = uint32_t(__builtin_popcount(pair[0]));
            const uint32_t pi = wave_inclusive_sum(pc);
            pair[1] = pi - pc - 1u;  // flags before this word, minus one
        }
        MARK("7_stage_write");

        // (c) the fetched quads and the exception literals go into their cells
        if (t.tile_staged) {
            if (lane < nfetch) {
                uint8_t* const c = stage + 16u * (e0.y & 0xFFFFu);
                *reinterpret_cast<u32x4*>(c) = q00;
                if (((e0.y >> 16) & 7u) > 1u) *reinterpret_cast<u32x4*>(c + 16) = q01;
            }
            if (nfetch > 64u && lane + 64u < nfetch) {
                uint8_t* const c = stage + 16u * (e1.y & 0xFFFFu);
                *reinterpret_cast<u32x4*>(c) = q10;
                if (((e1.y >> 16) & 7u) > 1u) *reinterpret_cast<u32x4*>(c + 16) = q11;
            }
            // size-16 cold codewords (rare): quads 2 and 3, fetched and waited for on the spot
            const bool big0 = lane < nfetch && ((e0.y >> 16) & 7u) > 2u;
            const bool big1 = nfetch > 64u && lane + 64u < nfetch && ((e1.y >> 16) & 7u) > 2u;
#ifdef DINT_EXP_NOFETCH
            if (false) {
#else
            if (__ballot(big0 || big1) != 0) {
#endif
                if (big0) {
                    uint8_t* const c = stage + 16u * (e0.y & 0xFFFFu);
                    *reinterpret_cast<u32x4*>(c + 32) = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e0.x + 32u, 0, 0);
                    if (((e0.y >> 16) & 7u) > 3u)
                        *reinterpret_cast<u32x4*>(c + 48) = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e0.x + 48u, 0, 0);
                }
                if (big1) {
                    uint8_t* const c = stage + 16u * (e1.y & 0xFFFFu);
                    *reinterpret_cast<u32x4*>(c + 32) = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e1.x + 32u, 0, 0);
                    if (((e1.y >> 16) & 7u) > 3u)
                        *reinterpret_cast<u32x4*>(c + 48) = __builtin_amdgcn_raw_buffer_load_b128(rs_table, e1.x + 48u, 0, 0);
                }
            }
            if (t.tile_exc) {
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k)
                    if ((t.live[k] & inbM & t.lit[k]) != 0)
                        *reinterpret_cast<uint32_t*>(stage + 16u * (t.qb - qdone + (t.cpre[k] & 0xFFFFu))) = t.excval[k];
            }
        }
        wave_lds_fence();

        // the wait point of the tile: the staged data is in, nothing has been stored yet. The caller
        // parks the waits for its own prefetches here and issues the next ones right behind them.
        // (the fetch registers too, on every path: a load the compiler cannot prove consumed would make it
        // wait wherever that register is next written — after the stores)
        asm volatile("" : "+v"(q00), "+v"(q01), "+v"(q10), "+v"(q11));
        before_stores();
        MARK("8_expand");
        // (d) expansion, GROUPS * 256 outputs per round: each lane takes 4 consecutive outputs of
        // every 256-output group; every source is an LDS byte address by now. Stores are whole
        // 16-byte quads: the descriptor clips what lies past the segment's n integers (range
        // checking is per dword), and what a quad writes past this batch's end inside the segment
        // is rewritten by the batches and tiles that follow (same wave, program order).
#pragma unroll
        for (uint32_t rd = 0; rd != ROUNDS; ++rd) {
            if (rd * GROUPS * 4 * kWave < bt) {  // wave-uniform
#ifdef DINT_EXP_STORE_LOCAL  // timing experiment: same store instructions, all into the segment's first 4 KB
                const uint32_t obyte = 0;
#else
                const uint32_t obyte = 4 * (out_int0 + done) + rd * GROUPS * 16 * kWave;  // output byte offset of the round
#endif
                uint32_t x[GROUPS][4];
#pragma unroll
                for (uint32_t g = 0; g != GROUPS; ++g) {
                    if ((rd * GROUPS + g) * 4 * kWave < bt) {  // wave-uniform
                        const u32x2 pr = *reinterpret_cast<const u32x2*>(fw + (rd * GROUPS + g) * 64 + pair_byte);
                        const uint32_t w = pr.x;
                        const uint32_t base = pr.y + uint32_t(__builtin_popcount(__builtin_amdgcn_ubfe(w, 0u, sh)));  // flags below its nibble
                        const uint32_t nib = w >> sh;
                        uint32_t r[4];
                        r[0] = base + (nib & 1u);
                        r[1] = base + uint32_t(__builtin_popcount(nib & 3u));
                        r[2] = base + uint32_t(__builtin_popcount(nib & 7u));
                        r[3] = base + uint32_t(__builtin_popcount(nib & 15u));
                        const uint32_t pos4 = (rd * GROUPS + g) * 16 * kWave + 16 * lane;  // byte position in the batch
#pragma unroll
                        for (int k = 0; k != 4; ++k) {
                            const uint32_t ad = *reinterpret_cast<const uint32_t*>(delta + 4 * r[k]) + pos4 + 4 * k;
                            x[g][k] = *reinterpret_cast<const uint32_t*>(lds_bytes + ad);
                        }
                    }
                }
                MARK("9_stores");
                // the prefetched registers must have landed before the first store is issued
#pragma unroll
                for (uint32_t g = 0; g != GROUPS; ++g) {
                    const uint32_t p0 = (rd * GROUPS + g) * 4 * kWave + 4 * lane;
                    if ((rd * GROUPS + g) * 4 * kWave < bt) {  // wave-uniform
#ifdef DINT_EXP_NOSTORE
                        if (x[g][0] == 0xDEADBEEFu && x[g][1] == 0x12345u) out[g] = x[g][2] + x[g][3];
#else
                        if (p0 < bt) {
                            u32x4 xv = {x[g][0], x[g][1], x[g][2], x[g][3]};
                            if (t.plus_one) xv += 1u;  // wave-uniform branch: nothing on the plain decode path
                            __builtin_amdgcn_raw_buffer_store_b128(xv, rs_out, 16 * g * kWave + 16 * lane, obyte, DINT_STORE_AUX);
                        }
#endif
                    }
                }
            }
        }
        wave_lds_fence();
        done = bend;
        rdone = rend;
        qdone = qend;
        wdone = wend;
    } while (done < t.total);

}

// ROUNDS x GROUPS x 256 = outputs per expansion batch: 2 x 4 for the long single-dictionary
// segments; a multi-dictionary segment is one block of at most 256 integers, 1 x 1.
template <int W, uint32_t ROUNDS, uint32_t GROUPS, bool CHAINED>
__device__ __forceinline__ uint64_t decode_segment(const decode_args& a, const uint32_t* lds, const uint16_t* cls,
                                                   uint32_t* scratch, const dict_desc& dd, uint64_t in_off,
                                                   uint32_t n, uint32_t* const out, uint32_t lane, chain_io& ch) {
    constexpr uint32_t kSlotBytes = W / 8;
    constexpr uint32_t kTileBytes = kTileSlots * kSlotBytes;
    const uint16_t* const rows = cls + (W == 16 ? 0 : kRows16);

    const uint32_t hot_k = dd.hot_k;
    const __amdgpu_buffer_rsrc_t rs_meta =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.dict.gmeta), 0, int(a.dict.gmeta_words * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_table =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.dict.gtable), 0, int(a.dict.gtable_words * 4), 0x00020000);
    // hardware bounds: nothing past this segment's n integers can be written
    const uint64_t out_bits = reinterpret_cast<uint64_t>(out);
    uint32_t* const out_u = reinterpret_cast<uint32_t*>((uint64_t(uniform(uint32_t(out_bits >> 32))) << 32) |
                                                        uniform(uint32_t(out_bits)));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, int(uniform(n) * 4), 0x00020000);

    // Metadata of the four slots of a lane: LDS for the hot codewords (unconditional reads, all four
    // in flight together: cold lanes read word 0), then L2 for the cold ones under their exec mask.
    // Two address spaces, never a pointer select (that would become one slow flat load).
    auto load_metas = [&](tile_regs& t) {
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.m[k] = lds[t.s[k] < hot_k ? dd.hot_base + t.s[k] : 0u];
        asm volatile("" : "+v"(t.m[0]), "+v"(t.m[1]), "+v"(t.m[2]), "+v"(t.m[3]));  // keep the DS reads DS reads
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
#ifdef DINT_EXP_NOMETA  // timing experiment: no L2 metadata reads (results are wrong)
            if (t.s[k] >= hot_k)
                t.m[k] = (lds[dd.hot_base + 7u + (t.s[k] & 1023u)] & 0xFF000000u) | kColdBase | (t.s[k] & 0xFFFFu);
#else
            if (t.s[k] >= hot_k) t.m[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_meta, 4 * (dd.meta_base + t.s[k]), 0, 0);
#endif
        }
    };

    // pipeline: tile t in `cur` (slots + metadata), tile t+1 in `nxt`, tile t+2's slots in flight
    const uint64_t in_off_u = (uint64_t(uniform(uint32_t(in_off >> 32))) << 32) | uniform(uint32_t(in_off));
    uint64_t slot_byte = in_off_u;  // first byte of the tile whose slots are loaded next (wave-uniform)
    tile_regs cur, nxt;
    uint64_t raw1, raw2 = 0;
    if (CHAINED) {  // tiles 0 and 1 arrived with the previous block (or were requested by the caller)
        unpack_slots<W>(chain_tile<W>(ch.data, 0, lane), cur);
        raw1 = chain_tile<W>(ch.data, 1, lane);
        slot_byte += kTileBytes;  // the pipeline is one tile shorter: tile t+2's slots are requested in tile t
    } else {
        unpack_slots<W>(load_lane_slots<W>(a.enc, slot_byte, lane, a.enc_bytes), cur);
        slot_byte += kTileBytes;
        raw1 = load_lane_slots<W>(a.enc, slot_byte, lane, a.enc_bytes);
        slot_byte += kTileBytes;
        raw2 = load_lane_slots<W>(a.enc, slot_byte, lane, a.enc_bytes);
    }
    ch.valid = false;
    load_metas(cur);
    // Everything loaded so far has landed before the loop is entered: inside it, a wait may only
    // ever sit right before a tile's stores (see the prefetch note below), never after them.
    asm volatile("" ::"v"(raw1), "v"(raw2), "v"(cur.m[0]), "v"(cur.m[1]), "v"(cur.m[2]), "v"(cur.m[3]));

    uint32_t produced = 0;
    uint32_t carry = 0;            // payload slots an exception of the previous tile still owns
    uint64_t tile_base = in_off_u; // byte offset of slot 0 of the current tile (wave-uniform)
    uint32_t end_slot = 0;
    MARK("loop_top");

    while (produced < n) {
        const uint32_t next_lo = uint32_t(raw1);  // first slots of the next tile (exception spill)
        // The far prefetch — the slots of the tile after next, straight from HBM — goes out first: it has to
        // be back before this tile's stores (every wait is a wait for everything), so it gets the whole tile.
        // (Issued a third of a tile before the wait, it cost a fifth of the kernel time whenever the stream
        // was not cached: any collection whose stream outgrows the 256 MB memory-side cache. Issuing it
        // right after the previous wait instead would keep its register in flight across the loop's
        // back edge, where the compiler copies it — and waits, after the stores.)
        slot_byte += kTileBytes;
        const uint64_t raw3 = load_lane_slots<W>(a.enc, slot_byte, lane, a.enc_bytes);
#if DINT_EARLY_METAS
        // the next tile's metadata too (L2 for its cold codewords): same reasoning
        unpack_slots<W>(raw1, nxt);
        load_metas(nxt);
#endif

        // Perturbation experiments (timing only): which resource does the kernel sit on? Pad every tile
        // with N independent instructions of one class and watch the time.
#ifdef DINT_EXP_PAD_VALU
        {
            uint32_t pad = lane;
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_VALU; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad));
            asm volatile("" ::"v"(pad));
        }
#endif
#ifdef DINT_EXP_PAD_SALU
        {
            uint32_t pad = 0;
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pad));
            asm volatile("" ::"s"(pad));
        }
#endif
#ifdef DINT_EXP_PAD_LDS
        {
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_LDS; ++i) {
                uint32_t v = lds[(lane * 33u + i * 67u) & 1023u];
                asm volatile("" : "+v"(v));
                acc += v;
            }
            asm volatile("" ::"v"(acc));
        }
#endif
#ifdef DINT_EXP_PAD_LDS_IND  // independent reads: LDS throughput, hardly any latency
        {
            uint32_t v[DINT_EXP_PAD_LDS_IND];
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_LDS_IND; ++i) v[i] = lds[(lane * 33u + i * 67u) & 1023u];
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_LDS_IND; ++i) { asm volatile("" : "+v"(v[i])); acc |= v[i]; }
            asm volatile("" ::"v"(acc));
        }
#endif
#ifdef DINT_EXP_PAD_LDS_IND128
        {
            u32x4 v[DINT_EXP_PAD_LDS_IND128];
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_LDS_IND128; ++i)
                v[i] = *reinterpret_cast<const u32x4*>(lds + ((lane * 4u + i * 260u) & 4095u));
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i != DINT_EXP_PAD_LDS_IND128; ++i) { asm volatile("" : "+v"(v[i])); acc |= v[i].x ^ v[i].w; }
            asm volatile("" ::"v"(acc));
        }
#endif
        MARK("1_classify");
        // ---- 1. classification: table lookup, repeated until the lane-to-lane carries agree ----
        uint32_t smin = cur.s[0];
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) smin = smin < cur.s[k] ? smin : cur.s[k];
        const bool special = __ballot(smin < 2) != 0 || carry != 0;
        uint32_t paybits = 0, excbits = 0, row = 0;
        uint32_t carry_out = 0;
        uint32_t excval[kSPL];  // set and read only when tile_exc
        bool tile_exc = false;
        if (special) {
            // base-3 digits of the four slots: 2 - min(slot, 2)
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t k = kSPL; k-- != 0;) lo = 3 * lo + (2u - (cur.s[k] < 2 ? cur.s[k] : 2u));
            uint32_t st_in = lane == 0 ? carry : 0u;
            for (;;) {
                row = rows[st_in * 81 + lo];
                uint32_t prev = __shfl_up((row >> 8) & 7u, 1);
                if (lane == 0) prev = carry;
                if (__ballot(prev != st_in) == 0) break;
                st_in = prev;
            }
            paybits = row & 15u;
            excbits = (row >> 4) & 15u;
            carry_out = readlane((row >> 8) & 7u, 63);
            tile_exc = __ballot(excbits != 0) != 0;
            if (tile_exc) {
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

        MARK("2_sizes");
        // ---- 2. sizes, offsets, ordinals (sizes and sources in BYTES of output / payload) --------
        // live[k]: all ones when slot k is a codeword header that belongs to this segment
        uint32_t sz4[kSPL], src4[kSPL], live[kSPL], lord[kSPL], lit[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            const uint32_t m = cur.m[k];
            sz4[k] = ((m >> 22) & 0x3FCu) + 4u;
            src4[k] = (m << 2) & 0x3FFFFFCu;  // cold metas carry kColdBase in their offset field
            live[k] = ~0u;
            lord[k] = k;
            lit[k] = 0;
        }
        uint32_t hdrcnt = 4;
        if (special) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t payM = uint32_t(int32_t(row << (31 - k)) >> 31);  // all ones: payload slot
                const uint32_t excM = uint32_t(int32_t(row << (27 - k)) >> 31);  // all ones: exception header
                sz4[k] = ((sz4[k] & ~excM) | (4u & excM)) & ~payM;
                lit[k] = excM;
                live[k] = ~payM;
            }
            lord[1] = (row >> 11) & 1u;
            lord[2] = (row >> 12) & 3u;
            lord[3] = (row >> 14) & 3u;
            hdrcnt = 4 - uint32_t(__builtin_popcount(paybits));
        }
        uint32_t off4[kSPL];
        off4[0] = 0;
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) off4[k] = off4[k - 1] + sz4[k - 1];
        uint32_t lsum = (off4[kSPL - 1] + sz4[kSPL - 1]) >> 2;
        const uint32_t packed = (hdrcnt << 24) | lsum;
        const uint32_t pincl = wave_inclusive_sum(packed);
        const uint32_t pexcl = pincl - packed;
        const uint32_t obase = pexcl & 0xFFFFFFu;  // first output of this lane's codewords
        const uint32_t rbase = pexcl >> 24;        // ordinal of this lane's first codeword
        const uint32_t remaining = n - produced;
        uint32_t total = readlane(pincl, 63) & 0xFFFFFFu;
        uint32_t nlive = hdrcnt;
        const bool last_tile = total >= remaining;
        if (last_tile) {  // last tile of the segment: clamp, and find where the stream ends
            total = remaining;
            uint32_t cand = 0;
            nlive = 0;
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t pos = obase + (off4[k] >> 2);
                const bool act = live[k] != 0 && pos < remaining;
                if (act) {
                    const uint32_t room4 = 4 * (remaining - pos);
                    sz4[k] = sz4[k] < room4 ? sz4[k] : room4;
                    const bool exc = (excbits >> k) & 1u;
                    cand = kSPL * lane + k + 1 + (exc ? (W == 16 ? cur.s[k] + 1 : 2 * cur.s[k] + 2) : 0u);
                    ++nlive;
                }
                live[k] = act ? ~0u : 0u;
            }
            lsum = obase < remaining ? (obase + lsum < remaining ? lsum : remaining - obase) : 0u;
            const uint64_t am = __ballot(cand != 0);
            end_slot = readlane(cand, 63u - uint32_t(__builtin_clzll(am | 1ull)));
        }
        // staging demand of each slot: 16-byte cells | (1 << 16 if they are fetched from the table).
        // Cold codewords (meta from L2: kColdBase set) take ceil(size / 4) cells, exception literals
        // one; hot codewords and runs (zero region) none.
        uint32_t pk[kSPL], cpre[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            const uint32_t coldM = live[k] & ~lit[k] & (0u - ((src4[k] >> 25) & 1u));
            pk[k] = (coldM & ((1u << 16) | ((sz4[k] + 12u) >> 4))) | (live[k] & lit[k] & 1u);
        }
        cpre[0] = 0;
#pragma unroll
        for (uint32_t k = 1; k != kSPL; ++k) cpre[k] = cpre[k - 1] + pk[k - 1];
        const uint32_t cl = cpre[kSPL - 1] + pk[kSPL - 1];
        const bool tile_staged = __ballot(cl != 0) != 0;
        uint32_t cexcl = 0;
        if (tile_staged) cexcl = wave_inclusive_sum(cl) - cl;
        const uint32_t qb = cexcl & 0xFFFFu, wb = cexcl >> 16;  // first cell / first fetch of this lane
        MARK("3_prefetch");
        // ---- prefetch: metadata of tile t+1 (its slots are already here), slots of tile t+2. Issued
        // before this tile's cold fetches and stores; waited for together with the fetches, right
        // before the stores (vmcnt is one in-order counter for loads AND stores on gfx950: a wait
        // placed after the stores would also wait for their acknowledgements).
        // The last tile of a segment has no successor to prefetch; a chained one asks for the next block.
        if (!last_tile) {
#if !DINT_EARLY_METAS
            unpack_slots<W>(raw1, nxt);
            load_metas(nxt);
#endif
        } else {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) nxt.s[k] = nxt.m[k] = 0;
            if (CHAINED) {
                const uint64_t nb = ch.next_off != ~0ull ? ch.next_off : tile_base + uint64_t(kSlotBytes) * end_slot;
                if (ch.more && nb + 1 + kChainBytes <= a.enc_bytes) chain_request(a.enc, nb, lane, ch);
            }
        }
        tile_slots t;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            t.live[k] = live[k];
            t.off4[k] = off4[k];
            t.lord[k] = lord[k];
            t.src4[k] = src4[k];
            t.pk[k] = pk[k];
            t.cpre[k] = cpre[k];
            t.lit[k] = lit[k];
            t.excval[k] = excval[k];
        }
        t.lsum = lsum, t.obase = obase, t.rbase = rbase, t.nlive = nlive, t.cl = cl, t.qb = qb, t.wb = wb;
        t.total = total, t.tile_exc = tile_exc, t.tile_staged = tile_staged;
        t.plus_one = a.plus_one;
        // the prefetched registers must have landed before the first store is issued
        // ("+v": from here on the values are the asm's, not a load's — nothing for the compiler to wait for later)
        uint64_t raw3w = raw3;
        expand_tile<ROUNDS, GROUPS>(t, produced, lds, scratch, rs_table, rs_out, out, lane, [&]() {
            asm volatile("" : "+v"(raw3w), "+v"(nxt.m[0]), "+v"(nxt.m[1]), "+v"(nxt.m[2]), "+v"(nxt.m[3]));
            if (CHAINED) asm volatile("" : "+v"(ch.sel), "+v"(ch.data.x), "+v"(ch.data.y), "+v"(ch.data.z), "+v"(ch.data.w));
        });

        MARK("10_rotate");
        produced += total;
        carry = carry_out;
        if (produced < n) tile_base += kTileBytes;

        // ---- rotate the pipeline ---------------------------------------------------
        cur = nxt;
        raw1 = CHAINED ? raw3w : raw2;
        raw2 = raw3w;
    }
    MARK("epilogue");
    return tile_base + uint64_t(kSlotBytes) * end_slot;
}

// A single-dictionary unit (rectangular or packed: the streams are byte-identical,
// only the dictionary source layout differed on the host) is one 16-bit segment.
__device__ __forceinline__ void decode_unit_single(const decode_args& a, const uint32_t* lds, const uint16_t* cls,
                                                   uint32_t* scratch, uint64_t unit_index, uint32_t lane, chain_io& ch,
                                                   uint64_t next_in_off) {
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || out_off + n > a.out_capacity || (a.only_full && n != 256)) {
        ch.valid = false;
        return;
    }
    const uint64_t in_off = (uint64_t(uniform(uint32_t(up->in_off >> 32))) << 32) | uniform(uint32_t(up->in_off));
    // chained unless the unit sits in the last kChainBytes of the buffer: its first two tiles were
    // requested by this wave's previous unit (or are requested here), and it requests the next unit's
    const bool chained = DINT_UNIT_CHAIN && in_off >= 1 && in_off + kChainBytes <= a.enc_bytes;
    ch.more = next_in_off != ~0ull && next_in_off >= 1;
    ch.next_off = next_in_off - 1;  // chain_request reads the byte before the data as a "selector"
    uint64_t end;
    if (chained) {
        if (!ch.valid) chain_request(a.enc, in_off - 1, lane, ch);
        end = decode_segment<16, kRounds, kGroups, true>(a, lds, cls, scratch, a.dict.first, in_off, n, a.out + out_off, lane, ch);
    } else {
        end = decode_segment<16, kRounds, kGroups, false>(a, lds, cls, scratch, a.dict.first, in_off, n, a.out + out_off, lane, ch);
    }
    if (a.end_off && lane == 0) a.end_off[unit_index] = end;
}

// ---- bundles of tiny units -----------------------------------------------------------------------
// Nine in ten posting lists of a Gov2-shaped collection hold at most 64 postings; as units of
// their own they are 0.2 % of the integers and a fifth of the tiles (each wave tile-step costs the
// same whether 5 or 250 of its slot positions are used). A bundle packs up to 64 consecutive tiny
// units into ONE tile: unit i takes ceil(bytes_i / 8) whole lanes, the front end runs segmented
// (per-lane slot address, carries cut at unit starts, sizes clamped at each unit's n), and because
// the units' outputs are consecutive the expansion is the ordinary one over the bundle's outputs.
constexpr uint32_t kBundleMaxInts = 256;   // a unit is bundled only if it decodes to at most this many integers
constexpr uint32_t kBundleMaxBytes = 256;  // ... and spans at most this many stream bytes (32 lanes)

// Host-launched before the decode kernel: sched[i] for every unit (see decode_args::sched). One
// workgroup per 256 units; bundles do not cross these blocks.
__global__ __launch_bounds__(256) void bundle_schedule_kernel(const dint_unit* units, const uint32_t* spans, uint64_t n_units,
                                                              const uint8_t* enc, uint64_t enc_bytes, uint64_t out_capacity,
                                                              uint32_t only_full, uint32_t multi, uint8_t* sched,
                                                              uint32_t* block_items) {
    __shared__ uint32_t lanes[256], pre[256];
    __shared__ uint8_t start[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t i = uint64_t(blockIdx.x) * 256 + tid;
    uint32_t L = 0;
    uint64_t in = 0, out = 0;
    uint32_t n = 0;
    if (i < n_units) {
        in = units[i].in_off;
        out = units[i].out_off;
        n = units[i].n;
        const uint64_t nxt = spans ? in + spans[i] : (i + 1 < n_units ? units[i + 1].in_off : enc_bytes);
        // (in-index launches decode only the full blocks; the other units stay on their own and are skipped)
        if (n >= 1 && n <= kBundleMaxInts && (!only_full || n == 256) && nxt > in && nxt - in <= kBundleMaxBytes &&
            nxt <= enc_bytes && out + n <= out_capacity) {
            if (!multi) {
                const uint32_t l = uint32_t((nxt - in + 7) >> 3);
                if (in + 8ull * l <= enc_bytes) L = l;  // every lane's 8-byte load stays inside the buffer
            } else {
                // a multi unit of <= 256 integers is one block: selector byte, then 16- or 8-bit slots
                // (4 to a lane: 8 or 4 bytes); every lane still loads 8 bytes
                const uint32_t sel = enc[in];
                const uint32_t stride = sel >= 6 ? 4u : 8u;
                const uint32_t l = uint32_t((nxt - in - 1 + stride - 1) / stride);
                if (sel < 12 && l >= 1 && l <= 32 && in + 1 + uint64_t(stride) * l + 8 <= enc_bytes) L = l;
            }
        }
    }
    lanes[tid] = L;
    // does this unit continue the previous one (both eligible, outputs consecutive)?
    pre[tid] = (L != 0 && tid != 0 && i < n_units && out == units[i - 1].out_off + units[i - 1].n) ? 1u : 0u;
    __syncthreads();
    // greedy packing, one thread per block of 256 units: a bundle takes units while their lanes fit a wave
    // (four threads, 64 units each — a bundle does not cross these quarters; the chain through in_use is
    // the only serial part, the LDS reads are unrolled ahead of it)
    if ((tid & 63u) == 0) {
        uint32_t in_use = 0, members = 0, prev_l = 0;
#pragma unroll 16
        for (uint32_t j = tid; j != tid + 64; ++j) {
            const uint32_t l = lanes[j];
            const bool cont = l != 0 && prev_l != 0 && pre[j] != 0 && in_use + l <= kWave - 1 && members < kWave;
            start[j] = cont ? 0 : 1;
            in_use = cont ? in_use + l : l;
            members = cont ? members + 1 : 1;
            prev_l = l;
        }
    }
    __syncthreads();
    const bool st = start[tid] != 0;
    const int starts = __syncthreads_count(st && i < n_units);
    if (tid == 0) block_items[blockIdx.x] = uint32_t(starts);
    if (i >= n_units) return;
    uint32_t c = 0;
    if (st) {
        c = 1;
        if (L != 0)
            while (tid + c < 256 && i + c < n_units && !start[tid + c]) ++c;
    }
    sched[i] = uint8_t(c);
}

// Work items = the units with sched != 0. block_items -> exclusive offsets (one workgroup), then every
// block of 256 units writes its items.
__global__ __launch_bounds__(1024) void bundle_offsets_kernel(uint32_t* block_items, uint32_t n_blocks, uint32_t* n_items) {
    __shared__ uint32_t part[1024];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_items[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += x;
            __syncthreads();
        }
        if (i < n_blocks) block_items[i] = carry + part[threadIdx.x] - v;
        carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_items = carry;
}

__global__ __launch_bounds__(256) void bundle_items_kernel(const uint8_t* sched, uint64_t n_units, const uint32_t* block_offsets,
                                                           uint32_t* items) {
    __shared__ uint32_t pre[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t i = uint64_t(blockIdx.x) * 256 + tid;
    const uint32_t f = i < n_units && sched[i] != 0 ? 1u : 0u;
    pre[tid] = f;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = tid >= d ? pre[tid - d] : 0;
        __syncthreads();
        pre[tid] += v;
        __syncthreads();
    }
    if (f) items[block_offsets[blockIdx.x] + pre[tid] - 1] = uint32_t(i);
}

// One tile over `cnt` (2..64) consecutive tiny units starting at unit u0. MULTI: every unit is one block
// of a multi-dictionary stream — its selector byte picks the dictionary and the slot width, per unit,
// hence per lane.
template <bool MULTI>
__device__ __forceinline__ void decode_bundle(const decode_args& a, const uint32_t* lds, const uint16_t* cls, uint32_t* scratch,
                                              uint64_t u0, uint32_t cnt, uint32_t lane) {
    const __amdgpu_buffer_rsrc_t rs_meta =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.dict.gmeta), 0, int(a.dict.gmeta_words * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_table =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.dict.gtable), 0, int(a.dict.gtable_words * 4), 0x00020000);

    // ---- the units' descriptors, one per lane; lanes and outputs of each by one packed scan ----
    const bool has = lane < cnt;
    const dint_unit* up = a.units + u0 + (has ? lane : 0u);
    const uint64_t my_in = up->in_off;
    const uint32_t my_n = has ? up->n : 0u;
    uint32_t nxt_lo = __shfl_down(uint32_t(my_in), 1), nxt_hi = __shfl_down(uint32_t(my_in >> 32), 1);
    if (lane + 1 == cnt) {
        const uint64_t e = u0 + cnt < a.n_units ? a.units[u0 + cnt].in_off : a.enc_bytes;
        nxt_lo = uint32_t(e);
        nxt_hi = uint32_t(e >> 32);
    }
    uint64_t nxt_in = (uint64_t(nxt_hi) << 32) | nxt_lo;
    if (a.spans && has) nxt_in = my_in + a.spans[u0 + lane];
    // the unit's dictionary and slot width (MULTI: from its selector byte)
    uint32_t my_narrow = 0, my_hot_base = a.dict.first.hot_base, my_hot_k = a.dict.first.hot_k, my_meta_base = a.dict.first.meta_base;
    uint32_t my_lanes = has ? uint32_t((nxt_in - my_in + 7) >> 3) : 0u;  // 1..32 by the schedule's test
    if (MULTI) {
        const uint32_t sel = has ? uint32_t(a.enc[my_in]) : 0u;
        my_narrow = sel >= 6 ? 1u : 0u;
        const dict_desc* dp = a.dict.descs + (my_narrow ? sel - 6 : sel) % 6;
        my_hot_base = dp->hot_base;
        my_hot_k = dp->hot_k;
        my_meta_base = dp->meta_base;
        const uint32_t stride = my_narrow ? 4u : 8u;
        my_lanes = has ? uint32_t((nxt_in - my_in - 1 + stride - 1) / stride) : 0u;
    }
    const uint32_t pk0 = my_n | (my_lanes << 16);
    const uint32_t inc0 = wave_inclusive_sum(pk0);
    const uint32_t my_lane0 = (inc0 - pk0) >> 16;      // first lane of this lane's unit
    const uint32_t my_out0 = (inc0 - pk0) & 0xFFFFu;   // its first output, relative to the bundle
    const uint32_t used = readlane(inc0, 63) >> 16;    // lanes in use, < 64
    const uint32_t total = readlane(inc0, 63) & 0xFFFFu;
    const uint64_t out0 = a.units[u0].out_off;
    if (total == 0 || out0 + total > a.out_capacity) return;

    // ---- lane -> unit: heads scattered into LDS, prefix maximum; then the unit's parameters ----
    uint32_t* const map = scratch;  // the flag pairs' space, free until expand_tile
    map[lane] = 0;
    wave_lds_fence();
    if (has) map[my_lane0] = lane + 1;
    wave_lds_fence();
    const uint32_t seg = wave_inclusive_max(map[lane]) - 1u;  // lane 0 is always a head
    wave_lds_fence();
    const bool lane_used = lane < used;
    const int sl = int(seg);
    const uint64_t seg_in = (uint64_t(uint32_t(__shfl(uint32_t(my_in >> 32), sl))) << 32) | uint32_t(__shfl(uint32_t(my_in), sl));
    const uint32_t seg_n = __shfl(my_n, sl);
    const uint32_t seg_lane0 = __shfl(my_lane0, sl);
    const uint32_t seg_out0 = __shfl(my_out0, sl);
    const bool seg_head = lane == seg_lane0;
    const bool narrow = MULTI && __shfl(my_narrow, sl) != 0;
    const uint32_t hot_base = MULTI ? uint32_t(__shfl(my_hot_base, sl)) : my_hot_base;
    const uint32_t hot_k = MULTI ? uint32_t(__shfl(my_hot_k, sl)) : my_hot_k;
    const uint32_t meta_base = MULTI ? uint32_t(__shfl(my_meta_base, sl)) : my_meta_base;
    const uint64_t slot0 = seg_in + (MULTI ? 1u : 0u);          // the unit's first slot
    const uint32_t stride = narrow ? 4u : 8u;                    // stream bytes per lane

    // ---- slots and metadata ---------------------------------------------------------------------
    tile_regs cur;
    uint32_t raw_lo = 0;  // the lane's first four bytes (the next lane's: what an exception at its end spills into)
    {
        uint64_t raw = 0;
        if (lane_used) {
            const u32x2 r = reinterpret_cast<const u32x2_a1*>(a.enc + slot0 + stride * (lane - seg_lane0))->v;
            raw = (uint64_t(r.y) << 32) | r.x;
        }
        raw_lo = uint32_t(raw);
        unpack_slots<16>(raw, cur);
        if (MULTI && narrow) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) cur.s[k] = (raw_lo >> (8 * k)) & 0xFFu;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) cur.m[k] = lds[cur.s[k] < hot_k ? hot_base + cur.s[k] : 0u];
    asm volatile("" : "+v"(cur.m[0]), "+v"(cur.m[1]), "+v"(cur.m[2]), "+v"(cur.m[3]));
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k)
        if (cur.s[k] >= hot_k) cur.m[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_meta, 4 * (meta_base + cur.s[k]), 0, 0);

    // ---- classification: as in decode_segment, the carries cut at every unit's first lane ----------
    tile_slots t;
    uint32_t row;
    {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t k = kSPL; k-- != 0;) lo = 3 * lo + (2u - (cur.s[k] < 2 ? cur.s[k] : 2u));
        uint32_t st_in = 0;
        const uint32_t rows_at = narrow ? kRows16 : 0u;  // the 8-bit rows follow the 16-bit ones
        for (;;) {
            row = cls[rows_at + st_in * 81 + lo];
            uint32_t prev = __shfl_up((row >> 8) & 7u, 1);
            if (seg_head) prev = 0;
            if (__ballot(prev != st_in) == 0) break;
            st_in = prev;
        }
    }
    const uint32_t excbits = (row >> 4) & 15u;
    t.tile_exc = __ballot(lane_used && excbits != 0) != 0;
    if (t.tile_exc) {
        const uint32_t nlo = __shfl_down(raw_lo, 1);  // an exception's payload never leaves its unit's lanes
        uint32_t e[kSPL + 2];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) e[k] = cur.s[k];
        e[kSPL] = nlo & 0xFFFFu;
        e[kSPL + 1] = nlo >> 16;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.excval[k] = e[k] == 0 ? e[k + 1] : (e[k + 1] | (e[k + 2] << 16));
        if (MULTI && narrow) {  // 8-bit slots: the value is the next 2 or 4 of them
            uint32_t b[kSPL + 4];
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) b[k] = cur.s[k];
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) b[kSPL + k] = (nlo >> (8 * k)) & 0xFFu;
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t lo16 = b[k + 1] | (b[k + 2] << 8);
                t.excval[k] = b[k] == 0 ? lo16 : (lo16 | (b[k + 3] << 16) | (b[k + 4] << 24));
            }
        }
    }

    // ---- sizes; positions inside each unit (one scan + the value at the unit's first lane); clamp ----
    uint32_t sz4[kSPL];
    const uint32_t usedM = lane_used ? ~0u : 0u;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t m = cur.m[k];
        const uint32_t payM = uint32_t(int32_t(row << (31 - k)) >> 31);
        const uint32_t excM = uint32_t(int32_t(row << (27 - k)) >> 31);
        sz4[k] = ((((m >> 22) & 0x3FCu) + 4u) & ~excM) | (4u & excM);
        sz4[k] &= ~payM & usedM;
        t.src4[k] = (m << 2) & 0x3FFFFFCu;
        t.lit[k] = excM;
        t.live[k] = ~payM & usedM;
    }
    t.lord[0] = 0;
    t.lord[1] = (row >> 11) & 1u;
    t.lord[2] = (row >> 12) & 3u;
    t.lord[3] = (row >> 14) & 3u;
    t.off4[0] = 0;
#pragma unroll
    for (uint32_t k = 1; k != kSPL; ++k) t.off4[k] = t.off4[k - 1] + sz4[k - 1];
    const uint32_t raw_sum = (t.off4[kSPL - 1] + sz4[kSPL - 1]) >> 2;
    const uint32_t inc1 = wave_inclusive_sum(raw_sum);
    // (the read is unconditional: ds_bpermute returns nothing from lanes that do not take part)
    const uint32_t inc1_before = uint32_t(__shfl(inc1, int(seg_lane0 + 63u) & 63));
    const uint32_t before_seg = seg_lane0 == 0 ? 0u : inc1_before;
    const uint32_t p0 = inc1 - raw_sum - before_seg;  // position of the lane's first codeword inside its unit
    uint32_t nlive = 0, lsum = 0;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t pos = p0 + (t.off4[k] >> 2);
        const bool act = t.live[k] != 0 && pos < seg_n;
        const uint32_t room4 = 4 * (seg_n - pos);
        sz4[k] = act ? (sz4[k] < room4 ? sz4[k] : room4) : 0u;
        t.live[k] = act ? ~0u : 0u;
        nlive += act ? 1u : 0u;
        lsum += sz4[k] >> 2;
    }
    t.lsum = lsum;
    t.nlive = nlive;
    t.obase = seg_out0 + p0;
    t.rbase = wave_inclusive_sum(nlive) - nlive;
    t.total = total;
    t.plus_one = a.plus_one;

    // ---- staging demand, as in decode_segment ---------------------------------------------------------
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t coldM = t.live[k] & ~t.lit[k] & (0u - ((t.src4[k] >> 25) & 1u));
        t.pk[k] = (coldM & ((1u << 16) | ((sz4[k] + 12u) >> 4))) | (t.live[k] & t.lit[k] & 1u);
    }
    t.cpre[0] = 0;
#pragma unroll
    for (uint32_t k = 1; k != kSPL; ++k) t.cpre[k] = t.cpre[k - 1] + t.pk[k - 1];
    t.cl = t.cpre[kSPL - 1] + t.pk[kSPL - 1];
    t.tile_staged = __ballot(t.cl != 0) != 0;
    uint32_t cexcl = 0;
    if (t.tile_staged) cexcl = wave_inclusive_sum(t.cl) - t.cl;
    t.qb = cexcl & 0xFFFFu;
    t.wb = cexcl >> 16;

#ifdef DINT_DEBUG_BUNDLE
    if (u0 <= 3 && lane < 6)
        printf("u0 %llu cnt %u lane %u seg %u in %llu n %u lane0 %u out0 %u | s %u %u %u %u row %x live %x %x %x %x sz %u %u %u %u p0 %u obase %u lsum %u nlive %u rbase %u exc %u %u %u %u pk %x %x %x %x total %u used %u\n",
               (unsigned long long)u0, cnt, lane, seg, (unsigned long long)seg_in, seg_n, seg_lane0, seg_out0, cur.s[0], cur.s[1], cur.s[2], cur.s[3], row,
               t.live[0] & 1, t.live[1] & 1, t.live[2] & 1, t.live[3] & 1, sz4[0], sz4[1], sz4[2], sz4[3], p0, t.obase, t.lsum, t.nlive, t.rbase,
               t.excval[0], t.excval[1], t.excval[2], t.excval[3], t.pk[0], t.pk[1], t.pk[2], t.pk[3], total, used);
#endif
    // where each unit's stream ends: after its last live codeword (and that one's payload). That codeword
    // sits in the unit's highest lane that has a live one: the next such lane belongs to another unit.
    if (a.end_off) {
        uint32_t last_end = 0;  // slots from the lane's first to the end of its last live codeword
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if (t.live[k] != 0)
                last_end = k + 1 + (((excbits >> k) & 1u) ? (narrow ? 2 * cur.s[k] + 2u : cur.s[k] + 1u) : 0u);
        const uint64_t havers = __ballot(last_end != 0);
        const uint64_t above = lane == 63 ? 0ull : havers & ~((2ull << lane) - 1ull);
        const uint32_t next_lane = above ? uint32_t(__builtin_ctzll(above)) : lane;
        const uint32_t next_seg = uint32_t(__shfl(seg, int(next_lane)));  // (unconditional: every lane takes part)
        if (last_end != 0 && (above == 0 || next_seg != seg))
            a.end_off[u0 + seg] = slot0 + uint64_t(stride) * (lane - seg_lane0) + (narrow ? 1ull : 2ull) * last_end;
    }

    uint32_t* const out = a.out + out0;
    const uint64_t out_bits = reinterpret_cast<uint64_t>(out);
    uint32_t* const out_u = reinterpret_cast<uint32_t*>((uint64_t(uniform(uint32_t(out_bits >> 32))) << 32) |
                                                        uniform(uint32_t(out_bits)));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, int(total * 4), 0x00020000);
    expand_tile<kRounds, kGroups>(t, 0u, lds, scratch, rs_table, rs_out, out, lane, []() {});
}

// A multi-dictionary unit: blocks of 256 integers (the last one shorter), each opened
// by a selector byte: < 6 -> 16-bit codewords against dictionary `selector`, else 8-bit
// codewords against dictionary `selector - 6` (vroom_env/dint_codecs.hpp:521-619).
// Blocks carry no length, so they are decoded one after the other.
__device__ __forceinline__ void decode_unit_multi(const decode_args& a, const uint32_t* lds, const uint16_t* cls,
                                                  uint32_t* scratch, uint64_t unit_index, uint32_t lane) {
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || out_off + n > a.out_capacity || (a.only_full && n != 256)) return;
    uint64_t pos = up->in_off;
    chain_io ch{};
    for (uint32_t done = 0; done < n;) {
        const uint32_t bsize = n - done < 256u ? n - done : 256u;
        pos = (uint64_t(uniform(uint32_t(pos >> 32))) << 32) | uniform(uint32_t(pos));
        // chained unless the block sits in the last kChainBytes of the buffer
        const bool chained = pos + 1 + kChainBytes <= a.enc_bytes;
        if (chained && !ch.valid) chain_request(a.enc, pos, lane, ch);  // first block of the unit
        uint32_t sel;
        if (chained) {
            sel = uniform(ch.sel) & 0xFFu;
        } else {
            const uint64_t sp = pos < a.enc_bytes ? pos : a.enc_bytes - 1;
            sel = uniform(a.enc[sp]);
        }
        const bool narrow = sel >= 6;
        const uint32_t d = (narrow ? sel - 6 : sel) % 6;
        dict_desc dd;
        dd.meta_base = uniform(a.dict.descs[d].meta_base);
        dd.hot_base = uniform(a.dict.descs[d].hot_base);
        dd.hot_k = uniform(a.dict.descs[d].hot_k);
        dd.pad = 0;
        uint32_t* const out = a.out + out_off + done;
        ch.more = done + bsize < n;
        ch.next_off = ~0ull;  // the next block starts where this one ends
        if (chained) {
            if (narrow) pos = decode_segment<8, 1, 1, true>(a, lds, cls, scratch, dd, pos + 1, bsize, out, lane, ch);
            else pos = decode_segment<16, 1, 1, true>(a, lds, cls, scratch, dd, pos + 1, bsize, out, lane, ch);
        } else {
            if (narrow) pos = decode_segment<8, 1, 1, false>(a, lds, cls, scratch, dd, pos + 1, bsize, out, lane, ch);
            else pos = decode_segment<16, 1, 1, false>(a, lds, cls, scratch, dd, pos + 1, bsize, out, lane, ch);
        }
        done += bsize;
    }
    if (a.end_off && lane == 0) a.end_off[unit_index] = pos;
}

// Units are handed out dynamically: their cost varies a lot (a sparse list full
// of exceptions takes several times longer than a dense one of the same length),
// so a static unit -> wave map leaves most of the chip idle behind the slowest
// waves. kQueueShards counters, one per group of workgroups; a wave draws its next
// work item while it is still decoding the current one.
#ifdef DINT_EXP_FINISH
__device__ unsigned long long g_finish[8192];
#endif
template <bool MULTI>
__device__ __forceinline__ void decode_kernel_body(const decode_args& a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < a.dict.hot_words; i += kBlockThreads) lds[i] = a.dict.lds_image[i];
    uint16_t* const cls = reinterpret_cast<uint16_t*>(lds + a.dict.hot_words);
    build_class_table(cls);
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x / kWave);
    uint32_t* scratch = lds + a.dict.hot_words + kClassTableWords + wave * kScratchWords;
    // Work queue. Every shard — the workgroups with the same blockIdx % n_shards: one XCD under
    // round-robin placement — walks its own CONTIGUOUS part of the work items and, when that is done,
    // helps with the next shards' parts. Contiguous, because an XCD that strides over the whole
    // stream and output touches every 2 MB page of them, and past ~2 GB the translations no longer stay
    // in its TLB (the time per integer rose by a fifth); stealing, because equal counts of work items
    // are not equal work.
    const uint32_t shard = blockIdx.x % a.n_shards;  // n_shards = min(kQueueShards, gridDim.x)
    // with a schedule the queue hands out work items (bundle leaders and units on their own)
    const uint64_t n_work = a.sched ? uint64_t(uniform(*a.n_items)) : a.n_units;
    const uint64_t per_shard = (n_work + a.n_shards - 1) / a.n_shards;
    uint32_t cur = shard, tried = 0;
    auto draw = [&]() -> uint64_t {  // next work item, or ~0: nothing left anywhere
        while (tried < a.n_shards) {
            const uint64_t first = per_shard * cur;
            const uint64_t size = first >= n_work ? 0 : (n_work - first < per_shard ? n_work - first : per_shard);
            uint32_t j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add(a.queue + cur * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            j = uniform(j);
            if (j < size) return first + j;
            cur = cur + 1 == a.n_shards ? 0 : cur + 1;
            ++tried;
        }
        return ~0ull;
    };
    chain_io ch{};
    uint64_t w = draw();
    while (w != ~0ull) {
        const uint64_t w_next = draw();
        const uint64_t u = a.sched ? uint64_t(uniform(a.items[w])) : w;
        const uint32_t cnt = a.sched ? uint32_t(uniform(a.sched[u])) : 1u;
        if (MULTI) {
            if (cnt > 1) decode_bundle<true>(a, lds, cls, scratch, u, cnt, lane);
            else decode_unit_multi(a, lds, cls, scratch, u, lane);
        } else if (cnt > 1) {
            decode_bundle<false>(a, lds, cls, scratch, u, cnt, lane);
        } else {
            uint64_t next_in = ~0ull;
            if (DINT_UNIT_CHAIN && w_next != ~0ull) {
                const uint64_t un = a.sched ? uint64_t(uniform(a.items[w_next])) : w_next;
                if (!a.sched || uniform(a.sched[un]) == 1u) next_in = a.units[un].in_off;
            }
            decode_unit_single(a, lds, cls, scratch, u, lane, ch, next_in);
        }
        w = w_next;
    }
#ifdef DINT_EXP_FINISH  // diagnostic: when did this wave run out of work? (tail of the kernel)
    if (lane == 0) g_finish[blockIdx.x * kWavesPerBlock + wave] = __builtin_amdgcn_s_memrealtime();
#endif
}

#ifndef DINT_MIN_WAVES
#define DINT_MIN_WAVES 1
#endif
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_single_kernel(decode_args a) {
    decode_kernel_body<false>(a);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_kernel(decode_args a) {
    decode_kernel_body<true>(a);
}

// ---- in-index path: helper kernels -------------------------------------------------------------

// unit table for the docs parts (in_off from the block table) or for the freqs parts (in_off =
// where the docs part ended)
__global__ void blocks_to_units_kernel(const dint_block_ref* blocks, const uint64_t* docs_end, uint64_t n_blocks,
                                       uint64_t index_bytes, dint_unit* units, uint32_t* spans) {
    const uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    dint_unit u;
    u.in_off = docs_end ? docs_end[b] : blocks[b].in_off;
    if (spans) {  // a part ends no later than where the next block of the table begins (exact for freqs parts)
        const uint64_t nxt = b + 1 < n_blocks && blocks[b + 1].in_off > u.in_off ? blocks[b + 1].in_off : index_bytes;
        const uint64_t sp = nxt > u.in_off ? nxt - u.in_off : 0;
        spans[b] = sp > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(sp);
    }
    u.out_off = blocks[b].out_off;
    u.n = blocks[b].n;
    u.list = blocks[b].list;
    units[b] = u;
}

// Binary interpolative decode of the blocks shorter than 256 (include/ds2i/interpolative_coding.hpp:
// 79-146, include/ds2i/block_codecs.hpp:130-150), one block per thread: the code is bit-serial and
// recursive (here: an explicit stack, node - left subtree - right subtree order), and there is at
// most one such block per posting list. docs parts have sum_of_values = max - base - (n - 1),
// freqs parts carry their sum as a leading vbyte.
struct tail_bits {
    const uint8_t* p;
    uint64_t limit;  // bytes readable from p
    uint64_t byte;   // next byte to fetch
    uint64_t buf;
    uint32_t avail;
    uint64_t pos;    // bits consumed
    __device__ uint32_t read(uint32_t len) {
        if (!len) return 0;
        if (avail < len) {
            uint32_t w = 0;
            if (byte + 4 <= limit) {
                w = reinterpret_cast<const u32_a1*>(p + byte)->v;  // one unaligned load
            } else {
                for (uint32_t i = 0; i != 4; ++i)
                    if (byte + i < limit) w |= uint32_t(p[byte + i]) << (8 * i);
            }
            byte += 4;
            buf |= uint64_t(w) << avail;
            avail += 32;
        }
        const uint32_t v = uint32_t(buf & ((uint64_t(1) << len) - 1));
        buf >>= len;
        avail -= len;
        pos += len;
        return v;
    }
    __device__ uint32_t read_int(uint32_t u) {
        const uint32_t b = 31u - uint32_t(__builtin_clz(u));
        const uint64_t m = (uint64_t(1) << (b + 1)) - u;
        uint32_t v = read(b);
        if (v >= m) v = (v << 1) + read(1) - uint32_t(m);
        return v;
    }
};

// The short blocks are one in fifteen of a block table; collected first, so that the bit-serial decoder
// below runs with full wavefronts (scattered over the table, four active lanes per wave made every
// wave last as long as its slowest decode).
__global__ void collect_tails_kernel(const dint_block_ref* blocks, uint64_t n_blocks, uint32_t* tails, uint32_t* n_tails) {
    const uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t n = blocks[b].n;
    if (n != 0 && n < 256) tails[atomicAdd(n_tails, 1u)] = uint32_t(b);
}

// One wavefront per 64 short blocks, one block per lane. The decoder's values and its explicit stack
// live in LDS (rows of odd stride: lane-private and conflict-free); the block is differenced there and
// the wave then copies the rows out together, coalesced — no pass over global memory but that one.
constexpr uint32_t kTailRow = 257;     // words per lane: up to 255 values
constexpr uint32_t kTailStack = 41;    // words per lane: 10 frames of 4 (depth <= log2(256) + 1)
constexpr uint32_t kTailLdsBytes = 64 * (kTailRow + kTailStack) * 4;

__global__ __launch_bounds__(64) void interpolative_tails_kernel(const uint8_t* index, uint64_t index_bytes,
                                                                 const dint_block_ref* blocks, const uint64_t* docs_end,
                                                                 const uint32_t* tails, const uint32_t* n_tails, uint32_t* out,
                                                                 uint64_t out_capacity, uint64_t* end_off, uint32_t plus_one) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tail_lds[];
    __shared__ uint32_t row_n[64];
    __shared__ uint64_t row_out[64];
    const uint32_t lane = threadIdx.x;
    if (uint64_t(blockIdx.x) * 64 >= *n_tails) return;  // (the grid is sized for the worst case)
    const uint64_t t = uint64_t(blockIdx.x) * 64 + lane;
    uint32_t* const o = tail_lds + lane * kTailRow;
    uint32_t* const stack = tail_lds + 64 * kTailRow + lane * kTailStack;
    uint32_t n = 0;
    uint64_t b = 0;
    if (t < *n_tails) {
        b = tails[t];
        n = blocks[b].n;
        if (n >= 256 || blocks[b].out_off + n > out_capacity) n = 0;
    }
    row_n[lane] = n;
    row_out[lane] = n ? blocks[b].out_off : 0;
    if (n != 0) {
        uint64_t pos = docs_end ? docs_end[b] : blocks[b].in_off;
        uint32_t sum;
        if (docs_end) {  // freqs: sum_of_values = -1 -> TightVariableByte sum first
            sum = 0;
            for (uint32_t shift = 0; pos < index_bytes; shift += 7) {
                const uint8_t c = index[pos++];
                sum += uint32_t(c & 127) << (shift & 31);
                if (c & 128) break;
            }
        } else {
            sum = blocks[b].max - blocks[b].base - (n - 1);
        }
        o[n - 1] = sum;
        uint64_t used = 0;
        if (n > 1) {
            tail_bits br{index + pos, index_bytes - pos, 0, 0, 0, 0};
            uint32_t top = 0;
            auto push = [&](uint32_t off, uint32_t cnt, uint32_t low, uint32_t high) {
                uint32_t* f = stack + 4 * top++;
                f[0] = off, f[1] = cnt, f[2] = low, f[3] = high;
            };
            push(0, n - 1, 0, sum);
            while (top) {
                const uint32_t* f = stack + 4 * --top;
                const uint32_t f_off = f[0], f_n = f[1], f_low = f[2], f_high = f[3];
                const uint32_t h = f_n / 2;
                const uint32_t val = f_low + br.read_int(f_high - f_low + 1);
                o[f_off + h] = val;
                if (f_n - h - 1) push(f_off + h + 1, f_n - h - 1, val, f_high);
                if (h) push(f_off, h, f_low, val);
            }
            for (uint32_t i = n - 1; i > 0; --i) o[i] -= o[i - 1];
            used = (br.pos + 7) / 8;
        }
        if (end_off) end_off[b] = pos + used;
    }
    __syncthreads();
    for (uint32_t j = 0; j != 64; ++j) {  // rows out, the whole wave on one row at a time
        const uint32_t nj = row_n[j];
        uint32_t* const dst = out + row_out[j];
        for (uint32_t i = lane; i < nj; i += 64) dst[i] = tail_lds[j * kTailRow + i] + plus_one;
    }
}

// gaps -> docIDs (docid_i = base + sum_{j<=i} gap_j + i, dict_posting_list.hpp:111-124) and
// freq - 1 -> freq; one wave per block, 4 consecutive postings per lane.
__global__ void finalize_postings_kernel(const dint_block_ref* blocks, uint64_t n_blocks, uint32_t* docids,
                                         uint32_t* freqs, uint64_t out_capacity) {
    const uint32_t lane = lane_id();
    const uint64_t b = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kWave;
    if (b >= n_blocks) return;
    const uint32_t n = blocks[b].n;
    const uint64_t at = blocks[b].out_off;
    if (n == 0 || n > 256 || at + n > out_capacity) return;
    uint32_t g[4], local = 0;
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const uint32_t i = 4 * lane + k;
        g[k] = i < n ? docids[at + i] + 1 : 0;
        local += g[k];
    }
    uint32_t run = blocks[b].base + wave_inclusive_sum(local) - local - 1;
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const uint32_t i = 4 * lane + k;
        run += g[k];
        if (i < n) {
            docids[at + i] = run;
            if (freqs) freqs[at + i] += 1;
        }
    }
}

// test hook: out[i] = inclusive prefix sum of in[0..i] over one wave
__global__ void debug_wave_scan_kernel(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]);
}

}  // namespace dint_dev
