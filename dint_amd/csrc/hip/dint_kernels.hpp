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
//  * per slot one metadata word ((size-1) << 24 | source): LDS for the hot codewords (a prefix
//    of the dictionary), L2 for the cold ones, looked up one tile ahead;
//  * the integers of a cold codeword come from L2, 16-byte HEADS addressed by the slot value alone (the metadata word and
//    the first six integers in one lane request, requested one tile ahead) and, for the few entries of more than six,
//    32-byte tails requested when the heads are in; both land in STAGING CELLS of the wave's LDS scratch, allocated in
//    slot order by one wave scan (a cell per exception literal, one to three per cold codeword);
//  * header/payload classification: a table-driven per-lane state machine, iterated until the
//    lane-to-lane carries agree (one or two rounds);
//  * a local prefix plus ONE DPP wave scan gives every codeword its output offset and ordinal;
//  * EXPANSION is output-centric: every codeword sets ONE bit at its first output position in a
//    per-wave flag bitmap (ds_or), stores `source - position` in a table indexed by its ordinal,
//    and the lane whose outputs cross a 32-output boundary writes the rank base of that flag word
//    (all in one LDS phase). Then each lane takes 4 consecutive output integers: flag word + rank
//    base -> 4 ranks -> 4 table reads -> 4 LDS gathers (u16) -> one 16-byte non-temporal store, so
//    every global store instruction covers 1 KB of consecutive output. Every source is in LDS by
//    then, 16 bits per integer: hot payloads, the zero region of the runs, the staging cells
//    (cold rows, exception literals). Exactly n integers are written per unit, nothing past them
//    (the reference needs a pre-zeroed buffer and a 256-word overflow area,
//    include/dint/dint_codecs.hpp:11, dict_posting_list.hpp:296);
//  * an exception literal sits in its staging cell as 32 bits: the gather takes the low half, and — told
//    by bit 31 of its codeword's delta entry (delta_word) — the upper half of a literal >= 65536 (four more
//    reads in the 256-output groups that hold one); a dictionary entry holding such a value, and whatever
//    finds no staging cell, are SLOW: expanded as zeros, then written by the codeword's own lane straight
//    to global memory (slow_stores), behind the tile's stores;
//  * WAITS: gfx950 counts loads and stores in one in-order counter, so a tile has exactly one
//    wait point — before its expansion — where everything prefetched is consumed (read-write
//    asm barriers, so that the compiler never adds a wait behind the stores, which would be a
//    wait for their acknowledgements).
//
// LDS (160 KB/CU, one 1024-thread workgroup per CU):
//   [ 256 u16 zeros | hot meta | hot payloads (u16) ]  <= kHotImageWords, shared by 16 waves
//   [ slot classification table, 1.3 KB ]
//   16 x [ {flag word, rank base} pairs | per-codeword delta table | kStageCells staging cells of 16 bytes ]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dint_hip.h"

namespace dint_dev {

// Cache policy of the output stores (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1). The decoded integers are
// written once and never read by this kernel: non-temporal stores keep the 4 bytes/integer output stream
// from evicting the dictionary's cold part and the block directories out of L2 and from queueing
// behind write-back traffic — 0.97 -> 0.72 ms on the 4e8-posting run, the largest single gain of round 1.
#ifndef DINT_STORE_AUX
#define DINT_STORE_AUX 2
#endif
#ifndef DINT_BLOCK_THREADS
#define DINT_BLOCK_THREADS 1024
#endif
#ifndef DINT_FF_OPEN
#define DINT_FF_OPEN 4  // bundles the multi-dictionary schedule keeps open while it packs a chunk (first fit)
#endif
#ifndef DINT_LEAN_SEGMENT
#define DINT_LEAN_SEGMENT 0  // 2: decode_single_kernel's long units through decode_segment_v4 (explicit vector-memory waits, heads and
                             // tails a tile ahead): measured in round 4 — as fast as decode_segment, no faster (profiles/r04_v4_ab.txt);
                             // off by default
#endif
#ifndef DINT_GATHER_AUX
#define DINT_GATHER_AUX 0  // cache policy of the metadata / row gathers (L2-resident tables, no reuse in L1)
#endif

constexpr uint32_t kWave = 64;
constexpr uint32_t kBlockThreads = DINT_BLOCK_THREADS;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;
constexpr uint32_t kBlocksPerCU = 1;
constexpr uint32_t kLdsWords = 160 * 1024 / 4;
constexpr uint32_t kSPL = 4;                          // slots per lane per tile
constexpr uint32_t kTileSlots = kWave * kSPL;         // 256 slots per tile
#ifndef DINT_GROUPS
#define DINT_GROUPS 2
#endif
constexpr uint32_t kGroups = DINT_GROUPS;             // 256-output groups expanded together (one round)
constexpr uint32_t kRounds = 8 / kGroups;             // rounds per batch (single-dictionary segments)
constexpr uint32_t kMaxCap = 2048;                    // outputs per expansion batch at most: the flag bitmap's bits
// per wave: 64 {flag word, rank base} pairs (+ spare), per-codeword delta table (+ 4 dummy
// entries for codewords that are not live in a batch), staging cells
constexpr uint32_t kFwWords = 2 * 64 + 4;              // 64 pairs: flag positions are taken mod 2048
constexpr uint32_t kDeltaWords = kTileSlots + 4;
// 16-byte staging cells of a tile: exception literals (one each), cold codewords' integers (one to three each). A tile of the
// bench stream takes about 105; what finds no cell goes through slow_stores (correct, slow). Every cell less is 64 bytes
// more of the dictionary in LDS: 256 -> 176 cells = 20 KB = 69 -> 74 % of the codewords on chip, -2.6 % time at 10^9
// postings (144: -2.9 %, 128: no better — the overflow path begins to show; profiles/r03_cells.txt).
#ifndef DINT_STAGE_CELLS
#define DINT_STAGE_CELLS 176
#endif
constexpr uint32_t kStageWords = 4 * DINT_STAGE_CELLS;
#ifdef DINT_PROFILE
constexpr uint32_t kProfWords = 16;
#else
constexpr uint32_t kProfWords = 0;
#endif
constexpr uint32_t kScratchWords = kFwWords + kDeltaWords + kStageWords + kProfWords;
constexpr uint32_t kClassTableWords = 328 + 24;       // slot classification table: 648 u16 rows, padded; then the (up to 6)
                                                      // dictionaries' descriptors, 4 words each (a block's selector byte
                                                      // picks one: an LDS read instead of a trip to L2 per block)
constexpr uint32_t kDescWordAt = 328;
constexpr uint32_t kHotImageWords = kLdsWords - kClassTableWords - kWavesPerBlock * kScratchWords;
constexpr uint32_t kZeroHalves = 256;                 // longest run codeword, in u16
// Single-dictionary images carry, right behind the zeros, one bit per codeword: "a COLD codeword whose integers do not fit its
// 16-byte head" (more than 6: it needs its 32-byte tail too). The slot value alone then says whether to ask for the tail,
// so decode_segment_v4 requests heads AND tails a tile ahead and a tile has ONE wait for the dictionary, not two.
constexpr uint32_t kLongBitmapWords = 65536 / 32;
constexpr uint32_t kLongBitmapWordAt = kZeroHalves / 2;
// metadata word of a codeword: (size - 1) << 24 | kMetaCold | kMetaSlow | cells << 20 | LDS byte offset
constexpr uint32_t kMetaCold = 1u << 23;              // the integers come through staging cells (row table)
constexpr uint32_t kMetaSlow = 1u << 22;              // ... or, with this bit, from gtable through slow_stores
constexpr uint32_t kMetaOffMask = (1u << 20) - 1;     // hot: byte offset of the integers (u16 each) in the LDS image; else 0
                                                      // bits 20-21: staging cells a cold codeword takes (1: up to 6 integers, 2: up to 14, 3)
constexpr uint32_t kMetaException = 1u << 20;         // what the two exception markers (slot values 0 and 1) look up: one integer, one
                                                      // staging cell — the literal that follows the marker in the stream goes there
constexpr uint32_t kQueueShards = 8;                  // dynamic unit queue: one counter per shard
constexpr uint32_t kQueueStride = 32;                 // words between counters (own 128-byte line each)
constexpr uint32_t kClockWordAt = 16;                 // in the chunk counter's line: shader-clock cycles of the launch's first wave (u64)
constexpr uint32_t kMaxUnitInts = 1u << 28;           // byte offsets inside a unit's output stay 32-bit

// One dictionary of the (possibly multi-) dictionary file.
struct dict_desc {
    uint32_t meta_base;    // first slot of this dictionary in gmeta / the row table
    uint32_t hot_base;     // LDS word offset of its hot meta table
    uint32_t hot_k;        // codewords < hot_k have meta + payload in the LDS image
    uint32_t pad;
};

// Device view of a dictionary file (layout: dint_hip.hip, stage_dictionary).
struct dict_view {
    const uint8_t* tables;      // heads (16 bytes per slot) | tails (32 bytes per slot) | goff (u32 per slot) | gtable (u32 payload words)
    const uint32_t* lds_image;  // [256 u16 zeros]{[hot meta of dictionary d]}[hot payloads as u16], hot_words long
    const dict_desc* descs;     // one per dictionary (multi: 6)
    uint32_t tables_bytes;
    uint32_t heads_base;        // byte offset of the heads inside `tables`: {metadata word, integers 0..5 as u16} per slot
    uint32_t tails_base;        // ... of the tails: integers 6..21 as u16 per slot
    uint32_t goff_base;         // ... of the slow path's offsets into gtable (u32 per slot)
    uint32_t gtable_base;       // ... of gtable
    uint32_t hot_words;         // multiple of 4
    dict_desc first;            // descs[0], for the single-dictionary kernel
    uint32_t long_bitmap_word;  // single-dictionary images: LDS word offset of the "long entry" bitmap (0: the image has none)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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
    const uint8_t* sched;  // nullable; per unit: c != 0 = a work item of the unit queue standing for c units (bundle_schedule_kernel)
    const uint32_t* items; // with sched: what the unit queue hands out, in order. Multi-dictionary kernel: the units on their
                           // own (sched == 1). Single-dictionary kernel: those and the bundles' first units —
    const uint8_t* item_cnt;  // ... with the units each item stands for (1: a unit on its own)
    const uint32_t* n_items;
    // with sched: the bundles, found chunk by chunk (64 consecutive units; a bundle does not cross chunks)
    const u32x4* urec;       // per unit: {in_off - chunk's in0, out_off - chunk's out0, packed (see bundle_schedule_kernel), 0}
    const uint64_t* cbase;   // per chunk: {in0, out0} = in_off / out_off of its first unit
    uint32_t* chunk_queue;   // one counter, zero at launch: the next chunk
    const uint32_t* spans; // nullable; per unit an upper bound of its stream bytes (else: up to the next unit's start)
    uint32_t plus_one;     // in-index freqs parts: every decoded integer + 1 (dict_posting_list.hpp:164-169)
    // in-index docs parts (units = 256-posting blocks): the gaps leave the kernel as docIDs — docid_i = base +
    // sum_{j<=i} (gap_j + 1) - 1 (dict_posting_list.hpp:111-124, :304), one wave scan per block in the expansion.
    const uint32_t* unit_base;  // nullable; per unit: the block's docID base
    uint8_t* gaps_left;         // with unit_base; per unit, zero at launch: set where a block had to be left as gaps (it
                                // held a slow codeword) for the flagged fix-up
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
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Inclusive prefix sum over the 64 lanes, in registers: four row_shr steps inside each row of 16, then
// row_bcast:15 / row_bcast:31 across rows (gfx9 DPP) — written out: the builtin form compiled to three
// instructions a step (mov 0, mov_dpp, add) wherever the combine pass gave up. A VALU result needs two wait
// states before a DPP instruction reads it.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
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

// The value of the lane below (lane 0: zero) / above (lane 63: zero): one DPP move instead of a trip
// through the LDS crossbar (ds_bpermute, what __shfl_up / __shfl_down compile to).
__device__ __forceinline__ uint32_t from_lane_below(uint32_t x) {
    return __builtin_amdgcn_update_dpp(0u, x, 0x138, 0xf, 0xf, true);  // wave_shr:1
}
__device__ __forceinline__ uint32_t from_lane_above(uint32_t x) {
    return __builtin_amdgcn_update_dpp(0u, x, 0x130, 0xf, 0xf, true);  // wave_shl:1
}

__device__ __forceinline__ uint32_t readlane(uint32_t x, uint32_t l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    return (uint64_t(uniform(uint32_t(x >> 32))) << 32) | uniform(uint32_t(x));
}

// Orders this wave's LDS traffic between phases that communicate across lanes.
// LDS operations of one wave execute in issue order, so no hardware barrier is
// needed; this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// kSPL consecutive slots of one lane (16-bit slots: 8 bytes; `narrow`, 8-bit slots: 4 bytes) from an
// arbitrary byte address. `tile_byte` is the (wave-uniform) offset of the tile's
// first slot: when the whole tile lies inside the buffer — every tile but the stream's last —
// this is one plain load whose result nothing touches until the tile is unpacked, two tiles
// later. Otherwise it never reads past the buffer: the tail lanes load the final bytes and
// shift (bytes past the end read as zero), which waits for the data on the spot.
// (`narrow` is a compile-time constant in the single-dictionary kernel and wave-uniform in the multi one.)
__device__ __forceinline__ uint64_t load_lane_slots(bool narrow, const uint8_t* enc, uint64_t tile_byte, uint32_t lane,
                                                    uint64_t enc_bytes) {
    const uint32_t kBytes = narrow ? 4u : 8u;
    const uint64_t byte_off = tile_byte + uint64_t(kBytes) * lane;
    if (tile_byte <= enc_bytes && enc_bytes - tile_byte >= uint64_t(kBytes) * kWave) {  // wave-uniform
        if (!narrow) {
            const u32x2 r = reinterpret_cast<const u32x2_a1*>(enc + byte_off)->v;
            return (uint64_t(r.y) << 32) | r.x;
        }
        return reinterpret_cast<const u32_a1*>(enc + byte_off)->v;
    }
    const uint64_t last_valid = enc_bytes - kBytes;  // enc_bytes >= 8 is checked by the host
    const uint64_t o = byte_off < last_valid ? byte_off : last_valid;
    const uint64_t over = byte_off - o;  // 0 for all but the tail lanes
    uint64_t q;
    if (!narrow) {
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
struct meta_regs {     // the metadata of a tile's hot slots on its way in (LDS); the cold slots' comes with their heads
    uint32_t h[kSPL];
};
struct head_regs {     // per cold slot: the 16-byte head of its dictionary entry — metadata word + its first 6 integers
    u32x4 q[kSPL];     // (later in the tile: the next 8 integers of the entries that have them)
};

__device__ __forceinline__ void unpack_slots(bool narrow, uint64_t raw, tile_regs& t) {
    if (!narrow) {
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.s[k] = uint32_t(raw >> (16 * k)) & 0xFFFFu;
    } else {
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.s[k] = uint32_t(raw >> (8 * k)) & 0xFFu;
    }
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
// Under -DDINT_PROFILE (a diagnostic build, tools/variants/) the marks also stamp the shader clock.
#ifdef DINT_PROFILE
#include "dint_profile.hpp"
#else
struct prof_t {};
#define SECTION(pf, id, name) MARK(name)
#endif

// What a wavefront carries through every unit it decodes.
struct wave_ctx {
    const uint32_t* lds;           // the workgroup's LDS (the dictionary image first)
    const uint16_t* cls;           // slot classification table
    const uint32_t* descs;         // the dictionaries' descriptors in LDS: {meta_base, hot_base, hot_k, pad} each
    uint32_t* scratch;             // this wave's {flag pairs | delta table | staging cells}
    uint32_t lane;
    __amdgpu_buffer_rsrc_t rs_dict;  // gmeta | rows | gtable: one descriptor, hardware bounds
    uint32_t heads_base, tails_base, goff_base, gtable_base;
};

__device__ __forceinline__ uint32_t* fw_of(uint32_t* scratch) { return scratch; }
__device__ __forceinline__ uint32_t* delta_of(uint32_t* scratch) { return scratch + kFwWords; }
__device__ __forceinline__ uint32_t* stage_of(uint32_t* scratch) { return scratch + kFwWords + kDeltaWords; }

// What the front end needs to know about the four slots of a lane, requested while the previous tile is
// being expanded: the metadata word of the hot codewords from the LDS image (unconditional reads, all four in
// flight together; the cold lanes read a dummy word), and for every cold slot the 16-byte HEAD of its
// dictionary entry from L2 — its metadata word and its first six integers (u16) in one lane request, addressed
// by the slot value alone. (Round 1 and the first versions of this round fetched the metadata word and the
// integers separately: two lane requests per cold codeword, and the vector-memory front end — 0.45 scattered
// lane requests per CU and cycle, tools/micro/gather_rate.hip — was what the kernel waited for.)
// `payload`: bit k = slot k is an exception's payload (decode_segment classifies a tile before it asks for its heads:
// a payload slot is a 16-bit piece of a literal and looks like a cold codeword more often than not — 13 wasted requests
// in a tile's 107); callers that do not know yet pass 0.
__device__ __forceinline__ void request_metas(const wave_ctx& c, uint32_t hot_base, uint32_t hot_k, uint32_t meta_base,
                                              const tile_regs& t, meta_regs& mr, head_regs& hr, uint32_t payload = 0) {
    // (the word behind a dictionary's hot metas is a dummy: the cold lanes read it, min instead of compare + select)
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) mr.h[k] = c.lds[hot_base + (t.s[k] < hot_k ? t.s[k] : hot_k)];
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k)
#ifdef DINT_EXP_HEADS_L1
        if (t.s[k] >= hot_k) hr.q[k] = __builtin_amdgcn_raw_buffer_load_b128(c.rs_dict, c.heads_base + 16 * (meta_base + hot_k + (t.s[k] & 63u)), 0, 0);
#else
        if (t.s[k] >= hot_k && ((payload >> k) & 1u) == 0)
            hr.q[k] = __builtin_amdgcn_raw_buffer_load_b128(c.rs_dict, c.heads_base + 16 * (meta_base + t.s[k]), 0, 0);
#endif
}
// ... where the wait is: everything has landed (the asm makes the values the asm's, not a load's: nothing
// for the compiler to wait for later)
__device__ __forceinline__ void take_metas(uint32_t hot_k, meta_regs& mr, head_regs& hr, tile_regs& t) {
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) asm volatile("" : "+v"(hr.q[k]));
    asm volatile("" : "+v"(mr.h[0]), "+v"(mr.h[1]), "+v"(mr.h[2]), "+v"(mr.h[3]));
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) t.m[k] = t.s[k] < hot_k ? mr.h[k] : hr.q[k].x;
}

// (LDS-DMA — buffer_load ... lds with per-lane offsets, no registers — delivers the right bytes,
// tools/micro/lds_dma.hip, but its destination is fixed by the lane number: a cell per SLOT instead of per
// cold codeword, 4 KB more LDS per wave — the LDS the dictionary's hot part lives on.)

// Segment chaining (multi-dictionary units): a block's bytes are known only when the previous block
// has been parsed, so a block on its own pays the full memory latency of its selector and its slots
// before it can start. Chained, the first kChainBytes of the next block (16 per lane, from the byte
// after its selector on) and its selector are requested as soon as the current block's end is known
// — after the scans, before its expansion and stores — and are re-laid-out lane to lane
// (ds_bpermute) into the first two tiles of the next segment.
constexpr uint32_t kChainBytes = 16 * kWave;
struct chain_io {
    u32x4 data;     // bytes [16 * lane, 16 * lane + 16) of the segment's slot stream
    uint32_t sel;   // the byte before them (the block's selector) in bits 0-7
    bool more;      // in: a segment follows this one
    bool valid;     // out: data / sel hold the next segment's bytes
};

__device__ __forceinline__ void chain_request(const uint8_t* enc, uint64_t selector_byte, uint32_t lane, chain_io& ch) {
    ch.sel = enc[selector_byte];
    ch.data = reinterpret_cast<const u32x4_a1*>(enc + selector_byte + 1 + 16u * lane)->v;
    ch.valid = true;
}

// slots of tile t (0 or 1) of a chained segment, in the lane layout load_lane_slots produces
__device__ __forceinline__ uint64_t chain_tile(bool narrow, const u32x4& d, uint32_t t, uint32_t lane) {
    if (!narrow) {  // lane l: bytes [512 t + 8 l, + 8) = half (l & 1) of lane 32 t + l / 2
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

// What a tile's front end hands to its tables and expansion, per lane (slot k = 0..3 of the lane) and per wave.
struct tile_slots {
    uint32_t off[kSPL];     // first output of slot k's codeword behind the lane's first (integers)
    uint32_t src2[kSPL];    // LDS byte address of its integers (u16 each): hot payload, zero region, its staging cell(s);
                            // bit 0: the cell holds a 32-bit exception literal
    uint32_t need[kSPL];    // staging cells it takes: 0; 1 (exception literal; cold codeword of up to 6 integers); 2 (up to 14); 3
    uint32_t lsum, obase;   // outputs of the lane's live codewords, position of the first
    uint32_t total;         // outputs of the tile (wave-uniform)
    // tiles that are not plain (an exception or a payload slot somewhere, the segment's last tile, a bundle):
    uint32_t row;           // the lane's classification row: bit 11 / bits 12-13 / bits 14-15 = ordinal of slot 1 / 2 / 3
                            // behind the lane's first codeword; bits 4-7: exception headers
    uint32_t liveb;         // bit k: slot k is a codeword header inside the segment
    uint32_t rbase, nlive;  // ordinal of the lane's first live codeword, how many it has
    __device__ __forceinline__ uint32_t lord(uint32_t k) const {
        return k == 0 ? 0u : k == 1 ? (row >> 11) & 1u : k == 2 ? (row >> 12) & 3u : (row >> 14) & 3u;
    }
};
constexpr uint32_t kPlainRow = (1u << 11) | (2u << 12) | (3u << 14);  // four codeword headers, nothing special
constexpr uint32_t kStageCells = kStageWords / 4;

// Staging cells of a tile: one 16-byte cell per exception literal; per cold codeword one for its head (metadata
// word + 6 integers: the integers start 4 bytes into the cell), a second for integers 6..13, a third for 14
// and 15; allocated in slot order by one wave scan over t.need. -> the cell's LDS byte
// address per slot; returns the cells the tile takes (wave-uniform).
__device__ __forceinline__ uint32_t allocate_cells(const tile_slots& t, uint32_t stage_byte0, uint32_t (&cell_addr)[kSPL]) {
    const uint32_t pre1 = t.need[0], pre2 = pre1 + t.need[1], pre3 = pre2 + t.need[2], mine = pre3 + t.need[3];
    const uint32_t incl = wave_inclusive_sum(mine);
    cell_addr[0] = stage_byte0 + 16 * (incl - mine);
    cell_addr[1] = cell_addr[0] + 16 * pre1;
    cell_addr[2] = cell_addr[0] + 16 * pre2;
    cell_addr[3] = cell_addr[0] + 16 * pre3;
    return readlane(incl, 63);
}

// A codeword's entry in the delta table: (its source - its first output position), both in u16 units, so that an output at
// position p of the batch reads its integer from LDS byte address 2 (delta + p). `src2` is the source's LDS byte address
// (even) with bit 0 set for a 32-bit exception literal; that bit travels in bit 31 of the entry (one rotate), the
// expansion's address arithmetic — (delta + p) << 1 — sheds it, and an OR over a lane's four entries says whether any
// of them needs its upper half. kDeltaBias keeps the entry non-negative (a source may lie below its position); the
// expansion's position carries - kDeltaBias.
constexpr uint32_t kDeltaBias = 4096;
__device__ __forceinline__ uint32_t delta_word(uint32_t src2, uint32_t rel) {
    return __builtin_rotateright32(src2, 1) - rel + kDeltaBias;
}

// ---- the flag / delta / rank-base tables of a batch: one LDS phase, written by the codewords' lanes ----
// A flag bit at each codeword's first output (the flag words are zero: cleared at the end of the previous
// batch), `source - position` by ordinal, and the rank base (codewords before the word, minus one) of every
// flag word that begins inside this lane's outputs.
//
// The rank base of every flag word — the codewords that start before it, minus one — is the running count of the
// flags themselves: lane w reads word w behind the ORs (a wave's LDS operations execute in order), one wave scan. (Until
// round 3 every lane wrote the bases of the words that begin inside its outputs, in a loop: as many rounds for the whole
// wave as the lane with the longest run needed, 17 vector instructions each — a fifth of the kernel's.)
__device__ __forceinline__ void rank_bases(uint8_t* fw, uint32_t lane) {
    const uint32_t flags = *reinterpret_cast<const uint32_t*>(fw + 8 * lane);
    const uint32_t cnt = uint32_t(__builtin_popcount(flags));
    *reinterpret_cast<uint32_t*>(fw + 8 * lane + 4) = wave_inclusive_sum(cnt) - cnt - 1u;
}

// A plain tile — four codeword headers in every lane, nothing clamped, one batch: ordinals are 4 lane + k, the
// four deltas of a lane are one 16-byte store, nothing is masked.
__device__ __forceinline__ void tables_plain(const tile_slots& t, uint8_t* fw, uint8_t* delta, uint32_t lane) {
    uint32_t rel[kSPL];
    rel[0] = t.obase;
#pragma unroll
    for (uint32_t k = 1; k != kSPL; ++k) rel[k] = t.obase + t.off[k];
    u32x4 d;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        uint32_t* const fword = reinterpret_cast<uint32_t*>(fw + ((rel[k] >> 2) & 0x1F8u));
        __hip_atomic_fetch_or(fword, 1u << (rel[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        d[k] = delta_word(t.src2[k], rel[k]);
    }
    *reinterpret_cast<u32x4*>(delta + 16 * lane) = d;
    rank_bases(fw, lane);
}

// Any tile, one batch of it: the lanes `inb`, outputs and ordinals counted from `done` / `rdone`. Every slot
// runs the same instructions: a codeword that is not live in this batch ORs a zero into an in-range flag word
// and parks its delta in a dummy.
__device__ __forceinline__ void tables_general(const tile_slots& t, uint8_t* fw, uint8_t* delta, bool inb, uint32_t done,
                                               uint32_t rdone, uint32_t lane) {
    const uint32_t inbM = inb ? ~0u : 0u;
    const uint32_t rel0 = t.obase - done, ord0 = t.rbase - rdone;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t lv = uint32_t(__builtin_amdgcn_sbfe(t.liveb, k, 1)) & inbM;
        const uint32_t rel = rel0 + t.off[k];
        uint32_t* const fword = reinterpret_cast<uint32_t*>(fw + ((rel >> 2) & 0x1F8u));
        __hip_atomic_fetch_or(fword, lv & (1u << (rel & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t ord = (lv & (ord0 + t.lord(k))) | (~lv & (kTileSlots + k));
        *reinterpret_cast<uint32_t*>(delta + 4 * ord) = delta_word(t.src2[k], rel);
    }
    rank_bases(fw, lane);
}

// A first-fit bundle (the multi-dictionary kernel, see bundle_map_first_fit) is any set of up to 8 units of one chunk;
// member m's outputs are positions [256 m, 256 m + n) of the tile — one expansion group — and are stored where the unit says.
struct group_out {
    uint32_t* base;            // the chunk's output base (wave-uniform)
    uint32_t unit_n, rel_out;  // lane g: member g's unit (its lane in the chunk) | its integers << 8; its outputs' offset from the base
};

// ---- expansion of a batch of `bt` outputs, GROUPS * 256 per round: each lane takes 4 consecutive outputs of
// every 256-output group — flag word + rank base -> 4 ranks -> 4 deltas -> 4 LDS gathers (u16) -> one 16-byte
// non-temporal store; every source is an LDS byte address by now. Stores are whole 16-byte quads: the
// descriptor clips what lies past the segment's n integers (range checking is per dword), and what a quad
// writes past this batch's end inside the segment is rewritten by the batches and tiles that follow (same
// wave, program order). The batch may hold 32-bit exception literals — bit 31 of a delta entry (delta_word)
// says the upper half follows the lower one in the staging cell.
template <uint32_t ROUNDS, uint32_t GROUPS>
__device__ __forceinline__ void expand_batch(uint32_t bt, uint32_t out_int, const uint8_t* lds_bytes, const uint8_t* fw,
                                             const uint8_t* delta, const __amdgpu_buffer_rsrc_t rs_out, uint32_t lane,
                                             uint32_t plus_one, const uint32_t* group_base, const group_out* go = nullptr) {
    // lane constants: this lane owns outputs 4*lane .. 4*lane+3 of every group
    const uint32_t sh = (4 * lane) & 31u;                 // bit position of its nibble in its flag word
    const uint32_t pair_byte = (lane >> 3) * 8;           // its {flag, base} pair inside a group's 8 pairs
    const uint32_t posb = 4 * lane - kDeltaBias;          // its first output's position in a group, minus the entries' bias
    out_int = uniform(out_int);  // (the stores' scalar offset: an SGPR, not a loop over the values a VGPR might hold)
#pragma unroll
    for (uint32_t rd = 0; rd != ROUNDS; ++rd) {
        if (rd * GROUPS * 4 * kWave < bt) {  // wave-uniform
            const uint32_t obyte = 4 * out_int + rd * GROUPS * 16 * kWave;  // output byte offset of the round
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
                    const uint32_t gbyte = (rd * GROUPS + g) * 8 * kWave;  // the group's source byte position in the batch
                    uint32_t d[4], ad[4];
#pragma unroll
                    for (int k = 0; k != 4; ++k) {
                        d[k] = *reinterpret_cast<const uint32_t*>(delta + 4 * r[k]);
                        ad[k] = (d[k] + posb) << 1;  // (bit 31 of the entry — "the upper half follows" — is shifted out)
                        // (the whole offset in 32 bits before it meets the pointer: ad is routinely a wrapped negative)
                        x[g][k] = *reinterpret_cast<const uint16_t*>(lds_bytes + uint32_t(ad[k] + gbyte + 2 * k));
                    }
                    // a 32-bit exception literal among the four (a few per tile): its upper half follows the lower one
                    if (__builtin_expect(__ballot(int32_t(d[0] | d[1] | d[2] | d[3]) < 0) != 0, 0)) {
#pragma unroll
                        for (int k = 0; k != 4; ++k) {
                            const uint32_t hi = *reinterpret_cast<const uint16_t*>(lds_bytes + uint32_t(ad[k] + gbyte + 2 * k + 2));
                            x[g][k] |= (int32_t(d[k]) < 0 ? hi : 0u) << 16;
                        }
                    }
                }
            }
            MARK("9_stores");
#pragma unroll
            for (uint32_t g = 0; g != GROUPS; ++g) {
                const uint32_t p0 = (rd * GROUPS + g) * 4 * kWave + 4 * lane;
                if ((rd * GROUPS + g) * 4 * kWave < bt) {  // wave-uniform
                    if (group_base) {  // wave-uniform: the group is one 256-posting block — gaps to docIDs
                        const uint32_t base = uniform(group_base[go ? readlane(go->unit_n, rd * GROUPS + g) & 255u : rd * GROUPS + g]);
                        const uint32_t v0 = x[g][0] + 1u, v1 = v0 + x[g][1] + 1u, v2 = v1 + x[g][2] + 1u, v3 = v2 + x[g][3] + 1u;
                        const uint32_t before = wave_inclusive_sum(v3) - v3 + base - 1u;
                        x[g][0] = before + v0, x[g][1] = before + v1, x[g][2] = before + v2, x[g][3] = before + v3;
                    }
                    if (go) {  // wave-uniform: the group's outputs have a place of their own, and end where its unit ends
                        u32x4 xv = {x[g][0], x[g][1], x[g][2], x[g][3]};
                        if (plus_one) xv += 1u;
                        uint32_t* const at = go->base + readlane(go->rel_out, rd * GROUPS + g);
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(at, 0, int(4 * (readlane(go->unit_n, rd * GROUPS + g) >> 8)), 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b128(xv, rs, 16 * lane, 0, DINT_STORE_AUX);
                    } else if (p0 < bt) {
                        u32x4 xv = {x[g][0], x[g][1], x[g][2], x[g][3]};
                        if (plus_one) xv += 1u;  // wave-uniform branch: nothing on the plain decode path
                        // (DINT_EXP_*: timing experiments, tools/build_variants.sh — their results are wrong by construction)
#ifdef DINT_EXP_NOSTORE
                        asm volatile("" : : "v"(xv));
#elif defined(DINT_EXP_STORE_LANES)
                        if (lane < DINT_EXP_STORE_LANES) __builtin_amdgcn_raw_buffer_store_b128(xv, rs_out, 16 * g * kWave + 16 * lane, obyte, DINT_STORE_AUX);
                        else asm volatile("" : : "v"(xv));
#else
                        __builtin_amdgcn_raw_buffer_store_b128(xv, rs_out, 16 * g * kWave + 16 * lane, obyte, DINT_STORE_AUX);
#endif
                    }
                }
            }
        }
    }
}

// Steps 3 and 4 of a tile: tables, expansion, stores — in one batch when the tile decodes to at most
// ROUNDS x GROUPS x 256 integers (every tile of a real stream), else in batches of lanes (a tile full of long
// runs: up to 256 x 256 integers). `plain`: tables_plain applies. `wide`: the tile holds a 32-bit exception
// literal. `before_gathers` runs once, before the first gathers: the caller lands there what it requested at
// the wait point, and requests the next tile's metadata (decode_segment).
// (The one-batch path is straight-line on purpose: inside a loop over batches everything the tables are built
// from would stay live through the expansion — 55 more registers, measured.)
template <uint32_t ROUNDS, uint32_t GROUPS, bool ONE_BATCH = false, class BeforeGathers>
__device__ __forceinline__ void expand_tile(const tile_slots& t, bool plain, bool wide, uint32_t plus_one,
                                            const uint32_t* group_base, uint32_t out_int0,
                                            const uint32_t* lds, uint32_t* scratch, const __amdgpu_buffer_rsrc_t rs_out,
                                            uint32_t lane, prof_t& pf, BeforeGathers&& before_gathers, const group_out* go = nullptr) {
    (void)pf;
    constexpr uint32_t kCap = ROUNDS * GROUPS * 256;
    static_assert(kCap <= kMaxCap, "the flag bitmap holds 2048 positions");
    // per-wave scratch (byte offsets): {flag word, rank base} pairs | delta table | staging cells
    uint8_t* const fw = reinterpret_cast<uint8_t*>(fw_of(scratch));        // 64 pairs of 8 bytes (+1 spare)
    uint8_t* const delta = reinterpret_cast<uint8_t*>(delta_of(scratch));  // 256 entries + 4 dummies
    const uint8_t* const lds_bytes = reinterpret_cast<const uint8_t*>(lds);
    SECTION(pf, 4, "4_tables");
    if (ONE_BATCH || __builtin_expect(t.total <= kCap, 1)) {  // (ONE_BATCH: the caller knows; what lies past the cap is not decoded)
        const uint32_t total = t.total < kCap ? t.total : kCap;
        if (plain) tables_plain(t, fw, delta, lane);
        else tables_general(t, fw, delta, t.lsum != 0, 0u, 0u, lane);
        wave_lds_fence();
        SECTION(pf, 7, "7_rows2");
        before_gathers();
        SECTION(pf, 9, "9_expand");
        (void)wide;
        expand_batch<ROUNDS, GROUPS>(total, out_int0, lds_bytes, fw, delta, rs_out, lane, plus_one, group_base, go);
        // the flag words go back to zero for the next batch (this wave's LDS operations execute in order)
        *reinterpret_cast<uint32_t*>(fw + 8 * lane) = 0;
        wave_lds_fence();
        return;
    }
    uint32_t done = 0, rdone = 0;
    do {
        // the lanes of this batch: as many as the flag bitmap has room for
        const bool inb = t.lsum != 0 && t.obase >= done && (t.obase + t.lsum - done) <= kCap;
        const uint64_t bm = __ballot(inb);
        const uint32_t last = 63u - uint32_t(__builtin_clzll(bm | 1ull));
        const uint32_t bend = readlane(t.obase + t.lsum, last);
        const uint32_t rend = readlane(t.rbase + t.nlive, last);
        if (bend <= done) {  // (malformed input: nothing decodable left in this tile)
            if (done == 0) before_gathers();
            break;
        }
        tables_general(t, fw, delta, inb, done, rdone, lane);
        wave_lds_fence();
        if (done == 0) before_gathers();
        // (batches of lanes do not end on block boundaries: no docIDs here — the caller leaves such a tile as gaps)
        expand_batch<ROUNDS, GROUPS>(bend - done, out_int0 + done, lds_bytes, fw, delta, rs_out, lane, plus_one, nullptr);
        *reinterpret_cast<uint32_t*>(fw + 8 * lane) = 0;
        wave_lds_fence();
        done = bend;
        rdone = rend;
    } while (done < t.total);
}

// The slow codewords of a tile, written by their own lanes straight to the output, behind the tile's
// stores (which put zeros there): dictionary entries that hold a value of 65536 or more, and whatever found
// no staging cell (more than 256 cells in one tile: no real stream, but a legal one) — looked up again here:
// rare enough not to ride through the expansion in registers. `slowb` bit k: slot k; `pos0` = the tile's first
// output inside the segment of `seg_n` integers; `slot_addr` = this lane's first slot in the stream (8 bits per
// slot if `narrow`, else 16).
__device__ __forceinline__ void slow_stores(bool narrow, const wave_ctx& c, const tile_slots& t, uint32_t slowb, uint32_t plus_one, uint32_t pos0,
                                            uint32_t seg_n, const uint8_t* slot_addr, uint32_t hot_base, uint32_t hot_k,
                                            uint32_t meta_base, const __amdgpu_buffer_rsrc_t rs_out, uint32_t store_shift = 0) {
    // (store_shift, per lane: what takes a position of the tile to the integer's place behind rs_out's base — first-fit
    // bundles, whose members' outputs are not the tile's positions)
    // the zeros must be in memory first: two stores of one wave to one address are only ordered by the wait
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        if ((slowb >> k) & 1u) {
            const uint32_t pos = pos0 + t.obase + t.off[k];
            const uint8_t* const sp = slot_addr + (narrow ? 1u : 2u) * k;
            const uint32_t sv = !narrow ? uint32_t(sp[0]) | (uint32_t(sp[1]) << 8) : uint32_t(sp[0]);
            if (sv < 2) {  // an exception whose literal found no staging cell: from the stream again
                const uint8_t* const lp = sp + (narrow ? 1 : 2);
                uint32_t v = uint32_t(lp[0]) | (uint32_t(lp[1]) << 8);
                if (sv == 1) v |= (uint32_t(lp[2]) << 16) | (uint32_t(lp[3]) << 24);
                __builtin_amdgcn_raw_buffer_store_b32(v + plus_one, rs_out, 4 * (pos + store_shift), 0, DINT_STORE_AUX);
            } else {
                const uint32_t m = sv < hot_k ? c.lds[hot_base + sv] : __builtin_amdgcn_raw_buffer_load_b32(c.rs_dict, c.heads_base + 16 * (meta_base + sv), 0, 0);
                const uint32_t goff = __builtin_amdgcn_raw_buffer_load_b32(c.rs_dict, c.goff_base + 4 * (meta_base + sv), 0, 0);
                const uint32_t size = (m >> 24) + 1u, room = seg_n - pos;
                const uint32_t cnt = size < room ? size : room;
#pragma nounroll
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(c.rs_dict, c.gtable_base + 4 * (goff + j), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v + plus_one, rs_out, 4 * (pos + j + store_shift), 0, DINT_STORE_AUX);
                }
            }
        }
    }
}

// ROUNDS x GROUPS x 256 = outputs per expansion batch: 2 x 4 for the long single-dictionary
// segments; a multi-dictionary segment is one block of at most 256 integers, 1 x 1.
// W = 16 / 8: the slot width; 0: `narrow_rt` says (wave-uniform). CHAINED = 1 / 0, or -1: `chained_rt` says.
// (The multi-dictionary kernel has ONE instantiation for both widths, chained or not: with four, inlined side by
// side, the register allocator spilled 127 registers to scratch.)
template <int W, uint32_t ROUNDS, uint32_t GROUPS, int CHAINED_T>
__device__ __forceinline__ uint64_t decode_segment(const decode_args& a, const wave_ctx& c, const dict_desc& dd, uint64_t in_off,
                                                   uint32_t n, uint32_t* const out, chain_io& ch, prof_t& pf,
                                                   bool narrow_rt = false, bool chained_rt = false,
                                                   const uint32_t* block_base = nullptr, uint8_t* gaps_left = nullptr) {
    SECTION(pf, 11, "segment_prologue");
    const bool narrow = W == 0 ? narrow_rt : W == 8;
    const bool CHAINED = CHAINED_T < 0 ? chained_rt : CHAINED_T != 0;
    const uint32_t kSlotBytes = narrow ? 1u : 2u;
    const uint32_t kTileBytes = kTileSlots * kSlotBytes;
    const uint16_t* const rows = c.cls + (!narrow ? 0 : kRows16);
    const uint32_t lane = c.lane;
    const uint32_t hot_k = dd.hot_k;
    // hardware bounds: nothing past this segment's n integers can be written
    uint32_t* const out_u = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(out)));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, int(uniform(n) * 4), 0x00020000);
    uint8_t* const lds_rw = reinterpret_cast<uint8_t*>(const_cast<uint32_t*>(c.lds));
    const uint32_t stage_byte0 = uint32_t(reinterpret_cast<const uint8_t*>(stage_of(c.scratch)) - lds_rw);  // the cells, as LDS byte addresses

    // pipeline: tile t in `cur` (slots + metadata; the rows of its cold slots on their way into `rr`), tile
    // t+1's slots in `nxt` (its metadata requested half-way through tile t, its rows at the end of tile t),
    // tile t+2's slots in raw2, tile t+3's requested at the top of tile t
    const uint64_t in_off_u = uniform64(in_off);
    uint64_t slot_byte = in_off_u;  // first byte of the tile whose slots are loaded next (wave-uniform)
    tile_regs cur, nxt;
    meta_regs mr;
    head_regs hr;  // (deliberately uninitialised: each register is written and read under the same lane predicate)
    uint64_t raw1, raw2 = 0;
    if (CHAINED) {  // tiles 0 and 1 arrived with the previous block (or were requested by the caller)
        unpack_slots(narrow, chain_tile(narrow, ch.data, 0, lane), cur);
        raw1 = chain_tile(narrow, ch.data, 1, lane);
        slot_byte += kTileBytes;  // the pipeline is one tile shorter: tile t+2's slots are requested in tile t
    } else {
        // (all three requested before the first is looked at: one trip to memory, not two)
        uint64_t raw0 = load_lane_slots(narrow, a.enc, slot_byte, lane, a.enc_bytes);
        slot_byte += kTileBytes;
        raw1 = load_lane_slots(narrow, a.enc, slot_byte, lane, a.enc_bytes);
        slot_byte += kTileBytes;
        raw2 = load_lane_slots(narrow, a.enc, slot_byte, lane, a.enc_bytes);
        asm volatile("" : "+v"(raw0), "+v"(raw1), "+v"(raw2));
        unpack_slots(narrow, raw0, cur);
    }
    ch.valid = false;
    // ---- classification of a tile's slots — which are an exception's payload, which exception headers — by table
    // lookup, repeated until the lane-to-lane carries agree. It needs the slots and the carry of the tile before and
    // nothing else, so it runs a tile AHEAD, before the tile's heads are requested: payload slots ask for none.
    struct tile_class {
        uint32_t row;        // the lane's classification row (kPlainRow: four codeword headers)
        uint32_t carry_out;  // payload slots the tile's last exception still owns in the next tile (wave-uniform)
        bool special;        // some slot is an exception header or payload (wave-uniform)
        bool tile_exc;       // some slot is an exception header (wave-uniform)
    };
    auto classify = [&](const tile_regs& tr, uint32_t carry_in) -> tile_class {
        tile_class k{kPlainRow, 0u, false, false};
        uint32_t smin = tr.s[0];
#pragma unroll
        for (uint32_t j = 1; j != kSPL; ++j) smin = smin < tr.s[j] ? smin : tr.s[j];
        k.special = __builtin_expect(__ballot(smin < 2) != 0 || carry_in != 0, 0);
        if (k.special) {
            // base-3 digits of the four slots: 2 - min(slot, 2)
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t j = kSPL; j-- != 0;) lo = 3 * lo + (2u - (tr.s[j] < 2 ? tr.s[j] : 2u));
            uint32_t st_in = lane == 0 ? carry_in : 0u;
            uint32_t row;
            for (;;) {
                row = rows[st_in * 81 + lo];
                uint32_t prev = from_lane_below((row >> 8) & 7u);
                if (lane == 0) prev = carry_in;
                if (__ballot(prev != st_in) == 0) break;
                st_in = prev;
            }
            k.row = row;
            k.carry_out = readlane((row >> 8) & 7u, 63);
            k.tile_exc = __ballot((row & 0xF0u) != 0) != 0;
        }
        return k;
    };
    tile_class kc = classify(cur, 0u);
    request_metas(c, dd.hot_base, hot_k, dd.meta_base, cur, mr, hr, kc.row & 15u);
    // Everything loaded so far has landed before the loop is entered: inside it, a wait may only
    // ever sit before a tile's stores (see the notes below), never right after them.
    asm volatile("" : "+v"(raw1), "+v"(raw2));
    unpack_slots(narrow, raw1, nxt);

    uint32_t produced = 0;
    uint64_t tile_base = in_off_u; // byte offset of slot 0 of the current tile (wave-uniform)
    uint32_t end_slot = 0;
    MARK("loop_top");

    while (produced < n) {
        // The far prefetch — the slots of the tile after next, straight from HBM — goes out first: it has to
        // be back before this tile's expansion (every wait is a wait for everything), so it gets the whole
        // front end. (Requested at the wait point instead, as the youngest load in flight there, it would have
        // a whole tile; measured: no gain — the kernel waits for the vector-memory front end, not for HBM.)
        slot_byte += kTileBytes;
        const uint64_t raw3 = load_lane_slots(narrow, a.enc, slot_byte, lane, a.enc_bytes);

        SECTION(pf, 1, "1_classify");
        // this tile's metadata: requested before the previous tile's stores (so this is no wait for them)
        take_metas(hot_k, mr, hr, cur);
        // ---- 1. classification: done a tile ago (kc) ----
        const bool special = kc.special;
        const uint32_t row = kc.row;
        const uint32_t carry_out = kc.carry_out;
        const bool tile_exc = kc.tile_exc;

        SECTION(pf, 2, "2_sizes");
        // ---- 2. sizes, offsets; where each codeword's integers are -----------------------------------
        tile_slots t;
        // A payload slot decodes to nothing and takes nothing: its metadata word (whatever its value looks up) becomes
        // zero, and the 1 of "size - 1" is its live bit. An exception header needs no special case here: the
        // metadata of the two markers says one integer from one staging cell (stage_dictionary, choose_hot_set).
        uint32_t lv1[kSPL] = {1u, 1u, 1u, 1u};
        if (special) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                cur.m[k] &= uint32_t(__builtin_amdgcn_sbfe(~row, k, 1));
                lv1[k] = __builtin_amdgcn_ubfe(~row, k, 1);
            }
        }
        // (behind the mask: a payload slot asked for no head, what its register holds is stale)
        const bool tile_slow_dict = __ballot(((cur.m[0] | cur.m[1] | cur.m[2] | cur.m[3]) & kMetaSlow) != 0) != 0;
        uint32_t e[kSPL];  // size - 1
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            e[k] = cur.m[k] >> 24;
            t.need[k] = __builtin_amdgcn_ubfe(cur.m[k], 20, 2);
            t.src2[k] = cur.m[k] & kMetaOffMask;  // hot: the image (runs: the zeros); cold, slow: zero for now
        }
        // An exception's literal rides into its staging cell the way a cold codeword's first integers do: in the second
        // word of the slot's head register (the integers of a cell start 4 bytes in), 32 bits wide. The expansion gathers
        // its low half like any other integer, and — told by bit 0 of the source address — the upper half of one >= 65536.
        bool wide[kSPL] = {false, false, false, false};
        if (tile_exc) {
            // the 32 bits of the stream behind each slot: the lane's later slots, then the next lane's first ones (lane
            // 63: the next tile's; what lane 0 of the next tile holds is read with every lane enabled: a cross-lane read
            // of a value computed under `lane == 63` would find lane 0's register untouched)
            uint32_t w32[kSPL];
            if (!narrow) {
                const uint32_t p01 = (cur.s[1] << 16) | cur.s[0], p23 = (cur.s[3] << 16) | cur.s[2];
                const uint32_t next0 = readlane((nxt.s[1] << 16) | nxt.s[0], 0);
                uint32_t nlo = from_lane_above(p01);
                if (lane == 63) nlo = next0;
                w32[0] = __builtin_amdgcn_alignbit(p23, p01, 16);
                w32[1] = p23;
                w32[2] = __builtin_amdgcn_alignbit(nlo, p23, 16);
                w32[3] = nlo;
            } else {
                const uint32_t b = cur.s[0] | (cur.s[1] << 8) | (cur.s[2] << 16) | (cur.s[3] << 24);
                const uint32_t next0 = readlane(nxt.s[0] | (nxt.s[1] << 8) | (nxt.s[2] << 16) | (nxt.s[3] << 24), 0);
                uint32_t nlo = from_lane_above(b);
                if (lane == 63) nlo = next0;
                w32[0] = __builtin_amdgcn_alignbit(nlo, b, 8);
                w32[1] = __builtin_amdgcn_alignbit(nlo, b, 16);
                w32[2] = __builtin_amdgcn_alignbit(nlo, b, 24);
                w32[3] = nlo;
            }
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t excM = uint32_t(__builtin_amdgcn_sbfe(row, 4 + k, 1));
                // marker 0: the 16 bits behind it; marker 1: all 32 (0 - marker = which)
                const uint32_t lit = w32[k] & ((0u - cur.s[k]) | 0xFFFFu) & excM;
                hr.q[k].y = (hr.q[k].y & ~excM) | lit;
                wide[k] = lit > 0xFFFFu;
            }
        }
        t.off[0] = 0;
        t.off[1] = e[0] + lv1[0];
        t.off[2] = t.off[1] + e[1] + lv1[1];
        t.off[3] = t.off[2] + e[2] + lv1[2];
        t.lsum = t.off[3] + e[3] + lv1[3];
        uint32_t hdrcnt = 4;
        uint32_t pincl;
        if (special) {
            hdrcnt = uint32_t(__builtin_popcount(~row & 15u));
            pincl = wave_inclusive_sum((hdrcnt << 24) | t.lsum);
        } else {
            pincl = wave_inclusive_sum(t.lsum);  // (the ordinals of a plain tile are 4 lane + k)
        }
        t.obase = (pincl & 0xFFFFFFu) - t.lsum;    // first output of this lane's codewords
        const uint32_t remaining = n - produced;
        t.total = readlane(pincl, 63) & 0xFFFFFFu;
        const bool last_tile = t.total >= remaining;
        const bool plain = !special && !last_tile && t.total <= ROUNDS * GROUPS * 256;
        uint32_t slowb = 0;  // bit k: slot k goes through slow_stores
        if (!plain) {
            t.row = row;
            t.liveb = ~row & 15u;
            // ordinal of this lane's first codeword (exclusive before the shift: the inclusive count can be 256)
            t.rbase = special ? (pincl - ((hdrcnt << 24) | t.lsum)) >> 24 : 4 * lane;
            t.nlive = hdrcnt;
            if (last_tile) {  // last tile of the segment: clamp, and find where the stream ends
                t.total = remaining;
                uint32_t cand = 0, lb = 0;
                t.nlive = 0;
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) {
                    const uint32_t pos = t.obase + t.off[k];
                    const bool act = ((t.liveb >> k) & 1u) != 0 && pos < remaining;
                    if (act) {
                        const bool exc = ((row >> (4 + k)) & 1u) != 0;
                        cand = kSPL * lane + k + 1 + (exc ? (!narrow ? cur.s[k] + 1 : 2 * cur.s[k] + 2) : 0u);
                        ++t.nlive;
                        lb |= 1u << k;
                    } else {
                        t.need[k] = 0;
                    }
                }
                t.liveb = lb;
                t.lsum = t.obase < remaining ? (t.obase + t.lsum < remaining ? t.lsum : remaining - t.obase) : 0u;
                const uint64_t am = __ballot(cand != 0);
                end_slot = readlane(cand, 63u - uint32_t(__builtin_clzll(am | 1ull)));
            }
            if (tile_slow_dict) {
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k)
                    slowb |= ((cur.m[k] >> 22) & (t.liveb >> k) & ~(row >> (4 + k)) & 1u) << k;
            }
        } else if (tile_slow_dict) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) slowb |= ((cur.m[k] >> 22) & 1u) << k;
        }
        // ---- staging cells: exception literals and cold codewords --------------------------------------
        uint32_t cell_addr[kSPL];
        const uint32_t cells = allocate_cells(t, stage_byte0, cell_addr);
        if (cells > kStageCells) {  // wave-uniform; what lies past the staging area turns slow: zeros, then slow_stores
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k)
                if (t.need[k] != 0 && cell_addr[k] + 16 * t.need[k] > stage_byte0 + 16 * kStageCells) {
                    t.need[k] = 0;
                    slowb |= 1u << k;
                }
        }
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.src2[k] = t.need[k] != 0 ? cell_addr[k] : t.src2[k];
        const bool tile_wide = tile_exc && __ballot(wide[0] || wide[1] || wide[2] || wide[3]) != 0;
        const bool tile_slow = __ballot(slowb != 0) != 0;
        const bool tile_big = __ballot((t.need[0] | t.need[1] | t.need[2] | t.need[3]) > 1u) != 0;
        if (CHAINED && last_tile) {  // a chained segment asks for the next block as soon as it knows where this one ends
            const uint64_t nb = tile_base + uint64_t(kSlotBytes) * end_slot;
            if (ch.more && nb + 1 + kChainBytes <= a.enc_bytes) chain_request(a.enc, nb, lane, ch);
        }
        // ---- the wait point of the tile: everything prefetched has landed — nothing has been stored yet,
        // so this is no wait for store acknowledgements ("+v": from here on the values are the asm's, not a
        // load's — nothing for the compiler to wait for later, behind the stores)
        SECTION(pf, 8, "8_wait");
        uint64_t raw3w = raw3;
        asm volatile("" : "+v"(raw3w));
        if (CHAINED) asm volatile("" : "+v"(ch.sel), "+v"(ch.data.x), "+v"(ch.data.y), "+v"(ch.data.z), "+v"(ch.data.w));
        // the heads of this tile's cold codewords into their cells (the integers start 4 bytes in); the tails
        // of the large ones requested: integers 6..13 into the same registers, 14 and 15 into one more each
        uint32_t t3[kSPL];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if (t.need[k] != 0) {
                *reinterpret_cast<u32x4*>(lds_rw + t.src2[k]) = hr.q[k];
                t.src2[k] += 4u + (wide[k] ? 1u : 0u);
            }
        if (tile_big) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t tail = c.tails_base + 32 * (dd.meta_base + cur.s[k]);
                if (t.need[k] > 1u) hr.q[k] = __builtin_amdgcn_raw_buffer_load_b128(c.rs_dict, tail, 0, 0);
                if (t.need[k] > 2u) t3[k] = __builtin_amdgcn_raw_buffer_load_b32(c.rs_dict, tail + 16, 0, 0);
            }
        }
        // (in-index docs part: the segment is one 256-posting block = one group of one tile; with a slow codeword in
        // it, or spread over two tiles — more than 256 slots: a block full of exceptions — it stays gaps and
        // the flagged fix-up (finalize_flagged_kernel, interpolative_tails_kernel) is told)
        const bool as_docids = block_base != nullptr && !tile_slow && produced == 0 && last_tile && t.total <= ROUNDS * GROUPS * 256;
        if (block_base != nullptr && !as_docids && lane == 0) *gaps_left = 1;
        expand_tile<ROUNDS, GROUPS>(t, plain, tile_wide, a.plus_one, as_docids ? block_base : nullptr, produced, c.lds, c.scratch, rs_out, lane, pf, [&]() {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) asm volatile("" : "+v"(hr.q[k]), "+v"(t3[k]));
            if (tile_big) {
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) {
                    if (t.need[k] > 1u) *reinterpret_cast<u32x4*>(lds_rw + t.src2[k] + 12) = hr.q[k];
                    if (t.need[k] > 2u) *reinterpret_cast<uint32_t*>(lds_rw + t.src2[k] + 28) = t3[k];
                }
            }
            wave_lds_fence();
            // ---- the next tile's metadata (its slots are already here): the hot codewords' from LDS, the cold
            // ones' heads from L2 — requested before this tile's stores, so that the wait for them at the top of
            // the next tile is no wait for the stores (vmcnt is one in-order counter for loads AND stores on
            // gfx950). The last tile of a segment has no successor.
            SECTION(pf, 3, "3_prefetch");
            if (!last_tile) {
                kc = classify(nxt, carry_out);
                request_metas(c, dd.hot_base, hot_k, dd.meta_base, nxt, mr, hr, kc.row & 15u);
            }
        });

        SECTION(pf, 10, "10_tail");
        if (tile_slow)
            slow_stores(narrow, c, t, slowb, a.plus_one, produced, n, a.enc + tile_base + uint64_t(kSPL * kSlotBytes) * lane, dd.hot_base,
                           hot_k, dd.meta_base, rs_out);
        SECTION(pf, 5, "10_rotate");
        produced += t.total;
        if (produced < n) tile_base += kTileBytes;

        // ---- rotate the pipeline ---------------------------------------------------
        cur = nxt;
        unpack_slots(narrow, CHAINED ? raw3w : raw2, nxt);
        raw2 = raw3w;
    }
    SECTION(pf, 13, "epilogue");
    return tile_base + uint64_t(kSlotBytes) * end_slot;
}

// One dword now: load and wait in one asm statement (the rare paths of decode_segment_v4: a load the compiler keeps
// books on, even on a path taken once in a million tiles, makes it place waits for "everything in flight" all over the
// loop — where control flow merges, its books take the rare path's registers for pending).
__device__ __forceinline__ uint32_t load_b32_now(const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    uint32_t v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(byte_off), "s"(rs) : "memory");
    return v;
}
// slow_stores for decode_segment_v4: the slots' values and the exception literals come in registers.
__device__ __forceinline__ void slow_stores_lean(const wave_ctx& c, const tile_slots& t, uint32_t slowb, uint32_t pos0, uint32_t seg_n,
                                                 const uint32_t (&sv)[kSPL], const uint32_t (&lit)[kSPL], uint32_t hot_base, uint32_t hot_k,
                                                 uint32_t meta_base, const __amdgpu_buffer_rsrc_t rs_out) {
    // the zeros must be in memory first: two stores of one wave to one address are only ordered by the wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        if ((slowb >> k) & 1u) {
            const uint32_t pos = pos0 + t.obase + t.off[k];
            if (sv[k] < 2) {  // an exception whose literal found no staging cell
                __builtin_amdgcn_raw_buffer_store_b32(lit[k], rs_out, 4 * pos, 0, DINT_STORE_AUX);
            } else {
                const uint32_t m = sv[k] < hot_k ? c.lds[hot_base + sv[k]] : load_b32_now(c.rs_dict, c.heads_base + 16 * (meta_base + sv[k]));
                const uint32_t goff = load_b32_now(c.rs_dict, c.goff_base + 4 * (meta_base + sv[k]));
                const uint32_t size = (m >> 24) + 1u, room = seg_n - pos;
                const uint32_t cnt = size < room ? size : room;
#pragma nounroll
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t v = load_b32_now(c.rs_dict, c.gtable_base + 4 * (goff + j));
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs_out, 4 * (pos + j), 0, DINT_STORE_AUX);
                }
            }
        }
    }
}

// ---- loads issued in inline asm, waited for by counted s_waitcnt (decode_segment_v4) ----------------------------------
// decode_segment leaves the waits to the compiler, and every one it places is a wait for everything the wave has in
// flight — the previous tile's stores, the far prefetch issued a moment ago. gfx950 completes a wave's buffer loads and
// stores in issue order and s_waitcnt vmcnt(N) waits for all but the N youngest (MI355X_MICROARCH.md), so a segment can
// ISSUE every load whose result outlives a phase in inline asm (the compiler does not know its destination is pending
// and adds no wait of its own) and wait for it with a counted s_waitcnt, also in asm. No register that a load is still
// writing may be touched in between (a copy would read it too early): tools/check_inflight.py scans the assembly.
__device__ __forceinline__ void issue_b128(u32x4& q, const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q) : "v"(byte_off), "s"(rs) : "memory");
}
__device__ __forceinline__ void reissue_b128(u32x4& q, const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {  // into a register that holds a value
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(q) : "v"(byte_off), "s"(rs) : "memory");
}
__device__ __forceinline__ void issue_b64(u32x2& q, const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(q) : "v"(byte_off), "s"(rs) : "memory");
}
__device__ __forceinline__ void issue_b32(uint32_t& q, const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off, uint32_t) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:16" : "=v"(q) : "v"(byte_off), "s"(rs) : "memory");
}
// ---- decode_segment_v4: the vroom kernel's segment (round 4) --------------------------------------------------------
// The same tile as decode_segment, its vector-memory traffic under explicit waits. A loop iteration is
//   C  heads AND tails of the tile whose front end runs next, the slots two tiles on   (requests only)
//   A  expansion and stores of the tile before
//   B  that front end: ONE wait for the dictionary — vmcnt(stores + 1): the far slots and this iteration's stores stay
//      in flight — classification (a tile ahead: payload slots ask for no head), sizes, cells, tables; at its end
//      vmcnt(0): the far slots have had a whole tile, the stores most of one — nothing is in flight across the back-edge.
// Whether a cold codeword needs its tail (more than 6 integers) is read in C from the long-entry bitmap of the LDS image,
// by the slot value alone, so both requests cross the whole expansion. (Round 3's first version of this, "lean", asked for
// the tails in the middle of B and drained everything at B's end, with a front end of its own: 4 % slower than this.)
// MEASURED (profiles/r04_v4_ab.txt, 1e9 postings, same process): 1.675 ms against decode_segment's 1.668 — the waits are
// gone from the wave and the kernel is no faster: it is bound by what the CU's waves share, not by what one
// wave waits for. Off by default (DINT_LEAN_SEGMENT=2 selects it; it costs the image 8 KB for the bitmap).
// A codeword of 15 or 16 integers that is cold (three staging cells) is written by slow_stores here: its last two
// integers would be a third request per slot.
struct tail_regs {
    u32x4 q[kSPL];   // integers 6..13
    uint32_t w[kSPL];  // integers 14 and 15
};
// all but the `keep` youngest vector-memory operations of this wave are done (keep <= 9; more: a longer wait)
__device__ __forceinline__ void wait_vmcnt_all_but(uint32_t keep, head_regs& hr, tail_regs& tr) {
    asm volatile(
        "s_cmp_lt_u32 %12, 4\n\t"
        "s_cbranch_scc1 .Ldv4_lo_%=\n\t"
        "s_cmp_lt_u32 %12, 6\n\t"
        "s_cbranch_scc1 .Ldv4_45_%=\n\t"
        "s_cmp_lt_u32 %12, 8\n\t"
        "s_cbranch_scc1 .Ldv4_67_%=\n\t"
        "s_cmp_eq_u32 %12, 8\n\t"
        "s_cbranch_scc1 .Ldv4_8_%=\n\t"
        "s_waitcnt vmcnt(9)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_8_%=:\n\t"
        "s_waitcnt vmcnt(8)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_67_%=:\n\t"
        "s_cmp_eq_u32 %12, 6\n\t"
        "s_cbranch_scc1 .Ldv4_6_%=\n\t"
        "s_waitcnt vmcnt(7)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_6_%=:\n\t"
        "s_waitcnt vmcnt(6)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_45_%=:\n\t"
        "s_cmp_eq_u32 %12, 4\n\t"
        "s_cbranch_scc1 .Ldv4_4_%=\n\t"
        "s_waitcnt vmcnt(5)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_4_%=:\n\t"
        "s_waitcnt vmcnt(4)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_lo_%=:\n\t"
        "s_cmp_lt_u32 %12, 2\n\t"
        "s_cbranch_scc1 .Ldv4_01_%=\n\t"
        "s_cmp_eq_u32 %12, 2\n\t"
        "s_cbranch_scc1 .Ldv4_2_%=\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_2_%=:\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_01_%=:\n\t"
        "s_cmp_eq_u32 %12, 1\n\t"
        "s_cbranch_scc1 .Ldv4_1_%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch .Ldv4_e_%=\n"
        ".Ldv4_1_%=:\n\t"
        "s_waitcnt vmcnt(1)\n"
        ".Ldv4_e_%=:"
        : "+v"(hr.q[0]), "+v"(hr.q[1]), "+v"(hr.q[2]), "+v"(hr.q[3]), "+v"(tr.q[0]), "+v"(tr.q[1]), "+v"(tr.q[2]), "+v"(tr.q[3]),
          "+v"(tr.w[0]), "+v"(tr.w[1]), "+v"(tr.w[2]), "+v"(tr.w[3])
        : "s"(keep)
        : "scc", "memory");
}

template <uint32_t ROUNDS, uint32_t GROUPS>
__device__ __forceinline__ uint64_t decode_segment_v4(const decode_args& a, const wave_ctx& c, const dict_desc& dd, uint64_t in_off,
                                                      uint32_t n, uint32_t* const out, prof_t& pf) {
    SECTION(pf, 11, "segment_prologue");
    constexpr uint32_t kCap = ROUNDS * GROUPS * 256;
    constexpr uint32_t kTileBytes = 2 * kTileSlots;
    n = uniform(n);
    const uint16_t* const rows = c.cls;
    const uint32_t lane = c.lane;
    const uint32_t hot_k = dd.hot_k, hot_base = dd.hot_base, meta_base = dd.meta_base;
    uint32_t* const out_u = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(out)));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, int(n * 4), 0x00020000);
    uint8_t* const lds_rw = reinterpret_cast<uint8_t*>(const_cast<uint32_t*>(c.lds));
    const uint8_t* const lds_bytes = lds_rw;
    uint8_t* const fw = reinterpret_cast<uint8_t*>(fw_of(c.scratch));
    uint8_t* const delta = reinterpret_cast<uint8_t*>(delta_of(c.scratch));
    const uint32_t stage_byte0 = uint32_t(reinterpret_cast<const uint8_t*>(stage_of(c.scratch)) - lds_rw);
    const uint32_t* const long_bits = c.lds + a.dict.long_bitmap_word;
    const uint64_t in_off_u = uniform64(in_off);
    const uint64_t seg_room = in_off_u <= a.enc_bytes ? a.enc_bytes - in_off_u : 0;
    const __amdgpu_buffer_rsrc_t rs_seg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.enc) + (in_off_u <= a.enc_bytes ? in_off_u : a.enc_bytes), 0,
        int(seg_room < 0x7FFFFFF0ull ? uint32_t(seg_room) : 0x7FFFFFF0u), 0x00020000);
    auto whole = [&](uint64_t tile_byte) { return tile_byte <= a.enc_bytes && a.enc_bytes - tile_byte >= kTileBytes; };  // wave-uniform
    auto issue_slots = [&](u32x2& raw, uint64_t tile_byte) { issue_b64(raw, rs_seg, uint32_t(tile_byte - in_off_u) + 8 * lane); };
    // A tile that is not wholly inside the buffer (the stream's last ones) reads zeros for the dwords past the end — the
    // descriptor clips per dword — and is read again, exactly, by load_lane_slots: compiler-tracked loads, waited for on the
    // spot. Every call sits right behind a wait for everything in flight, so the wait the compiler adds costs nothing.
    auto fix_slots = [&](u32x2& raw, uint64_t tile_byte) {
        if (__builtin_expect(!whole(tile_byte), 0)) {
            const uint64_t q = load_lane_slots(false, a.enc, tile_byte, lane, a.enc_bytes);
            raw.x = uint32_t(q), raw.y = uint32_t(q >> 32);
        }
    };
    auto slot_of = [](const u32x2& raw, uint32_t k) { return ((k < 2 ? raw.x : raw.y) >> (16 * (k & 1))) & 0xFFFFu; };
    // classification of a tile's slots (decode_segment): its row per lane, the carry it hands on, what kind of tile it is
    struct tile_class {
        uint32_t row;
        uint32_t carry_out;
        bool special, tile_exc;
    };
    auto classify = [&](const u32x2& raw, uint32_t carry_in) -> tile_class {
        tile_class k{kPlainRow, 0u, false, false};
        uint32_t sv[kSPL];
#pragma unroll
        for (uint32_t j = 0; j != kSPL; ++j) sv[j] = slot_of(raw, j);
        uint32_t smin = sv[0];
#pragma unroll
        for (uint32_t j = 1; j != kSPL; ++j) smin = smin < sv[j] ? smin : sv[j];
        k.special = __builtin_expect(__ballot(smin < 2) != 0 || carry_in != 0, 0);
        if (k.special) {
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t j = kSPL; j-- != 0;) lo = 3 * lo + (2u - (sv[j] < 2 ? sv[j] : 2u));
            uint32_t st_in = lane == 0 ? carry_in : 0u;
            uint32_t row;
            for (;;) {
                row = rows[st_in * 81 + lo];
                uint32_t prev = from_lane_below((row >> 8) & 7u);
                if (lane == 0) prev = carry_in;
                if (__ballot(prev != st_in) == 0) break;
                st_in = prev;
            }
            k.row = row;
            k.carry_out = readlane((row >> 8) & 7u, 63);
            k.tile_exc = __ballot((row & 0xF0u) != 0) != 0;
        }
        return k;
    };

    u32x2 rawA, rawB;
    uint64_t slot_byte = in_off_u;  // first byte of the tile whose slots were requested last (wave-uniform)
    issue_slots(rawA, slot_byte);
    slot_byte += kTileBytes;
    issue_slots(rawB, slot_byte);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawA), "+v"(rawB) : : "memory");
    fix_slots(rawA, in_off_u);
    fix_slots(rawB, slot_byte);
    tile_class kc = classify(rawA, 0u);

    uint32_t produced = 0, end_slot = 0;
    uint64_t tile_base = in_off_u;  // byte offset of slot 0 of the tile in the front end (wave-uniform)
    bool pending = false;           // a tile's tables are built, its expansion is due
    uint32_t pend_total = 0, pend_out = 0;
    bool more = true;
    MARK("loop_top");
    for (;;) {
        // ---- C: this tile's heads and tails, the slots two tiles on -------------------------------------------------
        head_regs hr;   // (deliberately uninitialised: each register is written and read under the same lane predicate)
        tail_regs tr;
        u32x2 rawC;
        uint32_t loads = 0;  // vector-memory loads this C issued (wave-uniform): the slots' and one per slot position that has any
        if (more) {
            SECTION(pf, 3, "3_prefetch");
            uint32_t sv[kSPL];
            bool cold[kSPL];
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                sv[k] = slot_of(rawA, k);
                cold[k] = sv[k] >= hot_k && ((kc.row >> k) & 1u) == 0;  // (a payload slot asks for nothing)
            }
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                if (__ballot(cold[k]) != 0) {  // wave-uniform: the count below is exact
                    if (cold[k]) issue_b128(hr.q[k], c.rs_dict, c.heads_base + 16 * (meta_base + sv[k]));
                    ++loads;
                }
            }
            // which of the cold codewords have a tail: one bit per codeword in the image
            uint32_t bw[kSPL];
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) bw[k] = long_bits[sv[k] >> 5];
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const bool lng = cold[k] && ((bw[k] >> (sv[k] & 31u)) & 1u) != 0;
                if (__ballot(lng) != 0) {
                    // (integers 6..13, and 14..15 — the same 64-byte line: the second request rides on the first's)
                    if (lng) {
                        issue_b128(tr.q[k], c.rs_dict, c.tails_base + 32 * (meta_base + sv[k]));
                        issue_b32(tr.w[k], c.rs_dict, c.tails_base + 32 * (meta_base + sv[k]), 16);
                    }
                    loads += 2;
                }
            }
            slot_byte += kTileBytes;
            issue_slots(rawC, slot_byte);
        }
        // ---- A: expansion of the tile whose tables phase B built ---------------------------------------------------
        uint32_t stores = 0;
        if (pending) {
            SECTION(pf, 9, "9_expand");
            expand_batch<ROUNDS, GROUPS>(pend_total, pend_out, lds_bytes, fw, delta, rs_out, lane, 0u, nullptr);
            *reinterpret_cast<uint32_t*>(fw + 8 * lane) = 0;  // the flag words go back to zero for the next tile
            wave_lds_fence();
            stores = (pend_total + 255u) >> 8;
            pending = false;
        }
        if (!more) break;

        // ---- B: front end of the tile ------------------------------------------------------------------------------
        SECTION(pf, 1, "1_classify");
        tile_regs cur;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) cur.s[k] = slot_of(rawA, k);
        meta_regs mr;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) mr.h[k] = c.lds[hot_base + (cur.s[k] < hot_k ? cur.s[k] : hot_k)];
        // heads and tails are there: the slots requested behind them and this iteration's stores may stay in flight
        wait_vmcnt_all_but(uniform(stores) + 1u, hr, tr);
        (void)loads;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) cur.m[k] = cur.s[k] < hot_k ? mr.h[k] : hr.q[k].x;
        const bool special = kc.special;
        const uint32_t row = kc.row;
        const uint32_t carry_out = kc.carry_out;
        const bool tile_exc = kc.tile_exc;

        SECTION(pf, 2, "2_sizes");
        tile_slots t;
        uint32_t lv1[kSPL] = {1u, 1u, 1u, 1u};
        if (special) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                cur.m[k] &= uint32_t(__builtin_amdgcn_sbfe(~row, k, 1));
                lv1[k] = __builtin_amdgcn_ubfe(~row, k, 1);
            }
        }
        const bool tile_slow_dict = __ballot(((cur.m[0] | cur.m[1] | cur.m[2] | cur.m[3]) & kMetaSlow) != 0) != 0;
        uint32_t e[kSPL];  // size - 1
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            e[k] = cur.m[k] >> 24;
            t.need[k] = __builtin_amdgcn_ubfe(cur.m[k], 20, 2);
            t.src2[k] = cur.m[k] & kMetaOffMask;
        }
        bool wide[kSPL] = {false, false, false, false};
        if (tile_exc) {  // the literals ride into their cells in the second word of the slots' head registers (decode_segment)
            const uint32_t p01 = rawA.x, p23 = rawA.y;
            const uint32_t next0 = readlane(rawB.x, 0);
            uint32_t nlo = from_lane_above(p01);
            if (lane == 63) nlo = next0;
            uint32_t w32[kSPL];
            w32[0] = __builtin_amdgcn_alignbit(p23, p01, 16);
            w32[1] = p23;
            w32[2] = __builtin_amdgcn_alignbit(nlo, p23, 16);
            w32[3] = nlo;
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t excM = uint32_t(__builtin_amdgcn_sbfe(row, 4 + k, 1));
                const uint32_t lit = w32[k] & ((0u - cur.s[k]) | 0xFFFFu) & excM;
                hr.q[k].y = (hr.q[k].y & ~excM) | lit;
                wide[k] = lit > 0xFFFFu;
            }
        }
        t.off[0] = 0;
        t.off[1] = e[0] + lv1[0];
        t.off[2] = t.off[1] + e[1] + lv1[1];
        t.off[3] = t.off[2] + e[2] + lv1[2];
        t.lsum = t.off[3] + e[3] + lv1[3];
        uint32_t hdrcnt = 4;
        uint32_t pincl;
        if (special) {
            hdrcnt = uint32_t(__builtin_popcount(~row & 15u));
            pincl = wave_inclusive_sum((hdrcnt << 24) | t.lsum);
        } else {
            pincl = wave_inclusive_sum(t.lsum);
        }
        t.obase = (pincl & 0xFFFFFFu) - t.lsum;
        const uint32_t remaining = n - produced;
        t.total = readlane(pincl, 63) & 0xFFFFFFu;
        const bool last_tile = t.total >= remaining;
        const bool plain = !special && !last_tile && t.total <= kCap;
        uint32_t slowb = 0;  // bit k: slot k goes through slow_stores
        if (!plain) {
            t.row = row;
            t.liveb = ~row & 15u;
            t.rbase = special ? (pincl - ((hdrcnt << 24) | t.lsum)) >> 24 : 4 * lane;
            t.nlive = hdrcnt;
            if (last_tile) {
                t.total = remaining;
                uint32_t cand = 0, lb = 0;
                t.nlive = 0;
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) {
                    const uint32_t pos = t.obase + t.off[k];
                    const bool act = ((t.liveb >> k) & 1u) != 0 && pos < remaining;
                    if (act) {
                        const bool exc = ((row >> (4 + k)) & 1u) != 0;
                        cand = kSPL * lane + k + 1 + (exc ? cur.s[k] + 1 : 0u);
                        ++t.nlive;
                        lb |= 1u << k;
                    } else {
                        t.need[k] = 0;
                    }
                }
                t.liveb = lb;
                t.lsum = t.obase < remaining ? (t.obase + t.lsum < remaining ? t.lsum : remaining - t.obase) : 0u;
                const uint64_t am = __ballot(cand != 0);
                end_slot = readlane(cand, 63u - uint32_t(__builtin_clzll(am | 1ull)));
            }
            if (tile_slow_dict) {
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) slowb |= ((cur.m[k] >> 22) & (t.liveb >> k) & ~(row >> (4 + k)) & 1u) << k;
            }
        } else if (tile_slow_dict) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) slowb |= ((cur.m[k] >> 22) & 1u) << k;
        }
        uint32_t cell_addr[kSPL];
        const uint32_t cells = allocate_cells(t, stage_byte0, cell_addr);
        if (cells > kStageCells) {  // wave-uniform; what lies past the staging area turns slow: zeros, then slow_stores
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k)
                if (t.need[k] != 0 && cell_addr[k] + 16 * t.need[k] > stage_byte0 + 16 * kStageCells) {
                    t.need[k] = 0;
                    slowb |= 1u << k;
                }
        }
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) t.src2[k] = t.need[k] != 0 ? cell_addr[k] : t.src2[k];
        const bool tile_slow = __ballot(slowb != 0) != 0;
        const bool tile_big = __ballot((t.need[0] | t.need[1] | t.need[2] | t.need[3]) > 1u) != 0;
        SECTION(pf, 8, "8_wait");
        // heads (the integers start 4 bytes into a cell) and tails (integers 6..13 behind them) into their cells
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if (t.need[k] != 0) {
                *reinterpret_cast<u32x4*>(lds_rw + t.src2[k]) = hr.q[k];
                t.src2[k] += 4u + (wide[k] ? 1u : 0u);
            }
        if (tile_big) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                if (t.need[k] > 1u) *reinterpret_cast<u32x4*>(lds_rw + t.src2[k] + 12) = tr.q[k];
                if (t.need[k] > 2u) *reinterpret_cast<uint32_t*>(lds_rw + t.src2[k] + 28) = tr.w[k];
            }
        }
        SECTION(pf, 4, "4_tables");
        const bool one_batch = !tile_slow && t.total <= kCap;  // wave-uniform
        if (__builtin_expect(one_batch, 1)) {
            if (plain) tables_plain(t, fw, delta, lane);
            else tables_general(t, fw, delta, t.lsum != 0, 0u, 0u, lane);
            wave_lds_fence();
            pending = true;
            pend_total = t.total;
            pend_out = produced;
        } else {
            // more than one expansion batch (a tile full of long runs), or a slow codeword: finished here, the old way
            wave_lds_fence();
            expand_tile<ROUNDS, GROUPS>(t, plain, true, 0u, nullptr, produced, c.lds, c.scratch, rs_out, lane, pf, []() {});
            SECTION(pf, 10, "10_tail");
            if (tile_slow) {
                uint32_t lit[kSPL];
#pragma unroll
                for (uint32_t k = 0; k != kSPL; ++k) lit[k] = hr.q[k].y;  // (an exception's literal rode in here)
                slow_stores_lean(c, t, slowb, produced, n, cur.s, lit, hot_base, hot_k, meta_base, rs_out);
            }
        }
        SECTION(pf, 5, "10_rotate");
        produced += t.total;
        more = produced < n;
        if (more) {
            tile_base += kTileBytes;
            kc = classify(rawB, carry_out);
        }
        // everything in flight is there: the far slots (a tile old) and the previous tile's stores (most of a tile old)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawC) : : "memory");
        if (more) {
            fix_slots(rawC, slot_byte);
            rawA = rawB;
            rawB = rawC;
        }
    }
    SECTION(pf, 13, "epilogue");
    return tile_base + 2ull * end_slot;
}

// A single-dictionary unit (rectangular or packed: the streams are byte-identical,
// only the dictionary source layout differed on the host) is one 16-bit segment.
// (Chaining such units like the blocks of a multi-dictionary unit — the next unit's first tiles
// requested while the current one is expanded — was measured in round 1: 5 % slower.)
template <int LEAN>  // 0: decode_segment; 2: decode_segment_v4 (the vroom kernel only)
__device__ __forceinline__ void decode_unit_single(const decode_args& a, const wave_ctx& c, uint64_t unit_index, prof_t& pf) {
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || n > kMaxUnitInts || out_off > a.out_capacity || a.out_capacity - out_off < n || (a.only_full && n != 256)) return;
    const uint64_t in_off = uniform64(up->in_off);
    if (LEAN) {  // (the vroom kernel: plain d-gaps)
        const uint64_t end = decode_segment_v4<kRounds, kGroups>(a, c, a.dict.first, in_off, n, a.out + out_off, pf);
        if (a.end_off && c.lane == 0) a.end_off[unit_index] = end;
        return;
    }
    chain_io ch{};
    const uint64_t end = decode_segment<16, kRounds, kGroups, 0>(a, c, a.dict.first, in_off, n, a.out + out_off, ch, pf, false, false,
                                                                 a.unit_base ? a.unit_base + unit_index : nullptr,
                                                                 a.unit_base ? a.gaps_left + unit_index : nullptr);
    if (a.end_off && c.lane == 0) a.end_off[unit_index] = end;
}

// ---- bundles of tiny units -----------------------------------------------------------------------
// Nine in ten posting lists of a Gov2-shaped collection hold at most 64 postings; as units of
// their own they are 0.2 % of the integers and a fifth of the tiles (each wave tile-step costs the
// same whether 5 or 250 of its slot positions are used). A bundle packs up to 64 consecutive tiny
// units into ONE tile: unit i takes ceil(bytes_i / 8) whole lanes, the front end runs segmented
// (per-lane slot address, carries cut at unit starts, sizes clamped at each unit's n), and because
// the units' outputs are consecutive the expansion is the ordinary one over the bundle's outputs.
constexpr uint32_t kBundleMaxInts = 256;   // a unit is bundled only if it decodes to at most this many integers
constexpr uint32_t kBundleMaxBytes = 504;  // ... and spans at most this many stream bytes (63 lanes of 8)

// Host-launched before the decode kernel: sched[i] for every unit (see decode_args::sched), and what the decode
// kernel's bundle path reads instead of the unit table — per unit one 16-byte record, per chunk of 64 units the
// stream / output offsets of its first unit:
//   urec[i] = {in_off - in0, out_off - out0, packed, 0},  packed = (n - 1) & 255 | lanes << 8 | selector << 14 | sched << 18
// (lanes: 8-byte — 8-bit slots: 4-byte — lanes of the tile the unit takes; selector: a multi-dictionary block's
// selector byte, read here so that the decode kernel does not wait for it). One workgroup per 256 units = 4 chunks.
constexpr uint32_t kChunkUnits = 64;
__global__ __launch_bounds__(256) void bundle_schedule_kernel(const dint_unit* units, const uint32_t* spans, uint64_t n_units,
                                                              const uint8_t* enc, uint64_t enc_bytes, uint64_t out_capacity,
                                                              uint32_t only_full, uint32_t multi, uint8_t* sched,
                                                              uint32_t* block_items, u32x4* urec, uint64_t* cbase) {
    __shared__ uint32_t lanes[256], pre[256];  // lanes: the unit's lanes | its integers << 8
    __shared__ uint32_t place[256];            // first-fit launches: bundle id | first lane << 8 | member index << 16
    __shared__ uint8_t start[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t i = uint64_t(blockIdx.x) * 256 + tid;
    uint32_t L = 0, sel = 0;
    uint64_t in = 0, out = 0;
    uint32_t n = 0;
    if (i < n_units) {
        in = units[i].in_off;
        out = units[i].out_off;
        n = units[i].n;
    }
    // the chunk's base: its first unit (a chunk that exists has one)
    const uint64_t in0 = (uint64_t(uint32_t(__shfl(uint32_t(in >> 32), 0))) << 32) | uint32_t(__shfl(uint32_t(in), 0));
    const uint64_t out0 = (uint64_t(uint32_t(__shfl(uint32_t(out >> 32), 0))) << 32) | uint32_t(__shfl(uint32_t(out), 0));
    if (i < n_units) {
        const uint64_t nxt = spans ? in + spans[i] : (i + 1 < n_units ? units[i + 1].in_off : enc_bytes);
        // (in-index launches decode only the full blocks; the other units stay on their own and are skipped)
        if (n >= 1 && n <= kBundleMaxInts && (!only_full || n == 256) && nxt > in && nxt - in <= kBundleMaxBytes &&
            nxt <= enc_bytes && out <= out_capacity && out_capacity - out >= n && in >= in0 && in - in0 <= 0xFFFFFFFFull && out >= out0 &&
            out - out0 < (multi ? (1ull << 28) : (1ull << 32))) {
            if (!multi) {
                const uint32_t l = uint32_t((nxt - in + 7) >> 3);
                if (l <= kWave - 1 && in + 8ull * l <= enc_bytes) L = l;  // every lane's 8-byte load stays inside the buffer
            } else {
                // a multi unit of <= 256 integers is one block: selector byte, then 16- or 8-bit slots
                // (4 to a lane: 8 or 4 bytes); every lane still loads 8 bytes
                sel = enc[in];
                const uint32_t stride = sel >= 6 ? 4u : 8u;
                const uint32_t l = uint32_t((nxt - in - 1 + stride - 1) / stride);
                if (sel < 12 && l >= 1 && l <= kWave - 1 && in + 1 + uint64_t(stride) * l + 8 <= enc_bytes) L = l;
            }
        }
    }
    lanes[tid] = L | (n << 8);  // (n <= 256 where L != 0)
    // does this unit continue the previous one (both eligible, outputs consecutive)?
    pre[tid] = (L != 0 && tid != 0 && i < n_units && out == units[i - 1].out_off + units[i - 1].n) ? 1u : 0u;
    __syncthreads();
    // greedy packing, one thread per chunk: a bundle takes units while their lanes fit a wave (the chain through
    // in_use is the only serial part, the LDS reads are unrolled ahead of it)
    if (multi) {
        // (the multi-dictionary kernel's bundles are packed by bundle_pack_kernel, behind this kernel: here every unit
        // that fits a tile is a bundle member, the others are the unit queue's)
        start[tid] = L != 0 ? 0 : 1;
        place[tid] = 0;
    } else if ((tid & 63u) == 0) {
        uint32_t in_use = 0, ints = 0, prev_l = 0;
#pragma unroll 16
        for (uint32_t j = tid; j != tid + 64; ++j) {
            const uint32_t l = lanes[j] & 255u, m = lanes[j] >> 8;
            // (a bundle decodes to at most kMaxCap integers: one expansion batch; in-index: 8 blocks, groups = blocks)
            const bool cont = l != 0 && prev_l != 0 && pre[j] != 0 && in_use + l <= kWave && ints + m <= kMaxCap;
            start[j] = cont ? 0 : 1;
            in_use = cont ? in_use + l : l;
            ints = cont ? ints + m : m;
            prev_l = l;
        }
    }
    __syncthreads();
    uint32_t c = 0;
    if (start[tid] != 0 && i < n_units) {
        c = 1;
        if (L != 0 && !multi)
            while (tid + c < 256 && ((tid + c) & 63u) != 0 && i + c < n_units && !start[tid + c]) ++c;
    }
    // Multi-dictionary kernel: a unit that fits a tile goes through the bundle path even when it has no neighbour
    // to share the tile with, the unit queue gets the others. Single-dictionary kernel: the unit queue hands out
    // everything, bundles (of two units or more) included (see decode_kernel_body).
    const bool alone = c == 1 && (L == 0 || !multi);
    const bool item = alone || (!multi && c > 1);
    const int n_items = __syncthreads_count(item);
    if (tid == 0) block_items[blockIdx.x] = uint32_t(n_items);
    if (i >= n_units) return;
    sched[i] = uint8_t(item ? c : 0);
    u32x4 r;
    r.x = uint32_t(in - in0);
    r.y = uint32_t(out - out0);
    r.z = ((n - 1u) & 255u) | (L << 8) | (sel << 14) | (c << 18);
    r.w = multi ? place[tid] : 0u;
    urec[i] = r;
    if ((tid & 63u) == 0) {
        cbase[2 * (i >> 6)] = in0;
        cbase[2 * (i >> 6) + 1] = out0;
    }
}

// The multi-dictionary kernel's bundles need not be runs of consecutive units (every member's 256 outputs are an
// expansion group of their own, stored where the unit says): FIRST FIT over a few open bundles — a tile's cost
// does not depend on how full it is, and units of 15 to 40 lanes packed in order fill 51 of 64 lanes, first fit 57.
// One thread per chunk, 64 chunks to a wave (the packing of a chunk is a serial walk over its 64 units: as a loop of
// one lane per wave inside bundle_schedule_kernel it took 1.5 ms for 2e7 units; here the 64 lanes of a wave walk 64
// chunks together). Rewrites the units' records: bundle id | first lane << 8 | member index << 16 in .w, "first
// member" in the count field of .z.
__global__ __launch_bounds__(64) void bundle_pack_kernel(u32x4* urec, uint64_t n_units) {
    __shared__ uint32_t zs[64][65];  // the chunk's packed words, a row per thread (odd stride: conflict-free)
    constexpr uint32_t kOpen = DINT_FF_OPEN;
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t(blockIdx.x) * 64 + tid) * kChunkUnits;
    for (uint32_t j = 0; j != kChunkUnits; ++j) zs[tid][j] = base + j < n_units ? urec[base + j].z : 0u;
    uint32_t used[kOpen], mem[kOpen], id[kOpen], n_open = 0, next_id = 0;
#pragma unroll
    for (uint32_t k = 0; k != kOpen; ++k) used[k] = 0, mem[k] = 0, id[k] = 0;
    for (uint32_t j = 0; j != kChunkUnits; ++j) {
        const uint32_t z = zs[tid][j];
        const uint32_t l = (z >> 8) & 63u;
        uint32_t b = kOpen;
#pragma unroll
        for (uint32_t k = 0; k != kOpen; ++k)
            if (b == kOpen && k < n_open && used[k] + l <= kWave && mem[k] < 8) b = k;
        const bool fresh = b == kOpen;  // a new bundle: in a free slot, or in place of the fullest open one
        if (fresh) {
            b = n_open < kOpen ? n_open : 0u;
            if (n_open == kOpen) {
                uint32_t most = used[0];
#pragma unroll
                for (uint32_t k = 1; k != kOpen; ++k)
                    if (used[k] > most) most = used[k], b = k;
            }
        }
        uint32_t pl = 0;
#pragma unroll
        for (uint32_t k = 0; k != kOpen; ++k)
            if (k == b && l != 0) {
                if (fresh) id[k] = next_id, used[k] = 0, mem[k] = 0;
                pl = id[k] | (used[k] << 8) | (mem[k] << 16) | (mem[k] == 0 ? 1u << 31 : 0u);
                used[k] += l;
                mem[k] += 1;
            }
        if (fresh && l != 0) {
            next_id += 1;
            n_open += n_open < kOpen ? 1u : 0u;
        }
        zs[tid][j] = pl;  // (the row is this thread's own)
    }
    for (uint32_t j = 0; j != kChunkUnits; ++j) {
        const uint32_t pl = zs[tid][j];
        if (base + j < n_units && pl != 0) {  // (a member: its first-lane field or its "first" bit is set... or both are 0 only for member 0 of bundle 0 at lane 0, which has the bit)
            u32x4 r = urec[base + j];
            r.z = (r.z & ~(127u << 18)) | ((pl >> 31) << 18);
            r.w = pl & 0x7FFFFFFFu;
            urec[base + j] = r;
        }
    }
}

// Work items of the unit queue = the units with sched != 0. block_items -> exclusive offsets (one
// workgroup), then every block of 256 units writes its items.
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
                                                           uint32_t* items, uint8_t* item_cnt) {
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
    if (f) {
        const uint32_t at = block_offsets[blockIdx.x] + pre[tid] - 1;
        items[at] = uint32_t(i);
        item_cnt[at] = sched[i];
    }
}

// ---- a bundle: one tile over `cnt` (2..64) consecutive tiny units starting at unit u0. MULTI: every unit is one
// block of a multi-dictionary stream — its selector byte picks the dictionary and the slot width, per unit, hence
// per lane.
// What a lane knows about its part of a bundle's tile (bundle_map), and the wave about the bundle:
struct bundle_lane {
    uint32_t par;        // its unit's n - 1 | first lane << 8 | first output (inside the bundle) << 14 | 8-bit slots << 28 | dictionary << 29
    uint32_t slot_rel;   // its first slot: byte offset from the chunk's in0
    uint32_t seg;        // its unit: member index inside the bundle (| the unit's lane in its chunk << 8: first-fit bundles)
};
struct bundle_head {     // wave-uniform ...
    uint64_t in0;        // the chunk's stream base
    uint64_t out0;       // the bundle's first output (absolute, integers); first-fit bundles: the chunk's output base
    uint32_t u0, cnt;    // first unit (first-fit bundles: the chunk's first unit), units
    uint32_t used, total;  // lanes in use (<= 64), integers (first-fit bundles: 256 per member, see below)
    // ... except, for first-fit bundles, what lane g knows about member g (= output group g): its unit's lane in the
    // chunk | its integers << 8, and its outputs' offset from the chunk's output base
    uint32_t g_unit_n, g_rel_out;
};



// Lanes and outputs of each unit by one packed scan over the units' records (one per lane, from the chunk's
// registers: unit u0 + lane is chunk lane p + lane). Registers and cross-lane operations only: it runs in the
// middle of the previous bundle's tile.
template <bool MULTI>
__device__ __forceinline__ void bundle_map(const wave_ctx& c, const u32x4& rc, uint32_t chunk, uint32_t p, uint32_t cnt, uint64_t in0,
                                           uint64_t chunk_out0, bundle_lane& bl, bundle_head& bh) {
    const uint32_t lane = c.lane;
    const bool has = lane < cnt;
    const int src = int((p + lane) & 63u);
    const uint32_t my_rel = uint32_t(__shfl(rc.x, src));
    const uint32_t my_pk = uint32_t(__shfl(rc.z, src));
    const uint32_t my_n = has ? (my_pk & 255u) + 1u : 0u;
    const uint32_t my_lanes = has ? (my_pk >> 8) & 63u : 0u;
    const uint32_t pk0 = my_n | (my_lanes << 16);
    const uint32_t inc0 = wave_inclusive_sum(pk0);
    const uint32_t my_lane0 = (inc0 - pk0) >> 16;      // first lane of this lane's unit
    bh.used = readlane(inc0, 63) >> 16;
    bh.total = readlane(inc0, 63) & 0xFFFFu;
    bh.in0 = in0;
    bh.out0 = chunk_out0 + readlane(rc.y, p);
    bh.u0 = chunk * kChunkUnits + p;
    bh.cnt = cnt;
    // lane -> unit: a bit per unit at its first lane (distinct bits: their sum is their OR), units before a lane = bits below it
    uint32_t mlo, mhi;
    if (cnt <= 12) {  // (wave-uniform; the usual bundle: a handful of blocks) the bits gathered on the scalar side
        uint64_t heads = 0;
        for (uint32_t m = 0; m != cnt; ++m) heads |= 1ull << (readlane(inc0 - pk0, m) >> 16);
        mlo = uint32_t(heads), mhi = uint32_t(heads >> 32);
    } else {
        const uint32_t hlo = has && my_lane0 < 32 ? 1u << my_lane0 : 0u, hhi = has && my_lane0 >= 32 ? 1u << (my_lane0 - 32) : 0u;
        mlo = readlane(wave_inclusive_sum(hlo), 63), mhi = readlane(wave_inclusive_sum(hhi), 63);
    }
    const uint32_t own = lane < 32 ? (mlo >> lane) & 1u : (mhi >> (lane - 32)) & 1u;
    const uint32_t seg = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + own - 1u;  // lane 0 is always a head
    const int sl_ = int(seg);
    const uint32_t seg_rel = uint32_t(__shfl(my_rel, sl_));
    const uint32_t seg_pk = uint32_t(__shfl(my_pk, sl_));
    // (first lane and first output of the unit: one value, one shuffle)
    const uint32_t seg_at = uint32_t(__shfl(inc0 - pk0, sl_));
    const uint32_t seg_lane0 = seg_at >> 16, seg_out0 = seg_at & 0xFFFFu;
    const uint32_t sel = MULTI ? (seg_pk >> 14) & 15u : 0u;
    const uint32_t narrow = sel >= 6 ? 1u : 0u;
    const uint32_t dict = narrow ? sel - 6 : sel;
    bl.par = (seg_pk & 255u) | (seg_lane0 << 8) | (seg_out0 << 14) | (narrow << 28) | (dict << 29);
    bl.slot_rel = seg_rel + (MULTI ? 1u : 0u) + (narrow ? 4u : 8u) * (lane - seg_lane0);
    bl.seg = seg;
}

// The same for a first-fit bundle: the units of the chunk whose record names bundle `b` (their first lane and member
// index come from the schedule), gathered on the scalar side — at most 8.
__device__ __forceinline__ void bundle_map_first_fit(const wave_ctx& c, const u32x4& rc, uint32_t chunk, uint32_t b, uint64_t in0,
                                                     uint64_t chunk_out0, bundle_lane& bl, bundle_head& bh) {
    const uint32_t lane = c.lane;
    uint64_t members = __ballot(((rc.z >> 8) & 63u) != 0 && (rc.w & 255u) == b);
    uint64_t heads = 0, units_of = 0;  // a bit at every member's first lane; member m's chunk lane in byte m
    uint32_t cnt = 0, used = 0;
    while (members != 0 && cnt != 8) {  // (in unit order = member order = lane order)
        const uint32_t u = uint32_t(__builtin_ctzll(members));
        members &= members - 1;
        const uint32_t lane0 = (readlane(rc.w, u) >> 8) & 63u;
        heads |= 1ull << lane0;
        units_of |= uint64_t(u) << (8 * cnt);
        used = lane0 + ((readlane(rc.z, u) >> 8) & 63u);
        ++cnt;
    }
    bh.used = used;
    bh.total = 256 * cnt;
    bh.in0 = in0;
    bh.out0 = chunk_out0;
    bh.u0 = chunk * kChunkUnits;
    bh.cnt = cnt;
    const uint32_t mlo = uint32_t(heads), mhi = uint32_t(heads >> 32);
    const uint32_t own = lane < 32 ? (mlo >> lane) & 1u : (mhi >> (lane - 32)) & 1u;
    const uint32_t seg = (__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + own - 1u) & 7u;  // lane 0 is a head
    const uint32_t unit = uint32_t(units_of >> (8 * seg)) & 63u;
    const uint32_t seg_rel = uint32_t(__shfl(rc.x, int(unit)));
    const uint32_t seg_pk = uint32_t(__shfl(rc.z, int(unit)));
    const uint32_t seg_lane0 = (uint32_t(__shfl(rc.w, int(unit))) >> 8) & 63u;
    const uint32_t sel = (seg_pk >> 14) & 15u;
    const uint32_t narrow = sel >= 6 ? 1u : 0u;
    const uint32_t dict = narrow ? sel - 6 : sel;
    bl.par = (seg_pk & 255u) | (seg_lane0 << 8) | ((256u * seg) << 14) | (narrow << 28) | (dict << 29);
    bl.slot_rel = seg_rel + 1u + (narrow ? 4u : 8u) * (lane - seg_lane0);
    bl.seg = seg | (unit << 8);
    // lane g: member g (the shuffles with every lane enabled: a cross-lane read under `lane < cnt` finds the lanes
    // outside the branch silent)
    const uint32_t gu = uint32_t(units_of >> (8 * (lane & 7u))) & 63u;
    const uint32_t g_pk = uint32_t(__shfl(rc.z, int(gu)));
    bh.g_rel_out = uint32_t(__shfl(rc.y, int(gu)));
    bh.g_unit_n = gu | ((lane < cnt ? (g_pk & 255u) + 1u : 0u) << 8);
}

// the lane's 8 stream bytes (the schedule made sure they lie inside the buffer)
__device__ __forceinline__ uint64_t bundle_raw(const decode_args& a, const bundle_lane& bl, const bundle_head& bh, uint32_t lane) {
    uint64_t raw = 0;
    if (lane < bh.used) {
        const u32x2 r = reinterpret_cast<const u32x2_a1*>(a.enc + bh.in0 + bl.slot_rel)->v;
        raw = (uint64_t(r.y) << 32) | r.x;
    }
    return raw;
}

// The tile of a mapped bundle whose stream bytes are on their way (`raw_in`). `issue_next` runs once, behind the
// tile's last wait and before its gathers and stores: the caller requests the next bundle's stream bytes there,
// so that they travel while this tile is expanded and stored.
template <bool MULTI, class IssueNext>
__device__ __forceinline__ void bundle_process(const decode_args& a, const wave_ctx& c, const bundle_lane& bl, const bundle_head& bh,
                                               uint64_t raw_in, prof_t& pf, IssueNext&& issue_next) {
    SECTION(pf, 12, "bundle_front");
    const uint32_t lane = c.lane;
    uint32_t* const scratch = c.scratch;
    const uint32_t* const lds = c.lds;
    const uint16_t* const cls = c.cls;
    const uint64_t u0 = bh.u0;
    const uint32_t cnt = bh.cnt, used = bh.used, total = bh.total;
    const uint64_t out0 = bh.out0;
    // (first-fit bundles — the multi-dictionary kernel's: the schedule checked every member's place in the output)
    if (total == 0 || (!MULTI && (out0 > a.out_capacity || a.out_capacity - out0 < total))) {
        issue_next();
        return;
    }
    const bool has = lane < cnt;
    const bool lane_used = lane < used;
    const uint32_t seg = bl.seg & 255u;
    const uint32_t seg_unit = MULTI ? bl.seg >> 8 : seg;  // the lane's unit behind u0
    const uint32_t seg_n = (bl.par & 255u) + 1u;
    const uint32_t seg_lane0 = (bl.par >> 8) & 63u;
    const uint32_t seg_out0 = (bl.par >> 14) & 0x3FFFu;
    const bool seg_head = lane == seg_lane0;
    const bool narrow = MULTI && ((bl.par >> 28) & 1u) != 0;
    uint32_t hot_base = a.dict.first.hot_base, hot_k = a.dict.first.hot_k, meta_base = a.dict.first.meta_base;
    if (MULTI) {
        const uint32_t* dp = c.descs + 4 * (bl.par >> 29);
        meta_base = dp[0];
        hot_base = dp[1];
        hot_k = dp[2];
    }
    const uint64_t slot_lane = bh.in0 + bl.slot_rel;   // the lane's first slot

    // ---- slots, metadata, the cold slots' rows -----------------------------------------------------
    tile_regs cur;
    uint32_t raw_lo = 0;  // the lane's first four bytes (the next lane's: what an exception at its end spills into)
    {
        uint64_t raw = raw_in;
        asm volatile("" : "+v"(raw));  // landed
        raw_lo = uint32_t(raw);
        unpack_slots(false, raw, cur);
        if (MULTI && narrow) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) cur.s[k] = (raw_lo >> (8 * k)) & 0xFFu;
        }
    }
    meta_regs mr;
    head_regs hr;
    request_metas(c, hot_base, hot_k, meta_base, cur, mr, hr);
    take_metas(hot_k, mr, hr, cur);

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
            uint32_t prev = from_lane_below((row >> 8) & 7u);
            if (seg_head) prev = 0;
            if (__ballot(prev != st_in) == 0) break;
            st_in = prev;
        }
    }
    const uint32_t excbits = lane_used ? (row >> 4) & 15u : 0u;
    const bool tile_exc = __ballot(excbits != 0) != 0;
    uint32_t excval[kSPL] = {0, 0, 0, 0};
    if (tile_exc) {
        const uint32_t nlo = from_lane_above(raw_lo);  // an exception's payload never leaves its unit's lanes
        uint32_t e[kSPL + 2];
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) e[k] = cur.s[k];
        e[kSPL] = nlo & 0xFFFFu;
        e[kSPL + 1] = nlo >> 16;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) excval[k] = e[k] == 0 ? e[k + 1] : (e[k + 1] | (e[k + 2] << 16));
        if (MULTI && narrow) {  // 8-bit slots: the value is the next 2 or 4 of them
            uint32_t b[kSPL + 4];
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) b[k] = cur.s[k];
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) b[kSPL + k] = (nlo >> (8 * k)) & 0xFFu;
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                const uint32_t lo16 = b[k + 1] | (b[k + 2] << 8);
                excval[k] = b[k] == 0 ? lo16 : (lo16 | (b[k + 3] << 16) | (b[k + 4] << 24));
            }
        }
    }

    // ---- sizes; positions inside each unit (one scan + the value at the unit's first lane); clamp ----
    uint8_t* const lds_rw = reinterpret_cast<uint8_t*>(const_cast<uint32_t*>(lds));
    const uint32_t stage_byte0 = uint32_t(reinterpret_cast<const uint8_t*>(stage_of(scratch)) - lds_rw);
    uint32_t sz[kSPL];
    uint32_t slowb = 0;
    uint32_t liveb = lane_used ? ~row & 15u : 0u;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t m = cur.m[k];
        const bool exc = ((excbits >> k) & 1u) != 0, live = ((liveb >> k) & 1u) != 0;
        sz[k] = live ? (exc ? 1u : (m >> 24) + 1u) : 0u;
        t.need[k] = live ? (exc ? 1u : __builtin_amdgcn_ubfe(m, 20, 2)) : 0u;
        t.src2[k] = m & kMetaOffMask;
        slowb |= ((live && !exc && ((m >> 22) & 1u) != 0) ? 1u : 0u) << k;
    }
    t.off[0] = 0;
#pragma unroll
    for (uint32_t k = 1; k != kSPL; ++k) t.off[k] = t.off[k - 1] + sz[k - 1];
    const uint32_t raw_sum = t.off[kSPL - 1] + sz[kSPL - 1];
    const uint32_t inc1 = wave_inclusive_sum(raw_sum);
    // (the read is unconditional: ds_bpermute returns nothing from lanes that do not take part)
    const uint32_t inc1_before = uint32_t(__shfl(inc1, int(seg_lane0 + 63u) & 63));
    const uint32_t before_seg = seg_lane0 == 0 ? 0u : inc1_before;
    const uint32_t p0 = inc1 - raw_sum - before_seg;  // position of the lane's first codeword inside its unit
    uint32_t nlive = 0, lsum = 0, lb = 0;
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) {
        const uint32_t pos = p0 + t.off[k];
        const bool act = ((liveb >> k) & 1u) != 0 && pos < seg_n;
        const uint32_t room = seg_n - pos;
        lb |= (act ? 1u : 0u) << k;
        nlive += act ? 1u : 0u;
        lsum += act ? (sz[k] < room ? sz[k] : room) : 0u;
        if (!act) t.need[k] = 0;
    }
    liveb = lb;
    slowb &= liveb;
    t.row = row;
    t.liveb = liveb;
    t.lsum = lsum;
    t.nlive = nlive;
    t.obase = seg_out0 + p0;
    t.rbase = wave_inclusive_sum(nlive) - nlive;
    t.total = total;
    uint32_t cell_addr[kSPL];
    const uint32_t cells = allocate_cells(t, stage_byte0, cell_addr);
    if (cells > kStageCells) {  // (as in decode_segment)
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if (t.need[k] != 0 && cell_addr[k] + 16 * t.need[k] > stage_byte0 + 16 * kStageCells) {
                t.need[k] = 0;
                slowb |= 1u << k;
            }
    }
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k) t.src2[k] = t.need[k] != 0 ? cell_addr[k] : t.src2[k];
    bool tile_wide = false;
    if (tile_exc) {
        bool wide = false;
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if (((excbits >> k) & 1u) != 0 && t.need[k] != 0) {  // an exception's literal into its staging cell
                *reinterpret_cast<uint32_t*>(lds_rw + cell_addr[k]) = excval[k];
                t.src2[k] |= excval[k] > 0xFFFFu ? 1u : 0u;
                wide = wide || excval[k] > 0xFFFFu;
                t.need[k] = 0;
            }
        tile_wide = __ballot(wide) != 0;
    }
    const bool tile_slow = __ballot(slowb != 0) != 0;
    const bool tile_big = __ballot((t.need[0] | t.need[1] | t.need[2] | t.need[3]) > 1u) != 0;

    // where each unit's stream ends: after its last live codeword (and that one's payload). That codeword
    // sits in the unit's highest lane that has a live one: the next such lane belongs to another unit.
    if (a.end_off) {
        uint32_t last_end = 0;  // slots from the lane's first to the end of its last live codeword
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k)
            if ((liveb >> k) & 1u)
                last_end = k + 1 + (((excbits >> k) & 1u) ? (narrow ? 2 * cur.s[k] + 2u : cur.s[k] + 1u) : 0u);
        const uint64_t havers = __ballot(last_end != 0);
        const uint64_t above = lane == 63 ? 0ull : havers & ~((2ull << lane) - 1ull);
        const uint32_t next_lane = above ? uint32_t(__builtin_ctzll(above)) : lane;
        const uint32_t next_seg = uint32_t(__shfl(seg, int(next_lane)));  // (unconditional: every lane takes part)
        if (last_end != 0 && (above == 0 || next_seg != seg))
            a.end_off[u0 + seg_unit] = slot_lane + (narrow ? 1ull : 2ull) * last_end;
    }

    uint32_t* const out = a.out + out0;
    uint32_t* const out_u = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(out)));
    // (first-fit bundles: rs_out is the slow path's only — everything from the chunk's output base on, see store_shift)
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, MULTI ? 0x7FFFFFFC : int(total * 4), 0x00020000);
    group_out go{out_u, bh.g_unit_n, bh.g_rel_out};
    // the heads into their cells; the tails of the large ones behind the flag/delta phase (decode_segment)
    uint32_t t3[kSPL];
#pragma unroll
    for (uint32_t k = 0; k != kSPL; ++k)
        if (t.need[k] != 0) {
            *reinterpret_cast<u32x4*>(lds_rw + t.src2[k]) = hr.q[k];
            t.src2[k] += 4;
        }
    if (tile_big) {
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) {
            const uint32_t tail = c.tails_base + 32 * (meta_base + cur.s[k]);
            if (t.need[k] > 1u) hr.q[k] = __builtin_amdgcn_raw_buffer_load_b128(c.rs_dict, tail, 0, 0);
            if (t.need[k] > 2u) t3[k] = __builtin_amdgcn_raw_buffer_load_b32(c.rs_dict, tail + 16, 0, 0);
        }
    }
    // (in-index docs parts: every unit of the bundle is a 256-posting block, so group g of the expansion is unit
    // u0 + g; a slow codeword anywhere leaves the whole bundle as gaps for the flagged fix-up)
    const bool as_docids = a.unit_base != nullptr && !tile_slow && total <= kMaxCap;
    if (a.unit_base != nullptr && !as_docids && has) a.gaps_left[u0 + (MULTI ? bh.g_unit_n & 255u : lane)] = 1;
    // (a bundle is one batch by construction: the schedule packs at most kMaxCap integers into one)
#ifndef DINT_BUNDLE_GROUPS
#define DINT_BUNDLE_GROUPS DINT_GROUPS
#endif
    expand_tile<8 / DINT_BUNDLE_GROUPS, DINT_BUNDLE_GROUPS, true>(t, false, tile_wide, a.plus_one, as_docids ? a.unit_base + u0 : nullptr, 0u, lds, scratch, rs_out, lane, pf, [&]() {
#pragma unroll
        for (uint32_t k = 0; k != kSPL; ++k) asm volatile("" : "+v"(hr.q[k]), "+v"(t3[k]));
        if (tile_big) {
#pragma unroll
            for (uint32_t k = 0; k != kSPL; ++k) {
                if (t.need[k] > 1u) *reinterpret_cast<u32x4*>(lds_rw + t.src2[k] + 12) = hr.q[k];
                if (t.need[k] > 2u) *reinterpret_cast<uint32_t*>(lds_rw + t.src2[k] + 28) = t3[k];
            }
        }
        wave_lds_fence();
        issue_next();
    }, MULTI ? &go : nullptr);
    if (tile_slow) {
        // (positions in slow_stores are relative to the bundle; a unit's clamp is what `room` must be)
        const uint8_t* const my_slots = a.enc + slot_lane;
        // first-fit bundles: the lane's unit's outputs begin at its own offset, not at the tile's position 256 * member
        const uint32_t shift = MULTI ? uint32_t(__shfl(bh.g_rel_out, int(seg))) - seg_out0 : 0u;
        slow_stores(MULTI && narrow, c, t, slowb, a.plus_one, 0u, seg_out0 + seg_n, my_slots, hot_base, hot_k, meta_base, rs_out, shift);
    }
    SECTION(pf, 13, "epilogue");
}

// The bundle path of a launch: chunks of 64 units are drawn from one counter; a chunk's records arrive in ONE load
// (requested a chunk ahead), its bundles — the lanes whose record says "leads c units" — are mapped from
// registers, and every bundle's stream bytes are requested while the bundle before it is expanded. Round 2's
// first version walked a list of bundles instead and paid, per bundle of some 900 integers, one full memory
// round trip each for the list entry, the units' descriptors, their selector bytes and their slots: 36 % of the
// multi-dictionary kernel's time, and 19 % more waiting for the queue ticket behind the previous bundle's stores.
// One bundle on its own, named by its first unit (the single-dictionary kernel's unit queue hands bundles out between
// the long units: see decode_kernel_body): its chunk's records, then as below.
template <bool MULTI>
__device__ __forceinline__ void decode_bundle_listed(const decode_args& a, const wave_ctx& c, uint64_t u0, uint32_t cnt, prof_t& pf) {
    const uint32_t lane = c.lane;
    const uint32_t ch = uint32_t(u0 / kChunkUnits), p = uint32_t(u0 % kChunkUnits);
    const uint64_t i = uint64_t(ch) * kChunkUnits + lane;
    u32x4 rc = {0, 0, 0, 0};
    if (i < a.n_units) rc = a.urec[i];
    const u32x4 cb = *reinterpret_cast<const u32x4*>(a.cbase + 2 * uint64_t(ch));
    const uint64_t in0 = (uint64_t(uniform(cb.y)) << 32) | uniform(cb.x), out0 = (uint64_t(uniform(cb.w)) << 32) | uniform(cb.z);
    bundle_lane bl;
    bundle_head bh;
    bundle_map<MULTI>(c, rc, ch, p, cnt, in0, out0, bl, bh);
    const uint64_t raw = bundle_raw(a, bl, bh, lane);
    bundle_process<MULTI>(a, c, bl, bh, raw, pf, []() {});
}

// `tk`: a chunk ticket in flight (in: asked by the caller or left by the previous call; out: the next one);
// `budget`: chunks to decode at most. -> false: no chunks left.
__device__ __forceinline__ uint32_t chunk_ticket(const decode_args& a, uint32_t lane) {
    uint32_t j = 0;
    if (lane == 0) j = __hip_atomic_fetch_add(a.chunk_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return j;
}
template <bool MULTI>
__device__ __forceinline__ bool decode_bundle_chunks(const decode_args& a, const wave_ctx& c, prof_t& pf, uint32_t& tk, uint32_t budget) {
    const uint32_t lane = c.lane;
    const uint32_t n_chunks = uint32_t((a.n_units + kChunkUnits - 1) / kChunkUnits);
    auto ask = [&]() -> uint32_t { return chunk_ticket(a, lane); };
    auto records = [&](uint32_t chunk) -> u32x4 {
        const uint64_t i = uint64_t(chunk) * kChunkUnits + lane;
        u32x4 r = {0, 0, 0, 0};
        if (i < a.n_units) r = a.urec[i];
        return r;
    };
    auto bases = [&](uint32_t chunk) -> u32x4 {  // (every lane the same 16 bytes: one request)
        return *reinterpret_cast<const u32x4*>(a.cbase + 2 * uint64_t(chunk));
    };
    auto leaders = [&](const u32x4& r) -> uint64_t { return __ballot(((r.z >> 18) & 127u) != 0 && ((r.z >> 8) & 63u) != 0); };
    // chunk `ch` (records rc, bases in0 / out0, bundle leaders left: lm); the next chunk's ticket in flight. (The next
    // chunk's records are NOT requested ahead: eight more registers live through every tile of the chunk cost more
    // — as spills, each reload a wait for everything in flight — than one exposed round trip per chunk.)
    uint32_t ch = uniform(tk);
    if (ch >= n_chunks) return false;
    u32x4 rc = records(ch);
    uint64_t in0, out0;
    {
        const u32x4 cb = bases(ch);
        in0 = (uint64_t(uniform(cb.y)) << 32) | uniform(cb.x);
        out0 = (uint64_t(uniform(cb.w)) << 32) | uniform(cb.z);
    }
    tk = ask();
    --budget;
    uint64_t lm = leaders(rc);
    // -> the next bundle: (chunk registers current, p, cnt); false: nothing left
    uint32_t p = 0, cnt = 0;
    auto advance = [&]() -> bool {
        while (lm == 0) {
            if (budget == 0) return false;
            const uint32_t t2 = uniform(tk);
            if (t2 >= n_chunks) return false;
            ch = t2;
            rc = records(ch);
            const u32x4 cb = bases(ch);
            in0 = (uint64_t(uniform(cb.y)) << 32) | uniform(cb.x);
            out0 = (uint64_t(uniform(cb.w)) << 32) | uniform(cb.z);
            tk = ask();
            --budget;
            lm = leaders(rc);
        }
        p = uint32_t(__builtin_ctzll(lm));
        lm &= lm - 1;
        cnt = (readlane(rc.z, p) >> 18) & 127u;
        return true;
    };
    auto map_here = [&](bundle_lane& l, bundle_head& h) {
        if (MULTI) bundle_map_first_fit(c, rc, ch, readlane(rc.w, p) & 255u, in0, out0, l, h);
        else bundle_map<MULTI>(c, rc, ch, p, cnt, in0, out0, l, h);
    };
    if (!advance()) return true;
    bundle_lane bl;
    bundle_head bh;
    map_here(bl, bh);
    uint64_t raw = bundle_raw(a, bl, bh, lane);
    for (;;) {
        // the bundle after this one is found, mapped and its bytes requested in the middle of this one's tile: behind
        // its last wait (what that needs from memory — a chunk's records, spilled registers — is no wait for stores
        // there: this tile's have not been issued, the previous one's are long done), before its gathers and stores
        bundle_lane bl_n;
        bundle_head bh_n;
        bool have_n = false;
        uint64_t raw_n = 0;
        bundle_process<MULTI>(a, c, bl, bh, raw, pf, [&]() {
            have_n = advance();
            if (have_n) {
                map_here(bl_n, bh_n);
                raw_n = bundle_raw(a, bl_n, bh_n, lane);
            }
        });
        if (!have_n) break;
        bl = bl_n;
        bh = bh_n;
        raw = raw_n;
    }
    return true;
}

// A multi-dictionary unit: blocks of 256 integers (the last one shorter), each opened
// by a selector byte: < 6 -> 16-bit codewords against dictionary `selector`, else 8-bit
// codewords against dictionary `selector - 6` (vroom_env/dint_codecs.hpp:521-619).
// Blocks carry no length, so they are decoded one after the other.
__device__ __forceinline__ void decode_unit_multi(const decode_args& a, const wave_ctx& c, uint64_t unit_index, prof_t& pf) {
    const uint32_t lane = c.lane;
    const dint_unit* up = a.units + unit_index;
    const uint64_t out_off = up->out_off;
    const uint32_t n = up->n;
    if (n == 0 || n > kMaxUnitInts || out_off > a.out_capacity || a.out_capacity - out_off < n || (a.only_full && n != 256)) return;
    uint64_t pos = up->in_off;
    chain_io ch{};
    for (uint32_t done = 0; done < n;) {
        const uint32_t bsize = n - done < 256u ? n - done : 256u;
        pos = uniform64(pos);
        // chained unless the block sits in the last kChainBytes of the buffer
        const bool chained = pos <= a.enc_bytes && a.enc_bytes - pos >= 1 + kChainBytes;
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
        dd.meta_base = uniform(c.descs[4 * d]);
        dd.hot_base = uniform(c.descs[4 * d + 1]);
        dd.hot_k = uniform(c.descs[4 * d + 2]);
        dd.pad = 0;
        uint32_t* const out = a.out + out_off + done;
        ch.more = done + bsize < n;
        pos = decode_segment<0, 1, 1, -1>(a, c, dd, pos + 1, bsize, out, ch, pf, narrow, chained,
                                          a.unit_base ? a.unit_base + unit_index : nullptr,
                                          a.unit_base ? a.gaps_left + unit_index : nullptr);
        done += bsize;
    }
    if (a.end_off && lane == 0) a.end_off[unit_index] = pos;
}

}  // namespace dint_dev
#include "dint_query_kernels.hpp"  // round_tail: what a query round does behind its decode
namespace dint_dev {

// The pages of a query round (dint_query_kernels.hpp): page i = block ids[i] of the index's block table, decoded to
// docIDs at out[256 i ..). The query kernels below take their work from this instead of a unit table.
struct query_pages {
    const dint_block_ref* blocks;  // the index's block table
    const uint32_t* ids;           // page -> block
    const uint32_t* count;         // nullable: the pages in use (the device knows; the launch is sized for `bound`)
    uint64_t bound;
    uint32_t retire;               // candidate pages: the slots past a page's last posting are marked dead
    // candidate pages, term_blocks set: the first round's block-max search rides along (and_search_kernel's job, same
    // arguments) — the wave that decoded a page searches for its 256 candidates
    const uint32_t* page_query;
    const uint32_t* term_first;
    const uint32_t* term_blocks;
    const uint32_t* block_max;
    uint32_t* target;
    uint32_t* needed;
    uint32_t* rank;
    uint32_t* touched;
    uint32_t* n_touched;
};
template <bool MULTI>
__device__ __forceinline__ void decode_query_page(const decode_args& a, const wave_ctx& c, const query_pages& qp, uint64_t page, prof_t& pf);

// Units are handed out dynamically: their cost varies a lot (a sparse list full
// of exceptions takes several times longer than a dense one of the same length),
// so a static unit -> wave map leaves most of the chip idle behind the slowest
// waves. kQueueShards counters, one per group of workgroups; a wave draws its next
// work item while it is still decoding the current one.
// The kernel arguments arrive as one 16-register scalar load; left like that, the compiler keeps (and under
// pressure spills and reloads) the whole tuple whenever one field is live — 66 v_readlane per tile in
// round 1. Passing every field through an empty asm makes each its own 32- or 64-bit scalar.
// Every pointer of the arguments is to GLOBAL memory, which own_scalars' empty asm hides from the compiler: said again,
// what goes through them is global_load / global_store / global_atomic. As FLAT operations — "LDS or memory, may return
// out of order" — each makes the compiler's next wait for anything a wait for everything in flight, and the work
// queue's ticket (an atomic asked for before a unit is decoded, looked at after) is in flight all through a unit.
// (The vroom single-dictionary kernel only: its segment counts its own waits. The others leave their waits to the
// compiler, which places them worse with typed pointers: measured, -3 %.)
template <class T>
__device__ __forceinline__ T* known_global(T* p) {
    typedef __attribute__((address_space(1))) T global_T;
    return (T*)(global_T*)(uintptr_t)p;
}
__device__ __forceinline__ void all_global(decode_args& a) {
    a.dict.tables = known_global(a.dict.tables);
    a.dict.lds_image = known_global(a.dict.lds_image);
    a.dict.descs = known_global(a.dict.descs);
    a.enc = known_global(a.enc);
    a.units = known_global(a.units);
    a.out = known_global(a.out);
    a.end_off = known_global(a.end_off);
    a.queue = known_global(a.queue);
    a.sched = known_global(a.sched);
    a.items = known_global(a.items);
    a.item_cnt = known_global(a.item_cnt);
    a.n_items = known_global(a.n_items);
    a.urec = known_global(a.urec);
    a.cbase = known_global(a.cbase);
    a.chunk_queue = known_global(a.chunk_queue);
    a.spans = known_global(a.spans);
}
__device__ __forceinline__ decode_args own_scalars(const decode_args& k) {
    decode_args a = k;
    asm volatile("" : "+s"(a.dict.tables), "+s"(a.dict.lds_image), "+s"(a.dict.descs), "+s"(a.dict.tables_bytes),
                 "+s"(a.dict.heads_base), "+s"(a.dict.tails_base), "+s"(a.dict.goff_base), "+s"(a.dict.gtable_base), "+s"(a.dict.hot_words),
                 "+s"(a.dict.first.meta_base), "+s"(a.dict.first.hot_base), "+s"(a.dict.first.hot_k));
    if (DINT_LEAN_SEGMENT == 2) asm volatile("" : "+s"(a.dict.long_bitmap_word));
    asm volatile("" : "+s"(a.enc), "+s"(a.enc_bytes), "+s"(a.units), "+s"(a.n_units), "+s"(a.out),
                 "+s"(a.out_capacity), "+s"(a.end_off), "+s"(a.queue), "+s"(a.n_shards), "+s"(a.only_full));
    asm volatile("" : "+s"(a.sched), "+s"(a.items), "+s"(a.item_cnt), "+s"(a.n_items), "+s"(a.urec), "+s"(a.cbase), "+s"(a.chunk_queue), "+s"(a.spans),
                 "+s"(a.plus_one), "+s"(a.unit_base), "+s"(a.gaps_left));
    return a;
}

// INDEX: an in-index launch (256-posting blocks: docIDs formed in the expansion, freqs + 1, full blocks only). The
// vroom kernels are compiled without any of that: the per-group branches of the expansion, and the masks the compiler
// puts on every gathered integer because the docID arithmetic might read it, are gone from their loops.
// BUNDLES_ONLY: a multi-dictionary launch whose schedule left the unit queue empty (a block-granular unit table: every
// unit fits a tile) — compiled without the unit queue and decode_unit_multi's segment loop, which is the larger half of
// the general kernel and what its register allocation is shaped by.
template <bool MULTI, bool INDEX, bool QUERY = false, bool BUNDLES_ONLY = false>
__device__ __forceinline__ void decode_kernel_body(const decode_args& kernarg, const query_pages* qp = nullptr,
                                                   const round_tail* tail = nullptr) {
    decode_args a_ = own_scalars(kernarg);
    if (DINT_LEAN_SEGMENT && !INDEX && !MULTI && !QUERY) all_global(a_);
    if (!INDEX) {
        a_.unit_base = nullptr;
        a_.gaps_left = nullptr;
        a_.plus_one = 0;
        a_.only_full = 0;
    }
    const decode_args a = a_;
    // (the first wave of the launch notes the shader clock at both ends: cycles / kernel time = the clock the kernel
    // actually ran at — boxes of one pool differ by a tenth in speed for the same binary, and this says why)
    const uint64_t clock0 = __builtin_amdgcn_s_memtime();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // the dictionary's hot part, 16 bytes a thread and step (hot_words is a multiple of 4): five steps instead of
    // eighteen dependent round trips — nothing for a launch that decodes 10^9 integers, a third of one that decodes a
    // query's handful of pages
    // (the query kernels: all of a thread's loads in flight before the first lands in LDS — a launch of one
    // workgroup, a query's round, waits for this copy: one round trip instead of five)
    if (!QUERY) {
        for (uint32_t i = threadIdx.x; 4 * i < a.dict.hot_words; i += kBlockThreads)
            reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(a.dict.lds_image)[i];
    } else {
        constexpr uint32_t kSteps = (kHotImageWords / 4 + kBlockThreads - 1) / kBlockThreads;
        u32x4 part[kSteps];
#pragma unroll
        for (uint32_t k = 0; k != kSteps; ++k) {
            const uint32_t i = threadIdx.x + k * kBlockThreads;
            if (4 * i < a.dict.hot_words) part[k] = reinterpret_cast<const u32x4*>(a.dict.lds_image)[i];
        }
#pragma unroll
        for (uint32_t k = 0; k != kSteps; ++k) {
            const uint32_t i = threadIdx.x + k * kBlockThreads;
            if (4 * i < a.dict.hot_words) reinterpret_cast<u32x4*>(lds)[i] = part[k];
        }
    }
    uint16_t* const cls = reinterpret_cast<uint16_t*>(lds + a.dict.hot_words);
    build_class_table(cls);
    uint32_t* const descs = lds + a.dict.hot_words + kDescWordAt;
    if (threadIdx.x < 24) descs[threadIdx.x] = MULTI ? reinterpret_cast<const uint32_t*>(a.dict.descs)[threadIdx.x] : 0u;
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x / kWave);
    uint32_t* const scratch = lds + a.dict.hot_words + kClassTableWords + wave * kScratchWords;
    for (uint32_t i = lane; i < kFwWords; i += kWave) scratch[i] = 0;  // the flag words start out zero
    __syncthreads();
    wave_ctx c;
    c.lds = lds;
    c.cls = cls;
    c.descs = descs;
    c.scratch = scratch;
    c.lane = lane;
    c.rs_dict = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dict.tables), 0, int(a.dict.tables_bytes), 0x00020000);
    c.heads_base = a.dict.heads_base;
    c.tails_base = a.dict.tails_base;
    c.goff_base = a.dict.goff_base;
    c.gtable_base = a.dict.gtable_base;
    // Work queue. Every shard — the workgroups with the same blockIdx % n_shards: one XCD under
    // round-robin placement — walks its own CONTIGUOUS part of the work items and, when that is done,
    // helps with the next shards' parts. Contiguous, because an XCD that strides over the whole
    // stream and output touches every 2 MB page of them, and past ~2 GB the translations no longer stay
    // in its TLB (the time per integer rose by a fifth); stealing, because equal counts of work items
    // are not equal work.
    const uint32_t shard = blockIdx.x % a.n_shards;  // n_shards = min(kQueueShards, gridDim.x)
    // with a schedule the queue hands out the units that are decoded on their own; the bundles follow
    uint64_t n_work_ = a.sched ? uint64_t(uniform(*a.n_items)) : a.n_units;
    if (QUERY) {
        n_work_ = qp->bound;
        if (qp->count) n_work_ = uniform(*qp->count) < n_work_ ? uniform(*qp->count) : n_work_;
    }
    const uint64_t n_work = n_work_;
    const uint64_t per_shard = (n_work + a.n_shards - 1) / a.n_shards;
    uint32_t cur = shard, tried = 0;
    // A draw is two steps: the ticket (one returning atomic on the shard's counter) is asked for while the
    // previous work item is still being decoded, and only looked at when that one is done — the atomic's
    // round trip (2.5k cycles per work item, 6 % of the kernel in round 1's profile) hides behind the decode.
    auto ask = [&]() -> uint32_t {
        uint32_t j = 0;
        if (lane == 0) j = __hip_atomic_fetch_add(a.queue + cur * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return j;
    };
    auto take = [&](uint32_t ticket) -> uint64_t {  // the work item of a ticket, or ~0: nothing left anywhere
        uint32_t j = uniform(ticket);
        for (;;) {
            const uint64_t first = per_shard * cur;
            const uint64_t size = first >= n_work ? 0 : (n_work - first < per_shard ? n_work - first : per_shard);
            if (j < size) return first + j;
            cur = cur + 1 == a.n_shards ? 0 : cur + 1;  // this shard is done: help the next one
            if (++tried >= a.n_shards) return ~0ull;
            j = uniform(ask());
        }
    };
    prof_t pf;
#ifdef DINT_PROFILE
    prof_begin(pf, scratch + kScratchWords - kProfWords, lane);
#endif
    // The unit queue. In the single-dictionary kernel it hands out the bundles too, in stream order between the long
    // units: a bundle is a short, latency-bound piece of work, and with the bundles in a phase of their own — behind
    // the units, before them, or one chunk behind every eighth unit of a wave; all measured on the 1e9-posting
    // run: 2.5 %, 5 % and 3.5 % slower — too many waves wait for the same kind of round trip at the same time.
    // The multi-dictionary kernel decodes block-granular unit tables, which are bundles and nothing else: there
    // the chunks follow the (few) units on their own.
    // (Prefetching the next item's description as well — two items ahead — was measured: no gain; what a wave
    // waits for here is the vector-memory front end, not the round trip.)
    uint64_t w = BUNDLES_ONLY ? ~0ull : take(ask());
    while (!BUNDLES_ONLY && w != ~0ull) {
        SECTION(pf, 14, "draw");
        uint32_t ticket = ask();
        const uint64_t uu = uniform64(!QUERY && a.sched ? uint64_t(a.items[w]) : w);
        if (QUERY) {
            decode_query_page<MULTI>(a, c, *qp, uu, pf);
        } else if (MULTI) {
            decode_unit_multi(a, c, uu, pf);
        } else {
            const uint32_t cc = uniform(a.sched ? uint32_t(a.item_cnt[w]) : 1u);
            if (__builtin_expect(cc > 1, 0)) decode_bundle_listed<false>(a, c, uu, cc, pf);
            else decode_unit_single<(!INDEX && !QUERY) ? DINT_LEAN_SEGMENT : 0>(a, c, uu, pf);
        }
        asm volatile("" : "+v"(ticket));
        w = take(ticket);
    }
    if (MULTI && !QUERY && a.sched) {
        uint32_t chunk_tk = chunk_ticket(a, lane);
        (void)decode_bundle_chunks<true>(a, c, pf, chunk_tk, ~0u);
    }
#ifdef DINT_PROFILE
    prof_end(pf);
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *reinterpret_cast<uint64_t*>(a.chunk_queue + kClockWordAt) = __builtin_amdgcn_s_memtime() - clock0;
    if (QUERY) {
        if (!tail->done) return;  // (uniform)
        // the rest of the round: by the workgroup that finishes last, with every page of the launch in memory
        // (a launch of ONE workgroup — a round of up to 16 pages, the common case of a single query — needs the
        // barrier and nothing else; an agent-scope fence writes back and invalidates this XCD's L2, microseconds each)
        __shared__ uint32_t last;
        if (gridDim.x != 1) __threadfence();  // (every wave: its pages' stores)
        __syncthreads();
        if (gridDim.x != 1) {
            if (threadIdx.x == 0) last = atomicAdd(tail->done, 1u) == gridDim.x - 1 ? 1u : 0u;
            __syncthreads();
            if (!last) return;
            __threadfence();
        }
        and_round_tail(*tail);
    }
}

#ifndef DINT_MIN_WAVES
#define DINT_MIN_WAVES 1
#endif
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_single_kernel(decode_args a) {
    decode_kernel_body<false, false>(a);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_kernel(decode_args a) {
    decode_kernel_body<true, false>(a);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_bundles_kernel(decode_args a) {
    decode_kernel_body<true, false, false, true>(a);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_single_index_kernel(decode_args a) {
    decode_kernel_body<false, true>(a);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_index_kernel(decode_args a) {
    decode_kernel_body<true, true>(a);
}

// ---- in-index path: helper kernels -------------------------------------------------------------

// unit table for the docs parts (in_off from the block table) or for the freqs parts (in_off =
// where the docs part ended)
__global__ void blocks_to_units_kernel(const dint_block_ref* blocks, const uint64_t* docs_end, uint64_t n_blocks,
                                       uint64_t index_bytes, dint_unit* units, uint32_t* spans, uint32_t* bases = nullptr) {
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
    if (bases) bases[b] = blocks[b].base;
}

// After a table's first decode the docs parts' ends are known: their byte spans become exact (they were "up to
// the next block", freqs bytes included), and later launches pack four to six docs parts to a tile instead of two.
__global__ void exact_spans_kernel(const dint_unit* units, const uint64_t* ends, uint64_t n_blocks, uint32_t* spans) {
    const uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    if (units[b].n == 256 && ends[b] > units[b].in_off && ends[b] - units[b].in_off < spans[b]) spans[b] = uint32_t(ends[b] - units[b].in_off);
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

// One block of n < 256 integers in binary interpolative code at p -> its prefix sums in o[0 .. n) (o[n - 1] = sum);
// returns the bytes consumed. The reference recurses node - left subtree - right subtree
// (interpolative_coding.hpp:128-146); here an explicit stack (frames of 4 words in LDS) is walked in that order.
// (A variant with the bit reader one word ahead and the left child taken without a trip through the stack was
// measured: a third slower — the kernel lives on how many of these lanes are in flight, not on their length.)
// (ONE word per stack frame — the subrange's offset and length: its bounds are the values next to it in the row, decoded
// before the recursion descends into it (the element left of the range, zero before the first; the element right of it,
// the sum behind the last) — so that a block takes 268 words of LDS, not 298, and the 8 blocks of a wave 8.4 KB: the
// short blocks of a 10^8-posting index, 4 591 waves, are then all resident at once: 18 waves to a CU. `o[-1]` must be zero.)
__device__ __forceinline__ uint64_t interpolative_prefix_sums(const uint8_t* p, uint64_t limit, uint32_t n, uint32_t sum, uint32_t* o,
                                                            uint32_t* stack) {
    o[n - 1] = sum;
    if (n <= 1) return 0;
    tail_bits br{p, limit, 0, 0, 0, 0};
    uint32_t top = 0;
    stack[top++] = (n - 1) << 8;  // frame: offset | length << 8
    while (top) {
        const uint32_t f = stack[--top];
        const uint32_t f_off = f & 255u, f_n = f >> 8;
        const uint32_t f_low = o[int32_t(f_off) - 1], f_high = o[f_off + f_n];
        const uint32_t h = f_n / 2;
        const uint32_t val = f_low + br.read_int(f_high - f_low + 1);
        o[f_off + h] = val;
        if (f_n - h - 1) stack[top++] = (f_off + h + 1) | ((f_n - h - 1) << 8);
        if (h) stack[top++] = f_off | (h << 8);
    }
    return (br.pos + 7) / 8;
}

// The short blocks are one in fifteen of a block table; collected first, so that the bit-serial decoder
// below runs with full wavefronts (scattered over the table, four active lanes per wave made every
// wave last as long as its slowest decode).
__global__ void collect_tails_kernel(const dint_block_ref* blocks, uint64_t n_blocks, uint32_t* tails, uint32_t* n_tails) {
    const uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t n = blocks[b].n;
    if (n != 0 && n < 256) tails[atomicAdd(n_tails, 1u)] = uint32_t(b);
}

// One wavefront per kTailLanes short blocks, one block per lane. The decoder's values and its explicit stack
// live in LDS (rows of odd stride: lane-private and conflict-free); the block is differenced there and
// the wave then copies the rows out together, coalesced — no pass over global memory but that one.
// (8 blocks to a wave, not 64: the decoder is bit-serial and a wave lasts as long as its longest block, so what
// counts is how many waves the chip has to overlap — one short block in fifteen leaves it far from full.)
// What a launch decodes: the docs parts (docs_end null; as_docids: written as docIDs — the code IS the prefix
// sums), or the freqs parts (docs_end = where each block's docs part ended; sum_of_values = -1: a vbyte of the sum
// first), or — freqs_out set — both, the freqs part right behind its docs part in the same lane.
#ifndef DINT_TAIL_LANES
#define DINT_TAIL_LANES 8
#endif
constexpr uint32_t kTailLanes = DINT_TAIL_LANES;
constexpr uint32_t kTailRow = 257;     // words per lane and row: a zero in front of up to 255 values (odd stride: conflict-free)
constexpr uint32_t kTailStack = 11;    // words per lane: 10 one-word frames (depth <= log2(256) + 1)
constexpr uint32_t kTailLdsBytes = kTailLanes * (kTailRow + kTailStack) * 4;

__device__ __forceinline__ void interpolative_tails_wave(const uint8_t* index, uint64_t index_bytes,
                                                         const dint_block_ref* blocks, const uint64_t* docs_end,
                                                         const uint32_t* tails, const uint32_t* n_tails, uint32_t* out,
                                                         uint64_t out_capacity, uint64_t* end_off, uint32_t plus_one,
                                                         uint32_t as_docids, uint32_t* freqs_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tail_lds[];
    __shared__ uint32_t row_n[kTailLanes], row_base[kTailLanes];
    __shared__ uint64_t row_out[kTailLanes];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = uint64_t(blockIdx.x) * kTailLanes + lane;
    uint32_t* const o = tail_lds + (lane % kTailLanes) * kTailRow + 1;  // (o[-1]: the zero the first range's lower bound reads)
    uint32_t* const stack = tail_lds + kTailLanes * kTailRow + (lane % kTailLanes) * kTailStack;
    if (lane < kTailLanes) o[-1] = 0;
    uint32_t n = 0;
    uint64_t b = 0;
    if (lane < kTailLanes && t < *n_tails) {
        b = tails[t];
        n = blocks[b].n;
        if (n >= 256 || blocks[b].out_off + n > out_capacity) n = 0;
    }
    if (lane < kTailLanes) {
        row_n[lane] = n;
        row_out[lane] = n ? blocks[b].out_off : 0;
        row_base[lane] = n ? blocks[b].base : 0;
    }
    auto vbyte_sum = [&](uint64_t& at) {  // sum_of_values = -1 -> TightVariableByte sum first
        uint32_t sum = 0;
        for (uint32_t shift = 0; at < index_bytes; shift += 7) {
            const uint8_t c = index[at++];
            sum += uint32_t(c & 127) << (shift & 31);
            if (c & 128) break;
        }
        return sum;
    };
    uint64_t pos = 0;
    if (n != 0) {
        pos = docs_end ? docs_end[b] : blocks[b].in_off;
        const uint32_t sum = docs_end ? vbyte_sum(pos) : blocks[b].max - blocks[b].base - (n - 1);
        pos += interpolative_prefix_sums(index + pos, index_bytes - pos, n, sum, o, stack);
        // (the code stores prefix sums: docID i of the block is base + prefix_i + i — no differencing then)
        if (!as_docids)
            for (uint32_t i = n - 1; i > 0; --i) o[i] -= o[i - 1];
        if (end_off) end_off[b] = pos;
    }
    __syncthreads();
    for (uint32_t j = 0; j != kTailLanes; ++j) {  // rows out, the whole wave on one row at a time
        const uint32_t nj = row_n[j];
        uint32_t* const dst = out + row_out[j];
        const uint32_t add = as_docids ? row_base[j] : plus_one;
        for (uint32_t i = lane; i < nj; i += 64) dst[i] = tail_lds[j * kTailRow + 1 + i] + add + (as_docids ? i : 0u);
    }
    if (!freqs_out) return;
    __syncthreads();  // the rows are free again: the freqs parts, right behind the docs parts
    if (n != 0) {
        const uint32_t fsum = vbyte_sum(pos);
        interpolative_prefix_sums(index + pos, index_bytes - pos, n, fsum, o, stack);
        for (uint32_t i = n - 1; i > 0; --i) o[i] -= o[i - 1];
    }
    __syncthreads();
    for (uint32_t j = 0; j != kTailLanes; ++j) {
        const uint32_t nj = row_n[j];
        for (uint32_t i = lane; i < nj; i += 64) freqs_out[row_out[j] + i] = tail_lds[j * kTailRow + 1 + i] + 1u;
    }
}

__global__ __launch_bounds__(64) void interpolative_tails_kernel(const uint8_t* index, uint64_t index_bytes,
                                                                 const dint_block_ref* blocks, const uint64_t* docs_end,
                                                                 const uint32_t* tails, const uint32_t* n_tails, uint32_t* out,
                                                                 uint64_t out_capacity, uint64_t* end_off, uint32_t plus_one,
                                                                 uint32_t as_docids = 0, uint32_t* freqs_out = nullptr,
                                                                 const uint8_t* todo = nullptr, uint64_t n_blocks = 0) {
    // (the grid may be sized for the worst case)
    if (uint64_t(blockIdx.x) * kTailLanes < *n_tails)
        interpolative_tails_wave(index, index_bytes, blocks, docs_end, tails, n_tails, out, out_capacity, end_off, plus_one, as_docids, freqs_out);
    if (!todo) return;
    // ... and this wave's share of the blocks the decode kernels flagged as left in gaps (next to none): gaps -> docIDs
    const uint32_t lane = threadIdx.x;
    const uint64_t share = ((n_blocks + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
    for (uint64_t b0 = share * blockIdx.x; b0 < share * (blockIdx.x + 1) && b0 < n_blocks; b0 += 64) {
        uint64_t flagged = __ballot(b0 + lane < n_blocks && todo[b0 + lane] != 0);
        while (flagged) {  // wave-uniform
            const uint64_t b = b0 + uint32_t(__builtin_ctzll(flagged));
            flagged &= flagged - 1;
            const uint32_t n = blocks[b].n;
            const uint64_t at = blocks[b].out_off;
            if (n == 0 || n > 256 || at + n > out_capacity) continue;
            uint32_t g[4], local = 0;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                const uint32_t i = 4 * lane + k;
                g[k] = i < n ? out[at + i] + 1 : 0;
                local += g[k];
            }
            uint32_t run = blocks[b].base + wave_inclusive_sum(local) - local - 1;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                const uint32_t i = 4 * lane + k;
                run += g[k];
                if (i < n) out[at + i] = run;
            }
        }
    }
}

// gaps -> docIDs (docid_i = base + sum_{j<=i} gap_j + i, dict_posting_list.hpp:111-124) for the few blocks the decode
// kernels had to leave as gaps (flags in `todo`): one wave per 64 blocks,
// every lane looks at one flag; the rare block that has it set is summed by the whole wave.
__global__ __launch_bounds__(64) void finalize_flagged_kernel(const dint_block_ref* blocks, uint64_t n_blocks, uint32_t* docids,
                                                              uint64_t out_capacity, const uint8_t* todo) {
    const uint32_t lane = threadIdx.x;
    const uint64_t b0 = uint64_t(blockIdx.x) * 64;
    uint64_t flagged = __ballot(b0 + lane < n_blocks && todo[b0 + lane] != 0);
    while (flagged) {  // wave-uniform
        const uint64_t b = b0 + uint32_t(__builtin_ctzll(flagged));
        flagged &= flagged - 1;
        const uint32_t n = blocks[b].n;
        const uint64_t at = blocks[b].out_off;
        if (n == 0 || n > 256 || at + n > out_capacity) continue;
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
            if (i < n) docids[at + i] = run;
        }
    }
}

// What is left to do on a query's pages behind the decode kernel, in one launch (wave w: short blocks
// tails[8w .. 8w+8), pages 8w .. 8w+8): the short blocks' interpolative docs parts (as docIDs), the gaps -> docIDs of
// the pages the decode kernel flagged, and — `retire` — the slots past every page's last posting marked as holding no
// candidate (0xFFFFFFFF is no docID).
__global__ __launch_bounds__(64) void fix_pages_kernel(const uint8_t* index, uint64_t index_bytes, const dint_block_ref* pages,
                                                       uint64_t n_pages, const uint32_t* tails, const uint32_t* n_tails, uint32_t* docids,
                                                       uint64_t out_capacity, const uint8_t* todo, uint32_t retire) {
    if (uint64_t(blockIdx.x) * kTailLanes < *n_tails)
        interpolative_tails_wave(index, index_bytes, pages, nullptr, tails, n_tails, docids, out_capacity, nullptr, 0u, 1u, nullptr);
    const uint32_t lane = threadIdx.x;
    for (uint32_t j = 0; j != kTailLanes; ++j) {
        const uint64_t b = uint64_t(blockIdx.x) * kTailLanes + j;
        if (b >= n_pages) break;
        const uint32_t n = pages[b].n;
        const uint64_t at = pages[b].out_off;
        if (at + 256 > out_capacity) continue;
        if (todo[b] != 0 && n != 0 && n <= 256) {  // wave-uniform
            uint32_t g[4], local = 0;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                const uint32_t i = 4 * lane + k;
                g[k] = i < n ? docids[at + i] + 1 : 0;
                local += g[k];
            }
            uint32_t run = pages[b].base + wave_inclusive_sum(local) - local - 1;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                const uint32_t i = 4 * lane + k;
                run += g[k];
                if (i < n) docids[at + i] = run;
            }
        }
        if (retire)
            for (uint32_t i = lane; i < 256; i += 64)
                if (i >= n) docids[at + i] = 0xFFFFFFFFu;
    }
}

// One short block by a whole wavefront (the query kernels: ONE short block to a wave, and the launch waits for it).
// The code is serial — where a value's bits start depends on every value before it — but the ORDER the reference's
// recursion visits the positions in (node, left subtree, right subtree, interpolative_coding.hpp:128-146) depends on
// n alone, and so do the two neighbours whose values bound each node (low = the value at a - 1 or 0, high = the
// value at b). So: the lanes lay the traversal out in parallel (rank of every position by a walk down the implicit
// tree) and copy the block's bytes into LDS, and what is left of the serial loop is branch-free: two LDS reads for the
// bounds, a bit reader that always holds 32 valid bits (the next word is read one node ahead), one LDS write — no
// stack, no trip to memory, no branch per node. Every lane runs the loop on the same values (nothing crosses lanes).
// tmp: 3 x 256 + 2 words of LDS.   -> component k of lane l = prefix sum l + 64 k (n <= 255).
// (A version with the values in registers across the lanes, read by v_readlane, was no faster than the one-lane
// decoder with its LDS stack: picking one of four registers by a scalar index compiles to a ladder of branches.)
// (not inlined: next to decode_segment its live registers cost the page decode sixteen spills)
__device__ __attribute__((noinline)) u32x4 interpolative_block_wave(const uint8_t* p, uint64_t limit, uint32_t n, uint32_t sum, uint32_t lane,
                                                                    lds_u32* tmp) {
    lds_u32* const ord = tmp;            // [256]: the traversal, entry = position | (a + 1) << 8 | (b + 1) << 16
    lds_u32* const win = tmp + 256;      // [256 + 1]: the block's first 1024 bytes (254 values of at most 32 bits)
    lds_u32* const o = tmp + 256 + 257;  // [1 + 256]: o[0] = 0 (the lower bound of the leftmost nodes), o[1 + i] = prefix sum i
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const uint64_t byte = 4ull * (lane + 64 * k);
        uint32_t w = 0;
        if (byte + 4 <= limit) {
            w = reinterpret_cast<const u32_a1*>(p + byte)->v;
        } else {
            for (uint32_t i = 0; i != 4; ++i)
                if (byte + i < limit) w |= uint32_t(p[byte + i]) << (8 * i);
        }
        win[lane + 64 * k] = w;
        o[1 + lane + 64 * k] = lane + 64 * k == n - 1 ? sum : 0u;
        ord[lane + 64 * k] = 0;
    }
    if (lane == 0) win[256] = 0, o[0] = 0;
    wave_lds_fence();
    // the traversal: position m is visited as number rank(m)
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const uint32_t m = lane + 64 * k;
        if (m + 1 < n) {
            uint32_t off = 0, cnt = n - 1, rank = 0, a1 = 0, b = n - 1;
            for (;;) {
                const uint32_t h = cnt >> 1, mid = off + h;
                if (m == mid) break;
                if (m < mid) {  // into the left subtree: behind the node itself; bounded above by the node
                    rank += 1;
                    b = mid;
                    cnt = h;
                } else {        // into the right subtree: behind the node and its left subtree; bounded below by the node
                    rank += 1 + h;
                    a1 = mid + 1;
                    off = mid + 1;
                    cnt = cnt - h - 1;
                }
            }
            ord[rank] = m | (a1 << 8) | ((b + 1) << 16);
        }
    }
    wave_lds_fence();
    uint64_t buf = (uint64_t(win[1]) << 32) | win[0];  // (least significant bit first, tail_bits::read)
    uint32_t avail = 64, wi = 2;
    uint32_t next_w = win[2], e = ord[0];
    for (uint32_t s = 0; s + 1 < n; ++s) {
        const uint32_t mid = e & 255u, a1 = (e >> 8) & 255u, b1 = (e >> 16) & 511u;
        e = ord[(s + 1) & 255u];
        const uint32_t hi = o[b1], lo = o[a1];
        const uint32_t u = hi - lo + 1;
        const uint32_t bits = (31u - uint32_t(__builtin_clz(u | 1u))) & 31u;
        const uint32_t thr = uint32_t((uint64_t(2) << bits) - u);
        uint32_t v = uint32_t(buf) & ((1u << bits) - 1u);
        const uint32_t more = v >= thr ? 1u : 0u;  // one more bit: (v << 1) + bit - thr
        v = more ? (v << 1) + (uint32_t(buf >> bits) & 1u) - thr : v;
        buf >>= bits + more;
        avail -= bits + more;
        const bool refill = avail <= 32;
        buf |= refill ? uint64_t(next_w) << avail : uint64_t(0);
        avail += refill ? 32u : 0u;
        wi += refill ? 1u : 0u;
        next_w = win[wi < 256 ? wi : 256];
        o[1 + mid] = lo + v;
    }
    wave_lds_fence();
    return u32x4{o[1 + lane], o[65 + lane], o[129 + lane], o[193 + lane]};
}

// ---- a query's pages in ONE launch ----------------------------------------------------------------
// What prepare_pages_kernel + the decode kernel + fix_pages_kernel do in three launches, for the small rounds
// (a single query, a handful of pages) where the launches themselves are what the caller waits for: the wave that
// draws page i looks its block up itself, decodes a full block through the DINT front end and expansion (docIDs
// formed there), sums a block the expansion had to leave as gaps on the spot, and runs the interpolative code of
// a short block in its first lane (there is at most one short block per list).
template <bool MULTI>
__device__ __forceinline__ void decode_query_page(const decode_args& a, const wave_ctx& c, const query_pages& qp, uint64_t page, prof_t& pf) {
    const uint32_t lane = c.lane;
    const dint_block_ref* const r = qp.blocks + uniform(qp.ids[page]);
    const uint32_t n = uniform(r->n);
    const uint64_t at = page * 256;
    if (n == 0 || n > 256 || at + 256 > a.out_capacity) return;
    uint32_t* const out = a.out + at;
    const uint64_t in_off = uniform64(r->in_off);
    if (n == 256) {
        volatile uint8_t* const flag = a.gaps_left + page;  // written and read by this wave's first lane only
        if (lane == 0) *flag = 0;
        chain_io ch{};
        if (!MULTI) {
            decode_segment<16, kRounds, kGroups, 0>(a, c, a.dict.first, in_off, 256, out, ch, pf, false, false, &r->base,
                                                    a.gaps_left + page);
        } else {
            const uint64_t sp = in_off < a.enc_bytes ? in_off : a.enc_bytes - 1;
            const uint32_t sel = uniform(a.enc[sp]);
            const bool narrow = sel >= 6;
            const uint32_t d = (narrow ? sel - 6 : sel) % 6;
            dict_desc dd;
            dd.meta_base = uniform(c.descs[4 * d]);
            dd.hot_base = uniform(c.descs[4 * d + 1]);
            dd.hot_k = uniform(c.descs[4 * d + 2]);
            dd.pad = 0;
            decode_segment<0, 1, 1, -1>(a, c, dd, in_off + 1, 256, out, ch, pf, narrow, false, &r->base, a.gaps_left + page);
        }
        uint32_t left = 0;
        if (lane == 0) left = *flag;
        if (uniform(left) != 0) {
            // gaps -> docIDs (docid_i = base + sum_{j<=i} gap_j + i, dict_posting_list.hpp:111-124), behind the block's stores
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            uint32_t g[4], local = 0;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                g[k] = __hip_atomic_load(out + 4 * lane + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                local += g[k];
            }
            uint32_t run = uniform(r->base) + wave_inclusive_sum(local) - local - 1;
#pragma unroll
            for (uint32_t k = 0; k != 4; ++k) {
                run += g[k];
                out[4 * lane + k] = run;
            }
        }
    } else {
        // a short block: binary interpolative code (block_codecs.hpp:130-150) — the code IS the prefix sums
        // (the delta table and the staging cells, one stretch of the wave's scratch, are free between two segments; the
        // flag words in front of them stay zero)
        lds_u32* const tmp = (lds_u32*)delta_of(c.scratch);
        static_assert(3 * 256 + 2 <= kDeltaWords + kStageWords, "the interpolative decoder's tables live in the delta table + staging cells");
        const uint32_t base = uniform(r->base);
        u32x4 ov = {0, 0, 0, 0};
        if (in_off < a.enc_bytes)
            ov = interpolative_block_wave(a.enc + in_off, a.enc_bytes - in_off, n, uniform(r->max) - base - (n - 1), lane, tmp);
#pragma unroll
        for (uint32_t k = 0; k != 4; ++k) {
            const uint32_t i = lane + 64 * k;
            if (i < n) out[i] = ov[k] + base + i;
            else if (qp.retire) out[i] = 0xFFFFFFFFu;
        }
    }
    if (!qp.term_blocks) return;
    // ---- the first round's block-max search for this page's candidates (next_geq's skipping, dict_posting_list.hpp:
    // 126-147; and_search_kernel): lane l takes candidates 4 l .. 4 l + 3, read back behind the page's stores
    constexpr uint32_t kDead = 0xFFFFFFFFu;
    const uint32_t q = uniform(qp.page_query[page]);
    const uint32_t nb = uniform(qp.term_blocks[q]);
    if (nb == 0) return;  // the query has one term only: its candidates pass
    const uint32_t fb = uniform(qp.term_first[q]);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // (the four searches of a lane step together — four loads in flight per round trip, not four searches one
    // after the other: the block maxima of a long list are a dozen dependent trips to memory)
    uint32_t gb[4], cand[4], lo[4], len[4];
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        cand[k] = __hip_atomic_load(out + 4 * lane + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo[k] = 0;
        len[k] = cand[k] != kDead ? nb : 0u;
    }
    while ((len[0] | len[1] | len[2] | len[3]) != 0) {  // first block of the list whose maximum is >= the candidate
        uint32_t bm[4];
#pragma unroll
        for (uint32_t k = 0; k != 4; ++k) bm[k] = len[k] ? qp.block_max[fb + lo[k] + (len[k] >> 1)] : 0u;
#pragma unroll
        for (uint32_t k = 0; k != 4; ++k)
            if (len[k]) {
                const uint32_t half = len[k] >> 1;
                const bool right = bm[k] < cand[k];
                lo[k] = right ? lo[k] + half + 1 : lo[k];
                len[k] = right ? len[k] - half - 1 : half;
            }
    }
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const uint32_t i = 4 * lane + k;
        gb[k] = kDead;
        if (cand[k] != kDead) {
            if (lo[k] == nb) {
                out[i] = kDead;  // past the list's last block
            } else {
                gb[k] = fb + lo[k];
                qp.target[at + i] = gb[k];
            }
        }
    }
    // candidates are sorted, neighbours mostly fall into the same block: one claim per run
    const uint32_t before = __shfl_up(gb[3], 1);
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
        const bool lead = gb[k] != kDead && (k == 0 ? (lane == 0 || before != gb[0]) : gb[k - 1] != gb[k]);
        if (lead && atomicExch(&qp.needed[gb[k]], 1u) == 0u) {
            const uint32_t kk = atomicAdd(qp.n_touched, 1u);
            qp.touched[kk] = gb[k];
            qp.rank[gb[k]] = kk;
        }
    }
}

// ---- a whole small query in ONE launch ------------------------------------------------------------------------
// A query of a few candidate pages is a chain of dependent launches in the round-per-launch form — candidates (with the
// first search riding along), then per further term the touched pages and the round's tail — and each launch costs the
// caller ≈7.6 us (profiles/r02_query_trace_single.txt): 3.2 launches a query on the reference's log. Here ONE
// workgroup walks the whole chain: a step = {the pages of a decode, where they go, the round's tail}; the 16 waves
// take a step's pages in turn (wave w: pages w, w + 16, ...: no queue), a workgroup barrier separates a step's decode
// from its tail and the tail from the next step (one CU, one L1: what a wave stored its neighbours read behind the
// barrier), and the dictionary's LDS image is loaded once for all steps. How many pages a step has is what the
// previous step's tail counted (qp.count), bounded by qp.bound.
struct fused_step {
    uint32_t* out;
    uint64_t out_capacity;
    uint8_t* gaps_left;
    query_pages qp;
    round_tail rt;  // rt.done == null: the step has no tail (the candidate pages)
};

template <bool MULTI>
__device__ __forceinline__ void query_fused_body(const decode_args& kernarg, const fused_step* steps, uint32_t n_steps) {
    const decode_args a = own_scalars(kernarg);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    {
        constexpr uint32_t kSteps = (kHotImageWords / 4 + kBlockThreads - 1) / kBlockThreads;
        u32x4 part[kSteps];
#pragma unroll
        for (uint32_t k = 0; k != kSteps; ++k) {
            const uint32_t i = threadIdx.x + k * kBlockThreads;
            if (4 * i < a.dict.hot_words) part[k] = reinterpret_cast<const u32x4*>(a.dict.lds_image)[i];
        }
#pragma unroll
        for (uint32_t k = 0; k != kSteps; ++k) {
            const uint32_t i = threadIdx.x + k * kBlockThreads;
            if (4 * i < a.dict.hot_words) reinterpret_cast<u32x4*>(lds)[i] = part[k];
        }
    }
    uint16_t* const cls = reinterpret_cast<uint16_t*>(lds + a.dict.hot_words);
    build_class_table(cls);
    uint32_t* const descs = lds + a.dict.hot_words + kDescWordAt;
    if (threadIdx.x < 24) descs[threadIdx.x] = MULTI ? reinterpret_cast<const uint32_t*>(a.dict.descs)[threadIdx.x] : 0u;
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x / kWave);
    uint32_t* const scratch = lds + a.dict.hot_words + kClassTableWords + wave * kScratchWords;
    for (uint32_t i = lane; i < kFwWords; i += kWave) scratch[i] = 0;  // the flag words start out zero
    __syncthreads();
    wave_ctx c;
    c.lds = lds;
    c.cls = cls;
    c.descs = descs;
    c.scratch = scratch;
    c.lane = lane;
    c.rs_dict = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dict.tables), 0, int(a.dict.tables_bytes), 0x00020000);
    c.heads_base = a.dict.heads_base;
    c.tails_base = a.dict.tails_base;
    c.goff_base = a.dict.goff_base;
    c.gtable_base = a.dict.gtable_base;
    prof_t pf;
    for (uint32_t s = 0; s != n_steps; ++s) {
        // (through the constant address space: scalar loads — a step's forty-odd pointers and counts are wave-uniform
        // and belong in scalar registers; as ordinary loads they took 70 vector registers and spilled to scratch)
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(4))) const fused_step constant_step;
        constant_step* const st = (constant_step*)(uintptr_t)(steps + s);
#else
        const fused_step* const st = steps + s;  // (the host pass only parses this)
#endif
        decode_args as = a;
        as.out = st->out;
        as.out_capacity = st->out_capacity;
        as.gaps_left = st->gaps_left;
        query_pages qp;
        round_tail rt;
#if defined(__HIP_DEVICE_COMPILE__)
        {   // (member-wise: an address-space-4 struct has no copy constructor into a generic one)
            typedef __attribute__((address_space(4))) const uint64_t constant_u64;
            constant_u64* const src_q = (constant_u64*)(uintptr_t)&(steps + s)->qp;
            constant_u64* const src_t = (constant_u64*)(uintptr_t)&(steps + s)->rt;
            uint64_t* const dst_q = reinterpret_cast<uint64_t*>(&qp);
            uint64_t* const dst_t = reinterpret_cast<uint64_t*>(&rt);
            static_assert(sizeof(query_pages) % 8 == 0 && sizeof(round_tail) % 8 == 0, "copied as 64-bit words");
#pragma unroll
            for (uint32_t i = 0; i != sizeof(query_pages) / 8; ++i) dst_q[i] = src_q[i];
#pragma unroll
            for (uint32_t i = 0; i != sizeof(round_tail) / 8; ++i) dst_t[i] = src_t[i];
        }
#else
        qp = st->qp;
        rt = st->rt;
#endif
        uint64_t n_work = qp.bound;
        if (qp.count) n_work = uniform(*qp.count) < n_work ? uniform(*qp.count) : n_work;
        for (uint64_t page = wave; page < n_work; page += kWavesPerBlock) decode_query_page<MULTI>(as, c, qp, page, pf);
        __syncthreads();
        if (rt.done) {
            and_round_tail(rt);
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_single_query_fused_kernel(decode_args a, const fused_step* steps,
                                                                                                 uint32_t n_steps) {
    query_fused_body<false>(a, steps, n_steps);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_query_fused_kernel(decode_args a, const fused_step* steps,
                                                                                                uint32_t n_steps) {
    query_fused_body<true>(a, steps, n_steps);
}

__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_single_query_kernel(decode_args a, query_pages qp, round_tail t) {
    decode_kernel_body<false, true, true>(a, &qp, &t);
}
__global__ __launch_bounds__(kBlockThreads, DINT_MIN_WAVES) void decode_multi_query_kernel(decode_args a, query_pages qp, round_tail t) {
    decode_kernel_body<true, true, true>(a, &qp, &t);
}

// test hook: out[i] = inclusive prefix sum of in[0..i] over one wave
__global__ void debug_wave_scan_kernel(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]);
}

}  // namespace dint_dev
