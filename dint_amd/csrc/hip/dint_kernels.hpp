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
#ifndef DINT_MAX_STRAGGLERS
#define DINT_MAX_STRAGGLERS 4096
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
// metadata word of a codeword: (size - 1) << 24 | kMetaCold | kMetaSlow | cells << 20 | LDS byte offset
constexpr uint32_t kMetaCold = 1u << 23;              // the integers come through staging cells (row table)
constexpr uint32_t kMetaSlow = 1u << 22;              // ... or, with this bit, from gtable through slow_stores
constexpr uint32_t kMetaOffMask = (1u << 20) - 2;     // hot: byte offset of the integers (u16 each: even) in the LDS image; else 0
                                                      // bits 20-21: staging cells a cold codeword takes (1: up to 6 integers, 2: up to 14, 3)
constexpr uint32_t kMetaPacked8 = 1u;                // a COLD codeword of eight integers, all below 256: its 16-byte head holds them all, a byte each
                                                      // (two staging cells, no tail request) — 6.2 % of the bench stream's codewords are such,
                                                      // every one a 32-byte request less
constexpr uint32_t kMetaException = 1u << 20;         // what the two exception markers (slot values 0 and 1) look up: one integer, one
                                                      // staging cell — the literal that follows the marker in the stream goes there
constexpr uint32_t kQueueShards = 8;                  // dynamic unit queue: one counter per shard
constexpr uint32_t kQueueStride = 32;                 // words between counters (own 128-byte line each)
constexpr uint32_t kChunkShards = 32;                 // the bundle path's chunk tickets: counters (a line each), four to an XCD's workgroups
constexpr uint32_t kChunkCounterLineAt = 1;           // ... which begin this many lines behind decode_args::chunk_queue
// (the unit queue stays at one counter per shard: four to a shard, the waves of a workgroup spread over them, was measured
// on the 1e9-posting run — 1.496 against 1.478 ms, no gain: its tickets are asked for a work item ahead and are few)
constexpr uint32_t kQueueLines = kQueueShards + 1 + kChunkShards;  // a launch's counters: unit-queue shards | the clock's line | chunk counters
constexpr uint32_t kClockWordAt = 16;                 // in the chunk counter's line: shader-clock cycles of the launch's first wave (u64)
constexpr uint32_t kChunkDryWordAt = 24;              // ... a bit per chunk counter that a wave has found dry (next_open_counter)
constexpr uint32_t kUnitDryWordAt = 25;               // ... and per unit-queue shard
constexpr uint32_t kMaxUnitInts = 1u << 28;           // byte offsets inside a unit's output stay 32-bit

#include "kernels/wave_basics.inc"
#include "kernels/tile.inc"
#include "kernels/segment.inc"
#include "kernels/bundles.inc"

}  // namespace dint_dev
#include "dint_query_kernels.hpp"  // round_tail: what a query round does behind its decode
namespace dint_dev {

#include "kernels/index.inc"
#include "kernels/kernel_body.inc"
#include "kernels/query_pages.inc"

// test hook: out[i] = inclusive prefix sum of in[0..i] over one wave
__global__ void debug_wave_scan_kernel(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]);
}

}  // namespace dint_dev

