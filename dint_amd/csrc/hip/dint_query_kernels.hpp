// Conjunctive (AND) queries over the in-index layout, batched: the set-at-a-time form of
// and_query<false> (reference include/ds2i/queries.hpp:34-84) and of the document_enumerator's
// next_geq (include/dint/dict_posting_list.hpp:126-147).
//
// The reference walks one candidate at a time through per-list cursors. On the device the same
// result is computed a whole batch of queries at a time:
//   1. the shortest list of every query is decoded into candidate pages (256 slots per block);
//   2. for each further term, every live candidate binary-searches that list's block maxima
//      (the block-max skipping of next_geq), the touched blocks are collected without duplicates,
//      only those blocks are decoded, and every candidate probes its block;
//   3. the survivors are counted per query.
// A block no live candidate falls into is never read, like in the reference.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dint_hip.h"

namespace dint_dev {

constexpr uint32_t kDeadCandidate = 0xFFFFFFFFu;  // not a docID: docIDs are < num_docs <= 2^32 - 1
constexpr uint32_t kPageSlots = 256;              // one block per page

// sub[i] = blocks[ids[i]] relocated to page i
// (count set: page i >= *count is empty — n_pages is then what the host knows, an upper bound)
__global__ void gather_pages_kernel(const dint_block_ref* blocks, const uint32_t* ids, uint64_t n_pages,
                                    dint_block_ref* sub, const uint32_t* count = nullptr) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_pages) return;
    dint_block_ref r{};
    if (!count || i < *count) r = blocks[ids[i]];
    r.out_off = i * kPageSlots;
    sub[i] = r;
}

// Everything a decode of pages needs, in one launch: page i < *count (count null: i < bound) is block ids[i], relocated to
// page i; the others (bound is what the host knows — the device knows how many pages there are) are empty. -> the pages'
// block records, their docs parts as units (byte span: up to where the next block of the index begins), their docID
// bases, the list of the short (interpolative) ones, and cleared "left as gaps" flags.
__global__ void prepare_pages_kernel(const dint_block_ref* blocks, uint64_t n_blocks_total, uint64_t index_bytes, const uint32_t* ids,
                                     const uint32_t* count, uint64_t bound, dint_block_ref* sub, dint_unit* units, uint32_t* spans,
                                     uint32_t* bases, uint8_t* gaps_left, uint32_t* tails, uint32_t* n_tails) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    const uint64_t n_pages = count ? uint64_t(*count) : bound;
    dint_block_ref r{};
    dint_unit u{};
    uint32_t span = 0;
    if (i < n_pages) {
        const uint64_t gb = ids[i];
        r = blocks[gb];
        uint64_t nxt = gb + 1 < n_blocks_total ? blocks[gb + 1].in_off : index_bytes;
        if (nxt <= r.in_off || nxt > index_bytes) nxt = index_bytes;
        const uint64_t sp = nxt > r.in_off ? nxt - r.in_off : 0;
        span = sp > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(sp);
        if (r.n != 0 && r.n < 256) tails[atomicAdd(n_tails, 1u)] = uint32_t(i);
    }
    r.out_off = i * kPageSlots;
    u.in_off = r.in_off;
    u.out_off = r.out_off;
    u.n = r.n;
    u.list = r.list;
    sub[i] = r;
    units[i] = u;
    spans[i] = span;
    bases[i] = r.base;
    gaps_left[i] = 0;
}

// first index in [0, n) with a[i] >= key (n if none)
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, len = n;
    while (len) {
        const uint32_t half = len >> 1;
        const bool right = a[lo + half] < key;
        lo = right ? lo + half + 1 : lo;
        len = right ? len - half - 1 : half;
    }
    return lo;
}

// Claims: a round's searches name the blocks that must be decoded, each ONCE — whoever claims a block first appends it to the
// touched list, and its place there (its page in the probe buffer) is what every candidate of the block looks up afterwards.
// Two forms. Dense (hash_mask == 0): a flag and a rank per block of the INDEX (`needed`, `rank`: n_blocks words each) — the
// forms in which one set serves the whole call. Hashed (hash_mask = capacity - 1): `needed` holds block + 1 at the block's
// slot of an open-addressing table, `rank` the place — a table per WORKGROUP for the workgroup-per-query batch form, sized
// for a query's rounds (at most 4096 touched blocks), not for the index. The keys are written and read at the L2 (atomics):
// a workgroup walks query after query over the same table, and this CU's L1 may still hold a line of the query before.
__device__ __forceinline__ uint32_t claim_slot(uint32_t gb, uint32_t hash_mask) { return ((gb * 2654435761u) >> 9) & hash_mask; }
__device__ __forceinline__ void claim_block(uint32_t* needed, uint32_t* rank, uint32_t* touched, uint32_t* n_touched, uint32_t hash_mask,
                                            uint32_t gb) {
    if (hash_mask == 0) {
        if (atomicExch(&needed[gb], 1u) == 0u) {
            const uint32_t k = atomicAdd(n_touched, 1u);
            touched[k] = gb;
            rank[gb] = k;
        }
        return;
    }
    for (uint32_t h = claim_slot(gb, hash_mask);; h = (h + 1u) & hash_mask) {
        const uint32_t old = atomicCAS(&needed[h], 0u, gb + 1u);
        if (old == 0u) {
            const uint32_t k = atomicAdd(n_touched, 1u);
            touched[k] = gb;
            __hip_atomic_store(&rank[h], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (old == gb + 1u) return;
    }
}
// the place of a claimed block in the touched list (a later phase than its claim: a barrier lies between)
__device__ __forceinline__ uint32_t claimed_rank(const uint32_t* needed, const uint32_t* rank, uint32_t hash_mask, uint32_t gb) {
    if (hash_mask == 0) return rank[gb];
    for (uint32_t h = claim_slot(gb, hash_mask);; h = (h + 1u) & hash_mask) {
        const uint32_t key = __hip_atomic_load(&needed[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (key == gb + 1u) return __hip_atomic_load(&rank[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (key == 0u) return 0u;  // (not claimed: no caller asks)
    }
}

// A whole round behind its decode, in the decode's own launch (few candidates — a single query: a launch less is a
// tenth of what the caller waits): the workgroup of decode_*_query_kernel that finishes last probes every candidate in
// the pages just decoded (and_probe_release_kernel), and either searches the NEXT round's list for the survivors
// (and_search_kernel) or, behind the last round, counts them and hands the results to the host. Two rounds' claim
// flags / ranks / touched lists are live at once here — this round's are read and released while the next round's
// are written; a list may be one query's term in this round and another's in the next — so rounds alternate
// between two sets of them.
struct round_tail {
    uint32_t* done;  // null: no tail. Zero at launch: the workgroups that have finished
    uint32_t* cand;
    uint64_t n_slots;
    const uint32_t* page_query;
    const dint_block_ref* blocks;
    uint32_t* target;
    // this round
    const uint32_t* term_blocks;
    const uint32_t* rank;
    const uint32_t* probe;
    const uint32_t* touched;
    const uint32_t* n_touched;
    uint32_t* needed;
    // the next round (next_blocks null: this was the last one)
    const uint32_t* next_first;
    const uint32_t* next_blocks;
    const uint32_t* block_max;
    uint32_t* next_needed;
    uint32_t* next_rank;
    uint32_t* next_touched;
    uint32_t* next_n_touched;
    // the last round
    unsigned long long* counts;
    unsigned long long* host_counts;  // nullable (pinned host memory, as the device sees it)
    uint32_t n_queries;
    uint32_t q_first;                 // and_round_tail (one workgroup): the queries [q_first, q_first + n_queries) are handed over
    uint32_t hash_mask;               // and_round_tail: 0, or the claim tables are hashed ones of this capacity - 1 (claim_block)
};

// (every thread of one workgroup; n_slots is a multiple of 256, so wavefronts stay whole inside the loop)
__device__ __forceinline__ void and_round_tail(const round_tail& t) {
    const uint32_t nt = *t.n_touched;
    // (release of this round's claims: dense flags right away — the probes below read ranks, not flags; a hashed table holds
    // the ranks' keys and is cleared behind the probes, see the end)
    if (t.hash_mask == 0)
        for (uint32_t k = threadIdx.x; k < nt; k += blockDim.x) t.needed[t.touched[k]] = 0;
    for (uint64_t i = threadIdx.x; i < t.n_slots; i += blockDim.x) {
        uint32_t gb = kDeadCandidate;
        // (what a candidate needs is asked for in as few dependent trips as its data allow: {candidate, its query, its target
        // block} -> {the query's block counts in this round and the next} -> {the target's size and page} -> the probe -> the
        // next round's block maxima. K-ary searches — 15 pivots a step, 2 trips for a block instead of 8 — were measured:
        // 29.0 against 27.4 us a query, the one CU's load issue is what they cost)
        const uint32_t c = t.cand[i];
        const uint32_t q = t.page_query[i / kPageSlots];
        const uint32_t b_raw = t.target[i];  // (only meaningful where this round has a term for q: masked below)
        if (c != kDeadCandidate) {
            const uint32_t tb = t.term_blocks[q];
            uint32_t nb = 0, fb = 0;
            if (t.next_blocks) nb = t.next_blocks[q], fb = t.next_first[q];
            bool alive = true;
            if (tb != 0) {
                const uint32_t n = t.blocks[b_raw].n;
                const uint32_t* page = t.probe + uint64_t(claimed_rank(t.needed, t.rank, t.hash_mask, b_raw)) * kPageSlots;
                const uint32_t pos = lower_bound_u32(page, n, c);
                alive = pos != n && page[pos] == c;
            }
            if (alive && nb) {
                const uint32_t pos = lower_bound_u32(t.block_max + fb, nb, c);
                if (pos == nb) {
                    alive = false;
                } else {
                    gb = fb + pos;
                    t.target[i] = gb;
                }
            }
            if (!alive) t.cand[i] = kDeadCandidate;
            else if (!t.next_blocks) atomicAdd(&t.counts[q], 1ull);
        }
        const uint32_t prev = __shfl_up(gb, 1);
        const bool lead = gb != kDeadCandidate && ((threadIdx.x & 63u) == 0 || prev != gb);
        if (lead) claim_block(t.next_needed, t.next_rank, t.next_touched, t.next_n_touched, t.hash_mask, gb);
    }
    if (t.hash_mask != 0 && nt != 0) {  // every probe has read its rank: this round's table is cleared for the round after next
        __syncthreads();
        for (uint32_t h = threadIdx.x; h <= t.hash_mask; h += blockDim.x) t.needed[h] = 0;
    }
    if (t.next_blocks || !t.host_counts) return;
    __syncthreads();
    for (uint32_t q = t.q_first + threadIdx.x; q < t.q_first + t.n_queries; q += blockDim.x)
        t.host_counts[q] = __hip_atomic_load(&t.counts[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same for a BATCH (a thread per candidate slot, a workgroup per page): the probe of round r, the release of its claims
// and the block-max search of round r + 1 in ONE launch — until round 5 the batch path ran them as two (and_probe_release_kernel,
// and_search_kernel): a launch per round less, of three. The two rounds' claim flags / ranks / touched lists alternate
// between two sets, as in and_round_tail. Behind the last round: the survivors counted per query and handed to the host by
// the workgroup that finishes last (t.done: zero at launch).
__global__ void and_round_tail_kernel(round_tail t) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < *t.n_touched) t.needed[t.touched[i]] = 0;
    uint32_t gb = kDeadCandidate;
    bool alive = false;
    if (i < t.n_slots) {
        const uint32_t c = t.cand[i];
        if (c != kDeadCandidate) {
            const uint32_t q = t.page_query[i / kPageSlots];
            alive = true;
            if (t.term_blocks[q] != 0) {
                const uint32_t b = t.target[i];
                const uint32_t n = t.blocks[b].n;
                const uint32_t* page = t.probe + uint64_t(t.rank[b]) * kPageSlots;
                const uint32_t pos = lower_bound_u32(page, n, c);
                alive = pos != n && page[pos] == c;
            }
            if (alive && t.next_blocks) {
                const uint32_t nb = t.next_blocks[q];
                if (nb) {
                    const uint32_t fb = t.next_first[q];
                    const uint32_t pos = lower_bound_u32(t.block_max + fb, nb, c);
                    if (pos == nb) {
                        alive = false;  // next_geq past the last block: m_universe, dict_posting_list.hpp:128-131
                    } else {
                        gb = fb + pos;
                        t.target[i] = gb;
                    }
                }
            }
            if (!alive) t.cand[i] = kDeadCandidate;
        }
    }
    if (t.next_blocks) {  // (uniform) the next round's claims: one per run of candidates that fall into the same block
        const uint32_t prev = __shfl_up(gb, 1);
        const bool lead = gb != kDeadCandidate && ((threadIdx.x & 63u) == 0 || prev != gb);
        if (lead && atomicExch(&t.next_needed[gb], 1u) == 0u) {
            const uint32_t k = atomicAdd(t.next_n_touched, 1u);
            t.next_touched[k] = gb;
            t.next_rank[gb] = k;
        }
        return;
    }
    const int n = __syncthreads_count(alive);
    if (threadIdx.x == 0 && n) atomicAdd(&t.counts[t.page_query[i / kPageSlots]], (unsigned long long)n);
    if (!t.host_counts) return;  // (uniform)
    __shared__ uint32_t last;
    if (gridDim.x != 1) {
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(t.done, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!last) return;
        __threadfence();
    }
    for (uint32_t q = threadIdx.x; q < t.n_queries; q += blockDim.x)
        t.host_counts[q] = __hip_atomic_load(&t.counts[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Round step A: block-max search. term_first/term_blocks give, per query, the block range of
// this round's list (term_blocks == 0: the query has no such term and its candidates pass).
// Touched blocks are appended once to `touched`, and rank[block] is their position there.
__global__ void and_search_kernel(uint32_t* cand, uint64_t n_slots, const uint32_t* page_query,
                                  const uint32_t* term_first, const uint32_t* term_blocks,
                                  const uint32_t* block_max, uint32_t* target, uint32_t* needed, uint32_t* rank,
                                  uint32_t* touched, uint32_t* n_touched) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t gb = kDeadCandidate;
    if (i < n_slots) {
        const uint32_t c = cand[i];
        if (c != kDeadCandidate) {
            const uint32_t q = page_query[i / kPageSlots];
            const uint32_t nb = term_blocks[q];
            if (nb) {
                const uint32_t fb = term_first[q];
                const uint32_t pos = lower_bound_u32(block_max + fb, nb, c);
                if (pos == nb) {
                    cand[i] = kDeadCandidate;  // next_geq past the last block: m_universe, :128-131
                } else {
                    gb = fb + pos;
                    target[i] = gb;
                }
            }
        }
    }
    // neighbouring candidates are sorted, so they mostly fall into the same block: one claim per run
    const uint32_t prev = __shfl_up(gb, 1);
    const bool lead = gb != kDeadCandidate && ((threadIdx.x & 63u) == 0 || prev != gb);
    if (lead && atomicExch(&needed[gb], 1u) == 0u) {
        const uint32_t k = atomicAdd(n_touched, 1u);
        touched[k] = gb;
        rank[gb] = k;
    }
}

// Round steps B and C in one launch: each live candidate looks itself up in its (now decoded) block, and — thread k, for the k-th touched block — the claim flag cleared for
// the next round (the probe reads target / rank, not the flags). `counts` (the last round): the survivors of the
// workgroup's page — one query's — are added to the query's result on the spot (and_count_kernel's job);
// `host_counts`: ... and the results written to the host's (pinned) memory by the last workgroup to finish.
__global__ void and_probe_release_kernel(uint32_t* cand, uint64_t n_slots, const uint32_t* page_query, const uint32_t* term_blocks,
                                         const dint_block_ref* blocks, const uint32_t* target, const uint32_t* rank,
                                         const uint32_t* probe, const uint32_t* touched, const uint32_t* n_touched, uint32_t* needed,
                                         unsigned long long* counts, uint32_t* done, unsigned long long* host_counts, uint32_t n_queries) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < *n_touched) needed[touched[i]] = 0;
    bool alive = false;
    if (i < n_slots) {
        const uint32_t c = cand[i];
        if (c != kDeadCandidate) {
            alive = true;
            if (term_blocks[page_query[i / kPageSlots]] != 0) {
                const uint32_t gb = target[i];
                const uint32_t n = blocks[gb].n;
                const uint32_t* page = probe + uint64_t(rank[gb]) * kPageSlots;
                const uint32_t pos = lower_bound_u32(page, n, c);
                if (pos == n || page[pos] != c) {
                    cand[i] = kDeadCandidate;
                    alive = false;
                }
            }
        }
    }
    if (!counts) return;  // (uniform)
    const int n = __syncthreads_count(alive);
    if (threadIdx.x == 0 && n) atomicAdd(&counts[page_query[i / kPageSlots]], (unsigned long long)n);
    if (!host_counts) return;  // (uniform)
    // ... and the workgroup that finishes last hands the results to the host (pinned memory, written from here: no
    // copy back, one stream operation less for the caller to wait for)
    __shared__ uint32_t last;
    if (gridDim.x != 1) {  // (one workgroup: nobody to wait for, and no fence — an agent-scope fence costs microseconds here)
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!last) return;
        __threadfence();
    }
    for (uint32_t q = threadIdx.x; q < n_queries; q += blockDim.x)
        host_counts[q] = __hip_atomic_load(&counts[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// and_query<true>, step A per term: the block each match (surviving candidate) falls into — for the rarest term
// (first == nullptr) the candidate page's own block — claimed once, as in and_search_kernel.
__global__ void and_freq_search_kernel(const uint32_t* cand, uint64_t n_slots, const uint32_t* page_query, const uint32_t* page_block,
                                       const uint32_t* term_first, const uint32_t* term_blocks, const uint32_t* block_max,
                                       uint32_t* target, uint32_t* needed, uint32_t* rank, uint32_t* touched, uint32_t* n_touched) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t gb = kDeadCandidate;
    if (i < n_slots) {
        const uint32_t c = cand[i];
        if (c != kDeadCandidate) {
            if (!term_first) {
                gb = page_block[i / kPageSlots];
            } else {
                const uint32_t q = page_query[i / kPageSlots];
                const uint32_t nb = term_blocks[q];
                if (nb) gb = term_first[q] + lower_bound_u32(block_max + term_first[q], nb, c);  // (a match: always found)
            }
            target[i] = gb;
        }
    }
    const uint32_t prev = __shfl_up(gb, 1);
    const bool lead = gb != kDeadCandidate && ((threadIdx.x & 63u) == 0 || prev != gb);
    if (lead && atomicExch(&needed[gb], 1u) == 0u) {
        const uint32_t k = atomicAdd(n_touched, 1u);
        touched[k] = gb;
        rank[gb] = k;
    }
}

// ... step B: every match finds its docID in its block's decoded page and adds the freq at that position to its query's
// sum (document_enumerator::freq(), dict_posting_list.hpp:164-169; queries.hpp:72-76 reads one per term and match).
__global__ void and_freq_gather_kernel(const uint32_t* cand, uint64_t n_slots, const uint32_t* page_query, const uint32_t* term_blocks,
                                       const dint_block_ref* blocks, const uint32_t* target, const uint32_t* rank,
                                       const uint32_t* probe, const uint32_t* fprobe, unsigned long long* freq_sums) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const uint32_t c = cand[i];
    if (c == kDeadCandidate) return;
    const uint32_t q = page_query[i / kPageSlots];
    if (term_blocks && term_blocks[q] == 0) return;  // the query has no such term
    const uint32_t gb = target[i];
    const uint32_t n = blocks[gb].n;
    const uint64_t page = uint64_t(rank[gb]) * kPageSlots;
    const uint32_t pos = lower_bound_u32(probe + page, n, c);
    if (pos < n && probe[page + pos] == c) atomicAdd(&freq_sums[q], (unsigned long long)fprobe[page + pos]);
}

__global__ void and_release_kernel(const uint32_t* touched, uint32_t n_touched, uint32_t* needed, const uint32_t* count = nullptr) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_touched && (!count || k < *count)) needed[touched[k]] = 0;
}
// results += 1 per surviving candidate (queries.hpp:72-76); a page belongs to one query
__global__ void and_count_kernel(const uint32_t* cand, uint64_t n_slots, const uint32_t* page_query,
                                 unsigned long long* counts) {
    // (a workgroup = the 256 slots of one page = one query: one atomic per page that has a survivor)
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool alive = i < n_slots && cand[i] != kDeadCandidate;
    const int n = __syncthreads_count(alive);
    if (threadIdx.x == 0 && n) atomicAdd(&counts[page_query[i / kPageSlots]], (unsigned long long)n);
}

}  // namespace dint_dev
