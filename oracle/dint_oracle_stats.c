/*
 * dint_oracle_stats.c — CPU restatement of the reference's dictionary CONSTRUCTION statistics:
 * the block selector, adjusted::collect (both flavours), the saving filter, the freq/length order and the
 * decreasing-static-frequencies cut. Plain C, part of liboracle.so.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see dint_oracle.h): the checker of the device's n-gram counting and selection
 * (dint_count_ngrams / dint_select_ngrams) and of the host library's statistics. PARITY UNPINNED: the reference
 * headers restated here (statistics_collectors.hpp, block_statistics.hpp, dictionary_builders.hpp) need util.hpp,
 * which needs <succinct/broadword.hpp> — absent; stand-ins are not allowed — so this follows the source as read;
 * every function cites its lines. The one piece that IS pinned to the reference is the key function: oracle_hash_u32s
 * equals ds2i::hash_bytes64 of oracle/_ref/libref_hash.so (tests/golden/murmur_vectors.json, tests/test_oracle_cpu.py).
 *
 * Where the reference's result depends on libstdc++ (the order of entries with equal frequency AND equal length:
 * unordered_map iteration order fed to std::sort, block_statistics.hpp:87-106) this file orders them by their integers,
 * lexicographically — the documented tie order of the product (dint/statistics.hpp); tests compare the selected sets
 * "up to ties at the cut": everything strictly above the cut's (frequency, length) class must agree, the class at the
 * cut must agree in size.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dint_oracle.h"

/* include/dint/dint_configuration.hpp:20,24-28; include/util.hpp:35 */
enum { NUM_SELECTORS = 6, MAX_ENTRY_SIZE = 16, NUM_TARGET_SIZES = 5, NUM_ENTRIES = 65536, BLOCK_SIZE = 256 };
static const uint32_t target_sizes[NUM_TARGET_SIZES] = {16, 8, 4, 2, 1};

/* util.hpp:67-70: ceil_log2(x) = (x > 1) ? msb(x - 1) + 1 : 0 */
static uint64_t ceil_log2_u64(uint64_t x) {
    if (x <= 1) return 0;
    uint64_t v = x - 1, msb = 0;
    while (v >>= 1) ++msb;
    return msb + 1;
}

/* statistics_collectors.hpp:21-40: selector::get — the block's maximum x; code = ceil_log2(ceil_log2(x + 1)), 0 if x <= 1.
 * `x + 1` is uint32_t arithmetic there (:23, :36): at x = 0xFFFFFFFF it wraps to 0, ceil_log2(0) is 0 in a release build
 * (util.hpp:67-70, the assert compiled out), and the block lands in context 0 — restated as written. */
uint32_t oracle_selector_get(const uint32_t* entry, size_t n) {
    uint32_t x = 0;
    for (const uint32_t* p = entry; p != entry + n; ++p)
        if (*p > x) x = *p;
    uint32_t selector_code = 0;
    if (x > 1) selector_code = (uint32_t)ceil_log2_u64(ceil_log2_u64((uint32_t)(x + 1u)));
    return selector_code;
}

/* hash_utils.hpp:7-71, :77-80: MurmurHash64A (published algorithm) of the integers' bytes, seed 0. The keys are whole
 * u32 words: len & 7 is 0 or 4, the tail switch's cases 4..1. */
uint64_t oracle_hash_u32s(const uint32_t* p, size_t n) {
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    const int r = 47;
    const size_t len = n * 4;
    uint64_t h = 0 ^ (len * m);
    const unsigned char* d = (const unsigned char*)p;
    for (size_t i = 0; i != len / 8; ++i, d += 8) {
        uint64_t k;
        memcpy(&k, d, 8);
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    switch (len & 7) {
        case 7: h ^= (uint64_t)d[6] << 48; /* fall through */
        case 6: h ^= (uint64_t)d[5] << 40; /* fall through */
        case 5: h ^= (uint64_t)d[4] << 32; /* fall through */
        case 4: h ^= (uint64_t)d[3] << 24; /* fall through */
        case 3: h ^= (uint64_t)d[2] << 16; /* fall through */
        case 2: h ^= (uint64_t)d[1] << 8;  /* fall through */
        case 1: h ^= (uint64_t)d[0]; h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

/* map_type (statistics_collectors.hpp:19): hash -> block_type{freq, data}; keyed by the hash ALONE (increase_frequency,
 * :66-80: a second n-gram with the same hash only counts, its integers are never compared). Open addressing here. */
typedef struct {
    uint64_t hash;
    uint64_t freq;
    uint64_t pos; /* where the first occurrence's integers are (block.data, copied at the first sight, :74-78) */
    uint32_t len; /* 0 = empty slot */
} slot_t;

typedef struct {
    slot_t* s;
    size_t cap, used;
} map_t;

struct oracle_stats {
    map_t maps[NUM_SELECTORS];
    uint32_t n_maps;
    uint64_t total_integers;
    const uint32_t* gaps; /* the caller's, must outlive the object */
};

static int map_grow(map_t* m) {
    const size_t ncap = m->cap ? m->cap * 2 : 1024;
    slot_t* ns = (slot_t*)calloc(ncap, sizeof(slot_t));
    if (!ns) return 0;
    for (size_t i = 0; i != m->cap; ++i)
        if (m->s[i].len) {
            size_t j = (size_t)(m->s[i].hash * 0x9E3779B97F4A7C15ULL >> 20) & (ncap - 1);
            while (ns[j].len) j = (j + 1) & (ncap - 1);
            ns[j] = m->s[i];
        }
    free(m->s);
    m->s = ns;
    m->cap = ncap;
    return 1;
}

/* statistics_collectors.hpp:66-80: increase_frequency */
static int increase_frequency(map_t* m, const uint32_t* base, uint64_t pos, uint32_t n) {
    if (m->used * 2 >= m->cap && !map_grow(m)) return 0;
    const uint64_t hash = oracle_hash_u32s(base + pos, n);
    size_t j = (size_t)(hash * 0x9E3779B97F4A7C15ULL >> 20) & (m->cap - 1);
    while (m->s[j].len && m->s[j].hash != hash) j = (j + 1) & (m->cap - 1);
    if (m->s[j].len) {
        m->s[j].freq += 1;
    } else {
        m->s[j].hash = hash;
        m->s[j].freq = 1; /* block_type() : freq(1), :9 */
        m->s[j].pos = pos;
        m->s[j].len = n;
        m->used += 1;
    }
    return 1;
}

oracle_stats* oracle_stats_create(int multi, const uint32_t* gaps) {
    oracle_stats* st = (oracle_stats*)calloc(1, sizeof(oracle_stats));
    if (!st) return NULL;
    st->n_maps = multi ? NUM_SELECTORS : 1;
    st->gaps = gaps;
    return st;
}

void oracle_stats_free(oracle_stats* st) {
    if (!st) return;
    for (uint32_t c = 0; c != NUM_SELECTORS; ++c) free(st->maps[c].s);
    free(st);
}

/* One list of the collection, gaps[first, first + n): block_statistics.hpp:62-81 / :229-248 (total_integers += n, then
 * Collector::collect on the list's gaps).
 * single: adjusted::collect(buf, block_map), statistics_collectors.hpp:109-118 — for every target size, the list's
 *         aligned n-grams (the remainder that does not fill one is dropped);
 * multi:  adjusted::collect(buf, block_maps), :90-107 — whole 256-integer blocks only, each into the map of its
 *         selector, every aligned n-gram of the block. */
int oracle_stats_collect(oracle_stats* st, uint64_t first, uint64_t n) {
    st->total_integers += n;
    const uint32_t* b = st->gaps;
    if (st->n_maps == 1) {
        for (uint32_t s = 0; s != NUM_TARGET_SIZES; ++s) {
            const uint32_t block_size = target_sizes[s];
            const uint64_t blocks = n / block_size;
            for (uint64_t i = 0, pos = 0; i != blocks; ++i, pos += block_size)
                if (!increase_frequency(&st->maps[0], b, first + pos, block_size)) return 0;
        }
        return 1;
    }
    const uint64_t blocks = n / BLOCK_SIZE;
    for (uint64_t i = 0, pos = 0; i != blocks; ++i, pos += BLOCK_SIZE) {
        const uint32_t index = oracle_selector_get(b + first + pos, BLOCK_SIZE);
        if (index >= NUM_SELECTORS) return 0; /* (the reference indexes block_maps[index] unchecked: x < 2^32 gives <= 5) */
        for (uint32_t s = 0; s != NUM_TARGET_SIZES; ++s) {
            const uint32_t jump_size = target_sizes[s];
            const uint32_t jumps = BLOCK_SIZE / jump_size;
            for (uint32_t j = 0, p = 0; j != jumps; ++j, p += jump_size)
                if (!increase_frequency(&st->maps[index], b, first + pos + p, jump_size)) return 0;
        }
    }
    return 1;
}

uint64_t oracle_stats_total(const oracle_stats* st) { return st->total_integers; }

uint64_t oracle_stats_distinct(const oracle_stats* st, uint32_t context) {
    return context < st->n_maps ? st->maps[context].used : 0;
}

/* every distinct n-gram of a context, in no particular order */
uint64_t oracle_stats_entries(const oracle_stats* st, uint32_t context, oracle_ngram* out, uint64_t cap) {
    if (context >= st->n_maps) return 0;
    const map_t* m = &st->maps[context];
    uint64_t k = 0;
    for (size_t i = 0; i != m->cap; ++i)
        if (m->s[i].len) {
            if (k < cap) {
                out[k].pos = m->s[i].pos;
                out[k].freq = m->s[i].freq;
                out[k].len = m->s[i].len;
                out[k].context = context;
            }
            ++k;
        }
    return k;
}

/* dictionary_builders.hpp:15-38: cost(), compute_saving(), cost_filter — with the uint32_t truncation of the frequency
 * the reference's signatures impose (block_frequency is a uint32_t parameter) */
static double compute_saving(uint32_t block_size, uint32_t block_frequency, uint64_t total_integers) {
    const double codeword_bits = log2((double)NUM_ENTRIES); /* :15 */
    const double initial_bpi = 3 * codeword_bits;           /* :16 */
    return block_frequency * (initial_bpi * block_size - codeword_bits) / total_integers;
}

static const uint32_t* g_sort_gaps;
/* statistics_collectors.hpp:57-64: freq_length_sorter — frequency descending, then length descending; equal in both: by
 * the integers (this file's header) */
static int by_freq_length(const void* a, const void* b) {
    const oracle_ngram* l = (const oracle_ngram*)a;
    const oracle_ngram* r = (const oracle_ngram*)b;
    if (l->freq != r->freq) return l->freq > r->freq ? -1 : 1;
    if (l->len != r->len) return l->len > r->len ? -1 : 1;
    for (uint32_t i = 0; i != l->len; ++i) {
        const uint32_t x = g_sort_gaps[l->pos + i], y = g_sort_gaps[r->pos + i];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

/* The entries decreasing_static_frequencies::build appends for one context (dictionary_builders.hpp:55-75): the blocks
 * that pass filter() (= cost_filter(eps / 1000), :50-53, eps = 0.0001 :17) or hold a single integer
 * (block_statistics.hpp:95-96 / :262-263), sorted by freq_length_sorter (:104-106 / :268-276), the first
 * min(num_entries, size) of them (:61-66). Returns how many pass the filter (before the cut); writes at most `cap`. NOT
 * thread-safe (qsort comparator state). */
uint64_t oracle_stats_select(const oracle_stats* st, uint32_t context, oracle_ngram* out, uint64_t cap) {
    if (context >= st->n_maps) return 0;
    const map_t* m = &st->maps[context];
    oracle_ngram* all = (oracle_ngram*)malloc((m->used ? m->used : 1) * sizeof(oracle_ngram));
    if (!all) return 0;
    const double threshold = 0.0001 / 1000;
    uint64_t k = 0;
    for (size_t i = 0; i != m->cap; ++i)
        if (m->s[i].len) {
            const slot_t* s = &m->s[i];
            if (compute_saving(s->len, (uint32_t)s->freq, st->total_integers) > threshold || s->len == 1) {
                all[k].pos = s->pos;
                all[k].freq = s->freq;
                all[k].len = s->len;
                all[k].context = context;
                ++k;
            }
        }
    g_sort_gaps = st->gaps;
    qsort(all, k, sizeof(oracle_ngram), by_freq_length);
    uint64_t n = NUM_ENTRIES;
    if (k < n) n = k;
    for (uint64_t i = 0; i != n && i != cap; ++i) out[i] = all[i];
    free(all);
    return k;
}
