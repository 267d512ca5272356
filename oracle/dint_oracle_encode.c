/*
 * dint_oracle_encode.c — CPU restatement of the reference's DINT ENCODE side and of the dictionary
 * PACKING (f2's last half): the builders' load / prepare_for_encoding / lookup with their hash-only quirks, the optimal-
 * parse and greedy encoders (whole-list and in-index block flavours), multi_opt_dint's exhaustive selector search,
 * interpolative_block::encode with its bit_writer, dict_posting_list::write, the vroom framing of jobs.hpp /
 * encode.cpp over a binary_collection, and builder::init / append / build / write with pack_policy::compact.
 * Plain C11, part of liboracle.so.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see dint_oracle.h): the checker of the product's encoders (dint/encoders.hpp,
 * dint/posting_list.hpp, dint/vroom_stream.hpp, dint/dictionaries.hpp) — "product bytes == oracle bytes".
 * PARITY UNPINNED: the reference headers restated here need <succinct/...> and Boost (absent; stand-ins are not
 * allowed), so this file follows the source as read, line by line, and every function cites the lines it follows.
 * It deliberately keeps the reference's shapes (the n + 2 node path with its dummy nodes, the reversed `encoding`
 * vector with the final dummy node, twelve candidate encodings per multi block, O(n^2) compaction, std::search per
 * entry) where the product restructured them: the two are independent statements of the same algorithm.
 *
 * Citations are relative to /root/reference.
 */
#include <stdlib.h>
#include <string.h>

#include "dint_oracle.h"

/* include/dint/dint_configuration.hpp:6,20,24-28; include/util.hpp:35 */
enum { EXCEPTIONS = 2, NUM_SELECTORS = 6, MAX_ENTRY_SIZE = 16, NUM_TARGET_SIZES = 5, NUM_ENTRIES = 65536, BLOCK_SIZE = 256 };
enum { RESERVED = EXCEPTIONS + 5 }; /* single_dictionary.hpp:22, multi_dictionary.hpp:23, rectangular_dictionary.hpp:20 */
static const uint32_t target_sizes[NUM_TARGET_SIZES] = {16, 8, 4, 2, 1};
#define INVALID_INDEX 0xFFFFFFFFu /* single_dictionary.hpp:21 */

/* ---- std::vector<uint8_t> -------------------------------------------------------------------------------------- */

void oracle_bytes_free(oracle_bytes* b) {
    if (!b) return;
    free(b->data);
    b->data = NULL;
    b->size = b->cap = 0;
}

static int bytes_reserve(oracle_bytes* b, size_t more) {
    if (b->size + more <= b->cap) return 1;
    size_t ncap = b->cap ? b->cap : 64;
    while (ncap < b->size + more) ncap *= 2;
    uint8_t* nd = (uint8_t*)realloc(b->data, ncap);
    if (!nd) return 0;
    b->data = nd;
    b->cap = ncap;
    return 1;
}
static int bytes_push(oracle_bytes* b, uint8_t v) {
    if (!bytes_reserve(b, 1)) return 0;
    b->data[b->size++] = v;
    return 1;
}
static int bytes_append(oracle_bytes* b, const void* p, size_t n) {
    if (!n) return 1;
    if (!bytes_reserve(b, n)) return 0;
    memcpy(b->data + b->size, p, n);
    b->size += n;
    return 1;
}
static int bytes_resize(oracle_bytes* b, size_t n) { /* vector::resize: new bytes are zero */
    if (n > b->size) {
        if (!bytes_reserve(b, n - b->size)) return 0;
        memset(b->data + b->size, 0, n - b->size);
    }
    b->size = n;
    return 1;
}

/* ---- std::unordered_map<uint64_t, uint32_t> as the builders use it: operator[] = (insert or OVERWRITE), find -------- */

typedef struct {
    uint64_t* keys;
    uint32_t* vals;
    uint8_t* used;
    size_t cap, n;
} hmap;

static void hmap_free(hmap* m) {
    free(m->keys);
    free(m->vals);
    free(m->used);
    memset(m, 0, sizeof *m);
}
static size_t hmap_slot(const hmap* m, uint64_t key) { return (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 17) & (m->cap - 1); }
static int hmap_set(hmap* m, uint64_t key, uint32_t val);
static int hmap_grow(hmap* m) {
    hmap n = {0};
    n.cap = m->cap ? m->cap * 2 : 64;
    n.keys = (uint64_t*)calloc(n.cap, 8);
    n.vals = (uint32_t*)calloc(n.cap, 4);
    n.used = (uint8_t*)calloc(n.cap, 1);
    if (!n.keys || !n.vals || !n.used) {
        hmap_free(&n);
        return 0;
    }
    for (size_t i = 0; i != m->cap; ++i)
        if (m->used[i]) hmap_set(&n, m->keys[i], m->vals[i]);
    hmap_free(m);
    *m = n;
    return 1;
}
static int hmap_set(hmap* m, uint64_t key, uint32_t val) {
    if (2 * (m->n + 1) > m->cap && !hmap_grow(m)) return 0;
    size_t j = hmap_slot(m, key);
    while (m->used[j] && m->keys[j] != key) j = (j + 1) & (m->cap - 1);
    if (!m->used[j]) {
        m->used[j] = 1;
        m->keys[j] = key;
        m->n += 1;
    }
    m->vals[j] = val; /* m_map[hash] = i: a later entry with the same hash replaces the earlier one */
    return 1;
}
static uint32_t hmap_find(const hmap* m, uint64_t key) {
    if (!m->cap) return INVALID_INDEX;
    size_t j = hmap_slot(m, key);
    while (m->used[j]) {
        if (m->keys[j] == key) return m->vals[j];
        j = (j + 1) & (m->cap - 1);
    }
    return INVALID_INDEX;
}

/* ---- the three builders: load, prepare_for_encoding, lookup ---------------------------------------------------- */

struct oracle_builder {
    int kind;
    uint32_t m_size;
    uint32_t* m_start_offsets; /* multi */
    uint32_t n_start;
    uint32_t* m_offsets; /* packed, multi */
    uint32_t n_offsets;
    uint32_t* m_table;
    size_t n_table;
    hmap maps[2 * NUM_SELECTORS]; /* single kinds: maps[0] = m_map; multi: m_maps */
};

static int rd32(const uint8_t** p, const uint8_t* end, uint32_t* v) {
    if (end - *p < 4) return 0;
    memcpy(v, *p, 4);
    *p += 4;
    return 1;
}
static uint32_t* rd32s(const uint8_t** p, const uint8_t* end, size_t n, size_t pad) {
    if ((size_t)(end - *p) < n * 4) return NULL;
    uint32_t* v = (uint32_t*)calloc(n + pad + 1, 4);
    if (!v) return NULL;
    memcpy(v, *p, n * 4);
    *p += n * 4;
    return v;
}

/* builder::size(i) / get(i): rectangular_dictionary.hpp:185-194; single_dictionary.hpp:204-216; multi_dictionary.hpp:261-279 */
static uint32_t b_size(const oracle_builder* b, uint32_t d, uint32_t i) {
    if (b->kind == ORACLE_RECT) return b->m_table[(size_t)i * (MAX_ENTRY_SIZE + 1) + MAX_ENTRY_SIZE];
    if (b->kind == ORACLE_SINGLE_PACKED) return (b->m_offsets[i] >> 24) + 1;
    return (b->m_offsets[b->m_start_offsets[d] + i] >> 24) + 1;
}
static const uint32_t* b_get(const oracle_builder* b, uint32_t d, uint32_t i) {
    if (b->kind == ORACLE_RECT) return &b->m_table[(size_t)i * (MAX_ENTRY_SIZE + 1)];
    if (b->kind == ORACLE_SINGLE_PACKED) return &b->m_table[b->m_offsets[i] & 0xFFFFFF];
    return &b->m_table[b->m_offsets[b->m_start_offsets[d] + i] & 0xFFFFFF];
}

void oracle_builder_free(oracle_builder* b) {
    if (!b) return;
    free(b->m_start_offsets);
    free(b->m_offsets);
    free(b->m_table);
    for (int i = 0; i != 2 * NUM_SELECTORS; ++i) hmap_free(&b->maps[i]);
    free(b);
}

/* builder.load(file) followed by builder.prepare_for_encoding(), as encode_dint does (vroom_env/encode.cpp:145-151)
 * and dict_freq_index::builder::build_model (dict_freq_index.hpp:52-66). */
oracle_builder* oracle_builder_load(int kind, const void* file_bytes, size_t len) {
    const uint8_t* p = (const uint8_t*)file_bytes;
    const uint8_t* end = p + len;
    oracle_builder* b = (oracle_builder*)calloc(1, sizeof *b);
    if (!b) return NULL;
    b->kind = kind;
    static uint32_t run[256]; /* std::vector<uint32_t> run(256, 0) */
    if (kind == ORACLE_RECT) {
        /* load, rectangular_dictionary.hpp:79-92: init() (:43-56) presets the reserved rows, then m_size rows are read */
        uint32_t size;
        if (!rd32(&p, end, &size) || size > NUM_ENTRIES) goto fail;
        b->n_table = (size_t)NUM_ENTRIES * (MAX_ENTRY_SIZE + 1);
        b->m_table = (uint32_t*)calloc(b->n_table, 4);
        if (!b->m_table) goto fail;
        uint32_t pos = MAX_ENTRY_SIZE + 1;
        for (int i = 0; i < EXCEPTIONS; ++i, pos += MAX_ENTRY_SIZE + 1) b->m_table[pos - 1] = 1;
        for (int i = 0, sz = 256; i < 5; ++i, pos += MAX_ENTRY_SIZE + 1, sz /= 2) b->m_table[pos - 1] = (uint32_t)sz;
        b->m_size = size;
        const size_t table_bytes = (size_t)size * (MAX_ENTRY_SIZE + 1) * 4;
        if ((size_t)(end - p) < table_bytes) goto fail;
        memcpy(b->m_table, p, table_bytes);
    } else if (kind == ORACLE_SINGLE_PACKED) {
        /* load, single_dictionary.hpp:88-107 */
        uint32_t n_off, n_tab;
        if (!rd32(&p, end, &b->m_size) || !rd32(&p, end, &n_off) || !rd32(&p, end, &n_tab)) goto fail;
        b->n_offsets = n_off;
        b->m_offsets = rd32s(&p, end, n_off, 0);
        b->m_table = rd32s(&p, end, n_tab, MAX_ENTRY_SIZE);
        b->n_table = n_tab;
        if (!b->m_offsets || !b->m_table || b->m_size > n_off) goto fail;
    } else if (kind == ORACLE_MULTI_PACKED) {
        /* load, multi_dictionary.hpp:93-121 */
        uint32_t n_start, n_off, n_tab;
        if (!rd32(&p, end, &b->m_size) || !rd32(&p, end, &n_start) || !rd32(&p, end, &n_off) || !rd32(&p, end, &n_tab)) goto fail;
        if (n_start != NUM_SELECTORS) goto fail;
        b->n_start = n_start;
        b->n_offsets = n_off;
        b->m_start_offsets = rd32s(&p, end, n_start, 0);
        b->m_offsets = rd32s(&p, end, n_off, 0);
        b->m_table = rd32s(&p, end, n_tab, MAX_ENTRY_SIZE);
        b->n_table = (size_t)n_tab + MAX_ENTRY_SIZE; /* :108 */
        if (!b->m_start_offsets || !b->m_offsets || !b->m_table) goto fail;
    } else {
        goto fail;
    }

    if (kind != ORACLE_MULTI_PACKED) {
        /* prepare_for_encoding, single_dictionary.hpp:154-165 = rectangular_dictionary.hpp:112-123: the five runs under
         * codewords 2..6, then every entry 7..size()-1 under the hash of its integers — nothing but the hash is kept, and an
         * entry whose hash equals an earlier one's (a run's included: a real entry of 16 zeros) takes the slot over. */
        uint32_t i = EXCEPTIONS;
        for (uint32_t n = 256; n >= 16; n /= 2, ++i)
            if (!hmap_set(&b->maps[0], oracle_hash_u32s(run, n), i)) goto fail;
        for (; i < b->m_size; ++i)
            if (!hmap_set(&b->maps[0], oracle_hash_u32s(b_get(b, 0, i), b_size(b, 0, i)), i)) goto fail;
    } else {
        /* prepare_for_encoding, multi_dictionary.hpp:187-217: two maps per dictionary (all entries: 16-bit codewords; the
         * entries below 256: 8-bit codewords); the scan of a dictionary stops `reserved` slots before its end (:201-206). */
        for (uint32_t dictionary_id = 0; dictionary_id != NUM_SELECTORS; ++dictionary_id) {
            uint32_t i = EXCEPTIONS;
            for (uint32_t n = 256; n >= 16; n /= 2, ++i) {
                const uint64_t hash = oracle_hash_u32s(run, n);
                if (!hmap_set(&b->maps[dictionary_id], hash, i)) goto fail;
                if (!hmap_set(&b->maps[dictionary_id + NUM_SELECTORS], hash, i)) goto fail;
            }
            const uint32_t n = (dictionary_id + 1 == NUM_SELECTORS ? b->n_offsets : b->m_start_offsets[dictionary_id + 1]) -
                               b->m_start_offsets[dictionary_id] - RESERVED;
            if (n > b->n_offsets) goto fail; /* (a dictionary of fewer than `reserved` slots: the reference's n wraps) */
            for (; i < n; ++i) {
                const uint64_t hash = oracle_hash_u32s(b_get(b, dictionary_id, i), b_size(b, dictionary_id, i));
                if (!hmap_set(&b->maps[dictionary_id], hash, i)) goto fail;
                if (i < 256 && !hmap_set(&b->maps[dictionary_id + NUM_SELECTORS], hash, i)) goto fail;
            }
        }
    }
    return b;
fail:
    oracle_builder_free(b);
    return NULL;
}

/* lookup: single_dictionary.hpp:167-175 = rectangular_dictionary.hpp:125-133 (dictionary_id and log2_num_entries ignored);
 * multi_dictionary.hpp:219-233. The n-gram's hash is all that is compared. */
uint32_t oracle_builder_lookup(const oracle_builder* b, uint32_t dictionary_id, const uint32_t* begin, uint32_t entry_size,
                               uint32_t log2_num_entries) {
    const uint64_t hash = oracle_hash_u32s(begin, entry_size);
    if (b->kind != ORACLE_MULTI_PACKED) return hmap_find(&b->maps[0], hash);
    return hmap_find(&b->maps[dictionary_id + (log2_num_entries == 8) * NUM_SELECTORS], hash);
}

/* ---- encoders ---------------------------------------------------------------------------------------------------- */

typedef struct { /* include/util.hpp:41-50 */
    uint32_t parent, codeword, cost;
} node;

/* write_index: vroom_env/dint_codecs.hpp:183-187, :325-329; include/dint/dint_codecs.hpp:134-138, :277-282, :513-518 */
static int write_index(uint32_t index, oracle_bytes* out, int b) {
    uint8_t ptr[4];
    memcpy(ptr, &index, 4);
    return bytes_append(out, ptr, (size_t)b / 8);
}

/* The optimal parse. One body for the reference's four textually equal copies, which differ in the lookup call alone:
 *   single_opt_dint::encode(builder, begin, n, out, b)                 vroom_env/dint_codecs.hpp:192-305
 *   multi_opt_dint::encode(builder, dictionary_id, begin, n, out, b)   vroom_env/dint_codecs.hpp:334-448
 *   opt_dint_single_dict_block::encode(builder, begin, n, out, b)      include/dint/dint_codecs.hpp:145-255
 *   opt_dint_multi_dict_block::encode(builder, dictionary_id, ...)     include/dint/dint_codecs.hpp:289-400
 * (line numbers in the comments below are the first copy's). */
static int opt_encode(const oracle_builder* builder, uint32_t dictionary_id, const uint32_t* begin, uint64_t n, oracle_bytes* out,
                      int b) {
    node* path = (node*)malloc((n + 2) * sizeof(node)); /* :195-196 */
    node* encoding = (node*)malloc((n + 2) * sizeof(node));
    if (!path || !encoding) {
        free(path);
        free(encoding);
        return 0;
    }
    path[0] = (node){0, 1, 0}; /* dummy node, :197 */
    for (uint32_t i = 1; i < n + 1; ++i) path[i] = (node){i - 1, 1, 3 * i}; /* :198-200 */

    for (uint32_t i = 0; i != n; ++i) { /* :202 */
        uint32_t longest_run_size = 0;
        uint32_t run_size = (uint32_t)(256 < n - i ? 256 : n - i);
        uint32_t index = EXCEPTIONS;

        for (uint32_t j = i; j != i + run_size; ++j) { /* :207-213 */
            if (begin[j] == 0)
                ++longest_run_size;
            else
                break;
        }

        if (longest_run_size >= 16) { /* :215-230 */
            uint32_t k = 256;
            while (longest_run_size < k && k > 16) {
                k /= 2;
                ++index;
            }
            while (k >= 16) {
                uint32_t c = path[i].cost + 1;
                if (path[i + k].cost > c) path[i + k] = (node){i, index, c};
                k /= 2;
                ++index;
            }
        }

        for (uint32_t s = 0; s < NUM_TARGET_SIZES; ++s) { /* :232-257 */
            uint32_t sub_block_size = target_sizes[s];
            uint32_t len = (uint32_t)(sub_block_size < n - i ? sub_block_size : n - i);
            index = oracle_builder_lookup(builder, dictionary_id, begin + i, len, (uint32_t)b);
            if (index != INVALID_INDEX) {
                uint32_t c = path[i].cost + 1;
                if (path[i + len].cost > c) path[i + len] = (node){i, index, c};
            } else {
                if (sub_block_size == 1) { /* exceptions */
                    uint32_t exception = begin[i];
                    uint32_t c = path[i].cost + 2; /* small exception cost */
                    index = 0;
                    if (exception > 65536 - 1) {
                        c += 1; /* large exception cost */
                        index = 1;
                    }
                    if (path[i + 1].cost > c) path[i + 1] = (node){i, index, c};
                }
            }
        }
    }

    size_t n_enc = 0; /* std::vector<node> encoding, :260-269 */
    {
        uint32_t i = (uint32_t)n;
        while (i != 0) {
            uint32_t parent = path[i].parent;
            encoding[n_enc++] = path[i];
            i = parent;
        }
    }
    for (size_t l = 0, r = n_enc; l + 1 < r; ++l, --r) { /* std::reverse */
        node t = encoding[l];
        encoding[l] = encoding[r - 1];
        encoding[r - 1] = t;
    }
    encoding[n_enc++] = (node){(uint32_t)n, 1, (uint32_t)-1}; /* final dummy node */

    int ok = 1;
    for (uint32_t i = 0, pos = 0; ok && i < n_enc - 1; ++i) { /* :271-302 */
        uint32_t index = encoding[i].codeword;
        uint32_t len = encoding[i + 1].parent - encoding[i].parent;
        if (index > 1) {
            ok = write_index(index, out, b);
        } else {
            uint32_t exception = begin[pos];
            uint8_t ptr[4];
            memcpy(ptr, &exception, 4);
            if (index == 0) {
                ok = bytes_push(out, 0);
                if (ok && b == 16) ok = bytes_push(out, 0);
                if (ok) ok = bytes_append(out, ptr, 2);
            } else {
                ok = bytes_push(out, 1);
                if (ok && b == 16) ok = bytes_push(out, 0);
                if (ok) ok = bytes_append(out, ptr, 4);
            }
        }
        pos += len;
    }
    free(path);
    free(encoding);
    return ok;
}

/* single_greedy_dint::encode, vroom_env/dint_codecs.hpp:110-171 = greedy_dint_single_dict_block::encode's loop,
 * include/dint/dint_codecs.hpp:65-123 */
static int greedy_encode(const oracle_builder* builder, const uint32_t* in, uint32_t n, oracle_bytes* out) {
    const uint32_t* begin = in;
    const uint32_t* end = begin + n;
    int ok = 1;
    while (ok && begin < end) {
        uint32_t longest_run_size = 0;
        uint32_t run_size = (uint32_t)(256 < end - begin ? 256 : end - begin);
        uint32_t index = EXCEPTIONS;
        for (const uint32_t* ptr = begin; ptr != begin + run_size; ++ptr) {
            if (*ptr == 0)
                ++longest_run_size;
            else
                break;
        }
        if (longest_run_size >= 16) {
            uint32_t k = 256;
            while (longest_run_size < k && k > 16) {
                ++index;
                k /= 2;
            }
            ok = write_index(index, out, 16);
            begin += k;
        } else {
            for (uint32_t s = 0; s < NUM_TARGET_SIZES; ++s) {
                uint32_t sub_block_size = target_sizes[s];
                uint32_t len = (uint32_t)(sub_block_size < end - begin ? sub_block_size : end - begin);
                index = oracle_builder_lookup(builder, 0, begin, len, 16);
                if (index != INVALID_INDEX) {
                    ok = write_index(index, out, 16);
                    begin += len;
                    break;
                }
            }
            if (index == INVALID_INDEX) {
                uint32_t exception = *begin;
                uint8_t ptr[4];
                memcpy(ptr, &exception, 4);
                if (exception < 65536) {
                    ok = bytes_push(out, 0) && bytes_push(out, 0) && bytes_append(out, ptr, 2);
                } else {
                    ok = bytes_push(out, 1) && bytes_push(out, 0) && bytes_append(out, ptr, 4);
                }
                begin += 1;
            }
        }
    }
    return ok;
}

/* One block of a multi-dictionary stream: "option 1: choose the best dictionary (exhaustive search)" —
 * vroom_env/dint_codecs.hpp:465-496 = include/dint/dint_codecs.hpp:411-432. Twelve encodings; per dictionary the 8-bit one
 * wins a tie (<=), across dictionaries the first strictly smaller wins (<). */
static int multi_block_encode(const oracle_builder* builder, const uint32_t* begin, uint64_t size, oracle_bytes* out) {
    oracle_bytes encoded[2 * NUM_SELECTORS];
    memset(encoded, 0, sizeof encoded);
    size_t best_size = (size_t)-1;
    uint32_t selector_code = 0;
    int ok = 1;
    for (uint32_t s = 0; ok && s != NUM_SELECTORS; ++s) {
        ok = opt_encode(builder, s, begin, size, &encoded[s], 16) && opt_encode(builder, s, begin, size, &encoded[s + NUM_SELECTORS], 8);
        size_t smallest_size = encoded[s].size;
        uint32_t sc = s;
        if (encoded[s + NUM_SELECTORS].size <= smallest_size) {
            smallest_size = encoded[s + NUM_SELECTORS].size;
            sc += NUM_SELECTORS;
        }
        if (smallest_size < best_size) {
            best_size = smallest_size;
            selector_code = sc;
        }
    }
    if (ok) ok = bytes_push(out, (uint8_t)selector_code) && bytes_append(out, encoded[selector_code].data, encoded[selector_code].size);
    for (int i = 0; i != 2 * NUM_SELECTORS; ++i) oracle_bytes_free(&encoded[i]);
    return ok;
}

/* Encoder::encode(builder, in, universe, n, out) of the vroom environment: single_opt_dint (vroom_env/dint_codecs.hpp:307-312),
 * single_greedy_dint (:110-171), multi_opt_dint (:450-518: blocks of 256 and a tail of n % 256). The dictionary kind of the
 * builder picks single or multi, as encode.cpp:312-320 pairs them. */
int oracle_encode_list(const oracle_builder* builder, int greedy, const uint32_t* in, uint32_t n, oracle_bytes* out) {
    if (builder->kind != ORACLE_MULTI_PACKED) return greedy ? greedy_encode(builder, in, n, out) : opt_encode(builder, 0, in, n, out, 16);
    const uint32_t* begin = in;
    uint64_t num_blocks = ((uint64_t)n + BLOCK_SIZE - 1) / BLOCK_SIZE; /* ceil_div */
    uint64_t tail = n - (n / BLOCK_SIZE * BLOCK_SIZE);
    for (uint64_t b = 0; b != num_blocks; ++b) {
        uint64_t size = BLOCK_SIZE;
        if (b == num_blocks - 1 && tail != 0) size = tail;
        if (!multi_block_encode(builder, begin, size, out)) return 0;
        begin += size;
    }
    return 1;
}

/* ---- interpolative_block::encode and its bit_writer ---------------------------------------------------------------- */

typedef struct { /* include/ds2i/interpolative_coding.hpp:10-77 */
    uint32_t* m_buf;
    size_t n_words, cap_words;
    size_t m_size;
    int failed;
} bit_writer;

static void bw_push(bit_writer* w, uint32_t bits) {
    if (w->n_words == w->cap_words) {
        size_t ncap = w->cap_words ? w->cap_words * 2 : 64;
        uint32_t* nb = (uint32_t*)realloc(w->m_buf, ncap * 4);
        if (!nb) {
            w->failed = 1;
            return;
        }
        w->m_buf = nb;
        w->cap_words = ncap;
    }
    w->m_buf[w->n_words++] = bits;
}

static void bw_write(bit_writer* w, uint32_t bits, uint32_t len) { /* :21-35 */
    if (!len || w->failed) return;
    uint32_t pos_in_word = (uint32_t)(w->m_size % 32);
    w->m_size += len;
    if (pos_in_word == 0) {
        bw_push(w, bits);
    } else {
        w->m_buf[w->n_words - 1] |= bits << pos_in_word; /* *m_cur_word: the back of the buffer */
        if (len > 32 - pos_in_word) bw_push(w, bits >> (32 - pos_in_word));
    }
}

static uint32_t msb64(uint64_t x) { /* succinct::broadword::msb: the position of the highest set bit (x > 0) */
    uint32_t r = 0;
    while (x >>= 1) ++r;
    return r;
}

static void bw_write_int(bit_writer* w, uint32_t val, uint32_t u) { /* :41-56 */
    uint32_t b = msb64(u);
    uint64_t m = ((uint64_t)1 << (b + 1)) - u;
    if (val < m) {
        bw_write(w, val, b);
    } else {
        val += (uint32_t)m;
        /* since we use little-endian we must split the writes */
        bw_write(w, val >> 1, b);
        bw_write(w, val & 1, 1);
    }
}

static void bw_write_interpolative(bit_writer* w, const uint32_t* in, size_t n, uint32_t low, uint32_t high) { /* :58-71 */
    if (!n) return;
    size_t h = n / 2;
    uint32_t val = in[h];
    bw_write_int(w, val - low, high - low + 1);
    bw_write_interpolative(w, in, h, low, val);
    bw_write_interpolative(w, in + h + 1, n - h - 1, val, high);
}

/* TightVariableByte::encode_single, include/ds2i/block_codecs.hpp:35-85 (twin vroom_env/codecs.hpp:37-92): the five cases
 * written out — 7 bits a byte, least significant group first, the LAST byte carries bit 7 */
static int vbyte_encode_single(uint32_t val, oracle_bytes* out) {
    uint8_t buf[5];
    size_t k = 0;
    if (val < (1U << 7)) {
        buf[k++] = (uint8_t)(val | (1U << 7));
    } else if (val < (1U << 14)) {
        buf[k++] = (uint8_t)((val >> 0) & 127);
        buf[k++] = (uint8_t)(val >> 7) | (1U << 7);
    } else if (val < (1U << 21)) {
        buf[k++] = (uint8_t)((val >> 0) & 127);
        buf[k++] = (uint8_t)((val >> 7) & 127);
        buf[k++] = (uint8_t)(val >> 14) | (1U << 7);
    } else if (val < (1U << 28)) {
        buf[k++] = (uint8_t)((val >> 0) & 127);
        buf[k++] = (uint8_t)((val >> 7) & 127);
        buf[k++] = (uint8_t)((val >> 14) & 127);
        buf[k++] = (uint8_t)(val >> 21) | (1U << 7);
    } else {
        buf[k++] = (uint8_t)((val >> 0) & 127);
        buf[k++] = (uint8_t)((val >> 7) & 127);
        buf[k++] = (uint8_t)((val >> 14) & 127);
        buf[k++] = (uint8_t)((val >> 21) & 127);
        buf[k++] = (uint8_t)(val >> 28) | (1U << 7);
    }
    return bytes_append(out, buf, k);
}

int oracle_vbyte_encode(uint32_t val, oracle_bytes* out) { return vbyte_encode_single(val, out); }

/* interpolative_block::encode, include/ds2i/block_codecs.hpp:104-128 */
int oracle_interpolative_encode(const uint32_t* in, uint32_t sum_of_values, size_t n, oracle_bytes* out) {
    if (n == 0 || n > BLOCK_SIZE) return 0;
    uint32_t inbuf[BLOCK_SIZE];
    inbuf[0] = *in;
    for (size_t i = 1; i < n; ++i) inbuf[i] = inbuf[i - 1] + in[i];
    if (sum_of_values == (uint32_t)-1) {
        sum_of_values = inbuf[n - 1];
        if (!vbyte_encode_single(sum_of_values, out)) return 0;
    }
    bit_writer bw = {0};
    bw_write_interpolative(&bw, inbuf, n - 1, 0, sum_of_values);
    int ok = !bw.failed && bytes_append(out, bw.m_buf, (bw.m_size + 7) / 8); /* ceil_div(bw.size(), 8) bytes of the words */
    free(bw.m_buf);
    return ok;
}

/* Coder::encode(builder, in, sum_of_values, n, out) of the index: opt_dint_single_dict_block (include/dint/dint_codecs.hpp:
 * 257-267), greedy_dint_single_dict_block (:56-63 + loop), opt_dint_multi_dict_block (:402-432). A block of fewer than 256
 * integers is binary-interpolative coded whatever the coder. */
int oracle_block_encode(const oracle_builder* builder, int greedy, const uint32_t* in, uint32_t sum_of_values, uint32_t n,
                        oracle_bytes* out) {
    if (n < BLOCK_SIZE) return oracle_interpolative_encode(in, sum_of_values, n, out);
    if (builder->kind == ORACLE_MULTI_PACKED) return multi_block_encode(builder, in, n, out);
    return greedy ? greedy_encode(builder, in, n, out) : opt_encode(builder, 0, in, n, out, 16 /* constants::log2_num_entries */);
}

/* dict_posting_list::write, include/dint/dict_posting_list.hpp:10-56 */
int oracle_posting_list_write(const oracle_builder* docs_dict_builder, const oracle_builder* freqs_dict_builder, int greedy, uint32_t n,
                              const uint32_t* docs_begin, const uint32_t* freqs_begin, oracle_bytes* out) {
    if (n == 0) return 0;
    if (!vbyte_encode_single(n, out)) return 0;
    uint64_t block_size = BLOCK_SIZE;
    uint64_t blocks = ((uint64_t)n + block_size - 1) / block_size;
    size_t begin_block_maxs = out->size;
    size_t begin_block_endpoints = begin_block_maxs + 4 * blocks;
    size_t begin_blocks = begin_block_endpoints + 4 * (blocks - 1);
    if (!bytes_resize(out, begin_blocks)) return 0;

    const uint32_t* docs_it = docs_begin;
    const uint32_t* freqs_it = freqs_begin;
    uint32_t docs_buf[BLOCK_SIZE], freqs_buf[BLOCK_SIZE];
    uint32_t last_doc = (uint32_t)-1;
    uint32_t block_base = 0;
    for (size_t b = 0; b < blocks; ++b) {
        uint32_t cur_block_size = ((b + 1) * block_size <= n) ? (uint32_t)block_size : (uint32_t)(n % block_size);
        for (size_t i = 0; i < cur_block_size; ++i) {
            uint32_t doc = *docs_it++;
            docs_buf[i] = doc - last_doc - 1;
            last_doc = doc;
            freqs_buf[i] = *freqs_it++ - 1;
        }
        memcpy(&out->data[begin_block_maxs + 4 * b], &last_doc, 4);
        if (!oracle_block_encode(docs_dict_builder, greedy, docs_buf, last_doc - block_base - (cur_block_size - 1), cur_block_size, out))
            return 0;
        if (!oracle_block_encode(freqs_dict_builder, greedy, freqs_buf, (uint32_t)-1, cur_block_size, out)) return 0;
        if (b != blocks - 1) {
            uint32_t endpoint = (uint32_t)(out->size - begin_blocks);
            memcpy(&out->data[begin_block_endpoints + 4 * b], &endpoint, 4);
        }
        block_base = last_doc + 1;
    }
    return 1;
}

/* ---- the vroom `encode` program over a binary_collection ------------------------------------------------------------- */

/* encode_dint, vroom_env/encode.cpp:133-191, with the jobs it queues (dint_sequence_adder::prepare / commit,
 * vroom_env/jobs.hpp:74-95) run in order, over `data` = the collection file's u32 words read the way
 * binary_collection::iterator::read does (include/ds2i/binary_collection.hpp:131-146: records `len, v[len]`, empty records
 * skipped, a truncated last record cut at the end of the file). docs != 0: the file is a .docs file — its first record
 * (`1, num_docs`) is skipped (encode.cpp:160-164) and the values are turned into d-gaps minus one, the first against -1;
 * docs == 0: a .freqs file, every value minus one. */
int oracle_encode_collection(const oracle_builder* builder, int greedy, const uint32_t* data, size_t data_size, int docs, oracle_bytes* output,
                             uint64_t* num_processed_lists, uint64_t* num_total_ints) {
    uint64_t lists = 0, ints = 0;
    size_t m_pos = 0;
    int first = 1;
    uint32_t* buf = NULL;
    size_t buf_cap = 0;
    int ok = 1;
    while (ok && m_pos != data_size) { /* it != input.end() */
        size_t n = 0;
        size_t pos = m_pos;
        while (pos < data_size && !(n = data[pos++])) { /* skip empty seqs */
        }
        if (n == 0) break; /* (nothing but empty records left: the reference reads past the end here) */
        if (n > data_size - pos) n = data_size - pos; /* file might be truncated */
        const uint32_t* begin = &data[pos];
        m_pos = pos + n; /* m_next_pos */
        if (docs && first) { /* ++it: the singleton sequence holding the number of documents */
            first = 0;
            continue;
        }
        first = 0;
        /* prepare(), jobs.hpp:74-87 */
        if (n > buf_cap) {
            free(buf);
            buf = (uint32_t*)malloc(n * 4);
            buf_cap = n;
            if (!buf) {
                ok = 0;
                break;
            }
        }
        uint32_t universe = 0;
        uint32_t prev = docs ? (uint32_t)-1 : 0;
        for (uint64_t i = 0; i != n; ++i, ++begin) {
            buf[i] = *begin - prev - 1;
            if (docs) prev = *begin;
            universe += buf[i];
        }
        oracle_bytes tmp = {0};
        ok = oracle_encode_list(builder, greedy, buf, (uint32_t)n, &tmp);
        /* commit(), jobs.hpp:89-95: header::write(n, universe) (vroom_env/codecs.hpp:110-115), then the payload */
        if (ok) ok = vbyte_encode_single((uint32_t)n, output) && vbyte_encode_single(universe, output) && bytes_append(output, tmp.data, tmp.size);
        oracle_bytes_free(&tmp);
        ++lists;
        ints += n;
    }
    free(buf);
    if (num_processed_lists) *num_processed_lists = lists;
    if (num_total_ints) *num_total_ints = ints;
    return ok;
}

/* ---- builder::init / append / build / write: packing a selection into a dictionary file ------------------------------- */

typedef struct { /* target_t, dictionary_building_utils.hpp:31-62 */
    const uint32_t* entry;
    uint32_t size;
    int valid;
} target_t;

static int target_less(const void* a, const void* b) { /* operator<, :35-43: by size, then lexicographically */
    const target_t* l = (const target_t*)a;
    const target_t* r = (const target_t*)b;
    if (l->size != r->size) return l->size < r->size ? -1 : 1;
    for (uint32_t i = 0; i != l->size; ++i)
        if (l->entry[i] != r->entry[i]) return l->entry[i] < r->entry[i] ? -1 : 1;
    return 0;
}

/* pack_policy::compact, dictionary_building_utils.hpp:241-292: all targets of all dictionaries sorted, duplicates removed,
 * every target that is a proper prefix of a longer VALID one dropped (the O(n^2) double loop as written), the survivors in
 * sorted order. Returns their number; `all` is rewritten in place. */
static size_t pack_compact(target_t* all, size_t n) {
    qsort(all, n, sizeof(target_t), target_less); /* std::sort(all_targets) — operator< is total up to equality */
    size_t u = 0;                                 /* std::unique + erase */
    for (size_t i = 0; i != n; ++i)
        if (u == 0 || target_less(&all[u - 1], &all[i]) != 0) all[u++] = all[i];
    n = u;
    for (size_t i = 0; i < n; i++) { /* "find prefix overlaps", :259-271 */
        target_t* cur = &all[i];
        for (size_t j = 0; j < n; j++) {
            target_t* other = &all[j];
            if (i != j && other->valid && cur->size < other->size) {
                if (memcmp(cur->entry, other->entry, (size_t)cur->size * 4) == 0) { /* prefix_overlap, :10-14 */
                    cur->valid = 0;
                    break;
                }
            }
        }
    }
    size_t k = 0; /* "remove prefix overlaps", :275-284 */
    for (size_t i = 0; i != n; ++i)
        if (all[i].valid) all[k++] = all[i];
    return k;
}

/* std::search(m_table.begin(), m_table.end(), entry.begin(), entry.end()) - m_table.begin() */
static uint32_t table_search(const uint32_t* table, size_t n_table, const uint32_t* entry, uint32_t size) {
    for (size_t t = 0; t + size <= n_table; ++t)
        if (memcmp(table + t, entry, (size_t)size * 4) == 0) return (uint32_t)t;
    return INVALID_INDEX;
}

static int put32(oracle_bytes* out, uint32_t v) { return bytes_append(out, &v, 4); }

/* What decreasing_static_frequencies::build leaves on disk (dictionary_builders.hpp:55-75 + try_store_to_file): the
 * builder's init(), one append() per selected n-gram — `words` holds them back to back, entry k has lens[k] integers and
 * goes to dictionary ctx[k] (entries of one dictionary in dictionary order; ctx may be NULL for the single kinds) — then
 * build() and write().
 *   rectangular: rectangular_dictionary.hpp:43-56 (init), :99-108 (append: refused once m_size == 65536), :72-77 (write)
 *   single packed: single_dictionary.hpp:40-56, :113-123, :125-152 (compact, table, one std::search per entry), :72-86
 *   multi packed: multi_dictionary.hpp:43-56, :127-137 (full() is GLOBAL: 6 x 65536), :139-185, :70-91 */
int oracle_pack_dictionary(int kind, const uint32_t* words, const uint32_t* lens, const uint32_t* ctx, size_t n_entries, oracle_bytes* file) {
    if (kind == ORACLE_RECT) {
        const uint32_t row = MAX_ENTRY_SIZE + 1;
        uint32_t* m_table = (uint32_t*)calloc((size_t)NUM_ENTRIES * row, 4);
        if (!m_table) return 0;
        uint32_t m_pos = RESERVED * row, m_size = RESERVED, pos = row;
        for (int i = 0; i < EXCEPTIONS; ++i, pos += row) m_table[pos - 1] = 1;
        for (int i = 0, size = 256; i < 5; ++i, pos += row, size /= 2) m_table[pos - 1] = (uint32_t)size;
        const uint32_t* e = words;
        for (size_t k = 0; k != n_entries; e += lens[k], ++k) {
            if (m_size == NUM_ENTRIES) continue; /* full(): append returns false, the caller goes on */
            if (lens[k] == 0 || lens[k] > MAX_ENTRY_SIZE) {
                free(m_table);
                return 0;
            }
            memcpy(&m_table[m_pos], e, (size_t)lens[k] * 4);
            m_pos += row;
            m_table[m_pos - 1] = lens[k];
            ++m_size;
        }
        int ok = put32(file, m_size) && bytes_append(file, m_table, (size_t)m_size * row * 4);
        free(m_table);
        return ok;
    }
    if (kind != ORACLE_SINGLE_PACKED && kind != ORACLE_MULTI_PACKED) return 0;
    const uint32_t num_dictionaries = kind == ORACLE_MULTI_PACKED ? NUM_SELECTORS : 1;
    const uint64_t capacity = (uint64_t)num_dictionaries * NUM_ENTRIES; /* full(): m_size == [num_dictionaries *] num_entries */
    target_t* m_targets = (target_t*)malloc((n_entries ? n_entries : 1) * sizeof(target_t)); /* in append order */
    uint32_t* owner = (uint32_t*)malloc((n_entries ? n_entries : 1) * 4);
    target_t* all = (target_t*)malloc((n_entries ? n_entries : 1) * sizeof(target_t));
    uint32_t* m_table = NULL;
    uint32_t* m_offsets = NULL;
    int ok = m_targets && owner && all;
    uint64_t m_size = RESERVED;
    size_t n_t = 0, table_words = MAX_ENTRY_SIZE;
    if (ok) {
        const uint32_t* e = words;
        for (size_t k = 0; k != n_entries; e += lens[k], ++k) {
            if (m_size == capacity) continue;
            const uint32_t d = ctx ? ctx[k] : 0;
            if (lens[k] == 0 || lens[k] > MAX_ENTRY_SIZE || d >= num_dictionaries) {
                ok = 0;
                break;
            }
            m_targets[n_t] = (target_t){e, lens[k], 1};
            owner[n_t] = d;
            ++n_t;
            ++m_size;
        }
    }
    size_t n_all = 0;
    if (ok) {
        /* compact(): the dictionaries' targets one dictionary after the other (std::copy per t, :246-248) */
        for (uint32_t d = 0; d != num_dictionaries; ++d)
            for (size_t k = 0; k != n_t; ++k)
                if (owner[k] == d) all[n_all++] = m_targets[k];
        n_all = pack_compact(all, n_all);
        for (size_t k = 0; k != n_all; ++k) table_words += all[k].size;
        m_table = (uint32_t*)calloc(table_words, 4); /* init(): max_entry_size zeros first */
        m_offsets = (uint32_t*)malloc((n_t + (size_t)num_dictionaries * RESERVED + 1) * 4);
        ok = m_table && m_offsets;
    }
    size_t n_off = 0;
    uint32_t m_start_offsets[NUM_SELECTORS];
    if (ok) {
        size_t w = MAX_ENTRY_SIZE;
        for (size_t k = 0; k != n_all; ++k) { /* "creating table..." */
            memcpy(m_table + w, all[k].entry, (size_t)all[k].size * 4);
            w += all[k].size;
        }
        for (uint32_t d = 0; ok && d != num_dictionaries; ++d) { /* "creating offsets..." */
            m_start_offsets[d] = (uint32_t)n_off;
            for (uint32_t i = 0; i != EXCEPTIONS; ++i) m_offsets[n_off++] = 0;
            for (uint32_t i = 0, size = 256; i != 5; ++i, size /= 2) m_offsets[n_off++] = (size - 1) << 24; /* offset is 0 */
            for (size_t k = 0; k != n_t; ++k) {
                if (owner[k] != d) continue;
                const uint32_t offset = table_search(m_table, table_words, m_targets[k].entry, m_targets[k].size);
                if (offset == INVALID_INDEX) {
                    ok = 0;
                    break;
                }
                m_offsets[n_off++] = ((m_targets[k].size - 1) << 24) | offset;
            }
        }
    }
    if (ok) {
        ok = put32(file, (uint32_t)m_size);
        if (kind == ORACLE_MULTI_PACKED) ok = ok && put32(file, num_dictionaries);
        ok = ok && put32(file, (uint32_t)n_off) && put32(file, (uint32_t)table_words);
        if (kind == ORACLE_MULTI_PACKED) ok = ok && bytes_append(file, m_start_offsets, 4 * (size_t)num_dictionaries);
        ok = ok && bytes_append(file, m_offsets, n_off * 4) && bytes_append(file, m_table, table_words * 4);
    }
    free(m_targets);
    free(owner);
    free(all);
    free(m_table);
    free(m_offsets);
    return ok;
}
