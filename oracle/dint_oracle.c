/*
 * dint_oracle.c — see dint_oracle.h (TEST INFRASTRUCTURE; PARITY UNPINNED).
 * Plain C11. Citations are relative to /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "dint_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ENTRIES 65536u     /* include/dint/dint_configuration.hpp:27 */
#define MAX_ENTRY 16u      /* :25 */
#define NUM_SELECTORS 6u   /* :20 */
#define EXCEPTIONS_ 2u     /* :6  */
#define BLOCK 256u         /* include/util.hpp:35 */
#define MAX_LIST 50000000u /* include/util.hpp:34 */

struct oracle_dict {
    int kind;
    /* rect: table = 65536 rows x 17 words (rectangular_dictionary.hpp:43-56) */
    /* packed: offsets[] + table[] (single_dictionary.hpp:230-238) */
    /* multi: start_offsets[6] + offsets[] + table[] (multi_dictionary.hpp:293-304) */
    uint32_t size;
    uint32_t* start_offsets;
    uint32_t n_start;
    uint32_t* offsets;
    uint32_t n_offsets;
    uint32_t* table;
    size_t n_table; /* including the padding added on load */
};

static int rd_u32(const uint8_t** p, const uint8_t* end, uint32_t* v) {
    if (end - *p < 4) return 0;
    memcpy(v, *p, 4);
    *p += 4;
    return 1;
}

static uint32_t* rd_u32s(const uint8_t** p, const uint8_t* end, size_t n, size_t pad) {
    if ((size_t)(end - *p) < n * 4) return NULL;
    uint32_t* v = (uint32_t*)calloc(n + pad + 1, 4);
    if (!v) return NULL;
    memcpy(v, *p, n * 4);
    *p += n * 4;
    return v;
}

oracle_dict* oracle_dict_load(int kind, const void* file_bytes, size_t len) {
    const uint8_t* p = (const uint8_t*)file_bytes;
    const uint8_t* end = p + len;
    oracle_dict* d = (oracle_dict*)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->kind = kind;
    if (kind == ORACLE_RECT) {
        /* builder::load, rectangular_dictionary.hpp:79-92: init() presets the
         * reserved rows (sizes 1,1,256,128,64,32,16), then m_size rows are read
         * over the start of the table. */
        uint32_t size;
        if (!rd_u32(&p, end, &size) || size > ENTRIES) goto fail;
        d->size = size;
        d->n_table = (size_t)ENTRIES * (MAX_ENTRY + 1);
        d->table = (uint32_t*)calloc(d->n_table, 4);
        if (!d->table) goto fail;
        for (uint32_t i = 0; i < EXCEPTIONS_; ++i) d->table[i * 17 + 16] = 1;
        for (uint32_t i = 0, s = 256; i < 5; ++i, s /= 2) d->table[(EXCEPTIONS_ + i) * 17 + 16] = s;
        size_t bytes = (size_t)size * 17 * 4;
        if ((size_t)(end - p) < bytes) goto fail;
        memcpy(d->table, p, bytes);
    } else if (kind == ORACLE_SINGLE_PACKED) {
        /* builder::load, single_dictionary.hpp:88-107. The reference does not
         * pad the table and copy() may read 15 words past it; here the
         * allocation is padded so the same 64-byte memcpy stays in bounds. */
        uint32_t n_off, n_tab;
        if (!rd_u32(&p, end, &d->size) || !rd_u32(&p, end, &n_off) || !rd_u32(&p, end, &n_tab)) goto fail;
        d->n_offsets = n_off;
        d->offsets = rd_u32s(&p, end, n_off, 0);
        d->table = rd_u32s(&p, end, n_tab, MAX_ENTRY);
        d->n_table = (size_t)n_tab + MAX_ENTRY;
        if (!d->offsets || !d->table) goto fail;
    } else if (kind == ORACLE_MULTI_PACKED) {
        /* builder::load, multi_dictionary.hpp:93-121 (table padded by 16, :108) */
        uint32_t n_start, n_off, n_tab;
        if (!rd_u32(&p, end, &d->size) || !rd_u32(&p, end, &n_start) || !rd_u32(&p, end, &n_off) ||
            !rd_u32(&p, end, &n_tab))
            goto fail;
        if (n_start != NUM_SELECTORS) goto fail;
        d->n_start = n_start;
        d->n_offsets = n_off;
        d->start_offsets = rd_u32s(&p, end, n_start, 0);
        d->offsets = rd_u32s(&p, end, n_off, 0);
        d->table = rd_u32s(&p, end, n_tab, MAX_ENTRY);
        d->n_table = (size_t)n_tab + MAX_ENTRY;
        if (!d->start_offsets || !d->offsets || !d->table) goto fail;
    } else {
        goto fail;
    }
    return d;
fail:
    oracle_dict_free(d);
    return NULL;
}

void oracle_dict_free(oracle_dict* d) {
    if (!d) return;
    free(d->start_offsets);
    free(d->offsets);
    free(d->table);
    free(d);
}

/* rectangular_dictionary::copy, rectangular_dictionary.hpp:206-213 */
static inline uint32_t copy_rect(const oracle_dict* d, uint32_t i, uint32_t* out) {
    const uint32_t* ptr = &d->table[(size_t)i * (MAX_ENTRY + 1)];
    memcpy(out, ptr, MAX_ENTRY * sizeof(uint32_t));
    return ptr[MAX_ENTRY];
}

/* single_dictionary::copy, single_dictionary.hpp:230-238 */
static inline uint32_t copy_packed(const oracle_dict* d, uint32_t i, uint32_t* out) {
    uint32_t size_and_offset = d->offsets[i];
    uint32_t offset = size_and_offset & 0xFFFFFF;
    uint32_t size = (size_and_offset >> 24) + 1;
    memcpy(out, &d->table[offset], MAX_ENTRY * sizeof(uint32_t));
    return size;
}

/* multi_dictionary::copy, multi_dictionary.hpp:293-304 */
static inline uint32_t copy_multi(const oracle_dict* d, uint32_t dict_id, uint32_t i, uint32_t* out) {
    uint32_t size_and_offset = d->offsets[d->start_offsets[dict_id] + i];
    uint32_t offset = size_and_offset & 0xFFFFFF;
    uint32_t size = (size_and_offset >> 24) + 1;
    memcpy(out, &d->table[offset], MAX_ENTRY * sizeof(uint32_t));
    return size;
}

uint32_t oracle_dict_copy(const oracle_dict* d, uint32_t dict_id, uint32_t i, uint32_t* out) {
    switch (d->kind) {
        case ORACLE_RECT: return copy_rect(d, i, out);
        case ORACLE_SINGLE_PACKED: return copy_packed(d, i, out);
        default: return copy_multi(d, dict_id, i, out);
    }
}

/* TightVariableByte::decode, vroom_env/codecs.hpp:93-107 (n = 1) */
const uint8_t* oracle_vbyte_read(const uint8_t* in, uint32_t* val) {
    uint32_t v = 0;
    for (unsigned shift = 0;; shift += 7) {
        uint8_t c = *in++;
        v += (uint32_t)(c & 127) << shift;
        if (c & 128) {
            *val = v;
            return in;
        }
    }
}

/* header::read, vroom_env/codecs.hpp:117-123 */
const uint8_t* oracle_header_read(const uint8_t* in, uint32_t* n, uint32_t* universe) {
    in = oracle_vbyte_read(in, n);
    return oracle_vbyte_read(in, universe);
}

static inline uint16_t ld16(const uint8_t* p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
static inline uint32_t ld32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

/* single_dint::decode, vroom_env/dint_codecs.hpp:37-107. One body per
 * dictionary type so the copy() is inlined as the template instantiation is. */
#define SINGLE_BODY(COPY)                                                        \
    const uint8_t* ptr = in;                                                     \
    for (size_t i = 0; i != n; ptr += 2) {                                       \
        uint32_t index = ld16(ptr);                                              \
        uint32_t decoded_ints = 1;                                               \
        if (__builtin_expect(index > EXCEPTIONS_ - 1, 1)) {                      \
            decoded_ints = COPY(d, index, out);                                  \
        } else if (index == 1) { /* 4-byte exception: 3 slots in all */          \
            ptr += 2;                                                            \
            *out = ld32(ptr);                                                    \
            ptr += 2;                                                            \
        } else { /* 2-byte exception: 2 slots in all */                          \
            ptr += 2;                                                            \
            *out = ld16(ptr);                                                    \
        }                                                                        \
        out += decoded_ints;                                                     \
        i += decoded_ints;                                                       \
    }                                                                            \
    return ptr;

static const uint8_t* decode_single_rect(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n) {
    SINGLE_BODY(copy_rect)
}
static const uint8_t* decode_single_packed(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n) {
    SINGLE_BODY(copy_packed)
}

const uint8_t* oracle_decode_single(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n) {
    return d->kind == ORACLE_RECT ? decode_single_rect(d, in, out, n) : decode_single_packed(d, in, out, n);
}

/* multi_opt_dint::decode, vroom_env/dint_codecs.hpp:521-619 */
const uint8_t* oracle_decode_multi(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n) {
    size_t num_blocks = (n + BLOCK - 1) / BLOCK;
    size_t tail = n - (n / BLOCK * BLOCK);
    for (size_t b = 0; b != num_blocks; ++b) {
        size_t size = BLOCK;
        if (b == num_blocks - 1 && tail != 0) size = tail;
        uint8_t selector_code = *in;
        if (selector_code < NUM_SELECTORS) {
            const uint8_t* ptr = in + 1;
            for (size_t i = 0; i != size; ptr += 2) {
                uint32_t index = ld16(ptr);
                uint32_t decoded_ints = 1;
                if (__builtin_expect(index > EXCEPTIONS_ - 1, 1)) {
                    decoded_ints = copy_multi(d, selector_code, index, out);
                } else if (index == 1) {
                    ptr += 2;
                    *out = ld32(ptr);
                    ptr += 2;
                } else {
                    ptr += 2;
                    *out = ld16(ptr);
                }
                out += decoded_ints;
                i += decoded_ints;
            }
            in = ptr;
        } else {
            selector_code -= NUM_SELECTORS;
            const uint8_t* ptr = in + 1;
            for (size_t i = 0; i != size; ++ptr) {
                uint32_t index = *ptr;
                uint32_t decoded_ints = 1;
                if (__builtin_expect(index > EXCEPTIONS_ - 1, 1)) {
                    decoded_ints = copy_multi(d, selector_code, index, out);
                } else if (index == 1) { /* 1 + 4 bytes */
                    ++ptr;
                    *out = ld32(ptr);
                    ptr += 3;
                } else { /* 1 + 2 bytes */
                    ++ptr;
                    *out = ld16(ptr);
                    ptr += 1;
                }
                out += decoded_ints;
                i += decoded_ints;
            }
            in = ptr;
        }
    }
    return in;
}

const uint8_t* oracle_decode_list(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n) {
    return d->kind == ORACLE_MULTI_PACKED ? oracle_decode_multi(d, in, out, n) : oracle_decode_single(d, in, out, n);
}

/* vroom_env/decode.cpp:139-150 framing + check_encoded_data.cpp:104-109 zeroing */
uint64_t oracle_decode_stream(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, uint32_t* out,
                              uint64_t out_cap, uint64_t* n_lists) {
    const uint8_t* begin = enc;
    const uint8_t* end = enc + enc_bytes;
    uint64_t total = 0, lists = 0;
    uint32_t* buf = NULL;
    size_t buf_cap = 0;
    while (begin != end) {
        uint32_t n, universe;
        begin = oracle_header_read(begin, &n, &universe);
        size_t need = (size_t)n + BLOCK + MAX_ENTRY;
        if (need > buf_cap) {
            free(buf);
            buf_cap = need * 2;
            buf = (uint32_t*)malloc(buf_cap * 4);
            if (!buf) return (uint64_t)-1;
        }
        memset(buf, 0, need * 4);
        begin = oracle_decode_list(d, begin, buf, n);
        if (out) {
            if (total + n > out_cap) {
                free(buf);
                return (uint64_t)-1;
            }
            memcpy(out + total, buf, (size_t)n * 4);
        }
        total += n;
        ++lists;
        if (begin > end) break;
    }
    free(buf);
    if (n_lists) *n_lists = lists;
    return total;
}

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* vroom_env/decode.cpp:125-155 */
double oracle_time_stream(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, uint64_t max_lists,
                          double max_seconds, uint64_t* ints_decoded, uint64_t* lists_decoded) {
    const uint8_t* begin = enc;
    const uint8_t* end = enc + enc_bytes;
    uint32_t* decoded = (uint32_t*)calloc((size_t)MAX_LIST + BLOCK + MAX_ENTRY, 4); /* zeroed ONCE, :128-129 */
    if (!decoded) return -1.0;
    double elapsed = 0.0;
    uint64_t ints = 0, lists = 0;
    while (begin != end) {
        uint32_t n, universe;
        begin = oracle_header_read(begin, &n, &universe);
        double t0 = now_sec();
        begin = oracle_decode_list(d, begin, decoded, n);
        double t1 = now_sec();
        elapsed += t1 - t0;
        ints += n;
        ++lists;
        if (max_lists && lists >= max_lists) break;
        if (max_seconds > 0 && elapsed >= max_seconds) break;
    }
    /* keep the result observable */
    volatile uint32_t sink = decoded[0];
    (void)sink;
    free(decoded);
    if (ints_decoded) *ints_decoded = ints;
    if (lists_decoded) *lists_decoded = lists;
    return elapsed;
}

/* The all-cores leg of the CPU baseline: the decode.cpp loop (vroom_env/decode.cpp:139-150: header::read,
 * then Decoder::decode into ONE reused, zeroed-once buffer) on every thread over its own contiguous range of lists; a
 * thread's buffer is persistent and as long as its longest list needs. Every thread walks its range again and again
 * until `seconds` have passed since the common start (looked at every 32 lists), so all of them end together; the
 * rate is the integers all threads decoded over the wall time from the first thread's start to the last one's end. */
typedef struct {
    const oracle_dict* d;
    const uint8_t* begin;
    const uint8_t* end;
    uint32_t* buf;
    double t_start, seconds, t_first, t_last;
    uint64_t ints, lists;
} par_job;

static void* par_worker(void* arg) {
    par_job* j = (par_job*)arg;
    j->t_first = now_sec();
    uint64_t ints = 0, lists = 0;
    int stop = j->begin == j->end;
    while (!stop) {
        const uint8_t* p = j->begin;
        while (p != j->end) {
            uint32_t n, universe;
            p = oracle_header_read(p, &n, &universe);
            p = oracle_decode_list(j->d, p, j->buf, n);
            ints += n;
            if ((++lists & 31u) == 0 && now_sec() - j->t_start >= j->seconds) {
                stop = 1;
                break;
            }
        }
        if (now_sec() - j->t_start >= j->seconds) stop = 1;
    }
    volatile uint32_t sink = j->buf[0];
    (void)sink;
    j->ints = ints;
    j->lists = lists;
    j->t_last = now_sec();
    return NULL;
}

double oracle_time_stream_parallel(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, const uint64_t* range_starts,
                                   uint32_t n_threads, double seconds, uint64_t* ints_decoded, uint64_t* lists_decoded) {
    if (n_threads == 0) return -1.0;
    par_job* jobs = (par_job*)calloc(n_threads, sizeof *jobs);
    pthread_t* th = (pthread_t*)calloc(n_threads, sizeof *th);
    if (!jobs || !th) {
        free(jobs);
        free(th);
        return -1.0;
    }
    double result = -1.0;
    uint32_t made = 0;
    uint32_t* scratch = (uint32_t*)calloc((size_t)MAX_LIST + BLOCK + MAX_ENTRY, 4);
    if (!scratch) goto done;
    /* set-up, untimed: the ranges and every thread's buffer (sized by a pass over its headers) */
    for (uint32_t k = 0; k != n_threads; ++k) {
        const uint64_t a = range_starts[k], b = k + 1 < n_threads ? range_starts[k + 1] : enc_bytes;
        if (a > b || b > enc_bytes) goto done;
        jobs[k].d = d;
        jobs[k].begin = enc + a;
        jobs[k].end = enc + b;
        jobs[k].seconds = seconds;
        uint32_t longest = 0;
        const uint8_t* p = jobs[k].begin; /* one walk to find the range's longest list (payloads carry no length) */
        while (p != jobs[k].end) {
            uint32_t n, universe;
            p = oracle_header_read(p, &n, &universe);
            p = oracle_decode_list(d, p, scratch, n);
            if (n > longest) longest = n;
        }
        jobs[k].buf = (uint32_t*)calloc((size_t)longest + BLOCK + MAX_ENTRY, 4);
        if (!jobs[k].buf) goto done;
    }
    {
        const double t0 = now_sec();
        for (uint32_t k = 0; k != n_threads; ++k) jobs[k].t_start = t0;
        for (; made != n_threads; ++made)
            if (pthread_create(&th[made], NULL, par_worker, &jobs[made]) != 0) break;
        for (uint32_t k = 0; k != made; ++k) pthread_join(th[k], NULL);
        if (made != n_threads) goto done;
        double first = jobs[0].t_first, last = jobs[0].t_last;
        uint64_t ints = 0, lists = 0;
        for (uint32_t k = 0; k != n_threads; ++k) {
            if (jobs[k].t_first < first) first = jobs[k].t_first;
            if (jobs[k].t_last > last) last = jobs[k].t_last;
            ints += jobs[k].ints;
            lists += jobs[k].lists;
        }
        if (ints_decoded) *ints_decoded = ints;
        if (lists_decoded) *lists_decoded = lists;
        result = last - first;
    }
done:
    free(scratch);
    for (uint32_t k = 0; k != n_threads; ++k) free(jobs[k].buf);
    free(jobs);
    free(th);
    return result;
}

/* ---- in-index path --------------------------------------------------------------------- */

/* bit_reader, include/ds2i/interpolative_coding.hpp:79-153 */
typedef struct {
    const uint8_t* in; /* next u32 to fetch (unaligned) */
    uint32_t avail;
    uint64_t buf;
    size_t pos;
} bit_reader;

static uint32_t br_read(bit_reader* r, uint32_t len) { /* :93-108 */
    if (!len) return 0;
    if (r->avail < len) {
        r->buf |= (uint64_t)ld32(r->in) << r->avail;
        r->in += 4;
        r->avail += 32;
    }
    uint32_t val = (uint32_t)(r->buf & (((uint64_t)1 << len) - 1));
    r->buf >>= len;
    r->avail -= len;
    r->pos += len;
    return val;
}

static uint32_t br_read_int(bit_reader* r, uint32_t u) { /* :110-122 */
    uint32_t b = 31u - (uint32_t)__builtin_clz(u); /* succinct::broadword::msb */
    uint64_t m = ((uint64_t)1 << (b + 1)) - u;
    uint32_t val = br_read(r, b);
    if (val >= m) val = (val << 1) + br_read(r, 1) - (uint32_t)m;
    return val;
}

static void br_read_interpolative(bit_reader* r, uint32_t* out, size_t n, uint32_t low, uint32_t high) { /* :124-146 */
    size_t h = n / 2;
    uint32_t val = low + br_read_int(r, high - low + 1);
    out[h] = val;
    if (n == 1) return;
    if (h) br_read_interpolative(r, out, h, low, val);
    if (n - h - 1) br_read_interpolative(r, out + h + 1, n - h - 1, val, high);
}

/* interpolative_block::decode, include/ds2i/block_codecs.hpp:130-150 */
const uint8_t* oracle_interpolative_decode(const uint8_t* in, uint32_t* out, uint32_t sum_of_values, size_t n) {
    const uint8_t* inbuf = in;
    if (sum_of_values == (uint32_t)-1) inbuf = oracle_vbyte_read(inbuf, &sum_of_values);
    out[n - 1] = sum_of_values;
    size_t read_bytes = 0;
    if (n > 1) {
        bit_reader r = {inbuf, 0, 0, 0};
        br_read_interpolative(&r, out, n - 1, 0, sum_of_values);
        for (size_t i = n - 1; i > 0; --i) out[i] -= out[i - 1];
        read_bytes = (r.pos + 7) / 8;
    }
    return inbuf + read_bytes;
}

/* dint_block::decode (:13-49) and opt_dint_multi_dict_block::decode (:460-510): a full block is
 * exactly one whole-list decode of 256 integers (single: 16-bit codewords; multi: selector byte
 * + one block), a shorter one is interpolative. */
const uint8_t* oracle_block_decode(const oracle_dict* d, const uint8_t* in, uint32_t* out, uint32_t sum_of_values,
                                   size_t n) {
    if (__builtin_expect(n < BLOCK, 0)) return oracle_interpolative_decode(in, out, sum_of_values, n);
    return oracle_decode_list(d, in, out, n);
}

uint32_t oracle_posting_list_decode(const oracle_dict* docs_dict, const oracle_dict* freqs_dict,
                                    const uint8_t* list, uint32_t* docids, uint32_t* freqs) {
    uint32_t n;
    const uint8_t* base = oracle_vbyte_read(list, &n); /* document_enumerator ctor, :90-107 */
    if (!docids && !freqs) return n;
    uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    const uint8_t* block_maxs = base;
    const uint8_t* block_endpoints = block_maxs + 4 * (size_t)blocks;
    const uint8_t* blocks_data = block_endpoints + 4 * (size_t)(blocks - 1);
    uint32_t docs_buf[BLOCK + 256], freqs_buf[BLOCK + 256];
    size_t pos = 0;
    for (uint32_t b = 0; b != blocks; ++b) {
        /* decode_docs_block, :284-309 */
        uint32_t endpoint = b ? ld32(block_endpoints + 4 * (size_t)(b - 1)) : 0;
        const uint8_t* block_data = blocks_data + endpoint;
        uint32_t cur_size = ((b + 1) * BLOCK <= n) ? BLOCK : (n % BLOCK);
        uint32_t cur_base = (b ? ld32(block_maxs + 4 * (size_t)(b - 1)) : (uint32_t)-1) + 1;
        uint32_t cur_max = ld32(block_maxs + 4 * (size_t)b);
        memset(docs_buf, 0, sizeof docs_buf); /* std::fill, :296 */
        const uint8_t* freqs_data =
            oracle_block_decode(docs_dict, block_data, docs_buf, cur_max - cur_base - (cur_size - 1), cur_size);
        docs_buf[0] += cur_base;
        /* decode_freqs_block, :311-318 */
        memset(freqs_buf, 0, sizeof freqs_buf);
        oracle_block_decode(freqs_dict, freqs_data, freqs_buf, (uint32_t)-1, cur_size);
        /* next(): m_cur_docid += m_docs_buf[pos] + 1 (:111-124); freq(): buf + 1 (:164-169) */
        uint32_t docid = docs_buf[0];
        for (uint32_t i = 0; i != cur_size; ++i) {
            if (i) docid += docs_buf[i] + 1;
            if (docids) docids[pos] = docid;
            if (freqs) freqs[pos] = freqs_buf[i] + 1;
            ++pos;
        }
    }
    return n;
}

/* ---- AND queries ------------------------------------------------------------------------ */

/* document_enumerator, include/dint/dict_posting_list.hpp:88-342 (docs side only) */
typedef struct {
    const oracle_dict* dict;
    uint32_t n, blocks;
    const uint8_t *block_maxs, *block_endpoints, *blocks_data;
    uint64_t universe;
    uint32_t cur_block, pos_in_block, cur_block_max, cur_block_size, cur_docid;
    uint32_t docs_buf[BLOCK + 256];
    /* freqs side (:164-169, :311-318): decoded lazily, once per block, on the first freq() */
    const oracle_dict* freqs_dict;
    const uint8_t* freqs_block_data;
    int freqs_decoded;
    uint64_t freqs_blocks_decoded;
    uint32_t freqs_buf[BLOCK + 256];
} oracle_enum;

static uint32_t en_block_max(const oracle_enum* e, uint32_t b) { return ld32(e->block_maxs + 4 * (size_t)b); }

static void en_decode_docs_block(oracle_enum* e, uint32_t block) { /* :284-309 */
    uint32_t endpoint = block ? ld32(e->block_endpoints + 4 * (size_t)(block - 1)) : 0;
    const uint8_t* block_data = e->blocks_data + endpoint;
    e->cur_block_size = ((block + 1) * BLOCK <= e->n) ? BLOCK : (e->n % BLOCK);
    uint32_t cur_base = (block ? en_block_max(e, block - 1) : (uint32_t)-1) + 1;
    e->cur_block_max = en_block_max(e, block);
    memset(e->docs_buf, 0, sizeof e->docs_buf);
    e->freqs_block_data = oracle_block_decode(e->dict, block_data, e->docs_buf,
                                              e->cur_block_max - cur_base - (e->cur_block_size - 1), e->cur_block_size);
    e->freqs_decoded = 0;
    e->docs_buf[0] += cur_base;
    e->cur_block = block;
    e->pos_in_block = 0;
    e->cur_docid = e->docs_buf[0];
}

static uint32_t en_freq(oracle_enum* e) { /* freq(), :164-169; decode_freqs_block, :311-318 */
    if (!e->freqs_decoded) {
        memset(e->freqs_buf, 0, sizeof e->freqs_buf);
        oracle_block_decode(e->freqs_dict, e->freqs_block_data, e->freqs_buf, (uint32_t)-1, e->cur_block_size);
        e->freqs_decoded = 1;
        e->freqs_blocks_decoded += 1;
    }
    return e->freqs_buf[e->pos_in_block] + 1;
}

static void en_init(oracle_enum* e, const oracle_dict* d, const uint8_t* data, uint64_t universe) { /* :90-109 */
    e->dict = d;
    e->freqs_dict = NULL;
    e->freqs_blocks_decoded = 0;
    const uint8_t* base = oracle_vbyte_read(data, &e->n);
    e->blocks = (e->n + BLOCK - 1) / BLOCK;
    e->block_maxs = base;
    e->block_endpoints = base + 4 * (size_t)e->blocks;
    e->blocks_data = e->block_endpoints + 4 * (size_t)(e->blocks - 1);
    e->universe = universe;
    en_decode_docs_block(e, 0);
}

static void en_next(oracle_enum* e) { /* :111-124 */
    ++e->pos_in_block;
    if (e->pos_in_block == e->cur_block_size) {
        if (e->cur_block + 1 == e->blocks) {
            e->cur_docid = (uint32_t)e->universe;
            return;
        }
        en_decode_docs_block(e, e->cur_block + 1);
    } else {
        e->cur_docid += e->docs_buf[e->pos_in_block] + 1;
    }
}

static void en_next_geq(oracle_enum* e, uint64_t lower_bound) { /* :126-147 */
    if (lower_bound > e->cur_block_max) {
        if (lower_bound > en_block_max(e, e->blocks - 1)) {
            e->cur_docid = (uint32_t)e->universe;
            return;
        }
        uint32_t block = e->cur_block + 1;
        while (en_block_max(e, block) < lower_bound) ++block;
        en_decode_docs_block(e, block);
    }
    while (e->cur_docid < lower_bound) e->cur_docid += e->docs_buf[++e->pos_in_block] + 1;
}

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : x > y;
}
static int cmp_enum_size(const void* a, const void* b) {
    const oracle_enum* x = *(oracle_enum* const*)a;
    const oracle_enum* y = *(oracle_enum* const*)b;
    return x->n < y->n ? -1 : x->n > y->n;
}

/* and_query<with_freqs>, include/ds2i/queries.hpp:34-84. freqs_dict != NULL: with_freqs = true — at every match the
 * freq() of every enumerator is read (:72-76); the reference throws the values away (do_not_optimize_away), here
 * their sum goes to *freq_sum and the number of freqs blocks that had to be decoded to *freqs_blocks. */
static uint64_t and_query_impl(const oracle_dict* docs_dict, const oracle_dict* freqs_dict, const uint8_t* index,
                               const uint64_t* list_offsets, uint64_t num_docs, const uint32_t* terms_in, size_t n_terms,
                               uint64_t* freq_sum, uint64_t* freqs_blocks) {
    if (freq_sum) *freq_sum = 0;
    if (freqs_blocks) *freqs_blocks = 0;
    if (!n_terms) return 0;
    uint32_t* terms = (uint32_t*)malloc(n_terms * 4);
    memcpy(terms, terms_in, n_terms * 4);
    qsort(terms, n_terms, 4, cmp_u32); /* remove_duplicate_terms, :28-31 */
    size_t m = 0;
    for (size_t i = 0; i != n_terms; ++i)
        if (!m || terms[i] != terms[m - 1]) terms[m++] = terms[i];
    oracle_enum* store = (oracle_enum*)malloc(m * sizeof(oracle_enum));
    oracle_enum** enums = (oracle_enum**)malloc(m * sizeof(oracle_enum*));
    for (size_t i = 0; i != m; ++i) {
        en_init(&store[i], docs_dict, index + list_offsets[terms[i]], num_docs);
        store[i].freqs_dict = freqs_dict;
        enums[i] = &store[i];
    }
    qsort(enums, m, sizeof(oracle_enum*), cmp_enum_size); /* sort by increasing frequency, :49-52 */
    uint64_t results = 0;
    uint64_t candidate = enums[0]->cur_docid;
    size_t i = 1;
    while (candidate < num_docs) {
        for (; i < m; ++i) {
            en_next_geq(enums[i], candidate);
            if (enums[i]->cur_docid != candidate) {
                candidate = enums[i]->cur_docid;
                i = 0;
                break;
            }
        }
        if (i == m) {
            results += 1;
            if (freqs_dict) {
                for (i = 0; i < m; ++i) *freq_sum += en_freq(enums[i]);
            }
            en_next(enums[0]);
            candidate = enums[0]->cur_docid;
            i = 1;
        }
    }
    if (freqs_blocks)
        for (size_t k = 0; k != m; ++k) *freqs_blocks += store[k].freqs_blocks_decoded;
    free(terms);
    free(store);
    free(enums);
    return results;
}

uint64_t oracle_and_query(const oracle_dict* docs_dict, const uint8_t* index, const uint64_t* list_offsets,
                          uint64_t num_docs, const uint32_t* terms_in, size_t n_terms) {
    return and_query_impl(docs_dict, NULL, index, list_offsets, num_docs, terms_in, n_terms, NULL, NULL);
}

uint64_t oracle_and_query_freqs(const oracle_dict* docs_dict, const oracle_dict* freqs_dict, const uint8_t* index,
                                const uint64_t* list_offsets, uint64_t num_docs, const uint32_t* terms_in, size_t n_terms,
                                uint64_t* freq_sum, uint64_t* freqs_blocks) {
    uint64_t dummy;
    return and_query_impl(docs_dict, freqs_dict, index, list_offsets, num_docs, terms_in, n_terms, freq_sum ? freq_sum : &dummy,
                          freqs_blocks);
}

/* ---- a query log on several cores: the honest CPU figure next to the device's batch rate ----------------------------
 * The reference runs its queries one after the other on one core (src/queries.cpp:15-61). A machine with more cores
 * answers a LOG of independent queries in parallel: thread k takes the queries k, k + T, k + 2T, ... (interleaved: a
 * log's heavy queries cluster), every one through the same and_query as above; `passes` times over; the result is the wall
 * time from the first thread's start to the last thread's end of ONE pass on average. One pthread per share INSIDE this
 * library: driving oracle_and_query from a Python thread pool measures the interpreter (round 4's 13.75 us per query
 * on 16 threads against 9 us on one). */
typedef struct {
    const oracle_dict* d;
    const uint8_t* index;
    const uint64_t* list_offsets;
    uint64_t num_docs;
    const uint32_t* terms;
    const uint64_t* offsets;
    uint64_t* counts;
    uint64_t n_queries;
    uint32_t k, n_threads, passes;
    double t_first, t_last;
} query_job;

static void* query_worker(void* arg) {
    query_job* j = (query_job*)arg;
    j->t_first = now_sec();
    for (uint32_t pass = 0; pass != j->passes; ++pass)
        for (uint64_t q = j->k; q < j->n_queries; q += j->n_threads)
            j->counts[q] = and_query_impl(j->d, NULL, j->index, j->list_offsets, j->num_docs, j->terms + j->offsets[q],
                                          (size_t)(j->offsets[q + 1] - j->offsets[q]), NULL, NULL);
    j->t_last = now_sec();
    return NULL;
}

double oracle_and_queries_parallel(const oracle_dict* docs_dict, const uint8_t* index, const uint64_t* list_offsets, uint64_t num_docs,
                                   const uint32_t* terms, const uint64_t* offsets, uint64_t n_queries, uint32_t n_threads,
                                   uint32_t passes, uint64_t* counts) {
    if (n_threads == 0 || passes == 0) return -1.0;
    query_job* jobs = (query_job*)calloc(n_threads, sizeof *jobs);
    pthread_t* th = (pthread_t*)calloc(n_threads, sizeof *th);
    double result = -1.0;
    if (jobs && th) {
        uint32_t made = 0;
        for (uint32_t k = 0; k != n_threads; ++k)
            jobs[k] = (query_job){docs_dict, index, list_offsets, num_docs, terms, offsets, counts, n_queries, k, n_threads, passes, 0, 0};
        for (; made != n_threads; ++made)
            if (pthread_create(&th[made], NULL, query_worker, &jobs[made]) != 0) break;
        for (uint32_t k = 0; k != made; ++k) pthread_join(th[k], NULL);
        if (made == n_threads) {
            double first = jobs[0].t_first, last = jobs[0].t_last;
            for (uint32_t k = 1; k != n_threads; ++k) {
                if (jobs[k].t_first < first) first = jobs[k].t_first;
                if (jobs[k].t_last > last) last = jobs[k].t_last;
            }
            result = (last - first) / passes;
        }
    }
    free(jobs);
    free(th);
    return result;
}
