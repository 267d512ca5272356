/*
 * dint_oracle.h — CPU restatement of the reference's DINT decode path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * (or, in bench.py, as the timed CPU baseline). The product path
 * (dint_amd/, include/dint_hip.h) never links, imports or calls it.
 *
 * PARITY UNPINNED. The reference (jermp/dint) cannot be built in this image:
 * its hot-path headers include <succinct/mappable_vector.hpp>,
 * <succinct/broadword.hpp> and Boost, whose submodules/packages are absent
 * (external/ is empty, /usr/include/boost does not exist), and writing
 * stand-ins for them is not allowed. The reference also ships no golden
 * vectors or tests for DINT (test/ covers the inherited ds2i codecs only).
 * This file is therefore a line-by-line restatement from reading the source;
 * every function cites the reference lines it follows. What IS checked:
 * hand-assembled known-answer streams built from the byte format
 * (tests/golden/), round trips encode -> decode == input (the reference's own
 * correctness contract, vroom_env/check_encoded_data.cpp:76-113), and the
 * MurmurHash64A key function against the one reference file that does compile
 * stand-alone (oracle/_ref, include/dint/hash_utils.hpp).
 */
#ifndef DINT_ORACLE_H
#define DINT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_RECT = 0, ORACLE_SINGLE_PACKED = 1, ORACLE_MULTI_PACKED = 2 };

typedef struct oracle_dict oracle_dict;

/* Parse a dictionary file image as the reference's builder::load does, then
 * "build(dict)" (steal the vectors). Returns NULL on a malformed image. */
oracle_dict* oracle_dict_load(int kind, const void* file_bytes, size_t len);
void oracle_dict_free(oracle_dict* d);
/* Dictionary::copy(i, out) / copy(dict_id, i, out): always writes 16 words,
 * returns the logical size. */
uint32_t oracle_dict_copy(const oracle_dict* d, uint32_t dict_id, uint32_t i, uint32_t* out);

/* TightVariableByte::decode of one value; returns the advanced pointer. */
const uint8_t* oracle_vbyte_read(const uint8_t* in, uint32_t* val);
/* header::read */
const uint8_t* oracle_header_read(const uint8_t* in, uint32_t* n, uint32_t* universe);

/* single_dint::decode (rect and single_packed dictionaries) and
 * multi_opt_dint::decode. `out` must be zero on entry and have room for
 * n + 256 words (the reference's overflow area). Return the advanced input
 * pointer. */
const uint8_t* oracle_decode_single(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n);
const uint8_t* oracle_decode_multi(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n);
/* dispatch on the dictionary kind */
const uint8_t* oracle_decode_list(const oracle_dict* d, const uint8_t* in, uint32_t* out, size_t n);

/* The decode.cpp loop over a whole vroom stream, with the output buffer
 * re-zeroed before every list as check_encoded_data.cpp does.
 * out (may be NULL) receives all lists back to back: capacity out_cap words.
 * Returns the number of integers decoded, or (uint64_t)-1 on overrun. */
uint64_t oracle_decode_stream(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, uint32_t* out,
                              uint64_t out_cap, uint64_t* n_lists);

/* The reference benchmark (vroom_env/decode.cpp:125-155): ONE output buffer of
 * max_size words zeroed once and reused, per-list steady-clock time around the
 * decode call only, summed. Stops after max_lists lists (0 = all) or when
 * max_seconds of summed decode time is exceeded (0 = no limit). */
double oracle_time_stream(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, uint64_t max_lists,
                          double max_seconds, uint64_t* ints_decoded, uint64_t* lists_decoded);

/* The all-cores leg of the same benchmark: thread k runs the decode.cpp loop over the lists whose headers start in
 * [range_starts[k], range_starts[k + 1]) (the last range ends at enc_bytes; an empty range is allowed), with its own
 * persistent zeroed-once buffer, again and again until `seconds` have passed. Returns the wall time from the first
 * thread's start to the last thread's end (buffers and ranges are set up before), or a negative value on failure. */
double oracle_time_stream_parallel(const oracle_dict* d, const uint8_t* enc, size_t enc_bytes, const uint64_t* range_starts,
                                   uint32_t n_threads, double seconds, uint64_t* ints_decoded, uint64_t* lists_decoded);

/* ---- in-index path ------------------------------------------------------------------- */

/* interpolative_block::decode (include/ds2i/block_codecs.hpp:130-150): n <= 256 values whose
 * prefix sums were binary-interpolative coded; sum_of_values == 0xFFFFFFFF means the sum is
 * vbyte-coded first. Returns the advanced input pointer. */
const uint8_t* oracle_interpolative_decode(const uint8_t* in, uint32_t* out, uint32_t sum_of_values, size_t n);

/* dint_block::decode / opt_dint_multi_dict_block::decode (include/dint/dint_codecs.hpp:13-49,
 * 460-510): one posting-list block; n < 256 -> interpolative. `out` zeroed, n + 256 words. */
const uint8_t* oracle_block_decode(const oracle_dict* d, const uint8_t* in, uint32_t* out, uint32_t sum_of_values,
                                   size_t n);

/* dict_posting_list::document_enumerator walked from the first to the last posting
 * (include/dint/dict_posting_list.hpp:88-342): decode_docs_block / decode_freqs_block for every
 * block, docid accumulation as next() does. docids/freqs must hold n values (n = first vbyte of
 * the list; call with NULL outputs to get it). Returns n. */
uint32_t oracle_posting_list_decode(const oracle_dict* docs_dict, const oracle_dict* freqs_dict,
                                    const uint8_t* list, uint32_t* docids, uint32_t* freqs);

/* and_query<false>::operator() (include/ds2i/queries.hpp:34-84) over document_enumerators with
 * the reference's next_geq / next (include/dint/dict_posting_list.hpp:111-147): block-max
 * skipping, lazy per-block decode. Lists are addressed through list_offsets (this repo's index
 * container); num_docs is the universe. Returns the number of documents that contain every term. */
uint64_t oracle_and_query(const oracle_dict* docs_dict, const uint8_t* index, const uint64_t* list_offsets,
                          uint64_t num_docs, const uint32_t* terms, size_t n_terms);

/* and_query<true>::operator() (queries.hpp:34-84 with :72-76 live): the same count; *freq_sum = sum over the matches and
 * over the query's enumerators of freq() (document_enumerator::freq, dict_posting_list.hpp:164-169, which decodes the
 * block's freqs part on first use, :311-318); *freqs_blocks = how many freqs parts were decoded. */
uint64_t oracle_and_query_freqs(const oracle_dict* docs_dict, const oracle_dict* freqs_dict, const uint8_t* index,
                                const uint64_t* list_offsets, uint64_t num_docs, const uint32_t* terms_in, size_t n_terms,
                                uint64_t* freq_sum, uint64_t* freqs_blocks);

/* A log of n_queries AND queries (query q: terms[offsets[q] .. offsets[q + 1])) answered by n_threads pthreads inside this
 * library, query q by thread q % n_threads, `passes` times over; counts[q] = its matches. Returns the wall seconds of one
 * pass (first thread's start to last thread's end, over passes), negative on failure. */
double oracle_and_queries_parallel(const oracle_dict* docs_dict, const uint8_t* index, const uint64_t* list_offsets, uint64_t num_docs,
                                   const uint32_t* terms, const uint64_t* offsets, uint64_t n_queries, uint32_t n_threads,
                                   uint32_t passes, uint64_t* counts);

/* ---- dictionary construction statistics (dint_oracle_stats.c) ----------------------------- */

/* selector::get (include/dint/statistics_collectors.hpp:21-40): the context of a block of n integers. */
uint32_t oracle_selector_get(const uint32_t* entry, size_t n);
/* hash_bytes64 over whole u32 words (include/dint/hash_utils.hpp:7-71, :77-80): the n-grams' only key. */
uint64_t oracle_hash_u32s(const uint32_t* p, size_t n);

typedef struct oracle_stats oracle_stats;
typedef struct {
    uint64_t pos;     /* the n-gram's integers: gaps[pos, pos + len) (its first occurrence) */
    uint64_t freq;    /* occurrences counted */
    uint32_t len;     /* 1, 2, 4, 8 or 16 */
    uint32_t context; /* 0 for single dictionaries, the block selector for multi */
} oracle_ngram;

/* block_statistics / block_multi_statistics construction (include/dint/block_statistics.hpp:45-108, :201-279):
 * create over the collection's gaps (the caller's array, kept by reference), collect() once per list, then read the
 * counts (entries) or the dictionary-ordered selection (select: filter, freq_length_sorter, the first 65536). */
oracle_stats* oracle_stats_create(int multi, const uint32_t* gaps);
void oracle_stats_free(oracle_stats* st);
int oracle_stats_collect(oracle_stats* st, uint64_t first, uint64_t n); /* adjusted::collect, statistics_collectors.hpp:90-118 */
uint64_t oracle_stats_total(const oracle_stats* st);
uint64_t oracle_stats_distinct(const oracle_stats* st, uint32_t context);
uint64_t oracle_stats_entries(const oracle_stats* st, uint32_t context, oracle_ngram* out, uint64_t cap);
/* dictionary_builders.hpp:15-38, :50-75: returns the number of entries that pass the filter; writes min(that, 65536, cap) */
uint64_t oracle_stats_select(const oracle_stats* st, uint32_t context, oracle_ngram* out, uint64_t cap);

/* ---- the ENCODE side and the dictionary packing (f2); dint_oracle_encode.c ---------------- */

/* std::vector<uint8_t>: zero-initialise, pass to the calls below (they append), release with oracle_bytes_free. */
typedef struct {
    uint8_t* data;
    size_t size, cap;
} oracle_bytes;
void oracle_bytes_free(oracle_bytes* b);

/* Dictionary::builder after load(file) and prepare_for_encoding(): rectangular_dictionary.hpp:79-92, :112-123;
 * single_dictionary.hpp:88-107, :154-165; multi_dictionary.hpp:93-121, :187-217 — with the hash-only maps of SURVEY H9. */
typedef struct oracle_builder oracle_builder;
oracle_builder* oracle_builder_load(int kind, const void* file_bytes, size_t len);
void oracle_builder_free(oracle_builder* b);
/* builder::lookup (single_dictionary.hpp:167-175, multi_dictionary.hpp:219-233): the codeword whose n-gram HASHES like
 * begin[0, entry_size), or 0xFFFFFFFF. dictionary_id / log2_num_entries (16 or 8) matter for the multi kind only. */
uint32_t oracle_builder_lookup(const oracle_builder* b, uint32_t dictionary_id, const uint32_t* begin, uint32_t entry_size,
                               uint32_t log2_num_entries);

/* The calls below return 1, or 0 when memory ran out / an argument is impossible. */

/* Encoder::encode of the vroom environment for one list of gaps: single_opt_dint (vroom_env/dint_codecs.hpp:192-312),
 * single_greedy_dint (:110-171, greedy != 0), multi_opt_dint (:334-518, picked by a multi builder; greedy ignored). */
int oracle_encode_list(const oracle_builder* b, int greedy, const uint32_t* in, uint32_t n, oracle_bytes* out);
/* TightVariableByte::encode_single (include/ds2i/block_codecs.hpp:35-85) */
int oracle_vbyte_encode(uint32_t val, oracle_bytes* out);
/* interpolative_block::encode (include/ds2i/block_codecs.hpp:104-128; bit_writer: interpolative_coding.hpp:10-77) */
int oracle_interpolative_encode(const uint32_t* in, uint32_t sum_of_values, size_t n, oracle_bytes* out);
/* Coder::encode of the index for one block of n <= 256 values: opt_dint_single_dict_block / greedy_dint_single_dict_block /
 * opt_dint_multi_dict_block (include/dint/dint_codecs.hpp:56-124, :145-267, :289-458) */
int oracle_block_encode(const oracle_builder* b, int greedy, const uint32_t* in, uint32_t sum_of_values, uint32_t n, oracle_bytes* out);
/* dict_posting_list::write (include/dint/dict_posting_list.hpp:10-56): docs = strictly increasing docIDs, freqs >= 1 */
int oracle_posting_list_write(const oracle_builder* docs_dict_builder, const oracle_builder* freqs_dict_builder, int greedy, uint32_t n,
                              const uint32_t* docs_begin, const uint32_t* freqs_begin, oracle_bytes* out);
/* The vroom `encode` program (vroom_env/encode.cpp:133-191 + jobs.hpp:74-95) over the u32 words of a collection file read as
 * binary_collection does (include/ds2i/binary_collection.hpp:131-146). docs != 0: a .docs file (record 0 skipped, values to
 * d-gaps); 0: a .freqs file (values minus one). */
int oracle_encode_collection(const oracle_builder* b, int greedy, const uint32_t* data, size_t data_size, int docs, oracle_bytes* output,
                             uint64_t* num_processed_lists, uint64_t* num_total_ints);
/* builder::init + append per selected n-gram + build + write: the dictionary FILE of a selection (words back to back, lens[k]
 * integers each, ctx[k] its dictionary — NULL for the single kinds). pack_policy::compact as written (O(n^2),
 * dictionary_building_utils.hpp:241-292), one std::search per entry (single_dictionary.hpp:138-151, multi_dictionary.hpp:
 * 165-181). */
int oracle_pack_dictionary(int kind, const uint32_t* words, const uint32_t* lens, const uint32_t* ctx, size_t n_entries, oracle_bytes* file);

#ifdef __cplusplus
}
#endif
#endif
