/*
 * dint_host.h — C ABI of the CPU-side (offline) half of the DINT path:
 * synthetic collections, dictionary construction and the vroom encoder.
 *
 * These are the producers of the decode path's inputs. In the reference they
 * are CPU C++ as well (vroom_env/encode.cpp, include/dint/dictionary_builders.hpp,
 * src/create_freq_index.cpp); nothing here runs on the device and nothing here
 * decodes. Python (bench.py, tests) binds this header with ctypes.
 */
#ifndef DINT_HOST_H
#define DINT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "dint_hip.h" /* dint_unit, dint_dict_kind, dint_status */

#ifdef __cplusplus
extern "C" {
#endif

/* Owned byte buffer returned by the functions below. */
typedef struct dinth_blob dinth_blob;
const void* dinth_blob_data(const dinth_blob* b);
size_t dinth_blob_size(const dinth_blob* b);
void dinth_blob_free(dinth_blob* b);

/* text of the last error raised on the calling thread */
const char* dinth_last_error(void);

typedef struct dinth_synth_params {
    uint64_t seed;
    uint32_t universe;
    uint32_t min_len;
    uint32_t max_len; /* 0 = universe / 3 */
    uint32_t reserved;
    double alpha;
    double stay_cluster;
    double stay_sparse;
    double p_cluster_min;
    double p_cluster_max;
} dinth_synth_params;

void dinth_synth_defaults(dinth_synth_params* p);
/* u32 list lengths summing exactly to target_postings */
int dinth_synth_lengths(const dinth_synth_params* p, uint64_t target_postings, dinth_blob** lens);
/* gaps_out must hold sum(lens) u32; list i of the collection has id first_list_id + i */
int dinth_synth_gaps(const dinth_synth_params* p, const uint32_t* lens, uint64_t n_lists,
                     uint64_t first_list_id, uint32_t* gaps_out, int threads);

/* Block statistics + DSF-65536-16 (reference dictionary_builders.hpp:40-76) over
 * the first lists whose lengths sum to <= max_sample_ints (0 = all lists).
 * Returns the dictionary FILE image (reference builder::write format). */
int dinth_build_dictionary(int kind, const uint32_t* gaps, const uint32_t* lens, uint64_t n_lists,
                           uint64_t max_sample_ints, int threads, dinth_blob** dict_file);

/* The selection half of the same construction, from n-gram counts made elsewhere (dint_count_ngrams on the device):
 * filter, frequency sort, DSF, packing — dictionary_builders.hpp:15-76 — over the entries' n-grams, read from
 * gaps[pos .. pos + len). total_ints: the integers of the sampled lists (the savings are relative to it,
 * block_statistics.hpp:250-262). Byte-identical to dinth_build_dictionary over the same lists. */
typedef struct dinth_ngram {  /* = dint_ngram of include/dint_hip.h */
    uint64_t pos;
    uint32_t freq;
    uint8_t len;
    uint8_t ctx;
    uint16_t pad;
} dinth_ngram;
int dinth_build_dictionary_from_ngrams(int kind, const uint32_t* gaps, uint64_t n_ints, uint64_t total_ints,
                                       const dinth_ngram* entries, uint64_t n_entries, dinth_blob** dict_file);

/* Packing only: `entries` are the dictionary's n-grams ALREADY selected and in dictionary order, context by context
 * (dint_select_ngrams on the device); each is appended as it comes and the table is packed and written
 * (builder::append / build / write: single_dictionary.hpp:109-160, dictionary_building_utils.hpp:241-292).
 * Byte-identical to dinth_build_dictionary_from_ngrams over the unselected counts. */
int dinth_pack_dictionary(int kind, const uint32_t* gaps, uint64_t n_ints, const dinth_ngram* entries, uint64_t n_entries,
                          dinth_blob** dict_file);

/* Encode lists into one vroom stream (reference vroom_env/encode.cpp:133-191).
 * kind selects the dictionary type of dict_file; greedy != 0 selects
 * single_greedy_dint instead of single_opt_dint (ignored for multi).
 * *units receives a dint_unit[] table cut every ~unit_ints integers
 * (0 = one unit per list). */
int dinth_encode_vroom(int kind, int greedy, const void* dict_file, size_t dict_len,
                       const uint32_t* gaps, const uint32_t* lens, uint64_t n_lists,
                       uint32_t unit_ints, int threads, dinth_blob** enc, dinth_blob** units);

/* Build an inverted index in the reference's in-index layout: every list is a
 * dict_posting_list (reference include/dint/dict_posting_list.hpp:10-56: vbyte n, block maxima,
 * block endpoints, per 256-posting block a docs part and a freqs part; blocks shorter than 256 are
 * binary-interpolative coded). docids: strictly increasing per list, freqs >= 1, both flat arrays
 * with lens[i] values per list. *index receives the lists back to back, *offsets a u64[n_lists + 1]
 * table of their byte offsets (this repo's own container; the reference freezes the same bytes with
 * succinct::mapper, which is absent here). */
int dinth_build_index(int kind, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                      size_t freqs_dict_len, const uint32_t* docids, const uint32_t* freqs, const uint32_t* lens,
                      uint64_t n_lists, int threads, dinth_blob** index, dinth_blob** offsets);

/* The same with the block coder named: greedy != 0 selects greedy_dint_single_dict_block (reference
 * include/dint/dint_codecs.hpp:52-139) for the single-dictionary kinds; the multi kind has the optimal coder only. */
int dinth_build_index_coder(int kind, int greedy, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                            size_t freqs_dict_len, const uint32_t* docids, const uint32_t* freqs, const uint32_t* lens,
                            uint64_t n_lists, int threads, dinth_blob** index, dinth_blob** offsets);

/* ---- ingest of a real collection (reference include/ds2i/binary_collection.hpp:13-157) --------------------------------
 * `words` / `n_words`: the u32 words of a ds2i collection file (records `len, v[len]`; empty records skipped, a truncated
 * last record cut at the end). docs != 0: a .docs file — record 0 (`1, num_docs`) is skipped and docIDs become d-gaps
 * minus one (first against -1); docs == 0: a .freqs file — every value minus one (vroom_env/jobs.hpp:74-84). */

/* The vroom `encode` program (reference vroom_env/encode.cpp:133-191): one list per record -> the encoded stream (+ the
 * unit-table sidecar, may be NULL). *n_lists / *n_ints (may be NULL): what the reference prints as num_sequences /
 * num_integers. */
int dinth_encode_collection(int kind, int greedy, const void* dict_file, size_t dict_len, const uint32_t* words, size_t n_words,
                            int docs, uint32_t unit_ints, int threads, dinth_blob** enc, dinth_blob** units, uint64_t* n_lists,
                            uint64_t* n_ints);
/* decreasing_static_frequencies::build over the statistics of the file's lists (reference dict_freq_index.hpp:139-161,
 * block_statistics.hpp:45-108 / :201-279, dictionary_builders.hpp:55-75) -> the dictionary file image. max_sample_ints: 0 =
 * every list (the reference), otherwise the first lists whose lengths sum to at most that. */
int dinth_build_dictionary_collection(int kind, const uint32_t* words, size_t n_words, int docs, uint64_t max_sample_ints,
                                      int threads, dinth_blob** dict_file);
/* dict_freq_index::builder over a .docs / .freqs pair (reference src/create_freq_index.cpp:54-110, dict_freq_index.hpp:30-49):
 * -> the index bytes, the u64[n_lists + 1] list offsets, *num_docs from record 0 of the .docs words. */
int dinth_build_index_collection(int kind, int greedy, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                                 size_t freqs_dict_len, const uint32_t* docs_words, size_t n_docs_words, const uint32_t* freqs_words,
                                 size_t n_freqs_words, int threads, dinth_blob** index, dinth_blob** offsets, uint64_t* num_docs);

/* MurmurHash64A(seed 0) of n u32 words (reference include/dint/hash_utils.hpp:7-80). */
uint64_t dinth_hash_u32s(const uint32_t* p, size_t n);

/* The compile-time constants this build was made with (dint/constants.hpp), for tests that pin them to the reference's
 * dint_configuration.hpp:6,20,24-28 / util.hpp:33-35: {exceptions, num_selectors, max_entry_size, num_entries,
 * num_target_sizes, target_sizes[0..4], block_size, reserved}. Returns how many there are; writes at most cap. */
int dinth_constants(uint32_t* out, int cap);
/* The context of a block (selector::get, reference statistics_collectors.hpp:21-40). */
uint32_t dinth_block_selector(const uint32_t* p, size_t n);

/* Dictionary file introspection for tests: writes up to cap (size, first
 * payload words) — returns number of entries of dictionary `d`. */
int dinth_dict_entry(int kind, const void* dict_file, size_t dict_len, uint32_t d, uint32_t index,
                     uint32_t* size_out, uint32_t* words16_out);
int dinth_dict_num_entries(int kind, const void* dict_file, size_t dict_len, uint32_t d,
                           uint32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* DINT_HOST_H */
