/*
 * dint_hip.h — C ABI of the MI355X (gfx950) DINT decode path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types. Each entry point names the reference interface (jermp/dint) it
 * replaces; the C++ adaptors that give these calls the reference's static
 * `Coder::decode(dict, in, out, sum, n) -> in_end` shape live in
 * dint_amd/csrc/host/dint/coders.hpp, and INTEGRATION.md shows the
 * reference-side binding.
 *
 * Threading: a dint_dict is immutable after creation and may be used from
 * several host threads with distinct streams (its launch bookkeeping — queue
 * slots, timing events — sits behind the handle's own `launch_mutex`). One
 * dint_dict lives on one device; multi-GPU = one dint_dict per device (the
 * dictionary is replicated, posting lists are partitioned, there is no
 * data-path collective). A dint_query_index may be called from several host
 * threads, each with its own stream: its calls SERIALISE on the handle's own
 * lock (`dint_query_index::mutex`, taken inside dint_and_queries /
 * dint_and_queries_freqs for the whole call — the handle's workspaces are one
 * set); two query indexes, or two dint_block_tables over one index and one
 * pair of dictionaries, share nothing mutable and run side by side. A
 * dint_block_table (and a dint_unit_table) belongs to one caller at a time:
 * its decodes are ordered on the stream they are enqueued on
 * (tests/test_gpu_queries.py::test_one_query_index_under_two_host_threads,
 * tests/test_gpu_index.py::test_two_block_tables_over_one_index_on_two_threads).
 *
 * All functions return DINT_OK (0) or a negative dint_status; none throws.
 */
#ifndef DINT_HIP_H
#define DINT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINT_ABI_VERSION 6

/* A unit decodes to at most this many integers (the kernels address a unit's output with 32-bit byte
 * offsets); dint_index_stream never cuts larger ones, dint_decode_units skips them. */
#define DINT_MAX_UNIT_INTS (1u << 28)

typedef enum dint_status {
    DINT_OK = 0,
    DINT_ERR_ARG = -1,       /* null pointer, bad enum, capacity too small          */
    DINT_ERR_FORMAT = -2,    /* dictionary file or encoded stream is malformed      */
    DINT_ERR_HIP = -3,       /* a HIP runtime call failed (see dint_last_hip_error) */
    DINT_ERR_NO_DEVICE = -4, /* no gfx950 device / device index out of range        */
    DINT_ERR_NOMEM = -5
} dint_status;

/* Dictionary flavours of the decode path (reference include/dint/dictionary_types.hpp:8-21). */
typedef enum dint_dict_kind {
    DINT_DICT_RECTANGULAR = 0,   /* single_rect_dint   */
    DINT_DICT_SINGLE_PACKED = 1, /* single_packed_dint */
    DINT_DICT_MULTI_PACKED = 2   /* multi_packed_dint  */
} dint_dict_kind;

/* Opaque device-resident dictionary.
 * Replaces: Dictionary::builder::load + builder.build(dict)
 *           (vroom_env/decode.cpp:116-123; single_dictionary.hpp:88-107,177-181;
 *            rectangular_dictionary.hpp:79-92; multi_dictionary.hpp:93-121). */
typedef struct dint_dict dint_dict;

/* One unit of decode work: a run of codewords that starts on a codeword
 * boundary and decodes to exactly `n` integers. The vroom stream has no sync
 * points (lists are `vbyte(n) vbyte(universe) payload` back to back,
 * vroom_env/jobs.hpp:89-91), so this table is the sidecar that makes the
 * stream parallel; it is produced by dint_index_stream() or by the encoder. */
typedef struct dint_unit {
    uint64_t in_off;  /* byte offset of the unit's first codeword in the encoded buffer */
    uint64_t out_off; /* index of the unit's first integer in the output buffer         */
    uint32_t n;       /* integers this unit decodes to (> 0)                             */
    uint32_t list;    /* ordinal of the posting list the unit belongs to                 */
} dint_unit;

typedef struct dint_dict_info {
    int32_t kind;           /* dint_dict_kind                                  */
    int32_t device;         /* HIP device ordinal                              */
    uint32_t num_dicts;     /* 1, or 6 for multi                               */
    uint32_t entries;       /* m_size of the file                              */
    uint32_t hot_entries;   /* codewords whose payload is staged in LDS        */
    uint32_t lds_bytes;     /* LDS image size in bytes                         */
    uint32_t table_words;   /* device table size in u32                        */
    uint32_t compute_units; /* CUs of the device the kernels are sized for     */
} dint_dict_info;

int dint_abi_version(void);

/* Process-wide switches for tests and measurements (the library reads no environment variable). Defaults are what a
 * caller wants; every entry point reads them with one relaxed atomic load, so a change takes effect from the next call
 * on and is safe against concurrent calls. dint_set_option refuses values outside an option's range (DINT_ERR_ARG). */
typedef enum dint_option {
    DINT_OPT_BUNDLES = 0,             /* 1 (default): tiny units share tiles (bundle schedule); 0: every unit on its own           */
    DINT_OPT_INDEX_CONCURRENT = 1,    /* 1 (default): dint_decode_block_table runs its launches side by side; 0: one stream      */
    DINT_OPT_QUERY_LEAN_PAGES = 2,    /* page decodes of at least this many pages take the three-launch form; -1 (default): none */
    DINT_OPT_QUERY_TAIL_PAGES = 3,    /* calls of at most this many candidate pages run a round per launch; default 4            */
    DINT_OPT_QUERY_FUSED_PAGES = 4,   /* ... of at most this many run as ONE launch; default 2, 0: never                         */
    DINT_OPT_INDEX_INLINE_TAILS = 5,  /* 1 (default): a created block table's short blocks are decoded inside its docs launch;   */
                                      /* 0: by a launch of their own                                                             */
    DINT_OPT_CHUNK_SPLIT = 6,         /* the bundle path hands out 1/2^n of a 64-unit chunk per ticket, n = 0..4; -1 (default):  */
                                      /* by the launch's size                                                                   */
    DINT_OPT_INDEX_PAIR = 7,          /* 1 (default): once a table's schedules are known, docs parts, short blocks and freqs     */
                                      /* parts of a decode are ONE launch; 0: a launch each                                      */
    DINT_OPT_QUERY_FUSED_COPY = 8,    /* 1 (default): the one-launch query form's workgroup fetches the call's inputs from the   */
                                      /* host's pinned memory itself; 0: a copy on the stream in front of the launch            */
    DINT_OPT_QUERY_BATCH_FUSED = 9,   /* 1 (default): a call whose queries all have few candidate pages runs as ONE launch, a      */
                                      /* workgroup per query; 0: the round-per-launch batch form                                */
    DINT_OPT_SPLIT_UNITS = 10,        /* 1 (default): a prepared multi-dictionary unit table cuts the units that fit no tile in two */
                                      /* records each (a second launch of the bundles kernel); 0: the general kernel decodes them */
    DINT_OPT_REFINE_UNITS = 11,       /* 1 (default): a prepared multi-dictionary unit table whose units hold several 256-integer  */
                                      /* blocks (at most 131072 integers each) finds the blocks once and decodes a table of      */
                                      /* blocks; 0: the units as they came (a wavefront decodes a unit's blocks one after another) */
    DINT_OPT_COUNT_ = 12
} dint_option;
int dint_set_option(int option, long long value);
int dint_get_option(int option, long long* value);
const char* dint_option_name(int option); /* "bundles", "index_concurrent", ... ; NULL past the last */
int dint_reset_options(void);             /* every option back to its default */

const char* dint_strerror(int status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char* dint_last_hip_error(void);
int dint_device_count(int* count);

/* Parse a dictionary file image (the bytes the reference's builder::write
 * produced) and stage it on `device`. */
int dint_dict_create(int kind, const void* file_bytes, size_t len, int device, dint_dict** out);
void dint_dict_destroy(dint_dict* dict);
int dint_dict_info_get(const dint_dict* dict, dint_dict_info* info);

/* Untimed host pre-pass over a whole vroom stream in host memory: reads every
 * list header, walks the codewords WITHOUT copying dictionary payloads, and
 * cuts each list into units of about `unit_ints` integers at codeword
 * boundaries (multi: at 256-integer block boundaries).
 * unit_ints = 0: one unit per list (a list of more than DINT_MAX_UNIT_INTS integers is still cut).
 * Replaces: the per-list framing loop of vroom_env/decode.cpp:139-150.
 * `*units` is malloc'ed; release with dint_free. */
int dint_index_stream(const dint_dict* dict, const uint8_t* enc, size_t enc_bytes,
                      uint32_t unit_ints, dint_unit** units, size_t* n_units,
                      uint64_t* total_ints, uint64_t* n_lists);
void dint_free(void* p);

/* Decode `n_units` units. All pointers except `dict` are DEVICE pointers on the
 * dictionary's device; `stream` is a hipStream_t (NULL = default stream). The
 * call is asynchronous. Exactly unit.n integers are written at
 * d_out[unit.out_off ...]; nothing else is touched (no pre-zeroed output, no
 * overflow area — unlike the reference, dint_codecs.hpp:11). If d_end_off is
 * not NULL, d_end_off[u] receives the byte offset one past unit u's last
 * consumed byte (the reference's returned `in` pointer).
 * Units may come in any order; runs of table-consecutive tiny units (<= 256
 * integers, <= 256 stream bytes up to the next unit's start, consecutive
 * outputs — the long tail of short posting lists) are decoded several to a
 * wavefront tile, which changes the speed, never the result.
 * Replaces: single_dint::decode / multi_opt_dint::decode
 *           (vroom_env/dint_codecs.hpp:37-107, :521-619). */
int dint_decode_units(const dint_dict* dict, const uint8_t* d_enc, size_t enc_bytes,
                      const dint_unit* d_units, size_t n_units, uint32_t* d_out,
                      size_t out_capacity, uint64_t* d_end_off, void* stream);

/* A unit table prepared for decoding. What depends on the unit table and the stream alone — which tiny units
 * share a wavefront tile, the work items of the unit queue, the selector bytes of a multi-dictionary stream's
 * blocks (the "bundle schedule": three small kernels and a read of 22 bytes per unit that dint_decode_units
 * runs before EVERY launch) — is computed once, here, like the sidecar itself (it is a property of the encoded
 * collection, not of a decode). The handle borrows `dict`, `d_enc` and `d_units`: they must outlive it and keep
 * their contents. `out_capacity` is the smallest output capacity later decodes may pass (the schedule's bounds
 * checks are made against it). Synchronises `stream`.
 * A multi-dictionary table whose units hold SEVERAL 256-integer blocks (at most 131072 integers each) is refined here: blocks
 * carry no length, so a unit of several is one wavefront's sequential work (242 G ints/s on the bench stream), where a table of
 * blocks packs three blocks into a tile (472-526 G). A lane per unit walks the unit's codewords once and the handle keeps a
 * unit record per block (24 bytes per 256 integers, + 22 bytes of schedule); decodes then run over the blocks, and d_end_off
 * still has one entry per unit of the CALLER's table. DINT_OPT_REFINE_UNITS = 0: the units as they came. Results are identical
 * either way. (Decodes of ONE table are ordered on one stream: its counters and, refined, its per-block end offsets are the
 * table's own.)
 * Replaces: nothing in the reference (its decode loop is sequential); it is the set-up half of
 * dint_decode_units, i.e. of vroom_env/decode.cpp:139-150. */
typedef struct dint_unit_table dint_unit_table;
int dint_unit_table_create(const dint_dict* dict, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units,
                           size_t n_units, size_t out_capacity, void* stream, dint_unit_table** out);
void dint_unit_table_destroy(dint_unit_table* table);
/* dint_decode_units over a prepared table: one kernel launch, asynchronous. out_capacity must be at least the
 * table's (DINT_ERR_ARG otherwise). Results are identical to dint_decode_units'. */
int dint_decode_unit_table(const dint_dict* dict, dint_unit_table* table, uint32_t* d_out, size_t out_capacity,
                           uint64_t* d_end_off, void* stream);
/* Where to put the output. The decode kernels run 10-17 % faster or slower depending on where the driver put the
 * stream they read relative to the output they write — a property of the PAIR of buffers, stable while both live,
 * which nothing but the decode kernel itself can see. A caller whose
 * output buffer lives for many decodes allocates a few candidates and lets this rank them: the table is decoded into
 * every candidate three times, kernel_ms[i] = the faster of the last two launches' kernel times, *fastest = the index
 * of the smallest. Synchronises `stream`; every candidate holds the decoded integers afterwards.
 * Replaces: nothing in the reference. */
int dint_unit_table_rank_outputs(const dint_dict* dict, dint_unit_table* table, uint32_t* const* d_outs, size_t n_outs,
                                 size_t out_capacity, void* stream, float* kernel_ms, size_t* fastest);
/* The same question for the price of a SAMPLE (round 6): which (stream copy, output buffer) pair do the decode kernels run
 * fastest on? Every candidate pair — n_encs copies of ONE encoded stream at different addresses, n_outs output buffers —
 * decodes the same evenly spread sample of the unit table (runs of 4 units, about `sample_ints` integers in all, 0: the
 * smallest launch that fills the device; the units keep their own places in stream and output, so the sample touches all of
 * both buffers) four times; kernel_ms[i * n_outs + j] = the fastest of the last three launches' kernel times on copy i and
 * output j. The slow / fast level of a pair is a property of where the driver put the two buffers and
 * shows in a sample as it does in the full decode (profiles/r06_placement_probe.txt); the stream is the small buffer of the
 * two (an eighth of the output), so a caller tries a few COPIES of it — allocated at different points of its set-up — against
 * the one output buffer it has: INTEGRATION.md section 6. Synchronises `stream`; the outputs hold the sample's integers.
 * Replaces: nothing in the reference. */
int dint_probe_placement(const dint_dict* dict, const uint8_t* const* d_encs, size_t n_encs, size_t enc_bytes,
                         const dint_unit* d_units, size_t n_units, uint32_t* const* d_outs, size_t n_outs, size_t out_capacity,
                         uint64_t sample_ints, void* stream, float* kernel_ms);

/* Host-pointer convenience with the reference's call shape: decode ONE
 * sequence of n integers starting at in[0]; *consumed = bytes read. Uploads,
 * runs one unit on the device, downloads, synchronises. n <= DINT_MAX_UNIT_INTS
 * (DINT_ERR_ARG beyond: index the stream and batch); a single wavefront wide:
 * batch through dint_decode_units for throughput.
 * Replaces: Coder::decode(dict, in, out, universe, n) at vroom_env/decode.cpp:143. */
int dint_decode_list_host(const dint_dict* dict, const uint8_t* in, size_t in_bytes,
                          uint32_t* out, size_t n, size_t* consumed);

/* ---- in-index path: posting lists in the dict_posting_list layout --------------------------- */

/* Host-pointer call with the reference's in-index BLOCK Coder shape: decode ONE block of n <= 256
 * integers starting at in[0]; *consumed = bytes read. n == 256: a DINT block — 16-bit codewords
 * (single dictionaries), or a selector byte and 16- / 8-bit codewords (multi) — through the DINT
 * kernels; n < 256: binary interpolative, `sum_of_values` being the sum of the block's integers or
 * 0xFFFFFFFF for "a vbyte of it comes first" (what the reference passes for freqs blocks), through
 * the interpolative kernel. Uploads, runs one block on the device, downloads, waits for the
 * dictionary's own stream (pinned staging and a device workspace kept by the dictionary: no
 * allocation after the first call of a size, no device-wide synchronisation): the reference's
 * granularity, one block per call — dint_list_cache_* decodes a list's blocks at once,
 * dint_decode_posting_blocks many lists'. Nothing past out[n - 1] is written and `out` need not be zeroed (the reference needs
 * both: block_size + overflow zeroed words, dict_posting_list.hpp:104-105, :296).
 * Replaces: dint_block::decode / opt_dint_single_dict_block::decode /
 *           opt_dint_multi_dict_block::decode (include/dint/dint_codecs.hpp:13-49, :269-274,
 *           :460-510) and interpolative_block::decode (include/ds2i/block_codecs.hpp:130-150), as
 *           called from dict_posting_list.hpp:298-301 and :313-315. */
int dint_decode_block_host(const dint_dict* dict, const uint8_t* in, size_t in_bytes, uint32_t* out,
                           uint32_t sum_of_values, size_t n, size_t* consumed);

/* A whole posting list (host pointer, the dict_posting_list layout below) decoded ONCE, its blocks then served from host
 * memory: what makes the block Coder usable under a document_enumerator, which calls Coder::decode per touched block
 * (dict_posting_list.hpp:298-301, :313-315) — one launch sequence per 256 postings otherwise. `create` uploads the
 * list, decodes every block's docs part (as the Coder returns it: d-gaps) and — freqs_dict not NULL — freqs part (as
 * stored: freq - 1) through the batched kernels on the dictionary's own stream, and downloads; it synchronises that
 * stream only. `decode` = the Coder call for the block part that starts `in_offset` bytes into the list: a memcpy;
 * *consumed = the part's bytes. DINT_ERR_ARG if no part of n integers starts there.
 * Replaces: dint_block::decode / interpolative_block::decode as called from dict_posting_list.hpp:298-301, :313-315,
 *           for a list at a time. */
typedef struct dint_list_cache dint_list_cache;
int dint_list_cache_create(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* list, size_t list_bytes,
                           dint_list_cache** out);
int dint_list_cache_decode(const dint_list_cache* cache, size_t in_offset, uint32_t* out, size_t n, size_t* consumed);
void dint_list_cache_destroy(dint_list_cache* cache);

/* One 256-posting block (the last block of a list may be shorter) of a posting list laid out as
 * reference include/dint/dict_posting_list.hpp:10-56:
 *   vbyte(n) | u32 block_max[B] | u32 block_endpoint[B-1] | { docs part, freqs part } x B
 * The block-max / endpoint arrays already make every block independently addressable, so no
 * sidecar is needed here: this table is just those arrays flattened over many lists. */
typedef struct dint_block_ref {
    uint64_t in_off;  /* byte offset of the block's docs part in the index buffer            */
    uint64_t out_off; /* index of the block's first posting in the output arrays             */
    uint32_t n;       /* postings in the block, 1..256                                        */
    uint32_t base;    /* docID base: previous block's max + 1 (0 for a list's first block)    */
    uint32_t max;     /* largest docID of the block                                           */
    uint32_t list;    /* ordinal of the list                                                  */
} dint_block_ref;

/* Host: flatten the block directories of n_lists posting lists (list i starts at byte
 * list_offsets[i] of `index`) into a block table. Replaces the pointer set-up of
 * document_enumerator's constructor (dict_posting_list.hpp:90-107). Release with dint_free. */
int dint_index_posting_lists(const uint8_t* index, size_t index_bytes, const uint64_t* list_offsets,
                             size_t n_lists, dint_block_ref** blocks, size_t* n_blocks,
                             uint64_t* total_postings);

/* A block table prepared for decoding. What depends on the table alone — the docs parts' unit table and
 * docID bases, the list of short (interpolative) blocks — is computed once, here; the workspace of a
 * decode lives in the handle too. `blocks` is the HOST table (dint_index_posting_lists); the handle keeps a
 * device copy. One decode at a time per handle (calls on one stream are ordered anyway). */
typedef struct dint_block_table dint_block_table;
int dint_block_table_create(const dint_dict* docs_dict, const dint_block_ref* blocks, size_t n_blocks,
                            size_t index_bytes, dint_block_table** out);
void dint_block_table_destroy(dint_block_table* table);

/* The sizing pass over the index, done at set-up instead of under the caller's first decodes: everything a table learns as
 * it is used — where each block's docs part ends (nothing in the index says: dict_posting_list.hpp:42-53 records the end of
 * docs + freqs only), the freqs parts' units, both bundle schedules, how many blocks they left to the unit queue — learnt
 * here by decoding the index on the device into a scratch output of the table's own (4 B x 2 per posting, released before
 * the call returns). SYNCHRONOUS (it is set-up): synchronises `stream`. After it the FIRST dint_decode_block_table of the
 * table is already the one launch. freqs_dict NULL: a table that will decode docIDs only. Optional: a table not taught
 * learns under its first two decodes, as before. The CONTENT STABILITY rule of dint_decode_block_table starts here. */
int dint_block_table_learn(dint_block_table* table, const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index,
                           size_t index_bytes, void* stream);
/* 1 when the next complete dint_decode_block_table of this table (with freqs iff with_freqs) takes the one-launch form. */
int dint_block_table_ready(const dint_block_table* table, int with_freqs);
/* What a table has learnt so far. */
typedef struct dint_block_table_info {
    uint64_t n_blocks, n_short_blocks;   /* blocks; those of fewer than 256 postings (binary-interpolative) */
    uint32_t complete_decodes;           /* decodes that covered every block (learning ones included) */
    uint32_t spans_exact;                /* 1: where every docs part ends is known */
    uint32_t freqs_units_ready;          /* 1: the freqs parts' units are built */
    uint32_t docs_schedule, freqs_schedule;        /* 1: the bundle schedule is kept and its work-item count read back */
    uint32_t docs_queue_items, freqs_queue_items;  /* full blocks that fit no tile (the unit queue's: a small second launch) */
    uint32_t short_block_tickets;        /* tickets the short blocks are dealt in inside the docs launch (0: a launch of their own) */
} dint_block_table_info;
int dint_block_table_info_get(const dint_block_table* table, dint_block_table_info* info);

/* Device: decode every block of the prepared table to docIDs (and, if d_freqs is not NULL, term
 * frequencies). ASYNCHRONOUS: enqueues on `stream` and returns — except that the ONE decode that builds a table's kept
 * schedules (the second complete decode of a table not taught by dint_block_table_learn, or the first after a decode with a
 * smaller out_capacity invalidated them) reads a 4-byte count back and synchronises `stream` (and the table's side stream)
 * once before it returns: do not capture that call into a graph; call dint_block_table_learn at set-up to have none. Full blocks go through the DINT kernels —
 * the docID prefix sums are formed in the expansion, one wave scan per block, the gaps never reach memory;
 * freq = value + 1 is added where the values are stored — blocks shorter than 256 through the
 * binary-interpolative decoder (whose code is the prefix sums already). A table learns as it is used: its first decode
 * finds where the docs parts end, its second builds the bundle schedules, and from the third on a decode is ONE launch
 * for the docs parts, the short blocks (inside it, by the waves' first lanes) and the freqs parts
 * (DINT_OPT_INDEX_PAIR, DINT_OPT_INDEX_INLINE_TAILS; plus a small launch for the few blocks that fit no tile). Before
 * that — and with those options off — the freqs launch and the short blocks' decoder run on streams the table owns,
 * beside the docs launch, forked from and joined to `stream` inside the call; to the caller everything is ordered on
 * `stream` either way (dint_set_option(DINT_OPT_INDEX_CONCURRENT, 0): one stream, one launch after the other).
 * ONE STREAM AT A TIME: a table's decodes share its launch counters (two sets, taken in turn: the one-launch decode clears
 * the set of the decode after it instead of a fill per call) — successive decodes of one table must be ordered on the GPU
 * (the same stream, or streams the caller orders); two tables over one index are independent.
 * CONTENT STABILITY: what the table learns in its first complete decode (exact byte spans, the freqs parts' units, both
 * bundle schedules — which bake in the blocks' selector bytes of a multi-dictionary index) is kept and keyed by the
 * dictionaries, the index POINTER and its size, not by the bytes: while the table lives, the index at d_index must keep
 * its contents, like the stream under a dint_unit_table. An index shard reloaded into the same buffer needs a new table.
 * Replaces: see dint_decode_posting_blocks. */
int dint_decode_block_table(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index,
                            size_t index_bytes, dint_block_table* table, uint32_t* d_docids, uint32_t* d_freqs,
                            size_t out_capacity, void* stream);

/* One-shot form of the two calls above over a DEVICE block table: prepares, decodes, synchronises
 * `stream`, releases. d_index / d_blocks / outputs are device pointers on the dictionaries' device;
 * both dictionaries must be of the same kind and live on the same device.
 * Replaces: document_enumerator::decode_docs_block / decode_freqs_block + the docid
 * accumulation of next() (dict_posting_list.hpp:111-124, 284-318), i.e. dint_block::decode /
 * opt_dint_multi_dict_block::decode (include/dint/dint_codecs.hpp:13-49, 460-510) and
 * interpolative_block::decode (include/ds2i/block_codecs.hpp:130-150), for many blocks per launch. */
int dint_decode_posting_blocks(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index,
                               size_t index_bytes, const dint_block_ref* d_blocks, size_t n_blocks,
                               uint32_t* d_docids, uint32_t* d_freqs, size_t out_capacity, void* stream);

/* ---- conjunctive queries over the in-index layout ------------------------------------------
 * Replaces: the index + and_query<false> pair of the reference's query path
 * (include/ds2i/queries.hpp:34-84, driven by src/queries.cpp:15-61 op_perftest), for a batch of
 * queries per call. The query index keeps a device copy of the block table (with its block
 * maxima packed for the block-max search of next_geq, dict_posting_list.hpp:126-147) and the
 * workspaces of the query rounds; it borrows d_index and docs_dict, which must outlive it. */
typedef struct dint_query_index dint_query_index;

/* blocks: HOST block table of ALL lists as produced by dint_index_posting_lists (lists in order,
 * each list's blocks contiguous); d_index: the index bytes on docs_dict's device. */
int dint_query_index_create(const dint_dict* docs_dict, const uint8_t* d_index, size_t index_bytes,
                            const dint_block_ref* blocks, size_t n_blocks, size_t n_lists,
                            dint_query_index** out);
void dint_query_index_destroy(dint_query_index* qi);

/* counts[q] = number of documents that contain every term of query q (duplicate terms count
 * once, queries.hpp:28-31; an empty query counts 0, :38). terms/query_offsets/counts are HOST
 * arrays: query q is terms[query_offsets[q] .. query_offsets[q+1]). A term >= n_lists is
 * DINT_ERR_ARG. The call enqueues on `stream` and returns after synchronising it: ONE launch for a single query of a
 * page or two of candidates and for a call whose queries all have at most 16 candidate pages (a workgroup per query); a
 * round per launch for a single query of a few pages; one copy in and two launches per round for a batch of larger
 * queries; a mixed call is split into its small and its other queries. Which form a call takes is moved,
 * for tests and measurements, by dint_set_option: DINT_OPT_QUERY_LEAN_PAGES, DINT_OPT_QUERY_TAIL_PAGES,
 * DINT_OPT_QUERY_FUSED_PAGES, DINT_OPT_QUERY_FUSED_COPY, DINT_OPT_QUERY_BATCH_FUSED. The batch form keeps two hashed
 * claim tables per workgroup (160 KB each workgroup, 40 MB in all, whatever the index's size), allocated at the first such call. */
int dint_and_queries(dint_query_index* qi, const uint32_t* terms, const uint64_t* query_offsets,
                     size_t n_queries, uint64_t* counts, void* stream);

/* and_query<true> (queries.hpp:72-76): the same counts, and freq_sums[q] = the sum, over the matches of query q and
 * over its (distinct) terms, of the term's frequency in the matching document — what the reference reads through
 * document_enumerator::freq() at every match. Lazy like the reference (dict_posting_list.hpp:164-169, :311-318): a
 * freqs part is decoded only for the blocks that hold a match; *freq_blocks_decoded (nullable) = how many that were. */
int dint_and_queries_freqs(dint_query_index* qi, const dint_dict* freqs_dict, const uint32_t* terms,
                           const uint64_t* query_offsets, size_t n_queries, uint64_t* counts, uint64_t* freq_sums,
                           uint64_t* freq_blocks_decoded, void* stream);

/* ---- block statistics on the device (dictionary construction, counting half) ----------------------------
 * Counts every aligned 16/8/4/2/1-gram of the given lists — multi != 0: of their whole 256-integer blocks, per block
 * context — keyed by the MurmurHash64A of its integers, as the reference's collectors do.
 * Replaces: adjusted::collect, include/dint/statistics_collectors.hpp:90-118 (context: :21-40), the per-thread
 * maps of block_statistics.hpp:82-106. The selection (filter, sort, DSF, packing: dictionary_builders.hpp:40-76) stays
 * on the host: dinth_build_dictionary_from_ngrams (include/dint_host.h) takes these entries.
 * d_gaps: device, n_ints u32 (d-gaps minus one, lists back to back); list_starts: host, n_lists + 1 offsets into d_gaps.
 * *entries (malloc'ed, release with dint_free): one per distinct (context, n-gram) — position of its first
 * occurrence in d_gaps, length, context, number of occurrences — in no particular order (see top_k below). */
typedef struct dint_ngram {
    uint64_t pos;
    uint32_t freq;
    uint8_t len;
    uint8_t ctx;
    uint16_t pad;
} dint_ngram;
int dint_count_ngrams(int device, int multi, const uint32_t* d_gaps, uint64_t n_ints, const uint64_t* list_starts,
                      uint64_t n_lists, uint32_t top_k, dint_ngram** entries, size_t* n_entries, float* kernel_ms);
/* top_k = 0: every distinct n-gram. top_k > 0 (65536 for DSF-65536-16): only those that can be among the first top_k
 * of their context in the selection's order (occurrences first) — per context the entries whose count reaches the
 * top_k-th largest count among the n-grams the reference's filter keeps (dictionary_builders.hpp:15-38), ties
 * included: the dictionary built from them is the same, the host sorts tens of thousands of entries instead of
 * tens of millions. */

/* The selection half on the device too: of `entries` (dint_count_ngrams' output; in place), the n-grams the reference's
 * filter keeps (dictionary_builders.hpp:15-38), every context's in dictionary order — most frequent first, then the
 * longer, then by their integers (block_statistics.hpp:246-276 freq_length sorter; ties made deterministic) — and of
 * those the first top_k (decreasing_static_frequencies::build, dictionary_builders.hpp:55-75: 65536). A filter kernel, one
 * rocPRIM merge sort whose comparator reads the integers in d_gaps, one scatter. *n_selected entries come back; the
 * host only packs them (dinth_pack_dictionary, include/dint_host.h). */
int dint_select_ngrams(int device, const uint32_t* d_gaps, uint64_t n_ints, uint64_t total_ints, dint_ngram* entries,
                       size_t n_entries, uint32_t top_k, size_t* n_selected);

/* Device time (ms) between the two events the library records around the decode kernel of the most
 * recent dint_decode_units on this dictionary (one event pair per in-flight launch: launches on
 * different streams do not disturb each other's); synchronises that launch. */
int dint_last_kernel_ms(const dint_dict* dict, float* ms);

/* The same for the most recent launches, oldest first: up to max_n of the last 64 (the library keeps
 * that many event pairs); *n = how many were written. Synchronises them. What bench.py reports its
 * per-launch kernel time from: the events of the timed launches themselves. */
int dint_recent_kernel_ms(const dint_dict* dict, float* ms, size_t max_n, size_t* n);
/* The shader clock the most recent decode kernel ran at, MHz: cycles counted by the launch's first wavefront
 * (s_memtime at both ends) over the kernel's duration from its event pair. (Boxes of a pool differ.) The count is read from
 * where that launch's counters live — a block table's launches keep theirs in the table: call while it exists — and is 0 for a
 * launch that recorded none (the small launch behind an in-index decode for the blocks that fit no tile). */
int dint_last_kernel_clock_mhz(const dint_dict* dict, float* mhz);

/* What a vroom stream is made of, by the dictionary's device layout (host pre-pass, like
 * dint_index_stream): codewords, exceptions, how many codewords find their integers on chip. */
typedef struct dint_stream_stats {
    uint64_t lists, ints, payload_bytes;
    uint64_t codewords;        /* dictionary codewords (runs included), exceptions not */
    uint64_t run_codewords;
    uint64_t exceptions16, exceptions32;
    uint64_t hot_codewords;    /* dictionary codewords whose integers are in the LDS image (runs included) */
    uint64_t hot_ints;         /* integers they decode to */
    uint64_t wide_blocks;      /* multi: blocks of 16-bit slots; narrow_blocks: of 8-bit slots */
    uint64_t narrow_blocks;
} dint_stream_stats;
int dint_stream_stats_get(const dint_dict* dict, const uint8_t* enc, size_t enc_bytes, dint_stream_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* DINT_HIP_H */
