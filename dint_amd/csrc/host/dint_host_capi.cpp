// C ABI of the offline CPU half (see include/dint_host.h).
#include "dint_host.h"

#include <cstring>
#include <exception>
#include <string>
#include <utility>
#include <vector>

#include "dint/binary_collection.hpp"
#include "dint/dictionaries.hpp"
#include "dint/encoders.hpp"
#include "dint/posting_list.hpp"
#include "dint/statistics.hpp"
#include "dint/synthetic.hpp"
#include "dint/vroom_stream.hpp"

struct dinth_blob {
    std::vector<uint8_t> bytes;
};

namespace {

thread_local std::string g_error;

template <typename Fn>
int guarded(Fn&& fn) {
    try {
        g_error.clear();
        return fn();
    } catch (std::bad_alloc const&) {
        g_error = "out of memory";
        return DINT_ERR_NOMEM;
    } catch (std::exception const& e) {
        g_error = e.what();
        return DINT_ERR_FORMAT;
    }
}

template <typename T>
dinth_blob* blob_of(std::vector<T> const& v) {
    auto b = new dinth_blob;
    auto p = reinterpret_cast<uint8_t const*>(v.data());
    b->bytes.assign(p, p + v.size() * sizeof(T));
    return b;
}

dint::synth_params to_params(dinth_synth_params const& c) {
    dint::synth_params p;
    p.seed = c.seed;
    p.universe = c.universe;
    p.alpha = c.alpha;
    p.min_len = c.min_len;
    p.max_len = c.max_len;
    p.stay_cluster = c.stay_cluster;
    p.stay_sparse = c.stay_sparse;
    p.p_cluster_min = c.p_cluster_min;
    p.p_cluster_max = c.p_cluster_max;
    return p;
}

template <typename Builder>
int build_dictionary(bool multi, uint32_t const* gaps, uint32_t const* lens, uint64_t n_lists,
                     uint64_t max_sample_ints, int threads, dinth_blob** out) {
    using namespace dint;
    // sample = prefix of the collection
    uint64_t n_sample_lists = 0, ints = 0;
    for (; n_sample_lists != n_lists; ++n_sample_lists) {
        if (max_sample_ints && n_sample_lists && ints + lens[n_sample_lists] > max_sample_ints) break;
        ints += lens[n_sample_lists];
    }
    std::vector<uint64_t> starts(n_sample_lists + 1, 0);
    for (uint64_t i = 0; i != n_sample_lists; ++i) starts[i + 1] = starts[i] + lens[i];
    auto stats = collect_statistics(
        multi, n_sample_lists, [&](uint64_t i) { return lens[i]; },
        [&](uint64_t i, std::vector<uint32_t>&) { return gaps + starts[i]; }, threads);
    Builder builder;
    build_dsf(builder, stats);
    auto b = new dinth_blob;
    builder.write(b->bytes);
    *out = b;
    return DINT_OK;
}

template <typename Builder>
int dictionary_from_ngrams(bool multi, uint32_t const* gaps, uint64_t n_ints, uint64_t total_ints,
                                  dinth_ngram const* entries, uint64_t n_entries, dinth_blob** out) {
    using namespace dint;
    ngram_statistics stats(multi ? kNumSelectors : 1);
    stats.add_total(total_ints);
    for (uint64_t i = 0; i != n_entries; ++i) {
        dinth_ngram const& e = entries[i];
        if (e.len == 0 || e.len > kMaxEntrySize || (e.len & (e.len - 1)) != 0 || e.pos > n_ints || n_ints - e.pos < e.len ||
            e.ctx >= stats.num_contexts() || e.freq == 0)
            return int(DINT_ERR_ARG);
        stats.set(e.ctx, gaps + e.pos, e.len, e.freq);
    }
    Builder builder;
    build_dsf(builder, stats);
    auto b = new dinth_blob;
    builder.write(b->bytes);
    *out = b;
    return DINT_OK;
}

template <typename Builder>
int pack_dictionary(bool multi, uint32_t const* gaps, uint64_t n_ints, dinth_ngram const* entries, uint64_t n_entries, dinth_blob** out) {
    using namespace dint;
    const uint32_t contexts = multi ? kNumSelectors : 1;
    Builder builder;
    builder.init();
    uint32_t prev_ctx = 0;
    for (uint64_t i = 0; i != n_entries; ++i) {
        dinth_ngram const& e = entries[i];
        if (e.len == 0 || e.len > kMaxEntrySize || (e.len & (e.len - 1)) != 0 || e.pos > n_ints || n_ints - e.pos < e.len ||
            e.ctx >= contexts || e.ctx < prev_ctx)
            return int(DINT_ERR_ARG);
        prev_ctx = e.ctx;
        builder.append(gaps + e.pos, e.len, e.ctx);
    }
    builder.build();
    auto b = new dinth_blob;
    builder.write(b->bytes);
    *out = b;
    return DINT_OK;
}

template <typename Encoder, typename Builder>
int encode_with(void const* dict_file, size_t dict_len, uint32_t const* gaps, uint32_t const* lens,
                uint64_t n_lists, uint32_t unit_ints, int threads, dinth_blob** enc, dinth_blob** units) {
    Builder builder;
    builder.load(static_cast<uint8_t const*>(dict_file), dict_len);
    builder.prepare_for_encoding();
    auto out = dint::encode_vroom<Encoder>(builder, gaps, lens, n_lists, unit_ints, threads);
    auto e = new dinth_blob;
    e->bytes.swap(out.bytes);
    *enc = e;
    if (units) *units = blob_of(out.units);
    return DINT_OK;
}

}  // namespace

namespace {
// list i: n = len_of(i) postings, lists_of(i) -> (docIDs, freqs)
template <typename Coder, typename Builder, typename LenOf, typename ListsOf>
int build_index_lists(const void* docs_dict, size_t docs_len, const void* freqs_dict, size_t freqs_len, uint64_t n_lists,
                      LenOf&& len_of, ListsOf&& lists_of, int threads, dinth_blob** index, dinth_blob** offsets) {
    Builder docs_builder, freqs_builder;
    docs_builder.load(static_cast<uint8_t const*>(docs_dict), docs_len);
    freqs_builder.load(static_cast<uint8_t const*>(freqs_dict), freqs_len);
    docs_builder.prepare_for_encoding();
    freqs_builder.prepare_for_encoding();
    std::vector<std::vector<uint8_t>> encoded(n_lists);
    const uint64_t group = 16;
    dint::parallel_for((n_lists + group - 1) / group, threads, [&](size_t g) {
        uint64_t end = std::min<uint64_t>(n_lists, (g + 1) * group);
        for (uint64_t i = g * group; i != end; ++i) {
            const uint32_t n = uint32_t(len_of(i));
            if (n == 0) continue;
            auto lists = lists_of(i);
            dint::write_posting_list<Coder>(docs_builder, freqs_builder, encoded[i], n, lists.first, lists.second);
        }
    });
    auto idx = new dinth_blob;
    std::vector<uint64_t> offs(n_lists + 1, 0);
    for (uint64_t i = 0; i != n_lists; ++i) {
        offs[i] = idx->bytes.size();
        idx->bytes.insert(idx->bytes.end(), encoded[i].begin(), encoded[i].end());
        std::vector<uint8_t>().swap(encoded[i]);
    }
    offs[n_lists] = idx->bytes.size();
    *index = idx;
    *offsets = blob_of(offs);
    return DINT_OK;
}

// the Coder / Builder pair of a dictionary kind (reference include/index_types.hpp:73-79; greedy: dint_codecs.hpp:52-139)
template <typename LenOf, typename ListsOf>
int build_index_kind(int kind, int greedy, const void* docs_dict, size_t docs_len, const void* freqs_dict, size_t freqs_len,
                     uint64_t n_lists, LenOf&& len_of, ListsOf&& lists_of, int threads, dinth_blob** index, dinth_blob** offsets) {
    using namespace dint;
    switch (kind) {
        case DINT_DICT_RECTANGULAR:
            return greedy ? build_index_lists<greedy_dint_single_dict_block, rectangular_builder>(
                                docs_dict, docs_len, freqs_dict, freqs_len, n_lists, len_of, lists_of, threads, index, offsets)
                          : build_index_lists<opt_dint_single_dict_block, rectangular_builder>(
                                docs_dict, docs_len, freqs_dict, freqs_len, n_lists, len_of, lists_of, threads, index, offsets);
        case DINT_DICT_SINGLE_PACKED:
            return greedy ? build_index_lists<greedy_dint_single_dict_block, single_packed_builder>(
                                docs_dict, docs_len, freqs_dict, freqs_len, n_lists, len_of, lists_of, threads, index, offsets)
                          : build_index_lists<opt_dint_single_dict_block, single_packed_builder>(
                                docs_dict, docs_len, freqs_dict, freqs_len, n_lists, len_of, lists_of, threads, index, offsets);
        case DINT_DICT_MULTI_PACKED:
            return build_index_lists<opt_dint_multi_dict_block, multi_packed_builder>(
                docs_dict, docs_len, freqs_dict, freqs_len, n_lists, len_of, lists_of, threads, index, offsets);
        default:
            return int(DINT_ERR_ARG);
    }
}

template <typename Encoder, typename Builder>
int encode_collection_with(void const* dict_file, size_t dict_len, dint::binary_collection const& input, bool docs,
                           uint32_t unit_ints, int threads, dinth_blob** enc, dinth_blob** units, uint64_t* n_lists,
                           uint64_t* n_ints) {
    Builder builder;
    builder.load(static_cast<uint8_t const*>(dict_file), dict_len);
    builder.prepare_for_encoding();
    auto out = dint::encode_vroom_collection<Encoder>(builder, input, docs, unit_ints, threads);
    if (n_ints) *n_ints = out.total_ints;
    if (n_lists) {
        auto lists = input.sequences();
        *n_lists = lists.size() - ((docs && !lists.empty()) ? 1 : 0);
    }
    auto e = new dinth_blob;
    e->bytes.swap(out.bytes);
    *enc = e;
    if (units) *units = blob_of(out.units);
    return DINT_OK;
}

template <typename Builder>
int build_dictionary_collection(bool multi, dint::binary_collection const& input, bool docs, uint64_t max_sample_ints,
                                int threads, dinth_blob** out) {
    using namespace dint;
    auto lists = input.sequences();
    if (docs && !lists.empty()) lists.erase(lists.begin());  // the number of documents (block_statistics.hpp:57-60)
    uint64_t n_sample = 0, ints = 0;
    for (; n_sample != lists.size(); ++n_sample) {
        if (max_sample_ints && n_sample && ints + lists[n_sample].size() > max_sample_ints) break;
        ints += lists[n_sample].size();
    }
    auto stats = collect_statistics(
        multi, n_sample, [&](uint64_t i) { return lists[i].size(); },
        [&](uint64_t i, std::vector<uint32_t>& scratch) {
            scratch.resize(lists[i].size());
            list_to_gaps(lists[i], docs, scratch.data());
            return static_cast<uint32_t const*>(scratch.data());
        },
        threads);
    Builder builder;
    build_dsf(builder, stats);
    auto b = new dinth_blob;
    builder.write(b->bytes);
    *out = b;
    return DINT_OK;
}
}  // namespace

extern "C" {

const void* dinth_blob_data(const dinth_blob* b) { return b ? b->bytes.data() : nullptr; }
size_t dinth_blob_size(const dinth_blob* b) { return b ? b->bytes.size() : 0; }
void dinth_blob_free(dinth_blob* b) { delete b; }
const char* dinth_last_error(void) { return g_error.c_str(); }

void dinth_synth_defaults(dinth_synth_params* c) {
    if (!c) return;
    dint::synth_params p;
    c->seed = p.seed;
    c->universe = p.universe;
    c->min_len = p.min_len;
    c->max_len = p.max_len;
    c->reserved = 0;
    c->alpha = p.alpha;
    c->stay_cluster = p.stay_cluster;
    c->stay_sparse = p.stay_sparse;
    c->p_cluster_min = p.p_cluster_min;
    c->p_cluster_max = p.p_cluster_max;
}

int dinth_synth_lengths(const dinth_synth_params* p, uint64_t target_postings, dinth_blob** lens) {
    if (!p || !lens || p->universe == 0) return DINT_ERR_ARG;
    return guarded([&] {
        *lens = blob_of(dint::synth_lengths(to_params(*p), target_postings));
        return DINT_OK;
    });
}

int dinth_synth_gaps(const dinth_synth_params* p, const uint32_t* lens, uint64_t n_lists, uint64_t first_list_id,
                     uint32_t* gaps_out, int threads) {
    if (!p || (!lens && n_lists) || (!gaps_out && n_lists) || p->universe == 0) return DINT_ERR_ARG;
    return guarded([&] {
        auto params = to_params(*p);
        std::vector<uint64_t> starts(n_lists + 1, 0);
        for (uint64_t i = 0; i != n_lists; ++i) starts[i + 1] = starts[i] + lens[i];
        const uint64_t group = 256;
        dint::parallel_for((n_lists + group - 1) / group, threads, [&](size_t g) {
            uint64_t end = std::min<uint64_t>(n_lists, (g + 1) * group);
            for (uint64_t i = g * group; i != end; ++i)
                dint::synth_gaps(params, first_list_id + i, lens[i], gaps_out + starts[i]);
        });
        return DINT_OK;
    });
}

int dinth_build_dictionary(int kind, const uint32_t* gaps, const uint32_t* lens, uint64_t n_lists,
                           uint64_t max_sample_ints, int threads, dinth_blob** dict_file) {
    if (!dict_file || (n_lists && (!gaps || !lens))) return DINT_ERR_ARG;
    return guarded([&] {
        switch (kind) {
            case DINT_DICT_RECTANGULAR:
                return build_dictionary<dint::rectangular_builder>(false, gaps, lens, n_lists, max_sample_ints,
                                                                   threads, dict_file);
            case DINT_DICT_SINGLE_PACKED:
                return build_dictionary<dint::single_packed_builder>(false, gaps, lens, n_lists, max_sample_ints,
                                                                     threads, dict_file);
            case DINT_DICT_MULTI_PACKED:
                return build_dictionary<dint::multi_packed_builder>(true, gaps, lens, n_lists, max_sample_ints,
                                                                    threads, dict_file);
            default:
                return int(DINT_ERR_ARG);
        }
    });
}

int dinth_build_dictionary_from_ngrams(int kind, const uint32_t* gaps, uint64_t n_ints, uint64_t total_ints,
                                       const dinth_ngram* entries, uint64_t n_entries, dinth_blob** dict_file) {
    if (!dict_file || (n_entries && (!gaps || !entries))) return DINT_ERR_ARG;
    return guarded([&] {
        switch (kind) {
            case DINT_DICT_RECTANGULAR:
                return dictionary_from_ngrams<dint::rectangular_builder>(false, gaps, n_ints, total_ints, entries, n_entries, dict_file);
            case DINT_DICT_SINGLE_PACKED:
                return dictionary_from_ngrams<dint::single_packed_builder>(false, gaps, n_ints, total_ints, entries, n_entries, dict_file);
            case DINT_DICT_MULTI_PACKED:
                return dictionary_from_ngrams<dint::multi_packed_builder>(true, gaps, n_ints, total_ints, entries, n_entries, dict_file);
            default:
                return int(DINT_ERR_ARG);
        }
    });
}

int dinth_pack_dictionary(int kind, const uint32_t* gaps, uint64_t n_ints, const dinth_ngram* entries, uint64_t n_entries,
                          dinth_blob** dict_file) {
    if (!dict_file || (n_entries && (!gaps || !entries))) return DINT_ERR_ARG;
    return guarded([&] {
        switch (kind) {
            case DINT_DICT_RECTANGULAR: return pack_dictionary<dint::rectangular_builder>(false, gaps, n_ints, entries, n_entries, dict_file);
            case DINT_DICT_SINGLE_PACKED: return pack_dictionary<dint::single_packed_builder>(false, gaps, n_ints, entries, n_entries, dict_file);
            case DINT_DICT_MULTI_PACKED: return pack_dictionary<dint::multi_packed_builder>(true, gaps, n_ints, entries, n_entries, dict_file);
            default: return int(DINT_ERR_ARG);
        }
    });
}

int dinth_encode_vroom(int kind, int greedy, const void* dict_file, size_t dict_len, const uint32_t* gaps,
                       const uint32_t* lens, uint64_t n_lists, uint32_t unit_ints, int threads, dinth_blob** enc,
                       dinth_blob** units) {
    if (!dict_file || !enc || (n_lists && (!gaps || !lens))) return DINT_ERR_ARG;
    return guarded([&] {
        using namespace dint;
        switch (kind) {
            case DINT_DICT_RECTANGULAR:
                return greedy ? encode_with<single_greedy_dint, rectangular_builder>(dict_file, dict_len, gaps, lens,
                                                                                     n_lists, unit_ints, threads, enc,
                                                                                     units)
                              : encode_with<single_opt_dint, rectangular_builder>(dict_file, dict_len, gaps, lens,
                                                                                  n_lists, unit_ints, threads, enc,
                                                                                  units);
            case DINT_DICT_SINGLE_PACKED:
                return greedy ? encode_with<single_greedy_dint, single_packed_builder>(dict_file, dict_len, gaps, lens,
                                                                                       n_lists, unit_ints, threads,
                                                                                       enc, units)
                              : encode_with<single_opt_dint, single_packed_builder>(dict_file, dict_len, gaps, lens,
                                                                                    n_lists, unit_ints, threads, enc,
                                                                                    units);
            case DINT_DICT_MULTI_PACKED:
                return encode_with<multi_opt_dint, multi_packed_builder>(dict_file, dict_len, gaps, lens, n_lists,
                                                                         unit_ints, threads, enc, units);
            default:
                return int(DINT_ERR_ARG);
        }
    });
}

int dinth_build_index_coder(int kind, int greedy, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                            size_t freqs_dict_len, const uint32_t* docids, const uint32_t* freqs, const uint32_t* lens,
                            uint64_t n_lists, int threads, dinth_blob** index, dinth_blob** offsets) {
    if (!docs_dict_file || !freqs_dict_file || !index || !offsets || (n_lists && (!docids || !freqs || !lens)))
        return DINT_ERR_ARG;
    return guarded([&] {
        std::vector<uint64_t> starts(n_lists + 1, 0);
        for (uint64_t i = 0; i != n_lists; ++i) starts[i + 1] = starts[i] + lens[i];
        return build_index_kind(
            kind, greedy, docs_dict_file, docs_dict_len, freqs_dict_file, freqs_dict_len, n_lists,
            [&](uint64_t i) { return lens[i]; },
            [&](uint64_t i) { return std::make_pair(docids + starts[i], freqs + starts[i]); }, threads, index, offsets);
    });
}

int dinth_build_index(int kind, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                      size_t freqs_dict_len, const uint32_t* docids, const uint32_t* freqs, const uint32_t* lens,
                      uint64_t n_lists, int threads, dinth_blob** index, dinth_blob** offsets) {
    return dinth_build_index_coder(kind, 0, docs_dict_file, docs_dict_len, freqs_dict_file, freqs_dict_len, docids, freqs, lens,
                                   n_lists, threads, index, offsets);
}

int dinth_encode_collection(int kind, int greedy, const void* dict_file, size_t dict_len, const uint32_t* words, size_t n_words,
                            int docs, uint32_t unit_ints, int threads, dinth_blob** enc, dinth_blob** units, uint64_t* n_lists,
                            uint64_t* n_ints) {
    if (!dict_file || !enc || (n_words && !words)) return DINT_ERR_ARG;
    return guarded([&] {
        using namespace dint;
        binary_collection input(words, n_words);
        switch (kind) {
            case DINT_DICT_RECTANGULAR:
                return greedy ? encode_collection_with<single_greedy_dint, rectangular_builder>(dict_file, dict_len, input, docs != 0,
                                                                                                unit_ints, threads, enc, units, n_lists, n_ints)
                              : encode_collection_with<single_opt_dint, rectangular_builder>(dict_file, dict_len, input, docs != 0,
                                                                                             unit_ints, threads, enc, units, n_lists, n_ints);
            case DINT_DICT_SINGLE_PACKED:
                return greedy ? encode_collection_with<single_greedy_dint, single_packed_builder>(dict_file, dict_len, input, docs != 0,
                                                                                                  unit_ints, threads, enc, units, n_lists, n_ints)
                              : encode_collection_with<single_opt_dint, single_packed_builder>(dict_file, dict_len, input, docs != 0,
                                                                                               unit_ints, threads, enc, units, n_lists, n_ints);
            case DINT_DICT_MULTI_PACKED:
                return encode_collection_with<multi_opt_dint, multi_packed_builder>(dict_file, dict_len, input, docs != 0, unit_ints,
                                                                                    threads, enc, units, n_lists, n_ints);
            default:
                return int(DINT_ERR_ARG);
        }
    });
}

int dinth_build_dictionary_collection(int kind, const uint32_t* words, size_t n_words, int docs, uint64_t max_sample_ints,
                                      int threads, dinth_blob** dict_file) {
    if (!dict_file || (n_words && !words)) return DINT_ERR_ARG;
    return guarded([&] {
        dint::binary_collection input(words, n_words);
        switch (kind) {
            case DINT_DICT_RECTANGULAR:
                return build_dictionary_collection<dint::rectangular_builder>(false, input, docs != 0, max_sample_ints, threads, dict_file);
            case DINT_DICT_SINGLE_PACKED:
                return build_dictionary_collection<dint::single_packed_builder>(false, input, docs != 0, max_sample_ints, threads, dict_file);
            case DINT_DICT_MULTI_PACKED:
                return build_dictionary_collection<dint::multi_packed_builder>(true, input, docs != 0, max_sample_ints, threads, dict_file);
            default:
                return int(DINT_ERR_ARG);
        }
    });
}

int dinth_build_index_collection(int kind, int greedy, const void* docs_dict_file, size_t docs_dict_len, const void* freqs_dict_file,
                                 size_t freqs_dict_len, const uint32_t* docs_words, size_t n_docs_words, const uint32_t* freqs_words,
                                 size_t n_freqs_words, int threads, dinth_blob** index, dinth_blob** offsets, uint64_t* num_docs) {
    if (!docs_dict_file || !freqs_dict_file || !index || !offsets || !docs_words || !freqs_words) return DINT_ERR_ARG;
    return guarded([&] {
        dint::binary_collection docs(docs_words, n_docs_words), freqs(freqs_words, n_freqs_words);
        auto d = docs.sequences();
        auto f = freqs.sequences();
        // binary_freq_collection (include/ds2i/binary_freq_collection.hpp:14-23)
        if (d.empty() || d.front().size() != 1)
            throw std::invalid_argument("First sequence should only contain number of documents");
        if (num_docs) *num_docs = *d.front().begin();
        d.erase(d.begin());
        if (d.size() != f.size()) throw std::runtime_error("docs and freqs files do not match");
        for (size_t i = 0; i != d.size(); ++i)
            if (d[i].size() != f[i].size()) throw std::runtime_error("docs and freqs files do not match");
        return build_index_kind(
            kind, greedy, docs_dict_file, docs_dict_len, freqs_dict_file, freqs_dict_len, d.size(),
            [&](uint64_t i) { return d[i].size(); }, [&](uint64_t i) { return std::make_pair(d[i].begin(), f[i].begin()); },
            threads, index, offsets);
    });
}

uint64_t dinth_hash_u32s(const uint32_t* p, size_t n) { return dint::hash_u32s(p, n); }

int dinth_constants(uint32_t* out, int cap) {
    const uint32_t v[] = {dint::kExceptions, dint::kNumSelectors, dint::kMaxEntrySize, dint::kNumEntries, dint::kNumTargetSizes,
                          dint::kTargetSizes[0], dint::kTargetSizes[1], dint::kTargetSizes[2], dint::kTargetSizes[3],
                          dint::kTargetSizes[4], dint::kBlockSize, dint::kReserved};
    const int n = int(sizeof(v) / sizeof(v[0]));
    for (int i = 0; out && i != n && i < cap; ++i) out[i] = v[i];
    return n;
}

uint32_t dinth_block_selector(const uint32_t* p, size_t n) { return p ? dint::block_selector(p, n) : 0; }

int dinth_dict_num_entries(int kind, const void* dict_file, size_t dict_len, uint32_t d, uint32_t* n_out) {
    if (!dict_file || !n_out) return DINT_ERR_ARG;
    return guarded([&] {
        auto bytes = static_cast<uint8_t const*>(dict_file);
        if (kind == DINT_DICT_RECTANGULAR) {
            dint::rectangular_builder b;
            b.load(bytes, dict_len);
            *n_out = b.size();
        } else if (kind == DINT_DICT_SINGLE_PACKED) {
            dint::single_packed_builder b;
            b.load(bytes, dict_len);
            *n_out = b.size();
        } else if (kind == DINT_DICT_MULTI_PACKED) {
            dint::multi_packed_builder b;
            b.load(bytes, dict_len);
            if (d >= dint::kNumSelectors) return int(DINT_ERR_ARG);
            *n_out = b.slots(d);
        } else {
            return int(DINT_ERR_ARG);
        }
        return int(DINT_OK);
    });
}

int dinth_dict_entry(int kind, const void* dict_file, size_t dict_len, uint32_t d, uint32_t index, uint32_t* size_out,
                     uint32_t* words16_out) {
    if (!dict_file || !size_out || !words16_out) return DINT_ERR_ARG;
    return guarded([&] {
        auto bytes = static_cast<uint8_t const*>(dict_file);
        uint32_t size = 0;
        uint32_t const* p = nullptr;
        uint32_t avail = 0;
        dint::rectangular_builder rb;
        dint::single_packed_builder sb;
        dint::multi_packed_builder mb;
        if (kind == DINT_DICT_RECTANGULAR) {
            rb.load(bytes, dict_len);
            if (index >= dint::kNumEntries) return int(DINT_ERR_ARG);
            size = rb.size(index);
            p = rb.get(index);
            avail = dint::kMaxEntrySize;
        } else if (kind == DINT_DICT_SINGLE_PACKED) {
            sb.load(bytes, dict_len);
            if (index >= sb.offsets().size()) return int(DINT_ERR_ARG);
            size = sb.size(index);
            p = sb.get(index);
            avail = uint32_t(sb.table().size() - sb.offset(index));
        } else if (kind == DINT_DICT_MULTI_PACKED) {
            mb.load(bytes, dict_len);
            if (d >= dint::kNumSelectors || index >= mb.slots(d)) return int(DINT_ERR_ARG);
            size = mb.size(d, index);
            p = mb.get(d, index);
            avail = uint32_t(mb.table().size() - mb.offset(d, index));
        } else {
            return int(DINT_ERR_ARG);
        }
        *size_out = size;
        for (uint32_t k = 0; k != dint::kMaxEntrySize; ++k) words16_out[k] = k < avail ? p[k] : 0;
        return int(DINT_OK);
    });
}

}  // extern "C"
