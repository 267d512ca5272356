// Dictionary construction: block statistics + decreasing-static-frequency pick.
//
// Follows the reference's offline pipeline for DSF-65536-16:
//   selector (context of a 256-block)        statistics_collectors.hpp:21-40
//   adjusted::collect (aligned 16/8/4/2/1-grams) statistics_collectors.hpp:90-118
//   filter + freq_length sort                block_statistics.hpp:82-106, :246-276
//   cost filter                              dictionary_builders.hpp:15-38, :50-53
//   decreasing_static_frequencies::build     dictionary_builders.hpp:55-75
// Like the reference, n-grams are keyed by their 64-bit hash only.
//
// Differences, both outside any byte format: ties in the frequency sort are
// broken deterministically (the reference's order there depends on libstdc++'s
// unordered_map iteration and std::sort), and statistics can be collected from
// a caller-chosen subset of lists (the reference always scans the collection).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <atomic>
#include <thread>
#include <vector>

#include "constants.hpp"
#include "hash.hpp"

namespace dint {

inline uint32_t ceil_log2_u64(uint64_t x) {  // util.hpp:61-64
    if (x <= 1) return 0;
    return 64 - uint32_t(__builtin_clzll(x - 1));
}

// context of a block = ceil_log2(ceil_log2(max + 1)), 0 when max <= 1. The reference adds the 1 in 32 bits
// (statistics_collectors.hpp:23,36): a block holding 0xFFFFFFFF wraps to ceil_log2(0) = 0 and is context 0 there,
// so it is here.
inline uint32_t block_selector(uint32_t const* p, size_t n) {
    uint32_t x = 0;
    for (size_t i = 0; i != n; ++i) x = std::max(x, p[i]);
    return x > 1 ? ceil_log2_u64(ceil_log2_u64(uint32_t(x + 1u))) : 0;
}

struct ngram_stat {
    uint64_t freq = 1;  // block_type(): a new block starts at frequency 1
    std::vector<uint32_t> data;
};

class ngram_statistics {
public:
    explicit ngram_statistics(uint32_t num_contexts) : m_maps(num_contexts) {}

    // single-dictionary flavour: every aligned n-gram of the whole list
    void collect_single(uint32_t const* gaps, size_t n) {
        m_total += n;
        for (uint32_t s = 0; s != kNumTargetSizes; ++s) {
            uint32_t len = kTargetSizes[s];
            size_t blocks = n / len;
            for (size_t i = 0, pos = 0; i != blocks; ++i, pos += len) bump(m_maps[0], gaps + pos, len);
        }
    }

    // multi-dictionary flavour: only whole 256-blocks, each into its context's map
    void collect_multi(uint32_t const* gaps, size_t n) {
        m_total += n;
        size_t blocks = n / kBlockSize;
        for (size_t b = 0, pos = 0; b != blocks; ++b, pos += kBlockSize) {
            auto& map = m_maps[block_selector(gaps + pos, kBlockSize)];
            for (uint32_t s = 0; s != kNumTargetSizes; ++s) {
                uint32_t len = kTargetSizes[s];
                for (uint32_t p = 0; p != kBlockSize; p += len) bump(map, gaps + pos + p, len);
            }
        }
    }

    void merge(ngram_statistics const& other) {
        m_total += other.m_total;
        for (size_t c = 0; c != m_maps.size(); ++c) {
            for (auto const& kv : other.m_maps[c]) {
                auto it = m_maps[c].find(kv.first);
                if (it == m_maps[c].end()) {
                    m_maps[c].emplace(kv.first, kv.second);
                } else {
                    // freq == number of occurrences seen by each collector
                    it->second.freq += kv.second.freq;
                }
            }
        }
    }

    // counts made elsewhere (the device): one distinct n-gram of context c, seen freq times
    void set(uint32_t c, uint32_t const* p, uint32_t len, uint64_t freq) {
        ngram_stat st;
        st.freq = freq;
        st.data.assign(p, p + len);
        m_maps[c][hash_u32s(p, len)] = std::move(st);
    }
    void add_total(uint64_t n) { m_total += n; }

    uint64_t total_integers() const { return m_total; }

    // selected blocks of context c in dictionary order (most frequent first)
    std::vector<ngram_stat> select(uint32_t c) const {
        static const double codeword_bits = std::log2(double(kNumEntries));
        static const double initial_bpi = 3 * codeword_bits;
        const double threshold = 0.0001 / 1000;  // decreasing_static_frequencies::filter()
        std::vector<ngram_stat> picked;
        for (auto const& kv : m_maps[c]) {
            auto const& b = kv.second;
            double saving = double(uint32_t(b.freq)) * (initial_bpi * double(b.data.size()) - codeword_bits) /
                            double(m_total);
            if (saving > threshold || b.data.size() == 1) picked.push_back(b);
        }
        std::sort(picked.begin(), picked.end(), [](ngram_stat const& l, ngram_stat const& r) {
            if (l.freq != r.freq) return l.freq > r.freq;
            if (l.data.size() != r.data.size()) return l.data.size() > r.data.size();
            return l.data < r.data;  // deterministic tie-break (see header)
        });
        return picked;
    }

    uint32_t num_contexts() const { return uint32_t(m_maps.size()); }

private:
    using map_t = std::unordered_map<uint64_t, ngram_stat>;
    static void bump(map_t& map, uint32_t const* p, uint32_t len) {
        uint64_t h = hash_u32s(p, len);
        auto it = map.find(h);
        if (it != map.end()) {
            ++it->second.freq;
        } else {
            ngram_stat st;
            st.data.assign(p, p + len);
            map.emplace(h, std::move(st));
        }
    }
    std::vector<map_t> m_maps;
    uint64_t m_total = 0;
};

// decreasing_static_frequencies::build — the first min(65536, |blocks|) blocks of
// every context are appended until the builder reports full.
template <typename Builder>
void build_dsf(Builder& builder, ngram_statistics const& stats) {
    builder.init();
    for (uint32_t c = 0; c != stats.num_contexts(); ++c) {
        auto picked = stats.select(c);
        size_t n = std::min<size_t>(kNumEntries, picked.size());
        for (size_t i = 0; i != n; ++i)
            builder.append(picked[i].data.data(), uint32_t(picked[i].data.size()), c);
    }
    builder.build();
}

// block_statistics / block_multi_statistics construction (block_statistics.hpp:45-108, :201-279) over any list source,
// in parallel: list i has len_of(i) integers, gaps_of(i, scratch) returns them. One collector per worker, merged at the end
// (a count is a sum: the result does not depend on the split).
template <typename LenOf, typename GapsOf>
ngram_statistics collect_statistics(bool multi, uint64_t n_lists, LenOf&& len_of, GapsOf&& gaps_of, int threads) {
    const uint32_t contexts = multi ? kNumSelectors : 1;
    const int workers = std::max(1, threads);
    std::vector<ngram_statistics> partial(static_cast<size_t>(workers), ngram_statistics{contexts});
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> pool;
    for (int w = 0; w != workers; ++w) {
        pool.emplace_back([&, w] {
            std::vector<uint32_t> scratch;
            for (uint64_t i; (i = next.fetch_add(1)) < n_lists;) {
                const size_t n = size_t(len_of(i));
                if (n == 0) continue;  // constants::min_size = 0: n > 0 only
                uint32_t const* g = gaps_of(i, scratch);
                if (multi) partial[size_t(w)].collect_multi(g, n);
                else partial[size_t(w)].collect_single(g, n);
            }
        });
    }
    for (auto& t : pool) t.join();
    for (int w = 1; w < workers; ++w) {
        partial[0].merge(partial[size_t(w)]);
        partial[size_t(w)] = ngram_statistics(contexts);
    }
    return std::move(partial[0]);
}

inline std::string dsf_type_name() {
    return "DSF-" + std::to_string(kNumEntries) + "-" + std::to_string(kMaxEntrySize);
}

}  // namespace dint
