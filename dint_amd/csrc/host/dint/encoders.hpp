// Whole-list DINT encoders of the vroom environment (CPU, offline).
//
//   single_opt_dint::encode    optimal parse (shortest path over positions)
//                              reference vroom_env/dint_codecs.hpp:192-305
//   single_greedy_dint::encode longest-match-first
//                              reference vroom_env/dint_codecs.hpp:110-171
//   multi_opt_dint::encode     per 256-block: best of 6 contexts x {16,8}-bit
//                              reference vroom_env/dint_codecs.hpp:334-518
//
// Edge costs of the parse: 1 per dictionary/run codeword, 2 for a 16-bit
// exception, 3 for a 32-bit one; relaxation order runs first, then sizes
// 16,8,4,2,1, strict improvement only — so ties resolve exactly as the
// reference's do and the emitted bytes are the same for the same dictionary.
//
// Every encoder can also report `sync points` — (byte offset, integer offset)
// pairs at codeword boundaries — which is what the decoder's unit table is
// built from (the stream itself carries none).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "constants.hpp"
#include "dictionaries.hpp"
#include "statistics.hpp"

namespace dint {

struct sync_point {
    uint64_t byte_off;  // relative to the first payload byte of the list
    uint64_t int_off;   // integers decoded before this point
};

namespace detail {

struct parse_node {
    uint32_t parent;
    uint32_t codeword;
    uint32_t cost;
};

// zr[i] = min(256, number of consecutive zeros starting at i)
inline void zero_runs(uint32_t const* in, size_t n, std::vector<uint16_t>& zr) {
    zr.assign(n + 1, 0);
    for (size_t i = n; i-- != 0;) zr[i] = in[i] == 0 ? uint16_t(std::min<uint32_t>(256, zr[i + 1] + 1u)) : 0;
}

inline void emit_codeword(uint32_t index, int b, std::vector<uint8_t>& out) {
    out.push_back(uint8_t(index & 0xFF));
    if (b == 16) out.push_back(uint8_t((index >> 8) & 0xFF));
}

inline void emit_exception(uint32_t value, bool large, int b, std::vector<uint8_t>& out) {
    out.push_back(large ? 1 : 0);
    if (b == 16) out.push_back(0);
    out.push_back(uint8_t(value));
    out.push_back(uint8_t(value >> 8));
    if (large) {
        out.push_back(uint8_t(value >> 16));
        out.push_back(uint8_t(value >> 24));
    }
}

// Shortest-path parse of in[0, n) against `lookup(ptr, len) -> index`.
// Appends the codeword stream to `out`; if `syncs` is given, records a sync
// point at the first codeword boundary at or after every multiple of
// `sync_every` integers (excluding 0 and n). `byte_base`/`int_base` shift the
// recorded offsets.
template <typename Lookup>
void optimal_parse(Lookup&& lookup, uint32_t const* in, size_t n, int b, std::vector<uint8_t>& out,
                   std::vector<sync_point>* syncs = nullptr, uint32_t sync_every = 0,
                   uint64_t byte_base = 0, uint64_t int_base = 0) {
    if (n == 0) return;
    std::vector<parse_node> path(n + 1);
    path[0] = {0, 1, 0};
    for (size_t i = 1; i <= n; ++i) path[i] = {uint32_t(i - 1), 1, uint32_t(3 * i)};
    std::vector<uint16_t> zr;
    zero_runs(in, n, zr);

    for (size_t i = 0; i != n; ++i) {
        uint32_t base_cost = path[i].cost;
        uint32_t run = zr[i];
        if (run >= 16) {
            uint32_t k = 256, index = kExceptions;
            while (run < k && k > 16) {
                k /= 2;
                ++index;
            }
            for (; k >= 16; k /= 2, ++index) {
                if (path[i + k].cost > base_cost + 1) path[i + k] = {uint32_t(i), index, base_cost + 1};
            }
        }
        for (uint32_t s = 0; s != kNumTargetSizes; ++s) {
            uint32_t target = kTargetSizes[s];
            uint32_t len = uint32_t(std::min<size_t>(target, n - i));
            uint32_t index = lookup(in + i, len);
            if (index != kInvalidIndex) {
                if (path[i + len].cost > base_cost + 1) path[i + len] = {uint32_t(i), index, base_cost + 1};
            } else if (target == 1) {
                bool large = in[i] > 65535;
                uint32_t c = base_cost + (large ? 3 : 2);
                if (path[i + 1].cost > c) path[i + 1] = {uint32_t(i), large ? 1u : 0u, c};
            }
        }
    }

    std::vector<uint32_t> cuts;  // positions where codewords start, back to front
    for (size_t i = n; i != 0; i = path[i].parent) cuts.push_back(uint32_t(i));
    std::reverse(cuts.begin(), cuts.end());  // cuts[k] = end position of codeword k

    size_t out_start = out.size();
    uint64_t next_sync = sync_every ? sync_every : ~uint64_t(0);
    uint32_t pos = 0;
    for (uint32_t end : cuts) {
        if (syncs && pos >= next_sync) {
            syncs->push_back({byte_base + (out.size() - out_start), int_base + pos});
            next_sync = (uint64_t(pos) / sync_every + 1) * sync_every;
        }
        uint32_t index = path[end].codeword;
        if (index >= kExceptions) {
            emit_codeword(index, b, out);
        } else {
            emit_exception(in[pos], index == 1, b, out);
        }
        pos = end;
    }
}

// Longest-match-first parse (16-bit codewords only).
template <typename Lookup>
void greedy_parse(Lookup&& lookup, uint32_t const* in, size_t n, std::vector<uint8_t>& out) {
    size_t i = 0;
    while (i < n) {
        size_t limit = std::min<size_t>(256, n - i);
        uint32_t run = 0;
        while (run < limit && in[i + run] == 0) ++run;
        if (run >= 16) {
            uint32_t k = 256, index = kExceptions;
            while (run < k && k > 16) {
                ++index;
                k /= 2;
            }
            emit_codeword(index, 16, out);
            i += k;
            continue;
        }
        uint32_t index = kInvalidIndex;
        for (uint32_t s = 0; s != kNumTargetSizes; ++s) {
            uint32_t len = uint32_t(std::min<size_t>(kTargetSizes[s], n - i));
            index = lookup(in + i, len);
            if (index != kInvalidIndex) {
                emit_codeword(index, 16, out);
                i += len;
                break;
            }
        }
        if (index == kInvalidIndex) {
            emit_exception(in[i], in[i] >= 65536, 16, out);
            i += 1;
        }
    }
}

}  // namespace detail

struct single_opt_dint {
    static constexpr char const* name = "single_opt_dint";
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t /*universe*/, uint32_t n,
                       std::vector<uint8_t>& out, std::vector<sync_point>* syncs = nullptr,
                       uint32_t sync_every = 0) {
        detail::optimal_parse([&](uint32_t const* p, uint32_t len) { return builder.lookup(p, len); }, in, n, 16,
                              out, syncs, sync_every);
    }
};

struct single_greedy_dint {
    static constexpr char const* name = "single_greedy_dint";
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t /*universe*/, uint32_t n,
                       std::vector<uint8_t>& out, std::vector<sync_point>* = nullptr, uint32_t = 0) {
        detail::greedy_parse([&](uint32_t const* p, uint32_t len) { return builder.lookup(p, len); }, in, n, out);
    }
};

struct multi_opt_dint {
    static constexpr char const* name = "multi_opt_dint";
    // A sync point is recorded at every block boundary that is a multiple of
    // `sync_every` integers (rounded up to whole blocks).
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t /*universe*/, uint32_t n,
                       std::vector<uint8_t>& out, std::vector<sync_point>* syncs = nullptr,
                       uint32_t sync_every = 0) {
        size_t out_start = out.size();
        uint32_t blocks_per_sync = sync_every ? std::max<uint32_t>(1, (sync_every + kBlockSize - 1) / kBlockSize) : 0;
        std::vector<uint8_t> cand, best;
        for (uint32_t pos = 0, b = 0; pos < n; pos += kBlockSize, ++b) {
            uint32_t size = std::min<uint32_t>(kBlockSize, n - pos);
            if (syncs && b != 0 && b % blocks_per_sync == 0)
                syncs->push_back({uint64_t(out.size() - out_start), pos});
            size_t best_size = size_t(-1);
            uint32_t selector_code = 0;
            best.clear();
            for (uint32_t s = 0; s != kNumSelectors; ++s) {
                std::vector<uint8_t> wide, narrow;
                detail::optimal_parse(
                    [&](uint32_t const* p, uint32_t len) { return builder.lookup(s, p, len, 16); }, in + pos, size,
                    16, wide);
                detail::optimal_parse(
                    [&](uint32_t const* p, uint32_t len) { return builder.lookup(s, p, len, 8); }, in + pos, size,
                    8, narrow);
                bool take_narrow = narrow.size() <= wide.size();
                auto& smallest = take_narrow ? narrow : wide;
                if (smallest.size() < best_size) {
                    best_size = smallest.size();
                    selector_code = s + (take_narrow ? kNumSelectors : 0);
                    best.swap(smallest);
                }
            }
            out.push_back(uint8_t(selector_code));
            out.insert(out.end(), best.begin(), best.end());
        }
    }
};

}  // namespace dint
