// In-index posting lists: binary interpolative tails and the dict_posting_list layout (CPU, build side).
//
//   bit_writer::write / write_int / write_interpolative   reference include/ds2i/interpolative_coding.hpp:10-77
//   interpolative_block::encode                            reference include/ds2i/block_codecs.hpp:104-128
//   opt_dint_single_dict_block / greedy_dint_single_dict_block / opt_dint_multi_dict_block::encode (blocks of 256: the same optimal parse as
//     the whole-list coders; shorter blocks: interpolative)  reference include/dint/dint_codecs.hpp:145-267, 289-458
//   dict_posting_list::write                               reference include/dint/dict_posting_list.hpp:10-56
//
// List layout:  vbyte(n) | u32 block_max[B] | u32 block_endpoint[B-1] | { docs_block freqs_block } x B,
// B = ceil(n / 256). docs blocks hold d-gaps minus one with sum_of_values = max - base - (size - 1);
// freqs blocks hold freq - 1 with sum_of_values = 0xFFFFFFFF (the sum is then vbyte-coded in front of an
// interpolative block).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "constants.hpp"
#include "encoders.hpp"
#include "vbyte.hpp"

namespace dint {

class bit_writer {
public:
    explicit bit_writer(std::vector<uint32_t>& buf) : m_buf(buf) { m_buf.clear(); }

    void write(uint32_t bits, uint32_t len) {
        if (!len) return;
        uint32_t pos_in_word = uint32_t(m_size % 32);
        m_size += len;
        if (pos_in_word == 0) {
            m_buf.push_back(bits);
        } else {
            m_buf.back() |= bits << pos_in_word;
            if (len > 32 - pos_in_word) m_buf.push_back(bits >> (32 - pos_in_word));
        }
    }
    // minimal binary code of val in [0, u)
    void write_int(uint32_t val, uint32_t u) {
        uint32_t b = 31 - uint32_t(__builtin_clz(u));  // msb(u)
        uint64_t m = (uint64_t(1) << (b + 1)) - u;
        if (val < m) {
            write(val, b);
        } else {
            val += uint32_t(m);
            write(val >> 1, b);  // little-endian bit order: the writes are split
            write(val & 1, 1);
        }
    }
    void write_interpolative(uint32_t const* in, size_t n, uint32_t low, uint32_t high) {
        if (!n) return;
        size_t h = n / 2;
        uint32_t val = in[h];
        write_int(val - low, high - low + 1);
        write_interpolative(in, h, low, val);
        write_interpolative(in + h + 1, n - h - 1, val, high);
    }
    size_t size() const { return m_size; }

private:
    std::vector<uint32_t>& m_buf;
    size_t m_size = 0;
};

struct interpolative_block {
    static void encode(uint32_t const* in, uint32_t sum_of_values, size_t n, std::vector<uint8_t>& out) {
        std::vector<uint32_t> prefix(n), words;
        prefix[0] = in[0];
        for (size_t i = 1; i < n; ++i) prefix[i] = prefix[i - 1] + in[i];
        if (sum_of_values == uint32_t(-1)) {
            sum_of_values = prefix[n - 1];
            vbyte::append(sum_of_values, out);
        }
        bit_writer bw(words);
        bw.write_interpolative(prefix.data(), n - 1, 0, sum_of_values);
        auto p = reinterpret_cast<uint8_t const*>(words.data());
        out.insert(out.end(), p, p + (bw.size() + 7) / 8);
    }
};

// In-index block coders: Coder::encode(builder, in, sum_of_values, n, out) for n <= 256.
struct opt_dint_single_dict_block {
    static constexpr uint64_t block_size = kBlockSize;
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t sum_of_values, uint32_t n,
                       std::vector<uint8_t>& out) {
        if (n < block_size) {
            interpolative_block::encode(in, sum_of_values, n, out);
            return;
        }
        detail::optimal_parse([&](uint32_t const* p, uint32_t len) { return builder.lookup(p, len); }, in, n, 16, out);
    }
};

// greedy_dint_single_dict_block (include/dint/dint_codecs.hpp:52-139): longest match first
struct greedy_dint_single_dict_block {
    static constexpr uint64_t block_size = kBlockSize;
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t sum_of_values, uint32_t n,
                       std::vector<uint8_t>& out) {
        if (n < block_size) {
            interpolative_block::encode(in, sum_of_values, n, out);
            return;
        }
        detail::greedy_parse([&](uint32_t const* p, uint32_t len) { return builder.lookup(p, len); }, in, n, out);
    }
};

struct opt_dint_multi_dict_block {
    static constexpr uint64_t block_size = kBlockSize;
    template <typename Builder>
    static void encode(Builder& builder, uint32_t const* in, uint32_t sum_of_values, uint32_t n,
                       std::vector<uint8_t>& out) {
        if (n < block_size) {
            interpolative_block::encode(in, sum_of_values, n, out);
            return;
        }
        multi_opt_dint::encode(builder, in, 0, n, out);  // one block: selector byte + best of 6 x {16, 8} bit
    }
};

// dict_posting_list::write — docs are docIDs (strictly increasing), freqs are >= 1.
template <typename Coder, typename Builder>
void write_posting_list(Builder& docs_builder, Builder& freqs_builder, std::vector<uint8_t>& out, uint32_t n,
                        uint32_t const* docs, uint32_t const* freqs) {
    vbyte::append(n, out);
    const uint64_t blocks = (uint64_t(n) + kBlockSize - 1) / kBlockSize;
    const size_t begin_block_maxs = out.size();
    const size_t begin_block_endpoints = begin_block_maxs + 4 * blocks;
    const size_t begin_blocks = begin_block_endpoints + 4 * (blocks - 1);
    out.resize(begin_blocks);
    std::vector<uint32_t> docs_buf(kBlockSize), freqs_buf(kBlockSize);
    uint32_t last_doc = uint32_t(-1), block_base = 0;
    for (size_t b = 0; b != blocks; ++b) {
        const uint32_t cur = ((b + 1) * kBlockSize <= n) ? kBlockSize : (n % kBlockSize);
        for (uint32_t i = 0; i != cur; ++i) {
            const uint32_t doc = docs[b * kBlockSize + i];
            docs_buf[i] = doc - last_doc - 1;
            last_doc = doc;
            freqs_buf[i] = freqs[b * kBlockSize + i] - 1;
        }
        std::memcpy(&out[begin_block_maxs + 4 * b], &last_doc, 4);
        Coder::encode(docs_builder, docs_buf.data(), last_doc - block_base - (cur - 1), cur, out);
        Coder::encode(freqs_builder, freqs_buf.data(), uint32_t(-1), cur, out);
        if (b != blocks - 1) {
            const uint32_t endpoint = uint32_t(out.size() - begin_blocks);
            std::memcpy(&out[begin_block_endpoints + 4 * b], &endpoint, 4);
        }
        block_base = last_doc + 1;
    }
}

}  // namespace dint
