// The on-disk container of an inverted index in the reference's in-index layout.
//
// The reference freezes a dict_freq_index with succinct::mapper (include/dint/dict_freq_index.hpp:208-214: m_params,
// m_size, m_num_docs, m_endpoints (Elias-Fano), m_lists, m_docs_dict, m_freqs_dict); succinct is absent here and its
// byte format is pinned by nothing in the reference tree, so this container is this repo's own. It holds
// the same members — the LISTS' bytes are the reference's (dict_posting_list::write), the endpoints are plain u64s:
//
//   char     magic[8] = "DINTIDX1"
//   u32      kind (dint_dict_kind), u32 coder (0 optimal parse, 1 greedy)
//   u64      num_docs, n_lists, docs_dict_bytes, freqs_dict_bytes, index_bytes
//   u64      offsets[n_lists + 1]           byte offset of every list inside the index bytes
//   u8       docs dictionary file image     (builder::write format), padded to 8 bytes
//   u8       freqs dictionary file image, padded to 8 bytes
//   u8       index bytes                    (the lists back to back)
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace dint {

struct index_file_header {
    char magic[8];
    uint32_t kind, coder;
    uint64_t num_docs, n_lists, docs_dict_bytes, freqs_dict_bytes, index_bytes;
};
static_assert(sizeof(index_file_header) == 56, "index_file_header layout");

struct index_file_view {  // pointers into a mapped or loaded file
    index_file_header header;
    uint64_t const* offsets = nullptr;
    uint8_t const* docs_dict = nullptr;
    uint8_t const* freqs_dict = nullptr;
    uint8_t const* index = nullptr;
};

inline size_t pad8(size_t n) { return (n + 7) & ~size_t(7); }

inline void write_index_file(std::string const& path, uint32_t kind, uint32_t coder, uint64_t num_docs, uint64_t n_lists,
                             uint64_t const* offsets, void const* docs_dict, size_t docs_dict_bytes, void const* freqs_dict,
                             size_t freqs_dict_bytes, void const* index, size_t index_bytes) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    index_file_header h;
    std::memcpy(h.magic, "DINTIDX1", 8);
    h.kind = kind;
    h.coder = coder;
    h.num_docs = num_docs;
    h.n_lists = n_lists;
    h.docs_dict_bytes = docs_dict_bytes;
    h.freqs_dict_bytes = freqs_dict_bytes;
    h.index_bytes = index_bytes;
    const char zeros[8] = {0};
    f.write(reinterpret_cast<char const*>(&h), sizeof h);
    f.write(reinterpret_cast<char const*>(offsets), std::streamsize((n_lists + 1) * 8));
    f.write(static_cast<char const*>(docs_dict), std::streamsize(docs_dict_bytes));
    f.write(zeros, std::streamsize(pad8(docs_dict_bytes) - docs_dict_bytes));
    f.write(static_cast<char const*>(freqs_dict), std::streamsize(freqs_dict_bytes));
    f.write(zeros, std::streamsize(pad8(freqs_dict_bytes) - freqs_dict_bytes));
    f.write(static_cast<char const*>(index), std::streamsize(index_bytes));
    if (!f) throw std::runtime_error("cannot write " + path);
}

inline index_file_view view_index_file(void const* data, size_t bytes) {
    index_file_view v;
    if (bytes < sizeof(index_file_header)) throw std::runtime_error("index file truncated");
    std::memcpy(&v.header, data, sizeof v.header);
    if (std::memcmp(v.header.magic, "DINTIDX1", 8) != 0) throw std::runtime_error("not a DINT index file");
    auto const& h = v.header;
    // every section against what is LEFT of the file (no sum of header fields that a hostile value could wrap)
    size_t left = bytes - sizeof h;
    auto take = [&](uint64_t n) {
        if (n > left) throw std::runtime_error("index file truncated");
        left -= size_t(n);
    };
    if (h.n_lists >= (uint64_t(1) << 60)) throw std::runtime_error("index file truncated");
    take((h.n_lists + 1) * 8);
    if (h.docs_dict_bytes > left || h.freqs_dict_bytes > left) throw std::runtime_error("index file truncated");
    take(pad8(size_t(h.docs_dict_bytes)));
    take(pad8(size_t(h.freqs_dict_bytes)));
    take(h.index_bytes);
    auto p = static_cast<uint8_t const*>(data) + sizeof h;
    v.offsets = reinterpret_cast<uint64_t const*>(p);
    p += (h.n_lists + 1) * 8;
    v.docs_dict = p;
    p += pad8(h.docs_dict_bytes);
    v.freqs_dict = p;
    p += pad8(h.freqs_dict_bytes);
    v.index = p;
    if (v.offsets[h.n_lists] != h.index_bytes) throw std::runtime_error("index file: offsets do not end at the index's size");
    for (uint64_t i = 0; i != h.n_lists; ++i)
        if (v.offsets[i] > v.offsets[i + 1]) throw std::runtime_error("index file: list offsets decrease");
    return v;
}

}  // namespace dint
