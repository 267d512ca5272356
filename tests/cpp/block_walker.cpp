// A dict_posting_list<Dictionary, Coder>::document_enumerator-shaped caller over the device block Coders:
// walks one posting list block by block exactly as the reference does (include/dint/dict_posting_list.hpp:
// 90-107 constructor, :284-309 decode_docs_block, :311-318 decode_freqs_block, :111-124 next) — Coder::block_size
// to cut the blocks, Coder::decode(*docs_dict, block_data, buf, max - base - (size - 1), size) returning where
// the freqs part begins, Coder::decode(*freqs_dict, ...) with sum_of_values = -1 — and compares every
// (docid, freq) with the expected arrays. Own code: only the call sequence is the reference's.
// usage: block_walker <kind 1|2> <docs dict> <freqs dict> <list bytes> <docids u32> <freqs u32> [scope|noend]
//   scope: the walk runs inside a Coder::list_scope (the list decoded once, every Coder::decode a memcpy) and uses the
//          reference's end-less decode(dict, in, out, sum, n); prints the walk's wall time and how many device / pinned
//          allocations the library made during it.   noend: the end-less overload per block, bounded by readable_end().
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

#include "dint/coders.hpp"
#include "dint/vbyte.hpp"

template <typename T>
std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<T> v(size_t(f.tellg()) / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), std::streamsize(v.size() * sizeof(T)));
    return v;
}

template <typename Dictionary, typename Coder>
class enumerator {
public:
    enumerator(Dictionary const* docs_dict, Dictionary const* freqs_dict, uint8_t const* data, uint8_t const* data_end, bool endless = false)
        : m_end(data_end), m_docs_dict(docs_dict), m_freqs_dict(freqs_dict), m_endless(endless) {
        m_base = dint::vbyte::read(data, data_end, &m_n);
        m_blocks = (m_n + Coder::block_size - 1) / Coder::block_size;
        m_block_maxs = m_base;
        m_block_endpoints = m_block_maxs + 4 * m_blocks;
        m_blocks_data = m_block_endpoints + 4 * (m_blocks - 1);
        // (poisoned, not zeroed: the device Coders must not depend on the reference's std::fill)
        m_docs_buf.assign(Coder::block_size + Coder::overflow, 0xDEADBEEFu);
        m_freqs_buf.assign(Coder::block_size + Coder::overflow, 0xDEADBEEFu);
        decode_docs_block(0);
    }
    uint64_t size() const { return m_n; }
    uint64_t docid() const { return m_cur_docid; }
    uint64_t freq() {
        if (!m_freqs_decoded) decode_freqs_block();
        return m_freqs_buf[m_pos_in_block] + 1;
    }
    bool next() {  // false: past the end
        ++m_pos_in_block;
        if (m_pos_in_block == m_cur_block_size) {
            if (m_cur_block + 1 == m_blocks) return false;
            decode_docs_block(m_cur_block + 1);
        } else {
            m_cur_docid += m_docs_buf[m_pos_in_block] + 1;
        }
        return true;
    }
    bool poisoned() const {  // the overflow area: never written by the device Coders
        for (size_t i = Coder::block_size; i != m_docs_buf.size(); ++i)
            if (m_docs_buf[i] != 0xDEADBEEFu || m_freqs_buf[i] != 0xDEADBEEFu) return false;
        return true;
    }

private:
    uint32_t u32_at(uint8_t const* p, uint64_t i) const {
        uint32_t v;
        std::memcpy(&v, p + 4 * i, 4);
        return v;
    }
    void decode_docs_block(uint64_t block) {
        const uint32_t endpoint = block ? u32_at(m_block_endpoints, block - 1) : 0;
        uint8_t const* block_data = m_blocks_data + endpoint;
        m_cur_block_size = ((block + 1) * Coder::block_size <= m_n) ? uint32_t(Coder::block_size) : uint32_t(m_n % Coder::block_size);
        const uint32_t cur_base = (block ? u32_at(m_block_maxs, block - 1) : uint32_t(-1)) + 1;
        const uint32_t cur_max = u32_at(m_block_maxs, block);
        m_freqs_block_data = m_endless ? Coder::decode(*m_docs_dict, block_data, m_docs_buf.data(),  // (dict_posting_list.hpp:298-301)
                                                       cur_max - cur_base - (m_cur_block_size - 1), m_cur_block_size)
                                       : Coder::decode(*m_docs_dict, block_data, m_end, m_docs_buf.data(),
                                                       cur_max - cur_base - (m_cur_block_size - 1), m_cur_block_size);
        m_docs_buf[0] += cur_base;
        m_cur_block = block;
        m_pos_in_block = 0;
        m_cur_docid = m_docs_buf[0];
        m_freqs_decoded = false;
    }
    void decode_freqs_block() {
        if (m_endless) Coder::decode(*m_freqs_dict, m_freqs_block_data, m_freqs_buf.data(), uint32_t(-1), m_cur_block_size);  // (:313-315)
        else Coder::decode(*m_freqs_dict, m_freqs_block_data, m_end, m_freqs_buf.data(), uint32_t(-1), m_cur_block_size);
        m_freqs_decoded = true;
    }

    uint32_t m_n = 0;
    uint8_t const* m_base;
    uint8_t const* m_end;
    uint64_t m_blocks;
    uint8_t const* m_block_maxs;
    uint8_t const* m_block_endpoints;
    uint8_t const* m_blocks_data;
    Dictionary const* m_docs_dict;
    Dictionary const* m_freqs_dict;
    bool m_endless = false;
    uint64_t m_cur_block = 0;
    uint32_t m_pos_in_block = 0, m_cur_block_size = 0;
    uint64_t m_cur_docid = 0;
    uint8_t const* m_freqs_block_data = nullptr;
    bool m_freqs_decoded = false;
    std::vector<uint32_t> m_docs_buf, m_freqs_buf;
};

extern "C" int dint_debug_alloc_count(uint64_t* count);  // test hook of libdint_hip.so

template <typename Dictionary, typename Coder>
int run(char** argv, const char* mode) {
    auto docs_file = slurp<uint8_t>(argv[2]), freqs_file = slurp<uint8_t>(argv[3]), list = slurp<uint8_t>(argv[4]);
    auto want_docs = slurp<uint32_t>(argv[5]), want_freqs = slurp<uint32_t>(argv[6]);
    static_assert(Coder::block_size == 256 && Coder::overflow == 256, "the reference's statics");
    typename Dictionary::builder db, fb;
    db.load(docs_file);
    fb.load(freqs_file);
    Dictionary docs_dict, freqs_dict;
    db.build(docs_dict);
    fb.build(freqs_dict);
    const size_t list_bytes = list.size();
    list.resize(list.size() + 16, 0);
    const bool scoped = std::strcmp(mode, "scope") == 0, noend = std::strcmp(mode, "noend") == 0;
    if (scoped) {  // warm the dictionaries' host-call workspaces (their first use allocates): a scope over the same list
        typename Coder::list_scope warm(docs_dict, &freqs_dict, list.data(), list.data() + list_bytes);
    }
    const bool scopes = std::strcmp(mode, "scopes") == 0;
    if (noend || scopes) Coder::readable_end() = list.data() + list.size();
    uint64_t allocs0 = 0, allocs1 = 0;
    dint_debug_alloc_count(&allocs0);
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_ptr<typename Coder::list_scope> scope;
    if (scoped) scope.reset(new typename Coder::list_scope(docs_dict, &freqs_dict, list.data(), list.data() + list_bytes));
    auto walk = [&](bool endless) -> int {
        enumerator<Dictionary, Coder> e(&docs_dict, &freqs_dict, list.data(), list.data() + list.size(), endless);
        if (e.size() != want_docs.size()) {
            std::cerr << "list holds " << e.size() << " postings, expected " << want_docs.size() << "\n";
            return 1;
        }
        for (size_t i = 0; i != want_docs.size(); ++i) {
            if (e.docid() != want_docs[i] || e.freq() != want_freqs[i]) {
                std::cerr << "posting " << i << ": (" << e.docid() << ", " << e.freq() << ") expected (" << want_docs[i] << ", "
                          << want_freqs[i] << ")\n";
                return 1;
            }
            if ((!scoped || i % 256 == 0) && !e.poisoned()) {  // (a timed walk looks once per block)
                std::cerr << "posting " << i << ": the decoder wrote past the block\n";
                return 1;
            }
            const bool more = e.next();
            if (more != (i + 1 != want_docs.size())) {
                std::cerr << "next() at posting " << i << " returned " << more << "\n";
                return 1;
            }
        }
        return 0;
    };
    if (scopes) {
        // Scopes held in a container die front to back, not innermost first; a scope without a freqs dictionary serves the
        // docs parts and leaves the freqs parts to the per-block call (the cache answers DINT_ERR_ARG: no exception).
        using scope_t = typename Coder::list_scope;
        std::vector<std::unique_ptr<scope_t>> held;
        held.emplace_back(new scope_t(docs_dict, static_cast<Dictionary const*>(nullptr), list.data(), list.data() + list_bytes));
        if (walk(true)) return 1;  // docs from the scope, freqs decoded block by block
        held.emplace_back(new scope_t(docs_dict, &freqs_dict, list.data(), list.data() + list_bytes));
        held.emplace_back(new scope_t(docs_dict, static_cast<Dictionary const*>(nullptr), list.data(), list.data() + list_bytes));
        if (walk(true)) return 1;
        held.erase(held.begin());      // the OUTERMOST first ...
        if (walk(true)) return 1;      // ... the other two still serve
        held.erase(held.begin() + 1);  // then the innermost
        if (walk(true)) return 1;
        held.clear();
        if (Coder::list_scope::find(list.data() + 1) != nullptr) {
            std::cerr << "a destroyed scope is still on this thread's list\n";
            return 1;
        }
        if (walk(true)) return 1;      // no scope at all: every block through the per-block call
        std::cout << "ok\n";
        return 0;
    }
    if (walk(scoped || noend)) return 1;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    dint_debug_alloc_count(&allocs1);
    std::cout << "ok\n";
    if (scoped) std::cout << "walk_ms " << ms << " allocations " << (allocs1 - allocs0) << " postings " << want_docs.size() << "\n";
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 7 && argc != 8) return 2;
    const char* mode = argc == 8 ? argv[7] : "";
    try {
        switch (std::atoi(argv[1])) {
            case 1: return run<dint::single_dictionary_packed_type, dint::opt_dint_single_dict_block_device>(argv, mode);
            case 2: return run<dint::multi_dictionary_packed_type, dint::opt_dint_multi_dict_block_device>(argv, mode);
        }
    } catch (std::exception const& ex) {
        std::cerr << "exception: " << ex.what() << "\n";
        return 1;
    }
    return 2;
}
