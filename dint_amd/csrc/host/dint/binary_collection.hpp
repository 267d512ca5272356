// Readers for the ds2i collection files (CPU, ingest side).
//
//   binary_collection        reference include/ds2i/binary_collection.hpp:13-157
//   binary_freq_collection   reference include/ds2i/binary_freq_collection.hpp:11-110
//
// A collection file is a stream of little-endian u32 records `len, v[len]` (README.md:43-51). Reading
// follows the reference's iterator (:131-146): records of length 0 are skipped, and a last record that
// claims more values than the file holds is cut at the end of the file. In a `.docs` file record 0 is the
// singleton `1, num_docs` and every further record a strictly increasing docID list; the `.freqs` file
// holds the matching term frequencies (>= 1), record for record, with no leading singleton.
//
// The file is memory-mapped read-only (the reference maps it through Boost); lists are handed out as
// pointer ranges into the mapping. `sequences()` gives random access (one pass over the length words),
// which is what the parallel encoder and the statistics need.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace dint {

class binary_collection {
public:
    typedef uint32_t posting_type;

    struct sequence {
        posting_type const* m_begin = nullptr;
        posting_type const* m_end = nullptr;
        posting_type const* begin() const { return m_begin; }
        posting_type const* end() const { return m_end; }
        posting_type back() const { return *(m_end - 1); }
        size_t size() const { return size_t(m_end - m_begin); }
    };

    explicit binary_collection(char const* filename) {
        m_fd = ::open(filename, O_RDONLY);
        if (m_fd < 0) throw std::runtime_error("Error opening file");  // the reference's message (:21)
        struct stat st;
        if (::fstat(m_fd, &st) != 0) {
            ::close(m_fd);
            throw std::runtime_error("Error opening file");
        }
        m_bytes = size_t(st.st_size);
        if (m_bytes) {
            void* p = ::mmap(nullptr, m_bytes, PROT_READ, MAP_PRIVATE, m_fd, 0);
            if (p == MAP_FAILED) {
                ::close(m_fd);
                throw std::runtime_error("Error opening file");
            }
            m_map = p;
            ::posix_madvise(p, m_bytes, POSIX_MADV_SEQUENTIAL);
        }
        m_data = static_cast<posting_type const*>(m_map);
        m_data_size = m_bytes / sizeof(posting_type);
    }
    // over words already in memory (tests; the C ABI)
    binary_collection(posting_type const* words, size_t n_words) : m_data(words), m_data_size(n_words) {}
    binary_collection(binary_collection const&) = delete;
    binary_collection& operator=(binary_collection const&) = delete;
    ~binary_collection() {
        if (m_map) ::munmap(m_map, m_bytes);
        if (m_fd >= 0) ::close(m_fd);
    }

    // the number of u32 words of the file, length words included (the reference's name, :42-44)
    size_t num_postings() const { return m_data_size; }

    class iterator {
    public:
        sequence const& operator*() const { return m_cur_seq; }
        sequence const* operator->() const { return &m_cur_seq; }
        iterator& operator++() {
            m_pos = m_next_pos;
            read();
            return *this;
        }
        bool operator==(iterator const& o) const { return m_pos == o.m_pos; }
        bool operator!=(iterator const& o) const { return m_pos != o.m_pos; }

    private:
        friend class binary_collection;
        iterator(binary_collection const* c, size_t pos) : m_collection(c), m_pos(pos), m_next_pos(pos) { read(); }
        void read() {
            size_t const size = m_collection->m_data_size;
            if (m_pos >= size) {
                m_pos = size;
                return;
            }
            size_t n = 0, pos = m_pos;
            while (pos < size && !(n = m_collection->m_data[pos++])) {  // skip empty seqs
            }
            if (n == 0) {  // nothing but empty records up to the end of the file (the reference reads past it)
                m_pos = m_next_pos = size;
                m_cur_seq = sequence();
                return;
            }
            if (n > size - pos) n = size - pos;  // file might be truncated
            m_cur_seq.m_begin = m_collection->m_data + pos;
            m_cur_seq.m_end = m_cur_seq.m_begin + n;
            m_next_pos = pos + n;
        }
        binary_collection const* m_collection;
        size_t m_pos, m_next_pos;
        sequence m_cur_seq;
    };

    iterator begin() const { return iterator(this, 0); }
    iterator end() const { return iterator(this, m_data_size); }

    // every sequence the iterator would visit, in order
    std::vector<sequence> sequences() const {
        std::vector<sequence> all;
        for (auto it = begin(); it != end(); ++it) all.push_back(*it);
        return all;
    }

private:
    int m_fd = -1;
    void* m_map = nullptr;
    size_t m_bytes = 0;
    posting_type const* m_data = nullptr;
    size_t m_data_size = 0;
};

// <basename>.docs + <basename>.freqs, walked in step (binary_freq_collection.hpp:14-23, :27-34)
class binary_freq_collection {
public:
    struct sequence {
        binary_collection::sequence docs, freqs;
    };

    explicit binary_freq_collection(char const* basename)
        : m_docs((std::string(basename) + ".docs").c_str()), m_freqs((std::string(basename) + ".freqs").c_str()) {
        auto firstseq = *m_docs.begin();
        if (firstseq.size() != 1)
            throw std::invalid_argument("First sequence should only contain number of documents");
        m_num_docs = *firstseq.begin();
    }
    uint64_t num_docs() const { return m_num_docs; }
    uint64_t num_postings() const { return m_docs.num_postings() + m_freqs.num_postings() - 2; }

    // every (docs, freqs) pair, the leading singleton of .docs skipped
    std::vector<sequence> sequences() const {
        std::vector<sequence> all;
        auto d = m_docs.begin();
        ++d;
        auto f = m_freqs.begin();
        for (; d != m_docs.end(); ++d, ++f) {
            if (f == m_freqs.end() || f->size() != d->size())
                throw std::runtime_error("docs and freqs files do not match");
            all.push_back({*d, *f});
        }
        return all;
    }
    binary_collection const& docs() const { return m_docs; }
    binary_collection const& freqs() const { return m_freqs; }

private:
    binary_collection m_docs, m_freqs;
    uint64_t m_num_docs = 0;
};

// gaps of one list as every DINT consumer wants them (vroom_env/jobs.hpp:74-84; block_statistics.hpp:70-77):
// docs: doc - prev - 1 with prev = -1 before the first; freqs: value - 1. Returns the u32 sum (the "universe").
inline uint32_t list_to_gaps(binary_collection::sequence const& list, bool docs, uint32_t* out) {
    uint32_t universe = 0;
    uint32_t prev = docs ? uint32_t(-1) : 0;
    size_t i = 0;
    for (auto p = list.begin(); p != list.end(); ++p, ++i) {
        out[i] = *p - prev - 1;
        if (docs) prev = *p;
        universe += out[i];
    }
    return universe;
}

}  // namespace dint
