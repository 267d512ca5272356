// The reference's Coder / Dictionary operator surface over the device path.
//
// A ds2i-style *Coder* is a struct of statics (reference include/dint/dint_codecs.hpp:
// 141-283; whole-list twins vroom_env/dint_codecs.hpp:109-331, :333-619):
//
//   encode(Builder&, uint32_t const* in, uint32_t universe, uint32_t n, std::vector<uint8_t>& out)
//   decode(Dictionary const&, uint8_t const* in, uint32_t* out, uint32_t universe, size_t n)
//       -> uint8_t const*            (pointer one past the consumed bytes)
//
// and a *Dictionary* carries a nested `builder` with load / build(dict)
// (single_dictionary.hpp:24-226). The types below have exactly those shapes, so code
// written against the reference (vroom_env/decode.cpp:95-155) compiles against them;
// `encode` runs the CPU encoders of encoders.hpp, `decode` goes through the C ABI of
// include/dint_hip.h to the HIP kernels. There is no CPU decode here.
//
// The per-call `decode` is the reference's granularity (one list per call): it uploads,
// runs ONE unit on the device and downloads — correct for any n, but one wavefront
// wide. Throughput comes from the batched entry point (`dint::decode_stream`, or
// dint_decode_units directly), which is what tools/dint_decode.cpp uses.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "dint_hip.h"
#include "dictionaries.hpp"
#include "encoders.hpp"
#include "posting_list.hpp"

namespace dint {

inline void check(int status, char const* what) {
    if (status != DINT_OK)
        throw std::runtime_error(std::string(what) + ": " + dint_strerror(status) +
                                 (status == DINT_ERR_HIP ? std::string(" — ") + dint_last_hip_error() : ""));
}

// Device-resident dictionary with the reference's `Dictionary` shape: default
// constructible, filled by builder::build(dict), movable.
template <typename HostBuilder>
class device_dictionary {
public:
    static const uint32_t num_entries = kNumEntries;
    static const uint32_t max_entry_size = kMaxEntrySize;
    static const uint32_t reserved = kReserved;

    // The nested builder: the host builder (file formats, append/build, encoder
    // lookup) plus build(dict), which stages the dictionary on a device.
    struct builder : HostBuilder {
        size_t load(std::vector<uint8_t> const& file_bytes) {
            HostBuilder::load(file_bytes.data(), file_bytes.size());
            m_file = file_bytes;
            return file_bytes.size();
        }
        void build(device_dictionary& dict, int device = 0) {
            if (m_file.empty()) HostBuilder::write(m_file);
            dint_dict* h = nullptr;
            check(dint_dict_create(int(HostBuilder::kind), m_file.data(), m_file.size(), device, &h),
                  "dint_dict_create");
            dict.reset(h);
        }

    private:
        std::vector<uint8_t> m_file;
    };

    device_dictionary() = default;
    device_dictionary(device_dictionary const&) = delete;
    device_dictionary& operator=(device_dictionary const&) = delete;
    device_dictionary(device_dictionary&& o) noexcept : m_handle(o.m_handle) { o.m_handle = nullptr; }
    ~device_dictionary() { reset(nullptr); }
    void swap(device_dictionary& o) { std::swap(m_handle, o.m_handle); }
    dint_dict* handle() const { return m_handle; }
    void reset(dint_dict* h) {
        if (m_handle) dint_dict_destroy(m_handle);
        m_handle = h;
    }

private:
    dint_dict* m_handle = nullptr;
};

using single_dictionary_rectangular_type = device_dictionary<rectangular_builder>;  // dictionary_types.hpp:8-9
using single_dictionary_packed_type = device_dictionary<single_packed_builder>;     // :10-12
using multi_dictionary_packed_type = device_dictionary<multi_packed_builder>;       // :19-21

namespace detail {
// Upper bound of the bytes n integers can occupy: every integer a 32-bit exception
// (6 bytes with 16-bit codewords), plus one selector byte per 256 integers.
inline size_t worst_case_bytes(size_t n) { return 6 * n + n / 256 + 16; }

template <typename Dictionary>
uint8_t const* decode_list(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                           size_t n) {
    size_t consumed = 0;
    check(dint_decode_list_host(dict.handle(), in, size_t(in_end - in), out, n, &consumed), "dint_decode_list_host");
    return in + consumed;
}
}  // namespace detail

// Device-backed Coders. `in_end` bounds what may be read (and uploaded); the
// reference-shaped overload assumes worst_case_bytes(n) are readable, which holds
// for a list inside a larger buffer but not for the last list of an mmap'ed file —
// use the bounded overload there.
struct single_opt_dint_device : single_opt_dint {
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint32_t* out, uint32_t /*universe*/,
                                 size_t n) {
        return detail::decode_list(dict, in, in + detail::worst_case_bytes(n), out, n);
    }
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                                 uint32_t /*universe*/, size_t n) {
        return detail::decode_list(dict, in, in_end, out, n);
    }
};
struct single_greedy_dint_device : single_greedy_dint {
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint32_t* out, uint32_t universe,
                                 size_t n) {
        return single_opt_dint_device::decode(dict, in, out, universe, n);
    }
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                                 uint32_t universe, size_t n) {
        return single_opt_dint_device::decode(dict, in, in_end, out, universe, n);
    }
};
struct multi_opt_dint_device : multi_opt_dint {
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint32_t* out, uint32_t /*universe*/,
                                 size_t n) {
        return detail::decode_list(dict, in, in + detail::worst_case_bytes(n), out, n);
    }
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                                 uint32_t /*universe*/, size_t n) {
        return detail::decode_list(dict, in, in_end, out, n);
    }
};

// ---- in-index block Coders (reference include/dint/dint_codecs.hpp:9-19, :269-274, :460-510) ----
// The statics a dict_posting_list<Dictionary, Coder>-shaped caller uses: block_size, overflow,
// encode(builder, in, sum_of_values, n, out) (CPU, posting_list.hpp) and
// decode(dict, in, out, sum_of_values, n) -> in_end, which goes through dint_decode_block_host: full
// blocks to the DINT kernels, short ones to the interpolative kernel. `overflow` is kept for callers
// that size their buffers with it (dict_posting_list.hpp:104-105); the device path writes exactly n
// integers and needs neither the extra words nor a zeroed buffer.
namespace detail {
template <typename Dictionary>
uint8_t const* decode_block(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                            uint32_t sum_of_values, size_t n) {
    size_t consumed = 0;
    check(dint_decode_block_host(dict.handle(), in, size_t(in_end - in), out, sum_of_values, n, &consumed),
          "dint_decode_block_host");
    return in + consumed;
}
}  // namespace detail

// A posting list decoded ONCE on the device, for the lifetime of a scope object: every Coder::decode call of this thread
// whose `in` lies inside the list is then a memcpy from host memory (dint_list_cache, include/dint_hip.h). The
// reference's document_enumerator decodes lazily, one 256-posting block per Coder::decode call
// (dict_posting_list.hpp:126-147 next_geq, :298-301, :313-315): per call the device path costs a launch sequence and
// a stream wait, tens of microseconds — a cursor that skips through a list pays that per touched block, or this once.
// Where the reference constructs an enumerator over a list (dict_posting_list.hpp:90-107) the binding adds one line:
//     typename Coder::list_scope scope(docs_dict, &freqs_dict, list_begin, list_end);
// Scopes nest (a query holds one per term); the innermost list that contains `in` serves the call.
class list_scope_base {
public:
    list_scope_base(list_scope_base const&) = delete;
    list_scope_base& operator=(list_scope_base const&) = delete;
    ~list_scope_base() {
        // unlink from wherever this scope sits: scopes held in a container are not destroyed innermost first
        // (a vector<unique_ptr<list_scope>> goes front to back)
        list_scope_base const** link = &head();
        while (*link && *link != this) link = &const_cast<list_scope_base*>(*link)->m_prev;
        if (*link == this) *link = m_prev;
        dint_list_cache_destroy(m_cache);
    }
    // the scope of this thread whose list holds `in`, or null
    static list_scope_base const* find(uint8_t const* in) {
        for (list_scope_base const* s = head(); s; s = s->m_prev)
            if (in >= s->m_begin && in < s->m_end) return s;
        return nullptr;
    }
    // nullptr: the decoded list does not serve this call — `in` is not the start of a block's part, or it is a freqs
    // part and the scope was made without a freqs dictionary (DINT_ERR_ARG) — and the caller decodes the block itself
    uint8_t const* decode(uint8_t const* in, uint32_t* out, size_t n) const {
        size_t consumed = 0;
        const int st = dint_list_cache_decode(m_cache, size_t(in - m_begin), out, n, &consumed);
        if (st == DINT_ERR_ARG) return nullptr;
        check(st, "dint_list_cache_decode");
        return in + consumed;
    }
    uint8_t const* end() const { return m_end; }

protected:
    list_scope_base(dint_dict const* docs, dint_dict const* freqs, uint8_t const* begin, uint8_t const* end) : m_begin(begin), m_end(end) {
        check(dint_list_cache_create(docs, freqs, begin, size_t(end - begin), &m_cache), "dint_list_cache_create");
        m_prev = head();
        head() = this;
    }

private:
    static list_scope_base const*& head() {
        thread_local list_scope_base const* h = nullptr;
        return h;
    }
    uint8_t const* m_begin;
    uint8_t const* m_end;
    dint_list_cache* m_cache = nullptr;
    mutable list_scope_base const* m_prev = nullptr;
};

template <typename HostBlockCoder>
struct dint_block_device : HostBlockCoder {
    static const uint64_t block_size = kBlockSize;
    static const uint64_t overflow = 256;  // dint_block::overflow
    struct list_scope : list_scope_base {
        template <typename Dictionary>
        list_scope(Dictionary const& docs_dict, Dictionary const* freqs_dict, uint8_t const* list_begin, uint8_t const* list_end)
            : list_scope_base(docs_dict.handle(), freqs_dict ? freqs_dict->handle() : nullptr, list_begin, list_end) {}
    };
    // The reference's shape (no end pointer). Inside a list_scope: served from the decoded list, nothing is read. Outside
    // one, what is uploaded is bounded by readable_end() when the caller has set it (the end of the mapped index), else
    // by the bytes a block can occupy — readable for any block but the last ones of a mapped file: set the end, or use
    // the bounded overload, there.
    static uint8_t const*& readable_end() {
        thread_local uint8_t const* end = nullptr;
        return end;
    }
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint32_t* out, uint32_t sum_of_values, size_t n) {
        if (list_scope_base const* s = list_scope_base::find(in))
            if (uint8_t const* served = s->decode(in, out, n)) return served;
        uint8_t const* end = in + detail::worst_case_bytes(n);
        if (readable_end() && readable_end() > in && readable_end() < end) end = readable_end();
        return detail::decode_block(dict, in, end, out, sum_of_values, n);
    }
    template <typename Dictionary>
    static uint8_t const* decode(Dictionary const& dict, uint8_t const* in, uint8_t const* in_end, uint32_t* out,
                                 uint32_t sum_of_values, size_t n) {
        if (list_scope_base const* s = list_scope_base::find(in))
            if (uint8_t const* served = s->decode(in, out, n)) return served;
        return detail::decode_block(dict, in, in_end, out, sum_of_values, n);
    }
};
using opt_dint_single_dict_block_device = dint_block_device<opt_dint_single_dict_block>;  // dint_codecs.hpp:141-283
using opt_dint_multi_dict_block_device = dint_block_device<opt_dint_multi_dict_block>;    // :285-510

}  // namespace dint
