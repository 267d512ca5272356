// Host-side DINT dictionaries: construction, file formats, encoder lookup.
//
// Three builders with the reference's builder interface (init / append / build /
// load / write / prepare_for_encoding / lookup / size / get), one per
// dictionary type the decode path supports:
//
//   rectangular_builder    include/dint/rectangular_dictionary.hpp:24-203
//   single_packed_builder  include/dint/single_dictionary.hpp:24-226   (pack_policy)
//   multi_packed_builder   include/dint/multi_dictionary.hpp:26-289    (pack_policy)
//
// File formats are byte-compatible with the reference's `write`/`load`
// (rect :72-92, single :72-107, multi :70-121). The *decode* side of a
// dictionary (`copy()`) deliberately does not exist on the host: decoding is
// the device's job (include/dint_hip.h); builders only need `get(i)`/`size(i)`
// for the encoder's hash map.
//
// Packing: `pack_policy::compact` (dictionary_building_utils.hpp:241-292) is
// O(n^2) and the offset search (`std::search` per entry, single_dictionary.hpp:
// 138-151) is O(n * table); here both are done with hashing in O(n) while
// producing the same table and the same first-occurrence offsets.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "constants.hpp"
#include "hash.hpp"

namespace dint {

using entry_t = std::vector<uint32_t>;

namespace detail {

// (size, lexicographic) order of target_t::operator< (dictionary_building_utils.hpp:36-43)
inline bool entry_less(entry_t const& a, entry_t const& b) {
    if (a.size() != b.size()) return a.size() < b.size();
    return std::lexicographical_compare(a.begin(), a.end(), b.begin(), b.end());
}

struct span_hash_key {
    uint64_t h;
    uint32_t len;
    bool operator==(span_hash_key const& o) const { return h == o.h && len == o.len; }
};
struct span_hash_key_hasher {
    size_t operator()(span_hash_key const& k) const { return size_t(k.h ^ (uint64_t(k.len) << 56)); }
};

// pack_policy::compact: sort, unique, drop every entry that is a proper prefix
// of a longer entry, keep the survivors in sorted order.
inline std::vector<entry_t> pack_compact(std::vector<entry_t> all) {
    std::sort(all.begin(), all.end(), entry_less);
    all.erase(std::unique(all.begin(), all.end()), all.end());
    // All longer entries sort after a shorter one and are still "valid" when the
    // reference visits the shorter one, so the drop test reduces to: does any
    // longer entry start with it.
    std::unordered_map<span_hash_key, std::vector<uint32_t>, span_hash_key_hasher> prefixes;
    for (uint32_t i = 0; i != all.size(); ++i) {
        auto const& e = all[i];
        for (uint32_t p = 1; p < e.size(); ++p) {
            prefixes[{hash_u32s(e.data(), p), p}].push_back(i);
        }
    }
    std::vector<entry_t> kept;
    kept.reserve(all.size());
    for (auto& e : all) {
        bool dropped = false;
        auto it = prefixes.find({hash_u32s(e.data(), e.size()), uint32_t(e.size())});
        if (it != prefixes.end()) {
            for (uint32_t j : it->second) {
                if (std::equal(e.begin(), e.end(), all[j].begin())) {
                    dropped = true;
                    break;
                }
            }
        }
        if (!dropped) kept.push_back(e);
    }
    return kept;
}

// For every entry, the first position in `table` where it occurs as a
// contiguous subsequence (what std::search returns).
inline std::vector<uint32_t> first_occurrences(std::vector<uint32_t> const& table,
                                               std::vector<entry_t const*> const& entries) {
    std::vector<uint32_t> pos(entries.size(), kInvalidIndex);
    std::unordered_map<span_hash_key, std::vector<uint32_t>, span_hash_key_hasher> wanted;
    std::vector<uint32_t> lens;
    for (uint32_t i = 0; i != entries.size(); ++i) {
        auto const& e = *entries[i];
        wanted[{hash_u32s(e.data(), e.size()), uint32_t(e.size())}].push_back(i);
        if (std::find(lens.begin(), lens.end(), uint32_t(e.size())) == lens.end())
            lens.push_back(uint32_t(e.size()));
    }
    for (uint32_t len : lens) {
        if (table.size() < len) continue;
        for (uint32_t t = 0; t + len <= table.size(); ++t) {
            auto it = wanted.find({hash_u32s(&table[t], len), len});
            if (it == wanted.end()) continue;
            for (uint32_t i : it->second) {
                if (pos[i] == kInvalidIndex &&
                    std::equal(entries[i]->begin(), entries[i]->end(), table.begin() + t)) {
                    pos[i] = t;
                }
            }
        }
    }
    return pos;
}

inline void put_u32(std::vector<uint8_t>& out, uint32_t v) {
    uint8_t b[4];
    std::memcpy(b, &v, 4);
    out.insert(out.end(), b, b + 4);
}
inline void put_u32s(std::vector<uint8_t>& out, std::vector<uint32_t> const& v) {
    auto p = reinterpret_cast<uint8_t const*>(v.data());
    out.insert(out.end(), p, p + v.size() * 4);
}

struct reader {
    uint8_t const* p;
    uint8_t const* end;
    uint32_t u32() {
        if (end - p < 4) throw std::runtime_error("dictionary file truncated");
        uint32_t v;
        std::memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    void u32s(std::vector<uint32_t>& dst, size_t n) {
        if (size_t(end - p) < n * 4) throw std::runtime_error("dictionary file truncated");
        dst.resize(n);
        if (n) std::memcpy(dst.data(), p, n * 4);
        p += n * 4;
    }
};

// Open-addressing u64 -> u32 map for the encoder's hot lookup loop. Same
// semantics as the reference's unordered_map<uint64_t, uint32_t> (insert
// overwrites, find by hash only), several times faster to probe.
class flat_map {
public:
    void clear() {
        m_keys.clear();
        m_vals.clear();
        m_used = 0;
    }
    void reserve_for(size_t n) {
        size_t cap = 16;
        while (cap < 2 * n + 2) cap <<= 1;
        m_keys.assign(cap, 0);
        m_vals.assign(cap, kInvalidIndex);
        m_used = 0;
        m_has_zero = false;
    }
    void set(uint64_t key, uint32_t val) {
        if (key == 0) {  // 0 marks an empty slot
            m_has_zero = true;
            m_zero_val = val;
            return;
        }
        if (2 * (m_used + 1) > m_keys.size()) grow();
        size_t mask = m_keys.size() - 1;
        for (size_t i = size_t(key) & mask;; i = (i + 1) & mask) {
            if (m_keys[i] == key) {
                m_vals[i] = val;
                return;
            }
            if (m_keys[i] == 0) {
                m_keys[i] = key;
                m_vals[i] = val;
                ++m_used;
                return;
            }
        }
    }
    uint32_t find(uint64_t key) const {
        if (key == 0) return m_has_zero ? m_zero_val : kInvalidIndex;
        if (m_keys.empty()) return kInvalidIndex;
        size_t mask = m_keys.size() - 1;
        for (size_t i = size_t(key) & mask;; i = (i + 1) & mask) {
            if (m_keys[i] == key) return m_vals[i];
            if (m_keys[i] == 0) return kInvalidIndex;
        }
    }

private:
    void grow() {
        std::vector<uint64_t> keys;
        std::vector<uint32_t> vals;
        keys.swap(m_keys);
        vals.swap(m_vals);
        m_keys.assign(std::max<size_t>(16, keys.size() * 2), 0);
        m_vals.assign(m_keys.size(), kInvalidIndex);
        m_used = 0;
        for (size_t i = 0; i != keys.size(); ++i)
            if (keys[i]) set(keys[i], vals[i]);
    }
    std::vector<uint64_t> m_keys;
    std::vector<uint32_t> m_vals;
    size_t m_used = 0;
    bool m_has_zero = false;
    uint32_t m_zero_val = kInvalidIndex;
};

inline uint32_t packed_offsets_word(uint32_t size, uint32_t offset) {
    return ((size - 1) << 24) | offset;  // single_dictionary.hpp:147-149
}

}  // namespace detail

// ---------------------------------------------------------------------------
// rectangular: row i = 16 payload words + 1 size word (rectangular_dictionary.hpp:39-56)
// ---------------------------------------------------------------------------
struct rectangular_builder {
    static constexpr uint32_t row = kMaxEntrySize + 1;
    static std::string type() { return "rectangular"; }
    static constexpr dict_kind kind = dict_kind::rectangular;

    void init() {
        m_size = kReserved;
        m_table.assign(size_t(kNumEntries) * row, 0);
        for (uint32_t i = 0; i != kExceptions; ++i) m_table[i * row + kMaxEntrySize] = 1;
        for (uint32_t i = kExceptions; i != kReserved; ++i)
            m_table[i * row + kMaxEntrySize] = run_size_of(i);
    }
    bool full() const { return m_size == kNumEntries; }
    bool append(uint32_t const* entry, uint32_t entry_size, uint32_t /*dictionary_id*/) {
        if (full()) return false;
        std::copy(entry, entry + entry_size, &m_table[size_t(m_size) * row]);
        m_table[size_t(m_size) * row + kMaxEntrySize] = entry_size;
        ++m_size;
        return true;
    }
    void build() {}

    // file: u32 m_size, u32 table[m_size * 17]   (:72-77)
    void write(std::vector<uint8_t>& out) const {
        detail::put_u32(out, m_size);
        auto p = reinterpret_cast<uint8_t const*>(m_table.data());
        out.insert(out.end(), p, p + size_t(m_size) * row * 4);
    }
    void load(uint8_t const* bytes, size_t len) {
        detail::reader r{bytes, bytes + len};
        uint32_t size = r.u32();
        if (size > kNumEntries) throw std::runtime_error("rectangular dictionary: bad size");
        init();  // reserved rows are preset, then overwritten by the file (:79-92)
        m_size = size;
        std::vector<uint32_t> rows;
        r.u32s(rows, size_t(size) * row);
        std::copy(rows.begin(), rows.end(), m_table.begin());
    }

    uint32_t size() const { return m_size; }
    uint32_t size(uint32_t i) const { return m_table[size_t(i) * row + kMaxEntrySize]; }
    uint32_t const* get(uint32_t i) const { return &m_table[size_t(i) * row]; }
    uint32_t num_dictionaries() const { return 1; }

    void prepare_for_encoding() { prepare_single(*this, m_map); }
    uint32_t lookup(uint32_t const* begin, uint32_t entry_size) const {
        return m_map.find(hash_u32s(begin, entry_size));
    }

    // shared by rectangular and single_packed (single_dictionary.hpp:154-165):
    // runs first, real entries afterwards, later insertions overwrite.
    template <typename B>
    static void prepare_single(B const& b, detail::flat_map& map) {
        map.reserve_for(b.size());
        std::vector<uint32_t> zeros(256, 0);
        uint32_t i = kExceptions;
        for (uint32_t n = 256; n >= 16; n /= 2, ++i) map.set(hash_u32s(zeros.data(), n), i);
        for (; i < b.size(); ++i) map.set(hash_u32s(b.get(i), b.size(i)), i);
    }

private:
    uint32_t m_size = kReserved;
    std::vector<uint32_t> m_table;
    detail::flat_map m_map;
};

// ---------------------------------------------------------------------------
// single packed: offsets[i] = (size-1)<<24 | table offset; shared packed table
// that starts with 16 zeros (single_dictionary.hpp:40-56, 230-238)
// ---------------------------------------------------------------------------
struct single_packed_builder {
    static std::string type() { return "single_packed"; }
    static constexpr dict_kind kind = dict_kind::single_packed;

    void init() {
        m_size = kReserved;
        m_offsets.clear();
        m_table.assign(kMaxEntrySize, 0);
        m_targets.clear();
        for (uint32_t i = 0; i != kExceptions; ++i) m_offsets.push_back(0);
        for (uint32_t i = kExceptions; i != kReserved; ++i)
            m_offsets.push_back(detail::packed_offsets_word(run_size_of(i), 0));
    }
    bool full() const { return m_size == kNumEntries; }
    bool append(uint32_t const* entry, uint32_t entry_size, uint32_t /*dictionary_id*/) {
        if (full()) return false;
        m_targets.emplace_back(entry, entry + entry_size);
        ++m_size;
        return true;
    }
    void build() {
        for (auto const& e : detail::pack_compact(m_targets))
            m_table.insert(m_table.end(), e.begin(), e.end());
        std::vector<entry_t const*> ptrs;
        for (auto const& e : m_targets) ptrs.push_back(&e);
        auto pos = detail::first_occurrences(m_table, ptrs);
        for (uint32_t i = 0; i != m_targets.size(); ++i)
            m_offsets.push_back(detail::packed_offsets_word(uint32_t(m_targets[i].size()), pos[i]));
        m_targets.clear();
    }

    // file: u32 m_size, u32 n_offsets, u32 n_table, offsets[], table[]   (:72-86)
    void write(std::vector<uint8_t>& out) const {
        detail::put_u32(out, m_size);
        detail::put_u32(out, uint32_t(m_offsets.size()));
        detail::put_u32(out, uint32_t(m_table.size()));
        detail::put_u32s(out, m_offsets);
        detail::put_u32s(out, m_table);
    }
    void load(uint8_t const* bytes, size_t len) {
        detail::reader r{bytes, bytes + len};
        m_size = r.u32();
        uint32_t n_off = r.u32(), n_tab = r.u32();
        r.u32s(m_offsets, n_off);
        r.u32s(m_table, n_tab);
        if (m_size > m_offsets.size()) throw std::runtime_error("single dictionary: bad size");
    }

    uint32_t size() const { return m_size; }
    uint32_t size(uint32_t i) const { return (m_offsets[i] >> 24) + 1; }
    uint32_t offset(uint32_t i) const { return m_offsets[i] & 0xFFFFFF; }
    uint32_t const* get(uint32_t i) const { return &m_table[offset(i)]; }
    uint32_t num_dictionaries() const { return 1; }
    std::vector<uint32_t> const& offsets() const { return m_offsets; }
    std::vector<uint32_t> const& table() const { return m_table; }

    void prepare_for_encoding() { rectangular_builder::prepare_single(*this, m_map); }
    uint32_t lookup(uint32_t const* begin, uint32_t entry_size) const {
        return m_map.find(hash_u32s(begin, entry_size));
    }

private:
    uint32_t m_size = kReserved;
    std::vector<entry_t> m_targets;
    std::vector<uint32_t> m_offsets;
    std::vector<uint32_t> m_table;
    detail::flat_map m_map;
};

// ---------------------------------------------------------------------------
// multi packed: kNumSelectors dictionaries, one shared packed table,
// start_offsets[d] = first slot of dictionary d inside offsets[]
// (multi_dictionary.hpp:43-56, 139-185, 293-304)
// ---------------------------------------------------------------------------
struct multi_packed_builder {
    static std::string type() { return "multi_packed"; }
    static constexpr dict_kind kind = dict_kind::multi_packed;

    void init() {
        m_targets.assign(kNumSelectors, {});
        m_size = kReserved;
        m_start_offsets.clear();
        m_offsets.clear();
        m_table.assign(kMaxEntrySize, 0);
    }
    // NOTE the reference's full() is global (multi_dictionary.hpp:123-125): the
    // per-context cap comes from the DSF builder taking at most kNumEntries
    // blocks per context (dictionary_builders.hpp:59-72).
    bool full() const { return m_size == kNumSelectors * kNumEntries; }
    bool append(uint32_t const* entry, uint32_t entry_size, uint32_t dictionary_id) {
        if (full()) return false;
        m_targets[dictionary_id].emplace_back(entry, entry + entry_size);
        ++m_size;
        return true;
    }
    void build() {
        std::vector<entry_t> all;
        for (auto const& t : m_targets) all.insert(all.end(), t.begin(), t.end());
        for (auto const& e : detail::pack_compact(std::move(all)))
            m_table.insert(m_table.end(), e.begin(), e.end());
        std::vector<entry_t const*> ptrs;
        for (auto const& t : m_targets)
            for (auto const& e : t) ptrs.push_back(&e);
        auto pos = detail::first_occurrences(m_table, ptrs);
        size_t k = 0;
        for (uint32_t d = 0; d != kNumSelectors; ++d) {
            m_start_offsets.push_back(uint32_t(m_offsets.size()));
            for (uint32_t i = 0; i != kExceptions; ++i) m_offsets.push_back(0);
            for (uint32_t i = kExceptions; i != kReserved; ++i)
                m_offsets.push_back(detail::packed_offsets_word(run_size_of(i), 0));
            for (auto const& e : m_targets[d])
                m_offsets.push_back(detail::packed_offsets_word(uint32_t(e.size()), pos[k++]));
        }
        m_targets.clear();
    }

    // file: u32 m_size, u32 n_start, u32 n_off, u32 n_tab, start[], offsets[], table[]  (:70-91)
    void write(std::vector<uint8_t>& out) const {
        detail::put_u32(out, m_size);
        detail::put_u32(out, uint32_t(m_start_offsets.size()));
        detail::put_u32(out, uint32_t(m_offsets.size()));
        detail::put_u32(out, uint32_t(m_table.size()));
        detail::put_u32s(out, m_start_offsets);
        detail::put_u32s(out, m_offsets);
        detail::put_u32s(out, m_table);
    }
    void load(uint8_t const* bytes, size_t len) {
        detail::reader r{bytes, bytes + len};
        m_size = r.u32();
        uint32_t n_start = r.u32(), n_off = r.u32(), n_tab = r.u32();
        r.u32s(m_start_offsets, n_start);
        r.u32s(m_offsets, n_off);
        r.u32s(m_table, n_tab);
        if (n_start != kNumSelectors) throw std::runtime_error("multi dictionary: bad selector count");
    }

    uint32_t size() const { return m_size; }
    uint32_t num_dictionaries() const { return kNumSelectors; }
    // number of offset slots (reserved ones included) dictionary d owns
    uint32_t slots(uint32_t d) const {
        uint32_t end = d + 1 == kNumSelectors ? uint32_t(m_offsets.size()) : m_start_offsets[d + 1];
        return end - m_start_offsets[d];
    }
    uint32_t size(uint32_t d, uint32_t i) const { return (m_offsets[m_start_offsets[d] + i] >> 24) + 1; }
    uint32_t offset(uint32_t d, uint32_t i) const { return m_offsets[m_start_offsets[d] + i] & 0xFFFFFF; }
    uint32_t const* get(uint32_t d, uint32_t i) const { return &m_table[offset(d, i)]; }
    std::vector<uint32_t> const& start_offsets() const { return m_start_offsets; }
    std::vector<uint32_t> const& offsets() const { return m_offsets; }
    std::vector<uint32_t> const& table() const { return m_table; }

    // Two maps per dictionary: all entries (16-bit codewords) and entries < 256
    // (8-bit codewords). Like the reference (:187-217) the scan stops `reserved`
    // slots before the end of each dictionary, which is also what keeps every
    // reachable index below 65536.
    void prepare_for_encoding() {
        m_maps.assign(2 * kNumSelectors, {});
        std::vector<uint32_t> zeros(256, 0);
        for (uint32_t d = 0; d != kNumSelectors; ++d) {
            m_maps[d].reserve_for(slots(d));
            m_maps[d + kNumSelectors].reserve_for(256);
            uint32_t i = kExceptions;
            for (uint32_t n = 256; n >= 16; n /= 2, ++i) {
                uint64_t h = hash_u32s(zeros.data(), n);
                m_maps[d].set(h, i);
                m_maps[d + kNumSelectors].set(h, i);
            }
            uint32_t own = slots(d);
            uint32_t n = own >= kReserved ? own - kReserved : 0;
            for (; i < n; ++i) {
                uint64_t h = hash_u32s(get(d, i), size(d, i));
                m_maps[d].set(h, i);
                if (i < 256) m_maps[d + kNumSelectors].set(h, i);
            }
        }
    }
    uint32_t lookup(uint32_t dictionary_id, uint32_t const* begin, uint32_t entry_size,
                    uint32_t log2_num_entries) const {
        auto const& map = m_maps[dictionary_id + (log2_num_entries == 8 ? kNumSelectors : 0)];
        return map.find(hash_u32s(begin, entry_size));
    }

private:
    uint32_t m_size = kReserved;
    std::vector<std::vector<entry_t>> m_targets;
    std::vector<uint32_t> m_start_offsets;
    std::vector<uint32_t> m_offsets;
    std::vector<uint32_t> m_table;
    std::vector<detail::flat_map> m_maps;
};

}  // namespace dint
