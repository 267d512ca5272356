// C ABI of the device decode path (include/dint_hip.h): dictionary staging,
// the host indexing pre-pass, and the kernel launches.
#include "dint_hip.h"

#include <hip/hip_runtime.h>

#include <cstring>  // (before rocPRIM: one of its headers calls the host memset without including it)
#include <rocprim/device/device_merge_sort.hpp>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "dint_kernels.hpp"
#include "dint_query_kernels.hpp"
#include "dint_stats_kernels.hpp"

namespace {

using namespace dint_dev;

constexpr uint32_t kEntries = 65536;   // reference dint_configuration.hpp:27
constexpr uint32_t kMaxEntry = 16;     // :25
constexpr uint32_t kSelectors = 6;     // :20
constexpr uint32_t kReserved = 7;      // EXCEPTIONS + 5 run codewords
constexpr uint32_t kBlock = 256;       // util.hpp:35

thread_local std::string g_hip_error;

// Every device / pinned allocation of the library goes through these: dint_debug_alloc_count says how many were made
// (the host-pointer calls — one block, one list at a time — must make none once their workspace is warm).
std::atomic<uint64_t> g_alloc_count{0};
template <class T>
hipError_t counted_malloc(T** p, size_t bytes) {
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc(reinterpret_cast<void**>(p), bytes);
}
template <class T>
hipError_t counted_host_malloc(T** p, size_t bytes) {
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    return hipHostMalloc(reinterpret_cast<void**>(p), bytes, hipHostMallocDefault);
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(call)                                  \
    do {                                               \
        if (!hip_ok((call), #call)) return DINT_ERR_HIP; \
    } while (0)

struct reader {
    const uint8_t* p;
    const uint8_t* end;
    bool u32(uint32_t* v) {
        if (end - p < 4) return false;
        std::memcpy(v, p, 4);
        p += 4;
        return true;
    }
    bool u32s(std::vector<uint32_t>& dst, size_t n) {
        if (size_t(end - p) < n * 4) return false;
        dst.resize(n);
        if (n) std::memcpy(dst.data(), p, n * 4);
        p += n * 4;
        return true;
    }
};

// Host-side normal form shared by the three file formats: per dictionary a list
// of (size, payload pointer) in codeword order.
struct parsed_dict {
    uint32_t num_dicts = 1;
    uint32_t entries = 0;                    // m_size of the file
    std::vector<uint32_t> start;             // first meta slot of each dictionary (+ end)
    std::vector<uint32_t> size;              // per meta slot
    std::vector<uint32_t> off;               // per meta slot: word offset into `table`
    std::vector<uint32_t> table;             // payload words
};

bool parse_rectangular(reader r, parsed_dict& d) {
    // u32 m_size, u32 table[m_size * 17]   (rectangular_dictionary.hpp:72-92)
    uint32_t m_size;
    if (!r.u32(&m_size) || m_size > kEntries) return false;
    std::vector<uint32_t> rows;
    if (!r.u32s(rows, size_t(m_size) * (kMaxEntry + 1))) return false;
    d.num_dicts = 1;
    d.entries = m_size;
    d.start = {0, kEntries};
    d.size.assign(kEntries, 1);
    d.off.assign(kEntries, 0);
    d.table.assign(kMaxEntry, 0);
    // builder::init() presets the reserved rows; the file then overwrites them
    for (uint32_t i = 2; i != kReserved; ++i) d.size[i] = 256u >> (i - 2);
    for (uint32_t i = 0; i != m_size; ++i) {
        const uint32_t* row = &rows[size_t(i) * (kMaxEntry + 1)];
        uint32_t s = row[kMaxEntry];
        if (s == 0 || s > 256) return false;
        d.size[i] = s;
        if (s <= kMaxEntry && i >= kReserved) {
            d.off[i] = uint32_t(d.table.size());
            d.table.insert(d.table.end(), row, row + s);
        } else {
            d.off[i] = 0;  // runs copy zeros
        }
    }
    return true;
}

bool parse_packed(reader r, bool multi, parsed_dict& d) {
    // single: u32 m_size, n_off, n_tab, offsets[], table[]          (single_dictionary.hpp:72-107)
    // multi : u32 m_size, n_start, n_off, n_tab, start[], offsets[], table[] (multi_dictionary.hpp:70-121)
    uint32_t m_size, n_start = 0, n_off, n_tab;
    if (!r.u32(&m_size)) return false;
    if (multi && !r.u32(&n_start)) return false;
    if (!r.u32(&n_off) || !r.u32(&n_tab)) return false;
    std::vector<uint32_t> starts, offsets;
    if (multi && (n_start != kSelectors || !r.u32s(starts, n_start))) return false;
    if (!r.u32s(offsets, n_off) || !r.u32s(d.table, n_tab)) return false;
    d.entries = m_size;
    if (multi) {
        d.num_dicts = kSelectors;
        d.start = starts;
        d.start.push_back(n_off);
        for (uint32_t k = 0; k != kSelectors; ++k)
            if (d.start[k] > d.start[k + 1]) return false;
    } else {
        d.num_dicts = 1;
        d.start = {0, std::max(n_off, kEntries)};
    }
    size_t slots = d.start.back();
    d.size.assign(slots, 1);
    d.off.assign(slots, 0);
    for (uint32_t i = 0; i != n_off; ++i) {
        uint32_t s = (offsets[i] >> 24) + 1, o = offsets[i] & 0xFFFFFF;
        d.size[i] = s;
        d.off[i] = o;
        // a copy() of entry i reads `s` words (runs: only ever zeros at offset 0)
        if (s <= kMaxEntry && size_t(o) + s > d.table.size()) return false;
        if (s > kMaxEntry && o != 0) return false;
    }
    return true;
}

}  // namespace

// A bundle schedule kept by its owner (a prepared block table: the same units launch after launch) instead of
// being rebuilt, three small kernels, before every launch.
struct sched_cache {
    void* d_mem = nullptr;  // (the layout launch_decode gives a slot's schedule workspace)
    size_t mem_bytes = 0;
    bool valid = false;
    // What the schedule was built from. bundle_schedule_kernel bakes this launch's bounds checks (stream bytes, output
    // capacity) and, multi-dictionary streams, the blocks' selector bytes into the unit records: a later launch with
    // another dictionary, stream, unit table, span table or a SMALLER capacity must not reuse them — it rebuilds.
    const void* dict = nullptr;
    const void* d_enc = nullptr;
    const void* d_units = nullptr;
    const void* d_spans = nullptr;
    size_t enc_bytes = 0, n_units = 0, out_capacity = 0;
    uint32_t only_full = 0;
    bool items_known = false;  // n_items has been read back (a prepared unit table does, once)
    uint32_t n_items = 0;      // work items of the unit queue: 0 = every unit is a bundle member
    bool matches(const void* dd, const void* enc, size_t eb, const void* units, size_t n, const void* spans, size_t cap,
                 uint32_t full) const {
        return valid && dict == dd && d_enc == enc && enc_bytes == eb && d_units == units && n_units == n && d_spans == spans &&
               cap >= out_capacity && only_full == full;
    }
};

struct dint_dict {
    int kind = 0;
    int device = 0;
    uint32_t num_dicts = 1;
    uint32_t entries = 0;
    uint32_t compute_units = 0;
    // host copies used by dint_index_stream
    std::vector<uint32_t> h_start;  // per dictionary first meta slot (+ end)
    std::vector<uint32_t> h_size;   // per meta slot
    std::vector<uint32_t> h_hot_k;  // per dictionary: codewords below it have their integers in the LDS image
    std::vector<uint8_t> h_slow;    // per meta slot: a slow entry (not on chip whatever its index)
    uint64_t launches = 0;          // decode launches so far (slot = launch % kQueueSlots)
    // device buffers
    void* d_block = nullptr;  // one allocation: gmeta | rows | gtable | LDS image | descriptors
    uint32_t* d_image = nullptr;
    dint_dev::dict_desc* d_descs = nullptr;
    uint32_t hot_entries = 0;
    uint32_t table_words = 0;
    dint_dev::dict_view view{};
    // timing events of the decode kernel: one pair per queue slot (launches on different streams may
    // interleave); dint_last_kernel_ms reads the pair of the most recent launch
    int last_slot = -1;
    // work-queue counters: one slot per in-flight launch, recycled round-robin
    // behind the event of the launch that used the slot last
    static constexpr uint32_t kQueueSlots = 64;
    uint32_t* d_queues = nullptr;
    // the bundle schedule of a launch (22 bytes per unit): a few workspaces taken in turn — a launch waits (on the host,
    // normally not at all) until the launch that last used its workspace is done. (One per queue slot was 64
    // allocations of half a gigabyte for a block-granular table of 2e7 units.)
    static constexpr uint32_t kSchedSlots = 4;
    uint8_t* d_sched[kSchedSlots] = {};
    size_t sched_cap[kSchedSlots] = {};
    int sched_user[kSchedSlots] = {-1, -1, -1, -1};  // the queue slot of the launch that used it last
    hipEvent_t slot_done[kQueueSlots] = {};
    hipEvent_t slot_start[kQueueSlots] = {}, slot_stop[kQueueSlots] = {};
    bool slot_used[kQueueSlots] = {};
    std::atomic<uint32_t> next_slot{0};
    std::mutex launch_mutex;
    // The host-pointer calls (dint_decode_list_host, dint_decode_block_host, dint_list_cache_create): a stream of their
    // own, one pinned and one device buffer that only grow — after the first call of a size no allocation, no
    // device-wide synchronisation, one wait for the stream per call. One call at a time per dictionary (host_mutex).
    std::mutex host_mutex;
    hipStream_t host_stream = nullptr;
    uint8_t* h_pin = nullptr;
    uint8_t* d_host = nullptr;
    size_t h_pin_cap = 0, d_host_cap = 0;
    sched_cache host_sched;  // (memory only: the schedule of a host-pointer call is rebuilt every time — other list, same addresses)
};

namespace {

// Device layout of a dictionary file. One allocation [heads | tails | goff | gtable | LDS image | descriptors],
// one record per offsets slot of the file (multi: the 6 dictionaries back to back):
//   heads[i] = 16 bytes: the metadata word (size-1) << 24 | kMetaCold | kMetaSlow, or | staging cells << 20
//              (1: up to 6 integers, 2: up to 14, 3), then the entry's first six integers as u16 — what a cold
//              slot fetches with ONE lane request, addressed by the slot value alone, and all a codeword of up
//              to 6 integers needs; runs carry only their size (their source is the zero region of the LDS image);
//   tails[i] = 32 bytes: integers 6..21 as u16 (the second and third request of the larger cold codewords);
//   goff[i]  = the entry's word offset into gtable (slow path only);
//   gtable   = [the file's payload words][16 words of padding], 32-bit: the slow path's source;
//   LDS image = [256 u16 zeros]{[hot meta of dictionary d: hot_k[d] words]}[hot payloads as u16];
//   hot meta = (size-1) << 24 | byte offset of the payload inside the image (runs -> the zeros), one dummy
//              word behind the last; the two exception markers: kMetaException (one integer, from one staging cell);
//   payloads = the union of the hot entries' table intervals, each word once (the packed formats nest
//              short entries inside long ones: single_dictionary.hpp:109-160), as u16.
// 16 bits per integer on chip: every value of a DSF dictionary built from d-gaps is far below 65536. An
// entry that does hold a larger value is SLOW: kMetaSlow in its metadata, never in the hot payloads or the
// rows; the kernel writes its integers straight from gtable (dint_kernels.hpp, slow_stores).
// Hot = codewords below hot_k[d]: the DSF builder appends entries in decreasing corpus n-gram
// frequency (dictionary_builders.hpp:61-72), so "index < K" is the hotness test, one compare
// in the kernel. (Picking the hot set by measured USE counts behind a bitmap + rank remap was
// tried in round 1: 67% instead of 60% of the lookups on chip, paid for by the longer lookup — no gain.)
// The image budget is split evenly between the dictionaries of a multi file.
struct hot_layout {
    std::vector<uint32_t> image;
    std::vector<dict_desc> descs;
    uint32_t hot_entries = 0;
    uint32_t long_bitmap_word = 0;  // single-dictionary files: where the long-entry bitmap sits in the image (words)
};

uint32_t entry_payload_words(parsed_dict const& pd, uint32_t d, uint32_t i) {
    const uint32_t sz = pd.size[pd.start[d] + i];
    return (i >= kReserved && sz <= kMaxEntry) ? sz : 0;  // runs and the exception rows copy zeros
}

bool entry_is_wide(parsed_dict const& pd, uint32_t d, uint32_t i) {
    const uint32_t pw = entry_payload_words(pd, d, i), o = pd.off[pd.start[d] + i];
    for (uint32_t w = 0; w != pw; ++w)
        if (pd.table[o + w] > 0xFFFFu) return true;
    return false;
}

int choose_hot_set(parsed_dict const& pd, hot_layout& out) {
    const uint32_t nd = pd.num_dicts;
    // (single-dictionary files: 8 KB of the image are the long-entry bitmap, kLongBitmapWords behind the zeros)
    const uint32_t bitmap_halves = (DINT_LEAN_SEGMENT == 2 && nd == 1) ? 2 * kLongBitmapWords : 0;  // (only a build with decode_segment_v4 reads it)
    const uint64_t share = (2 * uint64_t(kHotImageWords) - kZeroHalves - bitmap_halves - 8) / nd - 8;  // in u16 units
    std::vector<uint8_t> covered(pd.table.size(), 0);
    std::vector<uint32_t> hot_k(nd, 0);
    // pass 1: how many codewords of each dictionary fit its share (a meta word = 2 units, a payload integer = 1)
    for (uint32_t d = 0; d != nd; ++d) {
        const uint32_t n_entries = std::min<uint32_t>(pd.start[d + 1] - pd.start[d], kEntries);
        uint64_t units = 0;
        uint32_t k = 0;
        for (; k != n_entries; ++k) {
            const bool wide = entry_is_wide(pd, d, k);
            const uint32_t pw = wide ? 0 : entry_payload_words(pd, d, k);
            const uint32_t o = pd.off[pd.start[d] + k];
            uint32_t add = 2;  // (the dummy word behind the metas comes out of the 8 spare units per dictionary)
            for (uint32_t w = 0; w != pw; ++w) add += covered[o + w] ? 0u : 1u;
            if (units + add > share) break;
            units += add;
            for (uint32_t w = 0; w != pw; ++w) covered[o + w] = 1;
        }
        hot_k[d] = k;
    }
    // pass 2: lay the image out (u16 units; the metas sit on word boundaries)
    std::vector<uint16_t> halves(kZeroHalves + bitmap_halves, 0);
    out.long_bitmap_word = bitmap_halves ? kLongBitmapWordAt : 0;
    if (bitmap_halves) {
        // bit i: codeword i is cold and its integers do not fit its 16-byte head (7..16 of them: two or three staging cells)
        const uint32_t n_entries = std::min<uint32_t>(pd.start[1] - pd.start[0], kEntries);
        for (uint32_t i = std::max<uint32_t>(2, hot_k[0]); i < n_entries; ++i) {
            const uint32_t pw = entry_payload_words(pd, 0, i);
            if (pw > 6 && !entry_is_wide(pd, 0, i)) halves[kZeroHalves + 2 * (i >> 5) + ((i >> 4) & 1u)] |= uint16_t(1u << (i & 15u));
        }
    }
    out.descs.assign(nd, dict_desc{});
    out.hot_entries = 0;
    for (uint32_t d = 0; d != nd; ++d) {
        out.descs[d].meta_base = pd.start[d];
        // (at least the two exception markers count as hot: the kernel takes "slot value >= hot_k" for "this
        // slot's staging cell receives a row", and the cell of a marker holds its literal)
        out.descs[d].hot_k = std::max<uint32_t>(2, hot_k[d]);
        out.descs[d].hot_base = uint32_t(halves.size() / 2);  // word offset of the dictionary's hot metas
        out.descs[d].pad = 0;
        // (+1: a dummy word behind the metas, what the lanes of cold slots read)
        halves.resize(halves.size() + 2 * (std::max<uint32_t>(2, hot_k[d]) + 1), 0);
        out.hot_entries += hot_k[d];
    }
    std::vector<uint32_t> where(pd.table.size(), 0);
    for (size_t w = 0; w != pd.table.size(); ++w)
        if (covered[w]) {
            where[w] = uint32_t(halves.size());
            halves.push_back(uint16_t(pd.table[w]));
        }
    for (uint32_t d = 0; d != nd; ++d)
        for (uint32_t i = 0; i != std::max<uint32_t>(2, hot_k[d]); ++i) {
            const uint32_t sz = pd.size[pd.start[d] + i];
            uint32_t m;
            if (i < 2) m = kMetaException;  // (the exception markers)
            else if (entry_is_wide(pd, d, i)) m = ((sz - 1) << 24) | kMetaCold | kMetaSlow;
            else m = ((sz - 1) << 24) | (entry_payload_words(pd, d, i) ? 2 * where[pd.off[pd.start[d] + i]] : 0u);  // 0: the zero region
            const size_t at = 2 * (size_t(out.descs[d].hot_base) + i);
            halves[at] = uint16_t(m);
            halves[at + 1] = uint16_t(m >> 16);
        }
    while (halves.size() % 8) halves.push_back(0);
    out.image.assign(halves.size() / 2, 0);
    std::memcpy(out.image.data(), halves.data(), halves.size() * 2);
    if (out.image.size() > kHotImageWords || 2 * halves.size() > kMetaOffMask) return DINT_ERR_FORMAT;
    return DINT_OK;
}

int upload_hot_set(dint_dict& dd, hot_layout const& lay) {
    HIP_TRY(hipSetDevice(dd.device));
    HIP_TRY(hipMemcpy(dd.d_image, lay.image.data(), lay.image.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd.d_descs, lay.descs.data(), lay.descs.size() * sizeof(dict_desc), hipMemcpyHostToDevice));
    dd.view.hot_words = uint32_t(lay.image.size());
    dd.view.first = lay.descs[0];
    dd.view.long_bitmap_word = lay.long_bitmap_word;
    dd.hot_entries = lay.hot_entries;
    dd.h_hot_k.clear();
    for (auto const& d : lay.descs) dd.h_hot_k.push_back(d.hot_k);
    return DINT_OK;
}

int stage_dictionary(dint_dict& dd, parsed_dict const& pd) {
    const size_t slots = pd.size.size();
    dd.h_slow.assign(slots, 0);
    std::vector<uint32_t> heads(slots * 4, 0);   // per slot: metadata word, integers 0..5 as u16
    std::vector<uint16_t> tails(slots * 16, 0);  // per slot: integers 6..21 as u16
    std::vector<uint32_t> goff(slots, 0);
    std::vector<uint32_t> gtable(pd.table);
    gtable.resize(gtable.size() + kMaxEntry, 0);
    for (uint32_t d = 0; d != pd.num_dicts; ++d)
        for (uint32_t i = 0; i != pd.start[d + 1] - pd.start[d]; ++i) {
            const size_t slot = pd.start[d] + i;
            const uint32_t sz = pd.size[slot];
            if (sz == 0 || sz > 256) return DINT_ERR_FORMAT;
            const uint32_t pw = (i >= kReserved && sz <= kMaxEntry) ? sz : 0;
            if (pw == 0) {  // runs copy zeros: their source is the zero region at the start of the LDS image
                heads[slot * 4] = i < 2 ? kMetaException : (sz - 1) << 24;
                continue;
            }
            bool slow = false;
            for (uint32_t w = 0; w != pw; ++w) slow = slow || pd.table[pd.off[slot] + w] > 0xFFFFu;
            // (goff: every entry's offset into gtable — a tile with more cold codewords than staging cells sends
            // the surplus through the slow path too)
            const uint32_t cells = pw <= 6 ? 1u : pw <= 14 ? 2u : 3u;
            heads[slot * 4] = ((sz - 1) << 24) | kMetaCold | (slow ? kMetaSlow : cells << 20);
            if (slow) dd.h_slow[slot] = 1;
            goff[slot] = pd.off[slot];
            if (!slow) {
                uint16_t* const h = reinterpret_cast<uint16_t*>(&heads[slot * 4 + 1]);
                for (uint32_t w = 0; w != pw; ++w) (w < 6 ? h[w] : tails[slot * 16 + w - 6]) = uint16_t(pd.table[pd.off[slot] + w]);
            }
        }
    hot_layout lay;
    const int st = choose_hot_set(pd, lay);
    if (st != DINT_OK) return st;

    HIP_TRY(hipSetDevice(dd.device));
    // One allocation, a multiple of 2 MB, for everything the kernel reads at random (heads, tails, payload
    // table) and at start (LDS image, descriptors): the randomly gathered tables then sit in as few and as
    // large page-table fragments as the driver can give, whatever the state of the memory pool.
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_heads = up256(heads.size() * 4), b_tails = up256(tails.size() * 2), b_goff = up256(goff.size() * 4),
                 b_table = up256(gtable.size() * 4);
    const size_t b_image = up256(size_t(kHotImageWords) * 4), b_descs = up256(pd.num_dicts * sizeof(dict_desc));
    const size_t two_mb = size_t(2) << 20;
    const size_t b_tables = b_heads + b_tails + b_goff + b_table;
    const size_t total = (b_tables + b_image + b_descs + two_mb - 1) / two_mb * two_mb;
    if (b_tables >= (size_t(1) << 31)) return DINT_ERR_FORMAT;
    HIP_TRY(counted_malloc(&dd.d_block, total));
    uint8_t* base = static_cast<uint8_t*>(dd.d_block);
    dd.d_image = reinterpret_cast<uint32_t*>(base + b_tables);
    dd.d_descs = reinterpret_cast<dint_dev::dict_desc*>(base + b_tables + b_image);
    HIP_TRY(hipMemcpy(base, heads.data(), heads.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + b_heads, tails.data(), tails.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + b_heads + b_tails, goff.data(), goff.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + b_heads + b_tails + b_goff, gtable.data(), gtable.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(counted_malloc(&dd.d_queues, size_t(dint_dict::kQueueSlots) * (kQueueShards + 1) * kQueueStride * 4));
    for (uint32_t i = 0; i != dint_dict::kQueueSlots; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&dd.slot_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&dd.slot_start[i]));
        HIP_TRY(hipEventCreate(&dd.slot_stop[i]));
    }
    dd.view.tables = base;
    dd.view.tables_bytes = uint32_t(b_tables);
    dd.view.heads_base = 0;
    dd.view.tails_base = uint32_t(b_heads);
    dd.view.goff_base = uint32_t(b_heads + b_tails);
    dd.view.gtable_base = uint32_t(b_heads + b_tails + b_goff);
    dd.view.lds_image = dd.d_image;
    dd.view.descs = dd.d_descs;
    dd.table_words = uint32_t(gtable.size());
    return upload_hot_set(dd, lay);
}

inline uint16_t ld16(const uint8_t* p) {
    uint16_t v;
    std::memcpy(&v, p, 2);
    return v;
}

const uint8_t* read_vbyte(const uint8_t* in, const uint8_t* end, uint32_t* val) {
    uint32_t v = 0;
    for (unsigned shift = 0; in != end; shift += 7) {
        uint8_t c = *in++;
        v += uint32_t(c & 127) << (shift & 31);
        if (c & 128) {
            *val = v;
            return in;
        }
    }
    return nullptr;
}

}  // namespace

namespace {
template <class T>
struct device_buffer {  // grow-only workspace
    T* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t need) {
        if (need <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(need + need / 2, 1024);
        if (!hip_ok(counted_malloc(&p, want * sizeof(T)), "counted_malloc(workspace)")) return false;
        cap = want;
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
}  // namespace

struct dint_query_index {
    const dint_dict* docs = nullptr;
    const uint8_t* d_index = nullptr;
    size_t index_bytes = 0;
    size_t n_blocks = 0;
    std::vector<uint32_t> list_first;  // n_lists + 1
    std::vector<uint32_t> list_len;    // postings per list
    dint_block_ref* d_blocks = nullptr;
    uint32_t* d_block_max = nullptr;
    uint32_t* d_needed = nullptr;   // n_blocks, zero between rounds
    uint32_t* d_rank = nullptr;     // n_blocks
    uint32_t* d_touched = nullptr;  // n_blocks
    uint32_t* d_n_touched = nullptr;  // two counters: {blocks a round touched, short pages of a page decode}
    bool claims_dirty = false;        // d_needed may hold claim flags of a call that did not run to its end
    // one call = one host-to-device copy (everything the call's kernels read from the host, staged in pinned memory),
    // one clear (every counter the call's launches count in), the launches, one copy back
    void* h_stage = nullptr;
    void* d_stage = nullptr;
    size_t h_stage_cap = 0;
    device_buffer<uint32_t> inputs, ctrl;
    device_buffer<uint32_t> cand, target, probe, fprobe, tails, spans, bases;
    device_buffer<dint_block_ref> sub;
    device_buffer<dint_unit> units;
    device_buffer<uint64_t> ends;
    device_buffer<uint8_t> gaps_left;
    device_buffer<unsigned long long> freq_sums;
    device_buffer<uint32_t> freq_counts;  // and_query<true>: per term, the blocks its matches fell into
    std::mutex mutex;
};

// ---- options (dint_set_option): process-wide switches for tests and measurements, read by the entry points with one
// relaxed atomic load — no environment variable is looked at anywhere in this library -----------------------------------
namespace {
struct option_def {
    const char* key;
    long long def;
};
constexpr option_def kOptionDefs[DINT_OPT_COUNT_] = {
    {"bundles", 1},                 // DINT_OPT_BUNDLES
    {"index_concurrent", 1},        // DINT_OPT_INDEX_CONCURRENT
    {"query_lean_pages", -1},       // DINT_OPT_QUERY_LEAN_PAGES (-1: every page decode takes the one-launch form)
    {"query_tail_pages", 4},        // DINT_OPT_QUERY_TAIL_PAGES
    {"query_fused_pages", 2},       // DINT_OPT_QUERY_FUSED_PAGES
};
std::atomic<long long> g_options[DINT_OPT_COUNT_] = {{1}, {1}, {-1}, {4}, {2}};
inline long long opt(int which) { return g_options[which].load(std::memory_order_relaxed); }
}  // namespace

extern "C" {

int dint_abi_version(void) { return DINT_ABI_VERSION; }

int dint_set_option(int option, long long value) {
    if (option < 0 || option >= DINT_OPT_COUNT_) return DINT_ERR_ARG;
    if (option != DINT_OPT_QUERY_LEAN_PAGES && value < 0) return DINT_ERR_ARG;
    g_options[option].store(value, std::memory_order_relaxed);
    return DINT_OK;
}

int dint_get_option(int option, long long* value) {
    if (option < 0 || option >= DINT_OPT_COUNT_ || !value) return DINT_ERR_ARG;
    *value = opt(option);
    return DINT_OK;
}

const char* dint_option_name(int option) { return option < 0 || option >= DINT_OPT_COUNT_ ? nullptr : kOptionDefs[option].key; }

int dint_reset_options(void) {
    for (int i = 0; i != DINT_OPT_COUNT_; ++i) g_options[i].store(kOptionDefs[i].def, std::memory_order_relaxed);
    return DINT_OK;
}

const char* dint_strerror(int status) {
    switch (status) {
        case DINT_OK: return "ok";
        case DINT_ERR_ARG: return "bad argument";
        case DINT_ERR_FORMAT: return "malformed dictionary or stream";
        case DINT_ERR_HIP: return "HIP runtime error";
        case DINT_ERR_NO_DEVICE: return "no such device";
        case DINT_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char* dint_last_hip_error(void) { return g_hip_error.c_str(); }

int dint_device_count(int* count) {
    if (!count) return DINT_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return DINT_OK;
}

int dint_dict_create(int kind, const void* file_bytes, size_t len, int device, dint_dict** out) {
    if (!file_bytes || !out) return DINT_ERR_ARG;
    *out = nullptr;
    parsed_dict pd;
    reader r{static_cast<const uint8_t*>(file_bytes), static_cast<const uint8_t*>(file_bytes) + len};
    bool ok;
    switch (kind) {
        case DINT_DICT_RECTANGULAR: ok = parse_rectangular(r, pd); break;
        case DINT_DICT_SINGLE_PACKED: ok = parse_packed(r, false, pd); break;
        case DINT_DICT_MULTI_PACKED: ok = parse_packed(r, true, pd); break;
        default: return DINT_ERR_ARG;
    }
    if (!ok) return DINT_ERR_FORMAT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return DINT_ERR_NO_DEVICE;
    }
    dint_dict* dd = new (std::nothrow) dint_dict;
    if (!dd) return DINT_ERR_NOMEM;
    dd->kind = kind;
    dd->device = device;
    dd->num_dicts = pd.num_dicts;
    dd->entries = pd.entries;
    dd->h_start = pd.start;
    dd->h_size = pd.size;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) {
        delete dd;
        return DINT_ERR_HIP;
    }
    dd->compute_units = uint32_t(prop.multiProcessorCount);
    int st = stage_dictionary(*dd, pd);
    if (st != DINT_OK) {
        dint_dict_destroy(dd);
        return st;
    }
    // the kernels need the whole 160 KiB of LDS: dynamic = what their static __shared__ variables leave of it
    // (the query kernels carry a few static words for their round tail)
    auto all_lds = [](const void* f, size_t want) {
        hipFuncAttributes fa{};
        if (!hip_ok(hipFuncGetAttributes(&fa, f), "hipFuncGetAttributes")) return false;
        const size_t room = size_t(kLdsWords) * 4 - std::min<size_t>(fa.sharedSizeBytes, size_t(kLdsWords) * 4);
        return hip_ok(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(std::min(want, room))), "hipFuncSetAttribute");
    };
    const size_t full = size_t(kLdsWords) * 4;
    if (!all_lds(reinterpret_cast<const void*>(&decode_single_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_multi_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_multi_bundles_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_single_index_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_multi_index_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_single_query_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_multi_query_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_single_query_fused_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&decode_multi_query_fused_kernel), full) ||
        !all_lds(reinterpret_cast<const void*>(&interpolative_tails_kernel), kTailLdsBytes)) {
        dint_dict_destroy(dd);
        return DINT_ERR_HIP;
    }
    *out = dd;
    return DINT_OK;
}

void dint_dict_destroy(dint_dict* dd) {
    if (!dd) return;
    (void)hipSetDevice(dd->device);
    if (dd->d_block) (void)hipFree(dd->d_block);
    for (auto e : dd->slot_start)
        if (e) (void)hipEventDestroy(e);
    for (auto e : dd->slot_stop)
        if (e) (void)hipEventDestroy(e);
    if (dd->d_queues) (void)hipFree(dd->d_queues);
    if (dd->host_stream) (void)hipStreamDestroy(dd->host_stream);
    if (dd->h_pin) (void)hipHostFree(dd->h_pin);
    if (dd->d_host) (void)hipFree(dd->d_host);
    if (dd->host_sched.d_mem) (void)hipFree(dd->host_sched.d_mem);
    for (auto p : dd->d_sched)
        if (p) (void)hipFree(p);
    for (auto e : dd->slot_done)
        if (e) (void)hipEventDestroy(e);
    delete dd;
}

int dint_dict_info_get(const dint_dict* dd, dint_dict_info* info) {
    if (!dd || !info) return DINT_ERR_ARG;
    info->kind = dd->kind;
    info->device = dd->device;
    info->num_dicts = dd->num_dicts;
    info->entries = dd->entries;
    info->hot_entries = dd->hot_entries;
    info->lds_bytes = dd->view.hot_words * 4;
    info->table_words = dd->table_words;
    info->compute_units = dd->compute_units;
    return DINT_OK;
}

void dint_free(void* p) { std::free(p); }

int dint_index_stream(const dint_dict* dd, const uint8_t* enc, size_t enc_bytes, uint32_t unit_ints,
                      dint_unit** units_out, size_t* n_units, uint64_t* total_ints, uint64_t* n_lists) {
    if (!dd || (!enc && enc_bytes) || !units_out || !n_units) return DINT_ERR_ARG;
    std::vector<dint_unit> units;
    const uint8_t* p = enc;
    const uint8_t* end = enc + enc_bytes;
    uint64_t out_pos = 0, lists = 0;
    const bool multi = dd->kind == DINT_DICT_MULTI_PACKED;
    const uint32_t cut = unit_ints ? std::min<uint32_t>(unit_ints, DINT_MAX_UNIT_INTS) : DINT_MAX_UNIT_INTS;

    while (p != end) {
        uint32_t n, universe;
        p = read_vbyte(p, end, &n);
        if (p) p = read_vbyte(p, end, &universe);
        if (!p) return DINT_ERR_FORMAT;
        uint32_t unit_start_int = 0;
        const uint8_t* unit_start = p;
        auto close_unit = [&](const uint8_t* at, uint32_t upto) {
            if (upto == unit_start_int) return;
            units.push_back({uint64_t(unit_start - enc), out_pos + unit_start_int, upto - unit_start_int,
                             uint32_t(lists)});
            unit_start = at;
            unit_start_int = upto;
        };
        if (!multi) {
            const uint32_t* size = dd->h_size.data();
            uint32_t i = 0;
            while (i < n) {
                if (i - unit_start_int >= cut) close_unit(p, i);
                if (end - p < 2) return DINT_ERR_FORMAT;
                uint32_t idx = ld16(p);
                if (idx >= 2) {
                    i += size[idx];
                    p += 2;
                } else {
                    i += 1;
                    p += idx == 1 ? 6 : 4;
                }
                if (p > end) return DINT_ERR_FORMAT;
            }
            if (i != n) return DINT_ERR_FORMAT;
            close_unit(p, n);
        } else {
            // one selector byte per 256 integers (vroom_env/dint_codecs.hpp:521-619)
            uint32_t done = 0;
            const uint32_t blocks_per_unit =
                std::max<uint32_t>(1, (std::min<uint32_t>(unit_ints ? unit_ints : DINT_MAX_UNIT_INTS, DINT_MAX_UNIT_INTS)) / kBlock);
            uint32_t blocks_in_unit = 0;
            while (done < n) {
                if (blocks_in_unit == blocks_per_unit) {
                    close_unit(p, done);
                    blocks_in_unit = 0;
                }
                uint32_t bsize = std::min<uint32_t>(kBlock, n - done);
                if (p == end) return DINT_ERR_FORMAT;
                uint32_t sc = *p++;
                if (sc >= 2 * kSelectors) return DINT_ERR_FORMAT;
                const bool narrow = sc >= kSelectors;
                const uint32_t dsel = narrow ? sc - kSelectors : sc;
                const uint32_t base = dd->h_start[dsel];
                const uint32_t limit = dd->h_start[dsel + 1] - base;
                uint32_t i = 0;
                while (i < bsize) {
                    if (end - p < (narrow ? 1 : 2)) return DINT_ERR_FORMAT;
                    uint32_t idx = narrow ? *p : ld16(p);
                    if (idx >= 2) {
                        if (idx >= limit) return DINT_ERR_FORMAT;
                        i += dd->h_size[base + idx];
                        p += narrow ? 1 : 2;
                    } else {
                        i += 1;
                        p += (narrow ? 1 : 2) + (idx == 1 ? 4 : 2);
                    }
                    if (p > end) return DINT_ERR_FORMAT;
                }
                if (i != bsize) return DINT_ERR_FORMAT;
                done += bsize;
                ++blocks_in_unit;
            }
            close_unit(p, n);
        }
        out_pos += n;
        ++lists;
    }
    dint_unit* mem = static_cast<dint_unit*>(std::malloc(std::max<size_t>(1, units.size()) * sizeof(dint_unit)));
    if (!mem) return DINT_ERR_NOMEM;
    if (!units.empty()) std::memcpy(mem, units.data(), units.size() * sizeof(dint_unit));
    *units_out = mem;
    *n_units = units.size();
    if (total_ints) *total_ints = out_pos;
    if (n_lists) *n_lists = lists;
    return DINT_OK;
}

// workspace of a schedule: [unit records 16 B x n][chunk bases 16 B x chunks][items u32 x n][block counts/offsets u32 x blocks]
// [n_items u32][sched u8 x n][item counts u8 x n]
struct sched_layout {
    size_t n_units, n_blocks, n_chunks, need;
    u32x4* d_urec = nullptr;
    uint64_t* d_cbase = nullptr;
    uint32_t *d_items = nullptr, *d_block = nullptr, *d_n_items = nullptr;
    uint8_t *d_sch = nullptr, *d_item_cnt = nullptr;
    explicit sched_layout(size_t n) : n_units(n), n_blocks((n + 255) / 256), n_chunks((n + kChunkUnits - 1) / kChunkUnits) {
        need = 16 * n_units + 16 * n_chunks + 4 * n_units + 4 * n_blocks + 4 + 2 * n_units;
    }
    void place(void* mem) {
        d_urec = reinterpret_cast<u32x4*>(mem);
        d_cbase = reinterpret_cast<uint64_t*>(d_urec + n_units);
        d_items = reinterpret_cast<uint32_t*>(d_cbase + 2 * n_chunks);
        d_block = d_items + n_units;
        d_n_items = d_block + n_blocks;
        d_sch = reinterpret_cast<uint8_t*>(d_n_items + 1);
        d_item_cnt = d_sch + n_units;
    }
};

static void run_schedule_kernels(const dint_dict* dd, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units, size_t n_units,
                                 size_t out_capacity, uint32_t only_full, const uint32_t* d_spans, const sched_layout& L, hipStream_t s) {
    hipLaunchKernelGGL(bundle_schedule_kernel, dim3(uint32_t(L.n_blocks)), dim3(256), 0, s, d_units, d_spans, uint64_t(n_units), d_enc,
                       uint64_t(enc_bytes), uint64_t(out_capacity), only_full, uint32_t(dd->kind == DINT_DICT_MULTI_PACKED), L.d_sch,
                       L.d_block, L.d_urec, L.d_cbase);
    if (dd->kind == DINT_DICT_MULTI_PACKED)
        hipLaunchKernelGGL(bundle_pack_kernel, dim3(uint32_t((L.n_chunks + 63) / 64)), dim3(64), 0, s, L.d_urec, uint64_t(n_units));
    hipLaunchKernelGGL(bundle_offsets_kernel, dim3(1), dim3(1024), 0, s, L.d_block, uint32_t(L.n_blocks), L.d_n_items);
    hipLaunchKernelGGL(bundle_items_kernel, dim3(uint32_t(L.n_blocks)), dim3(256), 0, s, L.d_sch, uint64_t(n_units), L.d_block, L.d_items,
                       L.d_item_cnt);
}

// (Re)build a kept schedule on stream `s` and remember what it was built from.
static int build_schedule(const dint_dict* dd, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units, size_t n_units,
                          size_t out_capacity, uint32_t only_full, const uint32_t* d_spans, sched_cache* cache, hipStream_t s) {
    HIP_TRY(hipSetDevice(dd->device));
    sched_layout L(n_units);
    cache->valid = false;
    if (cache->mem_bytes < L.need) {
        if (cache->d_mem) HIP_TRY(hipFree(cache->d_mem));  // (hipFree waits for the launches that read it)
        cache->d_mem = nullptr;
        cache->mem_bytes = 0;
        HIP_TRY(counted_malloc(&cache->d_mem, L.need));
        cache->mem_bytes = L.need;
    }
    L.place(cache->d_mem);
    run_schedule_kernels(dd, d_enc, enc_bytes, d_units, n_units, out_capacity, only_full, d_spans, L, s);
    HIP_TRY(hipGetLastError());
    cache->valid = true;
    cache->items_known = false;
    cache->dict = dd, cache->d_enc = d_enc, cache->enc_bytes = enc_bytes, cache->d_units = d_units, cache->n_units = n_units;
    cache->d_spans = d_spans, cache->out_capacity = out_capacity, cache->only_full = only_full;
    return DINT_OK;
}

static int launch_decode(const dint_dict* dd, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units,
                         size_t n_units, uint32_t* d_out, size_t out_capacity, uint64_t* d_end_off, void* stream,
                         uint32_t only_full, const uint32_t* d_spans = nullptr, uint32_t plus_one = 0,
                         const uint32_t* d_unit_base = nullptr, uint8_t* d_gaps_left = nullptr, sched_cache* cache = nullptr,
                         size_t schedule_from = 2, uint32_t* d_zeroed_queue = nullptr) {
    if (!dd) return DINT_ERR_ARG;
    if (n_units == 0) return DINT_OK;
    if (!d_enc || !d_units || !d_out || enc_bytes < 8) return DINT_ERR_ARG;  // slots are fetched 8 bytes at a time
    HIP_TRY(hipSetDevice(dd->device));
    decode_args a{};
    a.dict = dd->view;
    a.enc = d_enc;
    a.enc_bytes = enc_bytes;
    a.units = d_units;
    a.n_units = n_units;
    a.out = d_out;
    a.out_capacity = out_capacity;
    a.end_off = d_end_off;
    a.only_full = only_full;
    a.plus_one = plus_one;
    a.unit_base = d_unit_base;
    a.gaps_left = d_gaps_left;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t blocks_needed = (uint64_t(n_units) + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t grid = uint32_t(std::min<uint64_t>(blocks_needed, std::max<uint32_t>(1, dd->compute_units) * kBlocksPerCU));
    const size_t lds_bytes = (size_t(dd->view.hot_words) + kClassTableWords + kWavesPerBlock * kScratchWords) * 4;
    dint_dict* mut = const_cast<dint_dict*>(dd);
    // (an in-index launch — blocks, docIDs, freqs + 1 — runs the kernels compiled for that; the vroom kernels carry none of it)
    const bool index_launch = only_full != 0 || plus_one != 0 || d_unit_base != nullptr || d_gaps_left != nullptr;
    bool bundles_only = false;  // (set below: a kept schedule that is known to have left the unit queue empty)
    auto launch_kernel = [&]() {
        const bool multi = dd->kind == DINT_DICT_MULTI_PACKED;
        if (index_launch)
            hipLaunchKernelGGL(multi ? decode_multi_index_kernel : decode_single_index_kernel, dim3(grid), dim3(kBlockThreads), lds_bytes, s, a);
        else if (multi && bundles_only)
            hipLaunchKernelGGL(decode_multi_bundles_kernel, dim3(grid), dim3(kBlockThreads), lds_bytes, s, a);
        else
            hipLaunchKernelGGL(multi ? decode_multi_kernel : decode_single_kernel, dim3(grid), dim3(kBlockThreads), lds_bytes, s, a);
    };
    if (d_zeroed_queue && (n_units < schedule_from || (opt(DINT_OPT_BUNDLES) == 0))) {
        // the lean launch: the caller brings the (zeroed) queue counters, nothing is timed, nothing scheduled — one
        // API call (a query's pages: the host-side cost of a launch sequence is what a single query waits for)
        a.queue = d_zeroed_queue;
        a.chunk_queue = a.queue + kQueueShards * kQueueStride;
        a.n_shards = std::min<uint32_t>(kQueueShards, grid);
        a.sched = nullptr;
        a.items = nullptr;
        a.n_items = nullptr;
        a.item_cnt = nullptr;
        a.urec = nullptr;
        a.cbase = nullptr;
        a.spans = d_spans;
        launch_kernel();
        HIP_TRY(hipGetLastError());
        return DINT_OK;
    }
    std::lock_guard<std::mutex> lock(mut->launch_mutex);
    const uint32_t slot = mut->next_slot.fetch_add(1) % dint_dict::kQueueSlots;
    mut->launches += 1;
    if (mut->slot_used[slot]) HIP_TRY(hipEventSynchronize(mut->slot_done[slot]));  // normally long complete
    a.queue = mut->d_queues + size_t(slot) * (kQueueShards + 1) * kQueueStride;
    a.chunk_queue = a.queue + kQueueShards * kQueueStride;  // (the bundle path's counter: a line of its own behind the shards')
    a.n_shards = std::min<uint32_t>(kQueueShards, grid);
    HIP_TRY(hipMemsetAsync(a.queue, 0, size_t(kQueueShards + 1) * kQueueStride * 4, s));
    // tiny consecutive units are decoded several to a tile: schedule them (single-dictionary streams)
    a.sched = nullptr;
    a.items = nullptr;
    a.n_items = nullptr;
    a.item_cnt = nullptr;
    a.urec = nullptr;
    a.cbase = nullptr;
    a.spans = d_spans;
    bool start_recorded = false;
    // (schedule_from: a caller that decodes a handful of units at a time — a query's pages — does without the three
    // schedule launches: with fewer units than waves nothing is gained by sharing tiles)
    if (n_units >= schedule_from && n_units < 0xFFFFFFFFull &&
        !(opt(DINT_OPT_BUNDLES) == 0)) {
        sched_layout L(n_units);
        if (cache) {
            if (!cache->matches(dd, d_enc, enc_bytes, d_units, n_units, d_spans, out_capacity, only_full)) {
                // not built yet, or built for other buffers / a larger capacity: (re)built on this stream, and timed with
                // the launch — the event pair spans what the call put on the stream
                HIP_TRY(hipEventRecord(mut->slot_start[slot], s));
                start_recorded = true;
                const int st = build_schedule(dd, d_enc, enc_bytes, d_units, n_units, out_capacity, only_full, d_spans, cache, s);
                if (st != DINT_OK) return st;
            }
            L.place(cache->d_mem);
            bundles_only = !index_launch && cache->items_known && cache->n_items == 0;
        } else {
            const uint32_t ss = uint32_t(mut->launches % dint_dict::kSchedSlots);
            const int prev = mut->sched_user[ss];
            if (prev >= 0 && prev != int(slot) && mut->slot_used[prev]) HIP_TRY(hipEventSynchronize(mut->slot_done[prev]));
            mut->sched_user[ss] = int(slot);
            if (mut->sched_cap[ss] < L.need) {
                if (mut->d_sched[ss]) HIP_TRY(hipFree(mut->d_sched[ss]));
                mut->d_sched[ss] = nullptr;
                mut->sched_cap[ss] = 0;
                const size_t want = L.need + L.need / 4 + 4096;
                HIP_TRY(counted_malloc(&mut->d_sched[ss], want));
                mut->sched_cap[ss] = want;
            }
            L.place(mut->d_sched[ss]);
            HIP_TRY(hipEventRecord(mut->slot_start[slot], s));
            start_recorded = true;
            run_schedule_kernels(dd, d_enc, enc_bytes, d_units, n_units, out_capacity, only_full, d_spans, L, s);
        }
        uint8_t* const d_sch = L.d_sch;
        uint32_t* const d_items = L.d_items;
        uint32_t* const d_n_items = L.d_n_items;
        uint8_t* const d_item_cnt = L.d_item_cnt;
        u32x4* const d_urec = L.d_urec;
        uint64_t* const d_cbase = L.d_cbase;
        a.sched = d_sch;
        a.items = d_items;
        a.n_items = d_n_items;
        a.item_cnt = d_item_cnt;
        a.urec = d_urec;
        a.cbase = d_cbase;
    }
    if (!start_recorded) HIP_TRY(hipEventRecord(mut->slot_start[slot], s));
    launch_kernel();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(mut->slot_stop[slot], s));
    HIP_TRY(hipEventRecord(mut->slot_done[slot], s));
    mut->slot_used[slot] = true;
    mut->last_slot = int(slot);
    return DINT_OK;
}

int dint_decode_units(const dint_dict* dd, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units,
                      size_t n_units, uint32_t* d_out, size_t out_capacity, uint64_t* d_end_off, void* stream) {
    return launch_decode(dd, d_enc, enc_bytes, d_units, n_units, d_out, out_capacity, d_end_off, stream, 0);
}

struct dint_unit_table {
    const dint_dict* dict = nullptr;
    const uint8_t* d_enc = nullptr;
    size_t enc_bytes = 0;
    const dint_unit* d_units = nullptr;
    size_t n_units = 0;
    size_t out_capacity = 0;
    sched_cache sched;
    std::mutex mutex;
};

int dint_unit_table_create(const dint_dict* dd, const uint8_t* d_enc, size_t enc_bytes, const dint_unit* d_units, size_t n_units,
                           size_t out_capacity, void* stream, dint_unit_table** out) {
    if (!dd || !out || (n_units && (!d_enc || !d_units || enc_bytes < 8))) return DINT_ERR_ARG;
    *out = nullptr;
    auto* t = new (std::nothrow) dint_unit_table();
    if (!t) return DINT_ERR_NOMEM;
    t->dict = dd;
    t->d_enc = d_enc;
    t->enc_bytes = enc_bytes;
    t->d_units = d_units;
    t->n_units = n_units;
    t->out_capacity = out_capacity;
    if (n_units >= 2 && n_units < 0xFFFFFFFFull && !(opt(DINT_OPT_BUNDLES) == 0)) {
        int st = build_schedule(dd, d_enc, enc_bytes, d_units, n_units, out_capacity, 0, nullptr, &t->sched, static_cast<hipStream_t>(stream));
        if (st == DINT_OK) {  // how many work items the unit queue got (none: the bundles-only kernel serves the table)
            sched_layout L(n_units);
            L.place(t->sched.d_mem);
            uint32_t n_items = 0;
            if (!hip_ok(hipMemcpyAsync(&n_items, L.d_n_items, 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)), "hipMemcpyAsync") ||
                !hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize"))
                st = DINT_ERR_HIP;
            t->sched.n_items = n_items;
            t->sched.items_known = st == DINT_OK;
        }
        if (st != DINT_OK) {
            dint_unit_table_destroy(t);
            return st;
        }
    }
    *out = t;
    return DINT_OK;
}

void dint_unit_table_destroy(dint_unit_table* t) {
    if (!t) return;
    if (t->dict) (void)hipSetDevice(t->dict->device);
    if (t->sched.d_mem) (void)hipFree(t->sched.d_mem);
    delete t;
}

int dint_decode_unit_table(const dint_dict* dd, dint_unit_table* t, uint32_t* d_out, size_t out_capacity, uint64_t* d_end_off,
                           void* stream) {
    if (!dd || !t || t->dict != dd || out_capacity < t->out_capacity) return DINT_ERR_ARG;
    std::lock_guard<std::mutex> lock(t->mutex);  // (the cache may be rebuilt: a dictionary created under DINT_NO_BUNDLES has none)
    return launch_decode(dd, t->d_enc, t->enc_bytes, t->d_units, t->n_units, d_out, out_capacity, d_end_off, stream, 0, nullptr, 0, nullptr,
                         nullptr, &t->sched);
}

int dint_index_posting_lists(const uint8_t* index, size_t index_bytes, const uint64_t* list_offsets,
                             size_t n_lists, dint_block_ref** blocks_out, size_t* n_blocks,
                             uint64_t* total_postings) {
    if ((!index && index_bytes) || (!list_offsets && n_lists) || !blocks_out || !n_blocks) return DINT_ERR_ARG;
    std::vector<dint_block_ref> blocks;
    uint64_t out_pos = 0;
    for (size_t i = 0; i != n_lists; ++i) {
        if (list_offsets[i] >= index_bytes) return DINT_ERR_FORMAT;
        const uint8_t* p = index + list_offsets[i];
        const uint8_t* end = index + index_bytes;
        uint32_t n;
        const uint8_t* base = read_vbyte(p, end, &n);  // document_enumerator ctor, dict_posting_list.hpp:90-107
        if (!base || n == 0) return DINT_ERR_FORMAT;
        const uint64_t nb = (uint64_t(n) + kBlock - 1) / kBlock;
        const uint8_t* maxs = base;
        const uint8_t* endpoints = maxs + 4 * nb;
        const uint8_t* data = endpoints + 4 * (nb - 1);
        if (data > end) return DINT_ERR_FORMAT;
        uint32_t prev_max = uint32_t(-1);
        for (uint64_t b = 0; b != nb; ++b) {
            uint32_t endpoint = 0, mx;
            if (b) std::memcpy(&endpoint, endpoints + 4 * (b - 1), 4);
            std::memcpy(&mx, maxs + 4 * b, 4);
            dint_block_ref r;
            r.in_off = uint64_t(data - index) + endpoint;
            r.out_off = out_pos;
            r.n = (b + 1) * kBlock <= n ? kBlock : n % kBlock;
            r.base = prev_max + 1;
            r.max = mx;
            r.list = uint32_t(i);
            if (r.in_off > index_bytes) return DINT_ERR_FORMAT;
            blocks.push_back(r);
            out_pos += r.n;
            prev_max = mx;
        }
    }
    auto mem = static_cast<dint_block_ref*>(std::malloc(std::max<size_t>(1, blocks.size()) * sizeof(dint_block_ref)));
    if (!mem) return DINT_ERR_NOMEM;
    if (!blocks.empty()) std::memcpy(mem, blocks.data(), blocks.size() * sizeof(dint_block_ref));
    *blocks_out = mem;
    *n_blocks = blocks.size();
    if (total_postings) *total_postings = out_pos;
    return DINT_OK;
}

// A block table prepared for decoding: what depends on the table alone is computed once — the docs parts' unit
// table, byte spans and docID bases, the list of short (interpolative) blocks — and the per-call workspace (where
// the docs parts end, the freqs parts' units) lives here too, so that a decode is a sequence of launches and
// nothing else.
struct dint_block_table {
    int device = 0;
    size_t n_blocks = 0, n_tails = 0;
    bool owns_blocks = false;
    const dint_block_ref* d_blocks = nullptr;
    void* d_ws = nullptr;  // one allocation: everything below
    dint_unit* d_units = nullptr;       // docs parts
    uint32_t* d_spans = nullptr;
    uint32_t* d_bases = nullptr;
    uint32_t* d_tails = nullptr;        // indices of the short blocks, then their number
    uint64_t* d_ends = nullptr;         // per call: where each docs part ended = where the freqs part begins
    dint_unit* d_funits = nullptr;      // per call: freqs parts
    uint32_t* d_fspans = nullptr;
    uint8_t* d_gaps_left = nullptr;     // per call: blocks the decode kernels left as gaps
    bool spans_exact = false;           // the docs parts' byte spans have been cut down to their ends (after the first decode)
    // From the third decode on nothing but the decode kernels, the interpolative decoder and the clean-up run:
    // the first decode learns where the docs parts end (exact spans, the freqs parts' units), the second builds
    // the two bundle schedules from them, and both are a property of the index, not of the call.
    uint32_t decodes = 0;
    sched_cache docs_sched, freqs_sched;
    bool freqs_units_ready = false;
    uint64_t max_out_end = 0;           // max over the blocks of out_off + n: a decode whose out_capacity is below it skips blocks
    // Once the freqs parts' units are known (second decode on), their launch depends on nothing the docs launch produces:
    // it runs on a stream of the table's own, beside the docs launch and the short blocks' decoder, between two events
    // on the caller's stream (each launch is a quarter of a millisecond of persistent workgroups: one's ramp and tail
    // under the other's body).
    hipStream_t side = nullptr, side2 = nullptr;  // (side2: the short blocks' decoder)
    hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
};

namespace {
int block_table_prepare(dint_block_table& t, const dint_block_ref* d_blocks, size_t n_blocks, size_t index_bytes, hipStream_t s) {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_units = up(n_blocks * sizeof(dint_unit)), b_u32 = up((n_blocks + 1) * 4), b_u64 = up(n_blocks * 8),
                 b_u8 = up(n_blocks);
    HIP_TRY(counted_malloc(&t.d_ws, 2 * b_units + 4 * b_u32 + b_u64 + b_u8));
    uint8_t* p = static_cast<uint8_t*>(t.d_ws);
    t.d_units = reinterpret_cast<dint_unit*>(p), p += b_units;
    t.d_funits = reinterpret_cast<dint_unit*>(p), p += b_units;
    t.d_ends = reinterpret_cast<uint64_t*>(p), p += b_u64;
    t.d_spans = reinterpret_cast<uint32_t*>(p), p += b_u32;
    t.d_fspans = reinterpret_cast<uint32_t*>(p), p += b_u32;
    t.d_bases = reinterpret_cast<uint32_t*>(p), p += b_u32;
    t.d_tails = reinterpret_cast<uint32_t*>(p), p += b_u32;
    t.d_gaps_left = p;
    t.d_blocks = d_blocks;
    t.n_blocks = n_blocks;
    const uint32_t tb = 256, grid = uint32_t((n_blocks + tb - 1) / tb);
    HIP_TRY(hipMemsetAsync(t.d_tails + n_blocks, 0, 4, s));
    HIP_TRY(hipMemsetAsync(t.d_ends, 0, n_blocks * 8, s));  // (a block a decode skips leaves its end offset untouched)
    hipLaunchKernelGGL(collect_tails_kernel, dim3(grid), dim3(tb), 0, s, d_blocks, uint64_t(n_blocks), t.d_tails, t.d_tails + n_blocks);
    hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, d_blocks, static_cast<const uint64_t*>(nullptr),
                       uint64_t(n_blocks), uint64_t(index_bytes), t.d_units, t.d_spans, t.d_bases);
    HIP_TRY(hipGetLastError());
    uint32_t n_tails = 0;
    HIP_TRY(hipMemcpyAsync(&n_tails, t.d_tails + n_blocks, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    t.n_tails = n_tails;
    return DINT_OK;
}

// docs parts -> docIDs, freqs parts -> freqs: launches only, nothing waited for
int block_table_decode(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index, size_t index_bytes,
                       const dint_block_table& t, uint32_t* d_docids, uint32_t* d_freqs, size_t out_capacity, hipStream_t s) {
    const size_t n_blocks = t.n_blocks;
    const uint32_t tb = 256, grid = uint32_t((n_blocks + tb - 1) / tb);
    const uint32_t tgrid = uint32_t((t.n_tails + kTailLanes - 1) / kTailLanes);  // one wave per kTailLanes short blocks
    HIP_TRY(hipMemsetAsync(t.d_gaps_left, 0, n_blocks, s));
    // docs parts of the full blocks through the DINT kernel (docIDs formed in the expansion); then, where they end,
    // their freqs parts; the short blocks — both parts of a block in one lane — through the interpolative decoder
    dint_block_table& mt = const_cast<dint_block_table&>(t);
    // What a decode learns for the next ones (exact spans, the freqs parts' units, the schedules) is kept only when
    // this call decodes EVERY block: with an out_capacity below some block's end the kernels skip that block and leave
    // its end offset unwritten — nothing may be derived from it.
    const bool covers = out_capacity >= t.max_out_end;
    const bool keep = t.owns_blocks && covers && t.decodes >= 1;  // (a one-shot table never reaches its second decode)
    const bool concurrent_ok = opt(DINT_OPT_INDEX_CONCURRENT) != 0;
    // From the second decode of a table on (the side streams are the table's): the freqs launch beside the docs launch once
    // its units are known, and the short blocks' decoder beside both.
    const bool side_freqs = d_freqs && t.freqs_units_ready && keep && concurrent_ok;
    const bool side_tails = tgrid != 0 && keep && concurrent_ok;
    if (side_freqs || side_tails) {
        if (!mt.side) {  // (all five or none: a table with half of them would trip over the missing ones on its next decode)
            hipStream_t s1 = nullptr, s2 = nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            int prio_low = 0, prio_high = 0;  // (the short blocks' few, long-lived waves first: the DINT launches fill in around them)
            const bool ok = hip_ok(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking), "hipStreamCreateWithFlags") &&
                            hip_ok(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high), "hipDeviceGetStreamPriorityRange") &&
                            hip_ok(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio_high), "hipStreamCreateWithPriority") &&
                            hip_ok(hipEventCreateWithFlags(&e0, hipEventDisableTiming), "hipEventCreateWithFlags") &&
                            hip_ok(hipEventCreateWithFlags(&e1, hipEventDisableTiming), "hipEventCreateWithFlags") &&
                            hip_ok(hipEventCreateWithFlags(&e2, hipEventDisableTiming), "hipEventCreateWithFlags");
            if (!ok) {
                if (s1) (void)hipStreamDestroy(s1);
                if (s2) (void)hipStreamDestroy(s2);
                for (hipEvent_t e : {e0, e1, e2})
                    if (e) (void)hipEventDestroy(e);
                return DINT_ERR_HIP;
            }
            mt.side = s1, mt.side2 = s2, mt.fork = e0, mt.join = e1, mt.join2 = e2;
        }
        HIP_TRY(hipEventRecord(mt.fork, s));  // (what the caller put on its stream before this call — the index — is there)
    }
    hipStream_t fs = s;  // the freqs launch's stream
    if (side_freqs) {
        fs = mt.side;
        HIP_TRY(hipStreamWaitEvent(fs, mt.fork, 0));
    }
    if (side_tails) {
        // (the short blocks' outputs are their own; the "left as gaps" flags wait for the docs launch: finalize_flagged_kernel)
        HIP_TRY(hipStreamWaitEvent(mt.side2, mt.fork, 0));
        hipLaunchKernelGGL(interpolative_tails_kernel, dim3(tgrid), dim3(64), kTailLdsBytes, mt.side2, d_index, uint64_t(index_bytes), t.d_blocks,
                           static_cast<const uint64_t*>(nullptr), t.d_tails, t.d_tails + n_blocks, d_docids, uint64_t(out_capacity),
                           static_cast<uint64_t*>(nullptr), 0u, 1u, d_freqs, static_cast<uint8_t*>(nullptr), uint64_t(0));
        HIP_TRY(hipEventRecord(mt.join2, mt.side2));
    }
    int st = launch_decode(docs_dict, d_index, index_bytes, t.d_units, n_blocks, d_docids, out_capacity, t.d_ends, s, 1, t.d_spans, 0,
                           t.d_bases, t.d_gaps_left, keep ? &mt.docs_sched : nullptr);
    auto fail = [&](int code) {  // (nothing of this call may still be running on the table's streams when the caller hears of it)
        if (side_freqs) (void)hipStreamSynchronize(mt.side);
        if (side_tails) (void)hipStreamSynchronize(mt.side2);
        return code;
    };
    if (st != DINT_OK) return fail(st);
    if (!t.spans_exact && covers) {  // (stream-ordered: the next decode on this table finds the exact spans)
        hipLaunchKernelGGL(exact_spans_kernel, dim3(grid), dim3(tb), 0, s, t.d_units, t.d_ends, uint64_t(n_blocks), t.d_spans);
        mt.spans_exact = true;
    }
    if (d_freqs) {
        if (!t.freqs_units_ready) {
            // (a skipped block's end offset is zero: its freqs unit then starts at the buffer's first byte and is
            // skipped in turn — same out_off, same capacity)
            hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, t.d_blocks, t.d_ends, uint64_t(n_blocks),
                               uint64_t(index_bytes), t.d_funits, t.d_fspans, static_cast<uint32_t*>(nullptr));
            mt.freqs_units_ready = t.owns_blocks && covers;
        }
        // (freq = decoded value + 1, dict_posting_list.hpp:164-169: added where the values are stored)
        st = launch_decode(freqs_dict, d_index, index_bytes, t.d_funits, n_blocks, d_freqs, out_capacity, nullptr, fs, 1, t.d_fspans, 1,
                           nullptr, nullptr, keep ? &mt.freqs_sched : nullptr);
        if (st != DINT_OK) return fail(st);
        if (fs != s) HIP_TRY(hipEventRecord(mt.join, fs));
    }
    // the short blocks' waves also look through the "left as gaps" flags (a slow codeword, a block of more than 256
    // slots: next to none), one share each; a table without short blocks gets the flags kernel alone
    if (side_tails) {
        hipLaunchKernelGGL(finalize_flagged_kernel, dim3(uint32_t((n_blocks + 63) / 64)), dim3(64), 0, s, t.d_blocks, uint64_t(n_blocks),
                           d_docids, uint64_t(out_capacity), t.d_gaps_left);
        HIP_TRY(hipStreamWaitEvent(s, mt.join2, 0));
    } else if (tgrid)
        hipLaunchKernelGGL(interpolative_tails_kernel, dim3(tgrid), dim3(64), kTailLdsBytes, s, d_index, uint64_t(index_bytes), t.d_blocks,
                           static_cast<const uint64_t*>(nullptr), t.d_tails, t.d_tails + n_blocks, d_docids, uint64_t(out_capacity),
                           static_cast<uint64_t*>(nullptr), 0u, 1u, d_freqs, t.d_gaps_left, uint64_t(n_blocks));
    else
        hipLaunchKernelGGL(finalize_flagged_kernel, dim3(uint32_t((n_blocks + 63) / 64)), dim3(64), 0, s, t.d_blocks, uint64_t(n_blocks),
                           d_docids, uint64_t(out_capacity), t.d_gaps_left);
    HIP_TRY(hipGetLastError());
    if (fs != s) HIP_TRY(hipStreamWaitEvent(s, mt.join, 0));  // (everything behind this call on the caller's stream sees the freqs)
    if (covers) mt.decodes += 1;
    return DINT_OK;
}
}  // namespace

int dint_block_table_create(const dint_dict* docs_dict, const dint_block_ref* blocks, size_t n_blocks, size_t index_bytes,
                            dint_block_table** out) {
    if (!docs_dict || !out || (!blocks && n_blocks)) return DINT_ERR_ARG;
    *out = nullptr;
    if (n_blocks >= 0xFFFFFFFFull) return DINT_ERR_ARG;
    uint64_t max_out_end = 0;
    for (size_t b = 0; b != n_blocks; ++b) {
        if (blocks[b].n == 0 || blocks[b].n > kBlock || blocks[b].in_off > index_bytes) return DINT_ERR_FORMAT;
        if (blocks[b].out_off + blocks[b].n >= blocks[b].out_off) max_out_end = std::max<uint64_t>(max_out_end, blocks[b].out_off + blocks[b].n);
        else max_out_end = ~uint64_t(0);  // (wraps: no capacity covers it)
    }
    auto* t = new (std::nothrow) dint_block_table();
    if (!t) return DINT_ERR_NOMEM;
    t->device = docs_dict->device;
    t->max_out_end = max_out_end;
    if (n_blocks == 0) {
        *out = t;
        return DINT_OK;
    }
    dint_block_ref* d_blocks = nullptr;
    int st = DINT_ERR_HIP;
    if (hip_ok(hipSetDevice(t->device), "hipSetDevice") && hip_ok(counted_malloc(&d_blocks, n_blocks * sizeof(dint_block_ref)), "counted_malloc(blocks)") &&
        hip_ok(hipMemcpy(d_blocks, blocks, n_blocks * sizeof(dint_block_ref), hipMemcpyHostToDevice), "hipMemcpy(blocks)")) {
        t->owns_blocks = true;
        st = block_table_prepare(*t, d_blocks, n_blocks, index_bytes, nullptr);
    }
    if (st != DINT_OK) {
        if (d_blocks && !t->d_blocks) (void)hipFree(d_blocks);
        dint_block_table_destroy(t);
        return st;
    }
    *out = t;
    return DINT_OK;
}

void dint_block_table_destroy(dint_block_table* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->d_ws) (void)hipFree(t->d_ws);
    if (t->side) {
        (void)hipStreamSynchronize(t->side);
        (void)hipStreamDestroy(t->side);
        (void)hipEventDestroy(t->fork);
        (void)hipEventDestroy(t->join);
        (void)hipStreamSynchronize(t->side2);
        (void)hipStreamDestroy(t->side2);
        (void)hipEventDestroy(t->join2);
    }
    if (t->docs_sched.d_mem) (void)hipFree(t->docs_sched.d_mem);
    if (t->freqs_sched.d_mem) (void)hipFree(t->freqs_sched.d_mem);
    if (t->owns_blocks && t->d_blocks) (void)hipFree(const_cast<dint_block_ref*>(t->d_blocks));
    delete t;
}

int dint_decode_block_table(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index, size_t index_bytes,
                            dint_block_table* table, uint32_t* d_docids, uint32_t* d_freqs, size_t out_capacity, void* stream) {
    if (!docs_dict || !table || (d_freqs && !freqs_dict)) return DINT_ERR_ARG;
    if (table->n_blocks == 0) return DINT_OK;
    if (!d_index || !d_docids || index_bytes < 8 || table->device != docs_dict->device) return DINT_ERR_ARG;
    if (freqs_dict && (freqs_dict->device != docs_dict->device || freqs_dict->kind != docs_dict->kind)) return DINT_ERR_ARG;
    HIP_TRY(hipSetDevice(docs_dict->device));
    return block_table_decode(docs_dict, freqs_dict, d_index, index_bytes, *table, d_docids, d_freqs, out_capacity,
                              static_cast<hipStream_t>(stream));
}

int dint_decode_posting_blocks(const dint_dict* docs_dict, const dint_dict* freqs_dict, const uint8_t* d_index,
                               size_t index_bytes, const dint_block_ref* d_blocks, size_t n_blocks,
                               uint32_t* d_docids, uint32_t* d_freqs, size_t out_capacity, void* stream) {
    if (!docs_dict || (d_freqs && !freqs_dict)) return DINT_ERR_ARG;
    if (n_blocks == 0) return DINT_OK;
    if (!d_index || !d_blocks || !d_docids || index_bytes < 8 || n_blocks >= 0xFFFFFFFFull) return DINT_ERR_ARG;
    if (freqs_dict && (freqs_dict->device != docs_dict->device || freqs_dict->kind != docs_dict->kind))
        return DINT_ERR_ARG;
    HIP_TRY(hipSetDevice(docs_dict->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one shot: a prepared table for this call only (dint_block_table_create + dint_decode_block_table keep it)
    dint_block_table t;
    t.device = docs_dict->device;
    int st = block_table_prepare(t, d_blocks, n_blocks, index_bytes, s);
    if (st == DINT_OK) st = block_table_decode(docs_dict, freqs_dict, d_index, index_bytes, t, d_docids, d_freqs, out_capacity, s);
    (void)hipStreamSynchronize(s);  // the workspace goes away with the call
    if (t.d_ws) (void)hipFree(t.d_ws);
    return st;
}

// ---- conjunctive queries ------------------------------------------------------------------------

void dint_query_index_destroy(dint_query_index* qi) {
    if (!qi) return;
    if (qi->docs) (void)hipSetDevice(qi->docs->device);
    for (void* p : {static_cast<void*>(qi->d_blocks), static_cast<void*>(qi->d_block_max),
                    static_cast<void*>(qi->d_needed), static_cast<void*>(qi->d_rank),
                    static_cast<void*>(qi->d_touched), static_cast<void*>(qi->d_n_touched)})
        if (p) (void)hipFree(p);
    if (qi->h_stage) (void)hipHostFree(qi->h_stage);
    qi->inputs.release();
    qi->ctrl.release();
    qi->cand.release();
    qi->target.release();
    qi->probe.release();
    qi->tails.release();
    qi->fprobe.release();
    qi->spans.release();
    qi->bases.release();
    qi->ends.release();
    qi->gaps_left.release();
    qi->freq_sums.release();
    qi->freq_counts.release();
    qi->sub.release();
    qi->units.release();
    delete qi;
}

int dint_query_index_create(const dint_dict* docs_dict, const uint8_t* d_index, size_t index_bytes,
                            const dint_block_ref* blocks, size_t n_blocks, size_t n_lists, dint_query_index** out) {
    if (!docs_dict || !out || (!blocks && n_blocks) || (!d_index && n_blocks)) return DINT_ERR_ARG;
    if (n_blocks >= 0xFFFFFFFFull || n_lists >= 0xFFFFFFFFull || (n_blocks && index_bytes < 8)) return DINT_ERR_ARG;
    *out = nullptr;
    auto* qi = new (std::nothrow) dint_query_index();
    if (!qi) return DINT_ERR_NOMEM;
    qi->docs = docs_dict;
    qi->d_index = d_index;
    qi->index_bytes = index_bytes;
    qi->n_blocks = n_blocks;
    qi->list_first.assign(n_lists + 1, 0);
    qi->list_len.assign(n_lists, 0);
    std::vector<uint32_t> maxs(n_blocks);
    uint32_t prev_list = 0;
    for (size_t b = 0; b != n_blocks; ++b) {
        const uint32_t l = blocks[b].list;
        if (l >= n_lists || l < prev_list || blocks[b].n == 0 || blocks[b].n > 256 ||
            blocks[b].in_off > index_bytes) {  // lists in order, each list's blocks contiguous
            delete qi;
            return DINT_ERR_FORMAT;
        }
        prev_list = l;
        qi->list_first[l + 1] += 1;
        qi->list_len[l] += blocks[b].n;
        maxs[b] = blocks[b].max;
    }
    for (size_t l = 0; l != n_lists; ++l) qi->list_first[l + 1] += qi->list_first[l];
    const size_t nb = std::max<size_t>(1, n_blocks);
    bool ok = hip_ok(hipSetDevice(docs_dict->device), "hipSetDevice") &&
              hip_ok(counted_malloc(&qi->d_blocks, nb * sizeof(dint_block_ref)), "counted_malloc(blocks)") &&
              hip_ok(counted_malloc(&qi->d_block_max, nb * 4), "counted_malloc(block_max)") &&
              hip_ok(counted_malloc(&qi->d_needed, 2 * nb * 4), "counted_malloc(needed)") &&  // (two sets of each: round_tail)
              hip_ok(counted_malloc(&qi->d_rank, 2 * nb * 4), "counted_malloc(rank)") &&
              hip_ok(counted_malloc(&qi->d_touched, 2 * nb * 4), "counted_malloc(touched)") &&
              hip_ok(counted_malloc(&qi->d_n_touched, 8), "counted_malloc(n_touched)") &&  // {touched blocks, short pages} of a round
              hip_ok(hipMemset(qi->d_needed, 0, 2 * nb * 4), "hipMemset(needed)");
    if (ok && n_blocks)
        ok = hip_ok(hipMemcpy(qi->d_blocks, blocks, n_blocks * sizeof(dint_block_ref), hipMemcpyHostToDevice), "hipMemcpy(blocks)") &&
             hip_ok(hipMemcpy(qi->d_block_max, maxs.data(), n_blocks * 4, hipMemcpyHostToDevice), "hipMemcpy(block_max)");
    if (!ok) {
        dint_query_index_destroy(qi);
        return DINT_ERR_HIP;
    }
    *out = qi;
    return DINT_OK;
}

// The pages of `sub` decoded, 256 slots per page, no sync: docs parts -> docIDs in d_docs (formed in the decode
// kernels' expansion, like dint_decode_block_table) and, with a freqs dictionary, freqs parts -> d_freqs.
static int decode_pages(dint_query_index* qi, size_t n_pages, uint32_t* d_docs, const dint_dict* freqs_dict, uint32_t* d_freqs,
                        hipStream_t s) {
    if (!qi->units.ensure(n_pages) || !qi->spans.ensure(n_pages) || !qi->bases.ensure(n_pages) || !qi->ends.ensure(n_pages) ||
        !qi->gaps_left.ensure(n_pages) || !qi->tails.ensure(n_pages + 1))
        return DINT_ERR_HIP;
    const uint32_t tb = 256;
    const uint32_t grid = uint32_t((n_pages + tb - 1) / tb);
    const uint64_t cap = uint64_t(n_pages) * kPageSlots;
    HIP_TRY(hipMemsetAsync(qi->gaps_left.p, 0, n_pages, s));
    HIP_TRY(hipMemsetAsync(qi->tails.p + n_pages, 0, 4, s));
    hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, qi->sub.p, static_cast<const uint64_t*>(nullptr),
                       uint64_t(n_pages), uint64_t(qi->index_bytes), qi->units.p, qi->spans.p, qi->bases.p);
    hipLaunchKernelGGL(collect_tails_kernel, dim3(grid), dim3(tb), 0, s, qi->sub.p, uint64_t(n_pages), qi->tails.p,
                       qi->tails.p + n_pages);
    int st = launch_decode(qi->docs, qi->d_index, qi->index_bytes, qi->units.p, n_pages, d_docs, cap, qi->ends.p, s, 1, qi->spans.p, 0,
                           qi->bases.p, qi->gaps_left.p);
    if (st != DINT_OK) return st;
    if (freqs_dict) {  // freqs parts of the full blocks: from where their docs parts ended
        hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, qi->sub.p, qi->ends.p, uint64_t(n_pages),
                           uint64_t(qi->index_bytes), qi->units.p, qi->spans.p, static_cast<uint32_t*>(nullptr));
        st = launch_decode(freqs_dict, qi->d_index, qi->index_bytes, qi->units.p, n_pages, d_freqs, cap, nullptr, s, 1, qi->spans.p, 1);
        if (st != DINT_OK) return st;
    }
    // (the grid is sized for "every page is a short block"; the waves past the list's end leave at once)
    hipLaunchKernelGGL(interpolative_tails_kernel, dim3(uint32_t((n_pages + kTailLanes - 1) / kTailLanes)), dim3(64), kTailLdsBytes, s,
                       qi->d_index, uint64_t(qi->index_bytes), qi->sub.p, static_cast<const uint64_t*>(nullptr), qi->tails.p,
                       qi->tails.p + n_pages, d_docs, cap, static_cast<uint64_t*>(nullptr), 0u, 1u, freqs_dict ? d_freqs : nullptr);
    hipLaunchKernelGGL(finalize_flagged_kernel, dim3(uint32_t((n_pages + 63) / 64)), dim3(64), 0, s, qi->sub.p, uint64_t(n_pages), d_docs,
                       cap, qi->gaps_left.p);
    HIP_TRY(hipGetLastError());
    return DINT_OK;
}

// Control words of one page decode (a call has one set for the candidates and one per round, cleared together):
// [0] blocks the round touched, [1] short pages, [kCtrlQueueAt ...) the decode kernel's queue counters.
constexpr size_t kCtrlQueueAt = 32;
constexpr size_t kCtrlWords = kCtrlQueueAt + (kQueueShards + 1) * kQueueStride;

// A round's pages are decoded by decode_pages_lean (one launch) below this many pages, else by decode_pages_counted's
// three launches. Measured on the 1e8-posting index, in one process: the one-launch form wins at every size — a
// single query 95 -> 53 us, the reference's query log as one batch 2.16 -> 1.53 us per query, the longest lists
// 5.6 -> 5.0 — so it is the default; the variable keeps the other form testable (tests/test_gpu_queries.py).
static size_t lean_pages() {
    const long long v = opt(DINT_OPT_QUERY_LEAN_PAGES);
    return v < 0 ? ~size_t(0) : size_t(v);
}

// (one workgroup walks all the candidates: past a few pages the probe and search launches, a thread per candidate, win)
static size_t tail_pages() { return size_t(opt(DINT_OPT_QUERY_TAIL_PAGES)); }
// (a query of at most this many candidate pages runs as ONE launch of one workgroup: query_fused_body)
static size_t fused_pages() { return size_t(opt(DINT_OPT_QUERY_FUSED_PAGES)); }
static int decode_pages_lean(dint_query_index* qi, const uint32_t* d_ids, const uint32_t* d_count, size_t bound, uint32_t* d_docs,
                             uint32_t* ctrl, uint32_t retire, hipStream_t s, const query_pages* search,
                             const round_tail* tail = nullptr);

// Pages -> docIDs, three launches on the stream: the pages are the blocks ids[0 .. *count) (count null: ids[0 .. bound)),
// `bound` >= their number is what the launches are sized for. `ctrl`: this decode's (cleared) control words.
// retire: the slots past each page's last posting are marked dead (candidate pages).
// `search` (candidate pages): what the first round's search needs; *searched says whether it was done along the way.
static int decode_pages_counted(dint_query_index* qi, const uint32_t* d_ids, const uint32_t* d_count, size_t bound, uint32_t* d_docs,
                                uint32_t* ctrl, uint32_t retire, hipStream_t s, const query_pages* search = nullptr,
                                bool* searched = nullptr) {
    if (searched) *searched = false;
    if (bound < lean_pages() && qi->index_bytes >= 8) {
        if (searched) *searched = search != nullptr;
        return decode_pages_lean(qi, d_ids, d_count, bound, d_docs, ctrl, retire, s, search);
    }
    if (!qi->sub.ensure(bound) || !qi->units.ensure(bound) || !qi->spans.ensure(bound) || !qi->bases.ensure(bound) ||
        !qi->ends.ensure(bound) || !qi->gaps_left.ensure(bound) || !qi->tails.ensure(bound + 1))
        return DINT_ERR_HIP;
    const uint32_t tb = 256;
    const uint64_t cap = uint64_t(bound) * kPageSlots;
    uint32_t* const d_n_tails = ctrl + 1;
    hipLaunchKernelGGL(prepare_pages_kernel, dim3(uint32_t((bound + tb - 1) / tb)), dim3(tb), 0, s, qi->d_blocks, uint64_t(qi->n_blocks),
                       uint64_t(qi->index_bytes), d_ids, d_count, uint64_t(bound), qi->sub.p, qi->units.p, qi->spans.p, qi->bases.p,
                       qi->gaps_left.p, qi->tails.p, d_n_tails);
    int st = launch_decode(qi->docs, qi->d_index, qi->index_bytes, qi->units.p, bound, d_docs, cap, nullptr, s, 1, qi->spans.p, 0,
                           qi->bases.p, qi->gaps_left.p, nullptr, 2048, ctrl + kCtrlQueueAt);
    if (st != DINT_OK) return st;
    // (the grid is sized for "every page is a short block": the waves with nothing to do leave at once)
    hipLaunchKernelGGL(fix_pages_kernel, dim3(uint32_t((bound + kTailLanes - 1) / kTailLanes)), dim3(64), kTailLdsBytes, s, qi->d_index,
                       uint64_t(qi->index_bytes), qi->sub.p, uint64_t(bound), qi->tails.p, d_n_tails, d_docs, cap, qi->gaps_left.p, retire);
    HIP_TRY(hipGetLastError());
    return DINT_OK;
}

// The same in ONE launch: decode_*_query_kernel looks the blocks up itself, sums what it has to leave as gaps and
// runs the short blocks' interpolative code in place — no prepare, no schedule, no fix-up launch (for a single query
// the launches are what it waits for; in a batch the short blocks' bit-serial decoder, a launch of its own in the
// three-launch form, runs beside the full blocks instead of behind them).
static int decode_pages_lean(dint_query_index* qi, const uint32_t* d_ids, const uint32_t* d_count, size_t bound, uint32_t* d_docs,
                             uint32_t* ctrl, uint32_t retire, hipStream_t s, const query_pages* search, const round_tail* tail) {
    if (!qi->gaps_left.ensure(bound)) return DINT_ERR_HIP;
    const dint_dict* dd = qi->docs;
    decode_args a{};
    a.dict = dd->view;
    a.enc = qi->d_index;
    a.enc_bytes = qi->index_bytes;
    a.n_units = bound;
    a.out = d_docs;
    a.out_capacity = uint64_t(bound) * kPageSlots;
    a.gaps_left = qi->gaps_left.p;
    const uint64_t blocks_needed = (uint64_t(bound) + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t grid = uint32_t(std::min<uint64_t>(blocks_needed, std::max<uint32_t>(1, dd->compute_units) * kBlocksPerCU));
    const size_t lds_bytes = (size_t(dd->view.hot_words) + kClassTableWords + kWavesPerBlock * kScratchWords) * 4;
    a.queue = ctrl + kCtrlQueueAt;
    a.chunk_queue = a.queue + kQueueShards * kQueueStride;
    a.n_shards = std::min<uint32_t>(kQueueShards, grid);
    query_pages qp{};
    if (search) qp = *search;  // (the first round's search, for the candidate pages)
    qp.blocks = qi->d_blocks;
    qp.ids = d_ids;
    qp.count = d_count;
    qp.bound = bound;
    qp.retire = retire;
    round_tail rt{};
    if (tail) rt = *tail;
    if (dd->kind == DINT_DICT_MULTI_PACKED)
        hipLaunchKernelGGL(decode_multi_query_kernel, dim3(grid), dim3(kBlockThreads), lds_bytes, s, a, qp, rt);
    else
        hipLaunchKernelGGL(decode_single_query_kernel, dim3(grid), dim3(kBlockThreads), lds_bytes, s, a, qp, rt);
    HIP_TRY(hipGetLastError());
    return DINT_OK;
}

static int and_queries_impl(dint_query_index* qi, const dint_dict* freqs_dict, const uint32_t* terms, const uint64_t* query_offsets,
                            size_t n_queries, uint64_t* counts, uint64_t* freq_sums, uint64_t* freq_blocks, void* stream);

int dint_and_queries(dint_query_index* qi, const uint32_t* terms, const uint64_t* query_offsets, size_t n_queries,
                     uint64_t* counts, void* stream) {
    return and_queries_impl(qi, nullptr, terms, query_offsets, n_queries, counts, nullptr, nullptr, stream);
}

int dint_and_queries_freqs(dint_query_index* qi, const dint_dict* freqs_dict, const uint32_t* terms, const uint64_t* query_offsets,
                           size_t n_queries, uint64_t* counts, uint64_t* freq_sums, uint64_t* freq_blocks_decoded, void* stream) {
    if (!freqs_dict || !freq_sums) return DINT_ERR_ARG;
    if (qi && (freqs_dict->device != qi->docs->device || freqs_dict->kind != qi->docs->kind)) return DINT_ERR_ARG;
    return and_queries_impl(qi, freqs_dict, terms, query_offsets, n_queries, counts, freq_sums, freq_blocks_decoded, stream);
}

static int and_queries_impl(dint_query_index* qi, const dint_dict* freqs_dict, const uint32_t* terms, const uint64_t* query_offsets,
                            size_t n_queries, uint64_t* counts, uint64_t* freq_sums, uint64_t* freq_blocks, void* stream) {
    if (!qi || (n_queries && (!query_offsets || !counts))) return DINT_ERR_ARG;
    if (freq_blocks) *freq_blocks = 0;
    if (n_queries == 0) return DINT_OK;
    if (n_queries >= 0xFFFFFFFFull) return DINT_ERR_ARG;
    const size_t n_lists = qi->list_len.size();
    // per query: distinct terms, rarest list first (queries.hpp:28-31, 49-52)
    std::vector<std::vector<uint32_t>> plan(n_queries);
    size_t rounds = 0;
    std::vector<uint32_t> h_page_block, h_page_query;
    for (size_t q = 0; q != n_queries; ++q) {
        if (query_offsets[q + 1] < query_offsets[q] || (query_offsets[q + 1] > query_offsets[q] && !terms)) return DINT_ERR_ARG;
        std::vector<uint32_t>& t = plan[q];
        t.assign(terms + query_offsets[q], terms + query_offsets[q + 1]);
        for (uint32_t term : t)
            if (term >= n_lists) return DINT_ERR_ARG;
        std::sort(t.begin(), t.end());
        t.erase(std::unique(t.begin(), t.end()), t.end());
        std::stable_sort(t.begin(), t.end(), [&](uint32_t a, uint32_t b) { return qi->list_len[a] < qi->list_len[b]; });
        counts[q] = 0;
        if (freq_sums) freq_sums[q] = 0;
        if (t.empty()) continue;
        if (t.size() == 1 && !freqs_dict) {  // one list: every posting is a result (and_query<false> would walk it and count)
            counts[q] = qi->list_len[t[0]];
            t.clear();
            continue;
        }
        rounds = std::max(rounds, t.size() - 1);
        for (uint32_t b = qi->list_first[t[0]]; b != qi->list_first[t[0] + 1]; ++b) {
            h_page_block.push_back(b);
            h_page_query.push_back(uint32_t(q));
        }
    }
    const size_t n_pages = h_page_block.size();
    if (n_pages == 0) return DINT_OK;
    const uint64_t n_slots = uint64_t(n_pages) * kPageSlots;
    std::vector<uint32_t> h_first(std::max<size_t>(1, rounds * n_queries), 0), h_blocks(std::max<size_t>(1, rounds * n_queries), 0);
    for (size_t q = 0; q != n_queries; ++q)
        for (size_t j = 1; j < plan[q].size(); ++j) {
            const uint32_t l = plan[q][j];
            h_first[(j - 1) * n_queries + q] = qi->list_first[l];
            h_blocks[(j - 1) * n_queries + q] = qi->list_first[l + 1] - qi->list_first[l];
        }

    std::lock_guard<std::mutex> lock(qi->mutex);
    HIP_TRY(hipSetDevice(qi->docs->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // inputs: {page -> block, page -> query, per round and query: first block and block count of the round's list},
    // and behind them — zeros, copied in with them: one copy instead of a copy and a clear — every counter the
    // call's launches count in, and the result counters (u64 each)
    const size_t in_words = (2 * n_pages + h_first.size() + h_blocks.size() + 31) / 32 * 32;
    const size_t ctrl_words = ((rounds + 1) * kCtrlWords + 2 * n_queries + 1) / 2 * 2;
    const size_t step_words = (rounds + 1) * ((sizeof(fused_step) + 7) / 8 * 2);  // (the one-launch form's steps, 8-byte aligned)
    const size_t up_words = in_words + ctrl_words + step_words;
    const size_t stage_bytes = std::max(up_words * 4, n_queries * sizeof(unsigned long long));
    if (qi->h_stage_cap < stage_bytes) {
        if (qi->h_stage) (void)hipHostFree(qi->h_stage);
        qi->h_stage = nullptr;
        qi->h_stage_cap = 0;
        const size_t want = stage_bytes + stage_bytes / 2 + 4096;
        HIP_TRY(counted_host_malloc(&qi->h_stage, want));
        qi->h_stage_cap = want;
        qi->d_stage = nullptr;  // the same memory as the kernels see it (the last probe writes the results there)
        if (hipHostGetDevicePointer(&qi->d_stage, qi->h_stage, 0) != hipSuccess) qi->d_stage = nullptr;
    }
    if (!qi->inputs.ensure(up_words) || !qi->cand.ensure(n_slots) || !qi->target.ensure(n_slots)) return DINT_ERR_HIP;
    {
        uint32_t* h = static_cast<uint32_t*>(qi->h_stage);
        std::memcpy(h, h_page_block.data(), n_pages * 4);
        std::memcpy(h + n_pages, h_page_query.data(), n_pages * 4);
        std::memcpy(h + 2 * n_pages, h_first.data(), h_first.size() * 4);
        std::memcpy(h + 2 * n_pages + h_first.size(), h_blocks.data(), h_blocks.size() * 4);
        std::memset(h + in_words, 0, ctrl_words * 4);
    }
    uint32_t* const d_page_block = qi->inputs.p;
    uint32_t* const d_page_query = d_page_block + n_pages;
    uint32_t* const d_term_first = d_page_query + n_pages;
    uint32_t* const d_term_blocks = d_term_first + h_first.size();
    uint32_t* const d_ctrl = qi->inputs.p + in_words;
    unsigned long long* const d_counts = reinterpret_cast<unsigned long long*>(d_ctrl + (rounds + 1) * kCtrlWords);
    // ---- what the host knows of the rounds before anything runs: a bound of the pages each decodes ------------------
    uint64_t round0_blocks = 0;
    if (rounds)
        for (size_t q = 0; q != n_queries; ++q) round0_blocks += h_blocks[q];
    std::vector<size_t> round_bound(rounds, 0);
    bool small_rounds = rounds != 0 && round0_blocks != 0 && n_pages < lean_pages() && qi->index_bytes >= 8;
    for (size_t r = 0; r != rounds; ++r) {
        uint64_t list_blocks = 0;
        for (size_t q = 0; q != n_queries; ++q) list_blocks += h_blocks[r * n_queries + q];
        round_bound[r] = size_t(std::min<uint64_t>(n_slots, list_blocks));
        small_rounds = small_rounds && list_blocks != 0 && round_bound[r] < lean_pages();
    }
    // A query of a page or two of candidates: the whole chain — candidates, then every round's pages and tail — in ONE
    // launch of one workgroup (query_fused_body; DINT_QUERY_FUSED_PAGES: at most that many candidate pages, 0: never).
    const bool fused_ok = small_rounds && n_pages <= tail_pages() && n_pages <= fused_pages();
    fused_step* const d_steps = reinterpret_cast<fused_step*>(qi->inputs.p + in_words + ctrl_words);
    if (fused_ok) {
        size_t max_pages = n_pages;
        for (size_t r = 0; r != rounds; ++r) max_pages = std::max(max_pages, round_bound[r]);
        if (!qi->probe.ensure(uint64_t(max_pages) * kPageSlots) || !qi->gaps_left.ensure(max_pages)) return DINT_ERR_HIP;
        fused_step* const h_steps = reinterpret_cast<fused_step*>(static_cast<uint32_t*>(qi->h_stage) + in_words + ctrl_words);
        const size_t nb = std::max<size_t>(1, qi->n_blocks);
        const bool to_host = !freqs_dict && qi->d_stage != nullptr;
        for (size_t k = 0; k != rounds + 1; ++k) {
            fused_step st{};
            st.gaps_left = qi->gaps_left.p;
            st.qp.blocks = qi->d_blocks;
            if (k == 0) {  // the candidate pages, the first round's search riding along (decode_pages_lean's candidate call)
                st.out = qi->cand.p;
                st.out_capacity = uint64_t(n_pages) * kPageSlots;
                st.qp.page_query = d_page_query;
                st.qp.term_first = d_term_first;
                st.qp.term_blocks = d_term_blocks;
                st.qp.block_max = qi->d_block_max;
                st.qp.target = qi->target.p;
                st.qp.needed = qi->d_needed;
                st.qp.rank = qi->d_rank;
                st.qp.touched = qi->d_touched;
                st.qp.n_touched = d_ctrl + kCtrlWords;
                st.qp.ids = d_page_block;
                st.qp.count = nullptr;
                st.qp.bound = n_pages;
                st.qp.retire = 1u;
            } else {  // round r: the touched pages, then the tail (the round-per-launch form's round_tail, below)
                const size_t r = k - 1, set = r & 1, next_set = set ^ 1;
                uint32_t* const ctrl = d_ctrl + (r + 1) * kCtrlWords;
                st.out = qi->probe.p;
                st.out_capacity = uint64_t(round_bound[r]) * kPageSlots;
                st.qp.ids = qi->d_touched + set * nb;
                st.qp.count = ctrl;
                st.qp.bound = round_bound[r];
                st.qp.retire = 0u;
                round_tail& t = st.rt;
                t.done = ctrl + 2;  // (not counted in: non-null says "this step has a tail")
                t.cand = qi->cand.p;
                t.n_slots = n_slots;
                t.page_query = d_page_query;
                t.blocks = qi->d_blocks;
                t.target = qi->target.p;
                t.term_blocks = d_term_blocks + r * n_queries;
                t.rank = qi->d_rank + set * nb;
                t.probe = qi->probe.p;
                t.touched = qi->d_touched + set * nb;
                t.n_touched = ctrl;
                t.needed = qi->d_needed + set * nb;
                if (r + 1 != rounds) {
                    t.next_first = d_term_first + (r + 1) * n_queries;
                    t.next_blocks = d_term_blocks + (r + 1) * n_queries;
                    t.block_max = qi->d_block_max;
                    t.next_needed = qi->d_needed + next_set * nb;
                    t.next_rank = qi->d_rank + next_set * nb;
                    t.next_touched = qi->d_touched + next_set * nb;
                    t.next_n_touched = d_ctrl + (r + 2) * kCtrlWords;
                } else {
                    t.counts = d_counts;
                    t.host_counts = to_host ? static_cast<unsigned long long*>(qi->d_stage) : nullptr;
                    t.n_queries = uint32_t(n_queries);
                }
            }
            std::memcpy(h_steps + k, &st, sizeof st);
        }
    }
    HIP_TRY(hipMemcpyAsync(qi->inputs.p, qi->h_stage, up_words * 4, hipMemcpyHostToDevice, s));

    const uint32_t tb = 256;
    const uint32_t slot_grid = uint32_t(n_pages);  // 256 slots per page = one workgroup
    if (qi->claims_dirty) {  // (a call that failed between a search and its release left claim flags behind)
        HIP_TRY(hipMemsetAsync(qi->d_needed, 0, 2 * std::max<size_t>(1, qi->n_blocks) * 4, s));
        qi->claims_dirty = false;
    }
    // candidates: the rarest list of every query
    query_pages search0{};
    search0.page_query = d_page_query;
    search0.term_first = d_term_first;
    search0.term_blocks = d_term_blocks;
    search0.block_max = qi->d_block_max;
    search0.target = qi->target.p;
    search0.needed = qi->d_needed;
    search0.rank = qi->d_rank;
    search0.touched = qi->d_touched;
    search0.n_touched = d_ctrl + kCtrlWords;
    bool searched0 = false;
    int st = DINT_OK;
    if (fused_ok) {
        const dint_dict* dd = qi->docs;
        decode_args a{};
        a.dict = dd->view;
        a.enc = qi->d_index;
        a.enc_bytes = qi->index_bytes;
        // lists of fewer than 256 postings are one interpolative block each and need no dictionary: a query of such lists
        // only (most of a query log's) runs without the 88 KB LDS image — nothing reads it
        bool any_full = false;
        for (size_t q = 0; q != n_queries; ++q)
            for (uint32_t term : plan[q]) any_full = any_full || qi->list_len[term] >= kBlock;
        if (!any_full) a.dict.hot_words = 0;
        const size_t lds_bytes = (size_t(a.dict.hot_words) + kClassTableWords + kWavesPerBlock * kScratchWords) * 4;
        if (dd->kind == DINT_DICT_MULTI_PACKED)
            hipLaunchKernelGGL(decode_multi_query_fused_kernel, dim3(1), dim3(kBlockThreads), lds_bytes, s, a, d_steps, uint32_t(rounds + 1));
        else
            hipLaunchKernelGGL(decode_single_query_fused_kernel, dim3(1), dim3(kBlockThreads), lds_bytes, s, a, d_steps, uint32_t(rounds + 1));
        HIP_TRY(hipGetLastError());
        searched0 = true;
    } else {
        st = decode_pages_counted(qi, d_page_block, nullptr, n_pages, qi->cand.p, d_ctrl, 1u, s, round0_blocks ? &search0 : nullptr, &searched0);
    }
    if (st != DINT_OK) {
        (void)hipStreamSynchronize(s);
        return st;
    }
    qi->claims_dirty = true;  // until the call has run to its end

    // A round: block-max search -> the touched blocks, without duplicates -> decoded -> every candidate probes its
    // block. How many blocks a round touches only the device knows; the host knows a bound (the live candidates at
    // most, and no more blocks than the round's lists have) and sizes the launches for that — nothing on the host
    // waits for a round: a call is one copy in, then per round search, page decode, probe. (Round 1 read the
    // count back every round and made fourteen API calls per round: a query at a time, the host's share was most of
    // the 200 us a query took.) Past kAsyncPages the count is read back after all: launches sized for a bound far
    // above the truth cost more than the wait (measured again with the one-launch decode at 131072: every workgroup
    // of a grid sized for the bound loads the dictionary image, 1.41 against 1.34 us per query).
    constexpr size_t kAsyncPages = 32768;
    size_t last_round = rounds;  // the last round that has anything to probe counts the survivors as well
    for (size_t r = 0; r != rounds; ++r)
        for (size_t q = 0; q != n_queries; ++q)
            if (h_blocks[r * n_queries + q]) {
                last_round = r;
                break;
            }
    bool counted = false;
    // the last probe hands the results over itself (a few pages: every workgroup of it passes through one counter)
    bool results_to_host = !freqs_dict && qi->d_stage != nullptr && n_pages <= 4096;
    // Few candidates, few pages in every round (a single query): one launch per round — round_tail.
    const bool tail_form = searched0 && small_rounds && n_pages <= tail_pages();
    if (fused_ok) {
        counted = true;  // (the launch above was the whole query)
        results_to_host = results_to_host && qi->d_stage != nullptr;
    }
    if (tail_form && !fused_ok) {
        const size_t nb = std::max<size_t>(1, qi->n_blocks);
        for (size_t r = 0; r != rounds; ++r) {
            const size_t set = r & 1, next_set = set ^ 1;
            uint32_t* const ctrl = d_ctrl + (r + 1) * kCtrlWords;
            if (!qi->probe.ensure(uint64_t(round_bound[r]) * kPageSlots)) {
                (void)hipStreamSynchronize(s);
                return DINT_ERR_HIP;
            }
            round_tail t{};
            t.done = ctrl + 2;
            t.cand = qi->cand.p;
            t.n_slots = n_slots;
            t.page_query = d_page_query;
            t.blocks = qi->d_blocks;
            t.target = qi->target.p;
            t.term_blocks = d_term_blocks + r * n_queries;
            t.rank = qi->d_rank + set * nb;
            t.probe = qi->probe.p;
            t.touched = qi->d_touched + set * nb;
            t.n_touched = ctrl;
            t.needed = qi->d_needed + set * nb;
            if (r + 1 != rounds) {
                t.next_first = d_term_first + (r + 1) * n_queries;
                t.next_blocks = d_term_blocks + (r + 1) * n_queries;
                t.block_max = qi->d_block_max;
                t.next_needed = qi->d_needed + next_set * nb;
                t.next_rank = qi->d_rank + next_set * nb;
                t.next_touched = qi->d_touched + next_set * nb;
                t.next_n_touched = d_ctrl + (r + 2) * kCtrlWords;
            } else {
                t.counts = d_counts;
                t.host_counts = results_to_host ? static_cast<unsigned long long*>(qi->d_stage) : nullptr;
                t.n_queries = uint32_t(n_queries);
            }
            st = decode_pages_lean(qi, t.touched, ctrl, round_bound[r], qi->probe.p, ctrl, 0u, s, nullptr, &t);
            if (st != DINT_OK) {
                (void)hipStreamSynchronize(s);
                return st;
            }
        }
        counted = true;
    }
    for (size_t r = 0; r != rounds && !tail_form && !fused_ok; ++r) {
        const uint32_t* first = d_term_first + r * n_queries;
        const uint32_t* nblk = d_term_blocks + r * n_queries;
        uint32_t* const ctrl = d_ctrl + (r + 1) * kCtrlWords;
        uint64_t list_blocks = 0;
        for (size_t q = 0; q != n_queries; ++q) list_blocks += h_blocks[r * n_queries + q];
        if (list_blocks == 0) continue;  // no query has a term for this round
        size_t bound = size_t(std::min<uint64_t>(n_slots, list_blocks));
        if (r != 0 || !searched0)
            hipLaunchKernelGGL(and_search_kernel, dim3(slot_grid), dim3(tb), 0, s, qi->cand.p, n_slots, d_page_query,
                               first, nblk, qi->d_block_max, qi->target.p, qi->d_needed, qi->d_rank, qi->d_touched, ctrl);
        const uint32_t* d_count = ctrl;
        if (bound > kAsyncPages) {
            uint32_t n_touched = 0;
            HIP_TRY(hipMemcpyAsync(&n_touched, ctrl, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (n_touched == 0) continue;  // nothing left to probe in this round
            bound = n_touched;
            d_count = nullptr;
        }
        if (!qi->probe.ensure(uint64_t(bound) * kPageSlots)) {
            (void)hipStreamSynchronize(s);
            return DINT_ERR_HIP;
        }
        st = decode_pages_counted(qi, qi->d_touched, d_count, bound, qi->probe.p, ctrl, 0u, s);
        if (st != DINT_OK) {
            (void)hipStreamSynchronize(s);
            return st;
        }
        hipLaunchKernelGGL(and_probe_release_kernel, dim3(slot_grid), dim3(tb), 0, s, qi->cand.p, n_slots, d_page_query, nblk, qi->d_blocks,
                           qi->target.p, qi->d_rank, qi->probe.p, qi->d_touched, ctrl, qi->d_needed,
                           r == last_round ? d_counts : static_cast<unsigned long long*>(nullptr), ctrl + 2,
                           r == last_round && results_to_host ? static_cast<unsigned long long*>(qi->d_stage)
                                                              : static_cast<unsigned long long*>(nullptr),
                           uint32_t(n_queries));
        counted = counted || r == last_round;
    }
    results_to_host = results_to_host && counted;
    if (!counted) hipLaunchKernelGGL(and_count_kernel, dim3(slot_grid), dim3(tb), 0, s, qi->cand.p, n_slots, d_page_query, d_counts);
    HIP_TRY(hipGetLastError());
    // ---- and_query<true> (queries.hpp:72-76): the freq of every term at every match. Lazily, like the reference's
    // freq(): a freqs part is decoded only for the blocks that hold a match — term by term, the blocks the
    // matches fall into (for the rarest term: the candidate pages themselves), their docs and freqs parts, then
    // every match reads its freq at the position of its docID.
    std::vector<unsigned long long> h_sums;
    std::vector<uint32_t> h_freq_counts;
    if (freqs_dict) {
        if (!qi->freq_sums.ensure(n_queries)) return DINT_ERR_HIP;
        HIP_TRY(hipMemsetAsync(qi->freq_sums.p, 0, n_queries * sizeof(unsigned long long), s));
        // (the blocks a term's matches fall into are counted on the device; the launches of a term are sized for what
        // the host knows — no more blocks than matches can exist, than the terms' lists hold, than the candidate pages for
        // the rarest term — and the pages past the count are empty. Past kAsyncPages the count is read back after all,
        // as in the rounds above. The counts themselves travel to the host with the results.)
        if (!qi->freq_counts.ensure(rounds + 1)) return DINT_ERR_HIP;
        HIP_TRY(hipMemsetAsync(qi->freq_counts.p, 0, (rounds + 1) * 4, s));
        for (size_t r = 0; r != rounds + 1; ++r) {  // r = 0: the rarest term; r >= 1: the term of round r - 1
            const uint32_t* first = r ? d_term_first + (r - 1) * n_queries : nullptr;
            const uint32_t* nblk = r ? d_term_blocks + (r - 1) * n_queries : nullptr;
            uint64_t list_blocks = n_pages;
            if (r) {
                list_blocks = 0;
                for (size_t q = 0; q != n_queries; ++q) list_blocks += h_blocks[(r - 1) * n_queries + q];
            }
            size_t bound = size_t(std::min<uint64_t>(n_slots, list_blocks));
            if (bound == 0) continue;
            uint32_t* const d_cnt = qi->freq_counts.p + r;
            hipLaunchKernelGGL(and_freq_search_kernel, dim3(slot_grid), dim3(tb), 0, s, qi->cand.p, n_slots, d_page_query,
                               d_page_block, first, nblk, qi->d_block_max, qi->target.p, qi->d_needed, qi->d_rank,
                               qi->d_touched, d_cnt);
            const uint32_t* d_count = d_cnt;
            if (bound > kAsyncPages) {
                uint32_t n_touched = 0;
                HIP_TRY(hipMemcpyAsync(&n_touched, d_cnt, 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (n_touched == 0) continue;
                bound = n_touched;
                d_count = nullptr;
            }
            if (!qi->sub.ensure(std::max<size_t>(n_pages, bound)) || !qi->probe.ensure(uint64_t(bound) * kPageSlots) ||
                !qi->fprobe.ensure(uint64_t(bound) * kPageSlots)) {
                (void)hipStreamSynchronize(s);
                return DINT_ERR_HIP;
            }
            const uint32_t tgrid = uint32_t((bound + tb - 1) / tb);
            hipLaunchKernelGGL(gather_pages_kernel, dim3(tgrid), dim3(tb), 0, s, qi->d_blocks, qi->d_touched, uint64_t(bound), qi->sub.p,
                               d_count);
            st = decode_pages(qi, bound, qi->probe.p, freqs_dict, qi->fprobe.p, s);
            if (st != DINT_OK) {
                (void)hipStreamSynchronize(s);
                return st;
            }
            hipLaunchKernelGGL(and_freq_gather_kernel, dim3(slot_grid), dim3(tb), 0, s, qi->cand.p, n_slots, d_page_query, nblk,
                               qi->d_blocks, qi->target.p, qi->d_rank, qi->probe.p, qi->fprobe.p, qi->freq_sums.p);
            hipLaunchKernelGGL(and_release_kernel, dim3(tgrid), dim3(tb), 0, s, qi->d_touched, uint32_t(bound), qi->d_needed, d_count);
        }
        HIP_TRY(hipGetLastError());
        h_sums.resize(n_queries);
        HIP_TRY(hipMemcpyAsync(h_sums.data(), qi->freq_sums.p, n_queries * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        h_freq_counts.resize(rounds + 1);
        HIP_TRY(hipMemcpyAsync(h_freq_counts.data(), qi->freq_counts.p, (rounds + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    unsigned long long* const h_counts = static_cast<unsigned long long*>(qi->h_stage);  // (the inputs have long been copied)
    if (!results_to_host) HIP_TRY(hipMemcpyAsync(h_counts, d_counts, n_queries * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    // (Watching a flag in pinned memory, written behind the results, instead of the stream was measured: no faster.)
    HIP_TRY(hipStreamSynchronize(s));
    qi->claims_dirty = false;
    for (size_t q = 0; q != n_queries; ++q)
        if (!plan[q].empty()) counts[q] = h_counts[q];
    if (freqs_dict) {
        for (size_t q = 0; q != n_queries; ++q) freq_sums[q] = h_sums[q];
        if (freq_blocks)
            for (uint32_t c : h_freq_counts) *freq_blocks += c;
    }
    return DINT_OK;
}

// ---- block statistics ------------------------------------------------------------------------------

int dint_count_ngrams(int device, int multi, const uint32_t* d_gaps, uint64_t n_ints, const uint64_t* list_starts,
                      uint64_t n_lists, uint32_t top_k, dint_ngram** entries, size_t* n_entries, float* kernel_ms) {
    if (!entries || !n_entries || (n_lists && (!list_starts || !d_gaps))) return DINT_ERR_ARG;
    *entries = nullptr;
    *n_entries = 0;
    if (kernel_ms) *kernel_ms = 0.f;
    int count = 0;
    if (!hip_ok(hipGetDeviceCount(&count), "hipGetDeviceCount") || device < 0 || device >= count) return DINT_ERR_NO_DEVICE;
    // chunks: 256 integers of a list, aligned to the list's start
    std::vector<uint64_t> h_start;
    std::vector<uint32_t> h_n;
    uint64_t ngrams = 0, total_ints = 0;
    for (uint64_t l = 0; l != n_lists; ++l) {
        if (list_starts[l + 1] < list_starts[l] || list_starts[l + 1] > n_ints) return DINT_ERR_ARG;
        const uint64_t n = list_starts[l + 1] - list_starts[l];
        total_ints += n;
        for (uint64_t at = 0; at < n; at += kBlock) {
            const uint32_t c = uint32_t(std::min<uint64_t>(kBlock, n - at));
            if (multi && c != kBlock) break;
            h_start.push_back(list_starts[l] + at);
            h_n.push_back(c);
            ngrams += c + c / 2 + c / 4 + c / 8 + c / 16;
        }
    }
    if (h_start.empty()) return DINT_OK;
    HIP_TRY(hipSetDevice(device));
    // table: at least twice the n-grams (their distinct ones are far fewer), a power of two, 2^31 slots at most
    uint64_t slots = 1024;
    while (slots < 2 * ngrams && slots < (1ull << 31)) slots <<= 1;
    const size_t n_chunks = h_start.size();
    unsigned long long *d_keys = nullptr, *d_info = nullptr, *d_n_out = nullptr;
    uint32_t *d_freq = nullptr, *d_chunk_n = nullptr, *d_overflow = nullptr;
    uint64_t* d_chunk_start = nullptr;
    dint_ngram* d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        for (void* p : {static_cast<void*>(d_keys), static_cast<void*>(d_info), static_cast<void*>(d_n_out), static_cast<void*>(d_freq),
                        static_cast<void*>(d_chunk_n), static_cast<void*>(d_overflow), static_cast<void*>(d_chunk_start),
                        static_cast<void*>(d_out)})
            if (p) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    auto fail = [&](int st) {
        cleanup();
        return st;
    };
    if (!hip_ok(counted_malloc(&d_keys, slots * 8), "counted_malloc(ngram keys)") || !hip_ok(counted_malloc(&d_info, slots * 8), "counted_malloc(ngram info)") ||
        !hip_ok(counted_malloc(&d_freq, slots * 4), "counted_malloc(ngram freq)") || !hip_ok(counted_malloc(&d_n_out, 8), "hipMalloc") ||
        !hip_ok(counted_malloc(&d_overflow, 4), "hipMalloc") || !hip_ok(counted_malloc(&d_chunk_start, n_chunks * 8), "counted_malloc(chunks)") ||
        !hip_ok(counted_malloc(&d_chunk_n, n_chunks * 4), "counted_malloc(chunks)"))
        return fail(DINT_ERR_HIP);
    if (!hip_ok(hipMemset(d_keys, 0, slots * 8), "hipMemset") || !hip_ok(hipMemset(d_info, 0xFF, slots * 8), "hipMemset") ||
        !hip_ok(hipMemset(d_freq, 0, slots * 4), "hipMemset") || !hip_ok(hipMemset(d_n_out, 0, 8), "hipMemset") ||
        !hip_ok(hipMemset(d_overflow, 0, 4), "hipMemset") ||
        !hip_ok(hipMemcpy(d_chunk_start, h_start.data(), n_chunks * 8, hipMemcpyHostToDevice), "hipMemcpy") ||
        !hip_ok(hipMemcpy(d_chunk_n, h_n.data(), n_chunks * 4, hipMemcpyHostToDevice), "hipMemcpy") ||
        !hip_ok(hipEventCreate(&e0), "hipEventCreate") || !hip_ok(hipEventCreate(&e1), "hipEventCreate"))
        return fail(DINT_ERR_HIP);
    ngram_table t{d_keys, d_info, d_freq, slots - 1, d_overflow};
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(count_ngrams_kernel, dim3(uint32_t((n_chunks + kStatsWaves - 1) / kStatsWaves)), dim3(64 * kStatsWaves), 0, nullptr,
                       d_gaps, d_chunk_start, d_chunk_n, uint64_t(n_chunks), uint32_t(multi != 0), t);
    (void)hipEventRecord(e1, nullptr);
    uint32_t overflow = 0;
    if (!hip_ok(hipGetLastError(), "count_ngrams_kernel") || !hip_ok(hipDeviceSynchronize(), "count_ngrams_kernel") ||
        !hip_ok(hipMemcpy(&overflow, d_overflow, 4, hipMemcpyDeviceToHost), "hipMemcpy"))
        return fail(DINT_ERR_HIP);
    if (overflow) return fail(DINT_ERR_NOMEM);  // more distinct n-grams than the table holds
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, e0, e1);
    // the occupied slots, compacted (first a count, then the entries)
    const uint64_t cap = std::min<uint64_t>(ngrams, slots);
    if (!hip_ok(counted_malloc(&d_out, cap * sizeof(dint_ngram)), "counted_malloc(ngram entries)")) return fail(DINT_ERR_HIP);
    hipLaunchKernelGGL(collect_ngrams_kernel, dim3(uint32_t((slots + 255) / 256)), dim3(256), 0, nullptr, t, d_out, d_n_out, cap);
    unsigned long long n_out = 0;
    if (!hip_ok(hipGetLastError(), "collect_ngrams_kernel") || !hip_ok(hipMemcpy(&n_out, d_n_out, 8, hipMemcpyDeviceToHost), "hipMemcpy"))
        return fail(DINT_ERR_HIP);
    if (n_out > cap) return fail(DINT_ERR_HIP);
    if (top_k != 0 && n_out > top_k) {
        // per context: the largest count c with at least top_k kept n-grams of count >= c (1 if there are fewer)
        uint32_t* d_at_least = nullptr;
        unsigned long long* d_counts = nullptr;
        dint_ngram* d_sel = nullptr;
        auto fail2 = [&](int st) {
            for (void* p : {static_cast<void*>(d_at_least), static_cast<void*>(d_counts), static_cast<void*>(d_sel)})
                if (p) (void)hipFree(p);
            return fail(st);
        };
        if (!hip_ok(counted_malloc(&d_at_least, 32), "hipMalloc") || !hip_ok(counted_malloc(&d_counts, 64), "hipMalloc")) return fail2(DINT_ERR_HIP);
        uint32_t lo[8], hi[8], mid[8];
        for (int c = 0; c != 8; ++c) lo[c] = 1, hi[c] = 0xFFFFFFFFu;  // invariant: count(>= lo) >= top_k or lo == 1
        const uint32_t grid = uint32_t((n_out + 255) / 256);
        for (int it = 0; it != 33; ++it) {
            bool open = false;
            for (int c = 0; c != 8; ++c) {
                mid[c] = lo[c] + uint32_t((uint64_t(hi[c]) - lo[c] + 1) / 2);
                open = open || lo[c] < hi[c];
            }
            if (!open) break;
            unsigned long long h_counts[8];
            if (!hip_ok(hipMemcpy(d_at_least, mid, 32, hipMemcpyHostToDevice), "hipMemcpy") || !hip_ok(hipMemset(d_counts, 0, 64), "hipMemset"))
                return fail2(DINT_ERR_HIP);
            hipLaunchKernelGGL(count_at_least_kernel, dim3(grid), dim3(256), 0, nullptr, d_out, uint64_t(n_out), double(total_ints), d_at_least,
                               d_counts);
            if (!hip_ok(hipMemcpy(h_counts, d_counts, 64, hipMemcpyDeviceToHost), "hipMemcpy")) return fail2(DINT_ERR_HIP);
            for (int c = 0; c != 8; ++c) {
                if (lo[c] >= hi[c]) continue;
                if (h_counts[c] >= top_k) lo[c] = mid[c];
                else hi[c] = mid[c] - 1;
            }
        }
        if (!hip_ok(hipMemcpy(d_at_least, lo, 32, hipMemcpyHostToDevice), "hipMemcpy") || !hip_ok(hipMemset(d_n_out, 0, 8), "hipMemset") ||
            !hip_ok(counted_malloc(&d_sel, n_out * sizeof(dint_ngram)), "counted_malloc(selected ngrams)"))
            return fail2(DINT_ERR_HIP);
        hipLaunchKernelGGL(keep_at_least_kernel, dim3(grid), dim3(256), 0, nullptr, d_out, uint64_t(n_out), d_at_least, d_sel, d_n_out);
        unsigned long long n_sel = 0;
        if (!hip_ok(hipGetLastError(), "keep_at_least_kernel") || !hip_ok(hipMemcpy(&n_sel, d_n_out, 8, hipMemcpyDeviceToHost), "hipMemcpy") ||
            n_sel > n_out)
            return fail2(DINT_ERR_HIP);
        (void)hipFree(d_out);
        d_out = d_sel;
        d_sel = nullptr;
        n_out = n_sel;
        (void)hipFree(d_at_least);
        (void)hipFree(d_counts);
    }
    dint_ngram* mem = static_cast<dint_ngram*>(std::malloc(std::max<size_t>(1, n_out) * sizeof(dint_ngram)));
    if (!mem) return fail(DINT_ERR_NOMEM);
    if (n_out && !hip_ok(hipMemcpy(mem, d_out, n_out * sizeof(dint_ngram), hipMemcpyDeviceToHost), "hipMemcpy")) {
        std::free(mem);
        return fail(DINT_ERR_HIP);
    }
    cleanup();
    *entries = mem;
    *n_entries = size_t(n_out);
    return DINT_OK;
}

int dint_select_ngrams(int device, const uint32_t* d_gaps, uint64_t n_ints, uint64_t total_ints, dint_ngram* entries, size_t n_entries,
                       uint32_t top_k, size_t* n_selected) {
    if (!n_selected || (n_entries && (!entries || !d_gaps)) || top_k == 0) return DINT_ERR_ARG;
    *n_selected = 0;
    if (n_entries == 0) return DINT_OK;
    int count = 0;
    if (!hip_ok(hipGetDeviceCount(&count), "hipGetDeviceCount") || device < 0 || device >= count) return DINT_ERR_NO_DEVICE;
    for (size_t i = 0; i != n_entries; ++i)  // (the comparator reads the integers: they must lie inside d_gaps)
        if (entries[i].len == 0 || entries[i].len > kMaxEntry || entries[i].pos > n_ints || n_ints - entries[i].pos < entries[i].len ||
            entries[i].ctx >= 8)
            return DINT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    dint_ngram *d_in = nullptr, *d_kept = nullptr, *d_sorted = nullptr;
    unsigned long long* d_ctl = nullptr;  // [0] kept count, [1..8] first index per context, [9..16] output base per context
    void* d_tmp = nullptr;
    auto cleanup = [&]() {
        for (void* p : {static_cast<void*>(d_in), static_cast<void*>(d_kept), static_cast<void*>(d_sorted), static_cast<void*>(d_ctl), d_tmp})
            if (p) (void)hipFree(p);
    };
    auto fail = [&](int st) {
        cleanup();
        return st;
    };
    const size_t bytes = n_entries * sizeof(dint_ngram);
    if (!hip_ok(counted_malloc(&d_in, bytes), "hipMalloc") || !hip_ok(counted_malloc(&d_kept, bytes), "hipMalloc") ||
        !hip_ok(counted_malloc(&d_sorted, bytes), "hipMalloc") || !hip_ok(counted_malloc(&d_ctl, 17 * 8), "hipMalloc") ||
        !hip_ok(hipMemcpy(d_in, entries, bytes, hipMemcpyHostToDevice), "hipMemcpy") || !hip_ok(hipMemset(d_ctl, 0, 8), "hipMemset"))
        return fail(DINT_ERR_HIP);
    const uint32_t grid = uint32_t((n_entries + 255) / 256);
    hipLaunchKernelGGL(keep_filtered_kernel, dim3(grid), dim3(256), 0, nullptr, d_in, uint64_t(n_entries), double(total_ints), d_kept, d_ctl);
    unsigned long long n_kept = 0;
    if (!hip_ok(hipGetLastError(), "keep_filtered_kernel") || !hip_ok(hipMemcpy(&n_kept, d_ctl, 8, hipMemcpyDeviceToHost), "hipMemcpy") ||
        n_kept > n_entries)
        return fail(DINT_ERR_HIP);
    if (n_kept == 0) {
        cleanup();
        return DINT_OK;
    }
    ngram_dictionary_order order{d_gaps};
    size_t tmp_bytes = 0;
    if (!hip_ok(rocprim::merge_sort(nullptr, tmp_bytes, d_kept, d_sorted, size_t(n_kept), order, nullptr), "rocprim::merge_sort(size)") ||
        !hip_ok(counted_malloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)), "hipMalloc") ||
        !hip_ok(rocprim::merge_sort(d_tmp, tmp_bytes, d_kept, d_sorted, size_t(n_kept), order, nullptr), "rocprim::merge_sort"))
        return fail(DINT_ERR_HIP);
    // the first top_k of every context: where its run starts, how many it gives, where they go
    unsigned long long ctl[17];
    for (int c = 0; c != 8; ++c) ctl[1 + c] = n_kept;
    if (!hip_ok(hipMemcpy(d_ctl + 1, ctl + 1, 64, hipMemcpyHostToDevice), "hipMemcpy")) return fail(DINT_ERR_HIP);
    const uint32_t kgrid = uint32_t((n_kept + 255) / 256);
    hipLaunchKernelGGL(context_starts_kernel, dim3(kgrid), dim3(256), 0, nullptr, d_sorted, uint64_t(n_kept), d_ctl + 1);
    if (!hip_ok(hipMemcpy(ctl + 1, d_ctl + 1, 64, hipMemcpyDeviceToHost), "hipMemcpy")) return fail(DINT_ERR_HIP);
    unsigned long long out_n = 0;
    for (int c = 0; c != 8; ++c) {
        unsigned long long end = n_kept;  // the run ends where the next context that has entries begins
        for (int d = c + 1; d != 8; ++d)
            if (ctl[1 + d] < n_kept) {
                end = ctl[1 + d];
                break;
            }
        const unsigned long long have = ctl[1 + c] < n_kept ? end - ctl[1 + c] : 0;
        ctl[9 + c] = out_n;
        out_n += std::min<unsigned long long>(have, top_k);
    }
    if (!hip_ok(hipMemcpy(d_ctl + 9, ctl + 9, 64, hipMemcpyHostToDevice), "hipMemcpy")) return fail(DINT_ERR_HIP);
    hipLaunchKernelGGL(take_top_kernel, dim3(kgrid), dim3(256), 0, nullptr, d_sorted, uint64_t(n_kept), d_ctl + 1, d_ctl + 9, top_k, d_in);
    if (!hip_ok(hipGetLastError(), "take_top_kernel") || !hip_ok(hipMemcpy(entries, d_in, out_n * sizeof(dint_ngram), hipMemcpyDeviceToHost), "hipMemcpy"))
        return fail(DINT_ERR_HIP);
    cleanup();
    *n_selected = size_t(out_n);
    return DINT_OK;
}

int dint_last_kernel_ms(const dint_dict* dd, float* ms) {
    if (!dd || !ms) return DINT_ERR_ARG;
    int slot;
    {
        std::lock_guard<std::mutex> lock(const_cast<dint_dict*>(dd)->launch_mutex);
        slot = dd->last_slot;
    }
    if (slot < 0) return DINT_ERR_ARG;
    HIP_TRY(hipEventSynchronize(dd->slot_stop[slot]));
    HIP_TRY(hipEventElapsedTime(ms, dd->slot_start[slot], dd->slot_stop[slot]));
    return DINT_OK;
}

// The host-pointer calls' workspace: `pin_bytes` of pinned and `dev_bytes` of device memory, grown when too small
// (the caller holds host_mutex).
static int host_workspace(dint_dict* dd, size_t pin_bytes, size_t dev_bytes) {
    HIP_TRY(hipSetDevice(dd->device));
    if (!dd->host_stream) HIP_TRY(hipStreamCreateWithFlags(&dd->host_stream, hipStreamNonBlocking));
    if (dd->h_pin_cap < pin_bytes) {
        HIP_TRY(hipStreamSynchronize(dd->host_stream));
        if (dd->h_pin) HIP_TRY(hipHostFree(dd->h_pin));
        dd->h_pin = nullptr, dd->h_pin_cap = 0;
        const size_t want = std::max<size_t>(pin_bytes + pin_bytes / 2, 64 << 10);
        HIP_TRY(counted_host_malloc(&dd->h_pin, want));
        dd->h_pin_cap = want;
    }
    if (dd->d_host_cap < dev_bytes) {
        HIP_TRY(hipStreamSynchronize(dd->host_stream));
        if (dd->d_host) HIP_TRY(hipFree(dd->d_host));
        dd->d_host = nullptr, dd->d_host_cap = 0;
        const size_t want = std::max<size_t>(dev_bytes + dev_bytes / 2, 64 << 10);
        HIP_TRY(counted_malloc(&dd->d_host, want));
        dd->d_host_cap = want;
    }
    return DINT_OK;
}

static size_t up256(size_t b) { return (b + 255) / 256 * 256; }

int dint_decode_block_host(const dint_dict* dd_c, const uint8_t* in, size_t in_bytes, uint32_t* out, uint32_t sum_of_values,
                           size_t n, size_t* consumed) {
    if (!dd_c || (!in && in_bytes) || (!out && n)) return DINT_ERR_ARG;
    if (consumed) *consumed = 0;
    if (n == 0) return DINT_OK;
    if (n > kBlock || in_bytes < 1) return DINT_ERR_ARG;
    if (n == kBlock) return dint_decode_list_host(dd_c, in, in_bytes, out, n, consumed);  // one unit of one block
    // a short block: binary interpolative, one lane of the tails kernel
    dint_dict* dd = const_cast<dint_dict*>(dd_c);
    std::lock_guard<std::mutex> lock(dd->host_mutex);
    struct head {
        dint_block_ref ref;
        uint64_t docs_end, end;
        uint32_t tail, n_tails;
    };
    const size_t padded = in_bytes + 8;  // the bit reader fetches whole words
    const size_t b_head = up256(sizeof(head)), b_in = up256(padded), b_out = up256(kBlock * 4);
    int st = host_workspace(dd, b_head + b_in + b_out, b_head + b_in + b_out);
    if (st != DINT_OK) return st;
    hipStream_t s = dd->host_stream;
    // pinned: [head][block bytes, zero padded] -> device (one copy); device -> pinned: [head (the end offset)][integers]
    head* h = reinterpret_cast<head*>(dd->h_pin);
    *h = head{};
    const bool vbyte_sum = sum_of_values == 0xFFFFFFFFu;
    h->ref = dint_block_ref{0, 0, uint32_t(n), 0, uint32_t(sum_of_values + uint32_t(n - 1)), 0};  // max - base - (n - 1) = the sum
    h->n_tails = 1;
    std::memcpy(dd->h_pin + b_head, in, in_bytes);
    std::memset(dd->h_pin + b_head + in_bytes, 0, b_in - in_bytes);
    HIP_TRY(hipMemcpyAsync(dd->d_host, dd->h_pin, b_head + b_in, hipMemcpyHostToDevice, s));
    uint8_t* const d_in = dd->d_host + b_head;
    auto* d_ref = reinterpret_cast<dint_block_ref*>(dd->d_host);
    auto* d_docs_end = reinterpret_cast<uint64_t*>(dd->d_host + offsetof(head, docs_end));
    auto* d_end = reinterpret_cast<uint64_t*>(dd->d_host + offsetof(head, end));
    auto* d_tail = reinterpret_cast<uint32_t*>(dd->d_host + offsetof(head, tail));
    auto* d_n_tails = reinterpret_cast<uint32_t*>(dd->d_host + offsetof(head, n_tails));
    auto* d_out = reinterpret_cast<uint32_t*>(dd->d_host + b_head + b_in);
    hipLaunchKernelGGL(interpolative_tails_kernel, dim3(1), dim3(64), kTailLdsBytes, s, d_in, uint64_t(padded), d_ref,
                       vbyte_sum ? d_docs_end : static_cast<uint64_t*>(nullptr), d_tail, d_n_tails, d_out, uint64_t(kBlock),
                       d_end, 0u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dd->h_pin, dd->d_host, b_head, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dd->h_pin + b_head, d_out, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(out, dd->h_pin + b_head, n * 4);
    if (consumed) *consumed = size_t(h->end);
    return DINT_OK;
}

int dint_last_kernel_clock_mhz(const dint_dict* dd, float* mhz) {
    if (!dd || !mhz) return DINT_ERR_ARG;
    int slot;
    {
        std::lock_guard<std::mutex> lock(const_cast<dint_dict*>(dd)->launch_mutex);
        slot = dd->last_slot;
    }
    if (slot < 0) return DINT_ERR_ARG;
    float ms = 0.f;
    uint64_t cycles = 0;
    HIP_TRY(hipSetDevice(dd->device));
    HIP_TRY(hipEventSynchronize(dd->slot_stop[slot]));
    HIP_TRY(hipEventElapsedTime(&ms, dd->slot_start[slot], dd->slot_stop[slot]));
    HIP_TRY(hipMemcpy(&cycles, dd->d_queues + size_t(slot) * (kQueueShards + 1) * kQueueStride + kQueueShards * kQueueStride + kClockWordAt, 8,
                      hipMemcpyDeviceToHost));
    *mhz = ms > 0.f ? float(double(cycles) / (double(ms) * 1e3)) : 0.f;
    return DINT_OK;
}

int dint_recent_kernel_ms(const dint_dict* dd, float* ms, size_t max_n, size_t* n_out) {
    if (!dd || (!ms && max_n) || !n_out) return DINT_ERR_ARG;
    int last;
    uint64_t launches;
    {
        std::lock_guard<std::mutex> lock(const_cast<dint_dict*>(dd)->launch_mutex);
        last = dd->last_slot;
        launches = dd->launches;
    }
    *n_out = 0;
    if (last < 0) return DINT_OK;
    const size_t have = size_t(std::min<uint64_t>(std::min<uint64_t>(launches, dint_dict::kQueueSlots), max_n));
    for (size_t i = 0; i != have; ++i) {  // oldest first
        const uint32_t slot = uint32_t((uint64_t(last) + dint_dict::kQueueSlots - (have - 1 - i)) % dint_dict::kQueueSlots);
        HIP_TRY(hipEventSynchronize(dd->slot_stop[slot]));
        HIP_TRY(hipEventElapsedTime(&ms[i], dd->slot_start[slot], dd->slot_stop[slot]));
    }
    *n_out = have;
    return DINT_OK;
}

int dint_stream_stats_get(const dint_dict* dd, const uint8_t* enc, size_t enc_bytes, dint_stream_stats* st) {
    if (!dd || (!enc && enc_bytes) || !st) return DINT_ERR_ARG;
    *st = dint_stream_stats{};
    const uint8_t* p = enc;
    const uint8_t* end = enc + enc_bytes;
    const bool multi = dd->kind == DINT_DICT_MULTI_PACKED;
    auto codeword = [&](uint32_t d, uint32_t idx) -> uint32_t {  // -> integers it decodes to
        const uint32_t slot = dd->h_start[d] + idx;
        const uint32_t sz = dd->h_size[slot];
        st->codewords += 1;
        if (idx >= 2 && idx < kReserved) st->run_codewords += 1;
        if (idx < dd->h_hot_k[d] && !dd->h_slow[slot]) {
            st->hot_codewords += 1;
            st->hot_ints += sz;
        }
        return sz;
    };
    while (p != end) {
        uint32_t n, universe;
        p = read_vbyte(p, end, &n);
        if (p) p = read_vbyte(p, end, &universe);
        if (!p) return DINT_ERR_FORMAT;
        const uint8_t* const payload = p;
        st->lists += 1;
        st->ints += n;
        uint32_t done = 0;
        while (done < n) {
            const uint32_t bsize = multi ? std::min<uint32_t>(kBlock, n - done) : n - done;
            uint32_t d = 0;
            bool narrow = false;
            if (multi) {
                if (p == end) return DINT_ERR_FORMAT;
                const uint32_t sc = *p++;
                if (sc >= 2 * kSelectors) return DINT_ERR_FORMAT;
                narrow = sc >= kSelectors;
                d = narrow ? sc - kSelectors : sc;
                (narrow ? st->narrow_blocks : st->wide_blocks) += 1;
            }
            const uint32_t limit = dd->h_start[d + 1] - dd->h_start[d];
            uint32_t i = 0;
            while (i < bsize) {
                if (end - p < (narrow ? 1 : 2)) return DINT_ERR_FORMAT;
                const uint32_t idx = narrow ? *p : ld16(p);
                if (idx >= 2) {
                    if (idx >= limit) return DINT_ERR_FORMAT;
                    i += codeword(d, idx);
                    p += narrow ? 1 : 2;
                } else {
                    (idx == 1 ? st->exceptions32 : st->exceptions16) += 1;
                    i += 1;
                    p += (narrow ? 1 : 2) + (idx == 1 ? 4 : 2);
                }
                if (p > end) return DINT_ERR_FORMAT;
            }
            if (i != bsize && multi) return DINT_ERR_FORMAT;
            done += bsize;
        }
        st->payload_bytes += uint64_t(p - payload);
    }
    return DINT_OK;
}

int dint_decode_list_host(const dint_dict* dd_c, const uint8_t* in, size_t in_bytes, uint32_t* out, size_t n,
                          size_t* consumed) {
    if (!dd_c || (!in && in_bytes) || (!out && n)) return DINT_ERR_ARG;
    if (consumed) *consumed = 0;
    if (n == 0) return DINT_OK;
    if (in_bytes < 2 || n > DINT_MAX_UNIT_INTS) return DINT_ERR_ARG;
    dint_dict* dd = const_cast<dint_dict*>(dd_c);
    std::lock_guard<std::mutex> lock(dd->host_mutex);
    const size_t padded = in_bytes < 8 ? 8 : in_bytes;
    // pinned / device: [unit 24 B | end offset 8 B][stream, zero padded][integers]
    const size_t b_head = 256, b_in = up256(padded), b_out = up256(n * 4);
    int st = host_workspace(dd, b_head + std::max(b_in, b_out), b_head + b_in + b_out);
    if (st != DINT_OK) return st;
    hipStream_t s = dd->host_stream;
    dint_unit u{0, 0, uint32_t(n), 0};
    std::memcpy(dd->h_pin, &u, sizeof u);
    std::memset(dd->h_pin + sizeof u, 0, 8);
    std::memcpy(dd->h_pin + b_head, in, in_bytes);
    std::memset(dd->h_pin + b_head + in_bytes, 0, b_in - in_bytes);
    HIP_TRY(hipMemcpyAsync(dd->d_host, dd->h_pin, b_head + b_in, hipMemcpyHostToDevice, s));
    auto* d_unit = reinterpret_cast<dint_unit*>(dd->d_host);
    auto* d_end = reinterpret_cast<uint64_t*>(dd->d_host + sizeof(dint_unit));
    uint8_t* const d_enc = dd->d_host + b_head;
    auto* d_out = reinterpret_cast<uint32_t*>(dd->d_host + b_head + b_in);
    st = dint_decode_units(dd, d_enc, padded, d_unit, 1, d_out, n, d_end, s);
    if (st != DINT_OK) return st;
    HIP_TRY(hipMemcpyAsync(dd->h_pin, dd->d_host, b_head, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dd->h_pin + b_head, d_out, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(out, dd->h_pin + b_head, n * 4);
    uint64_t end_off = 0;
    std::memcpy(&end_off, dd->h_pin + sizeof(dint_unit), 8);
    if (consumed) *consumed = size_t(end_off);
    return DINT_OK;
}

// ---- a posting list decoded once, its blocks then served from host memory ---------------------------------------
struct dint_list_cache {
    size_t n_blocks = 0, list_bytes = 0;
    bool with_freqs = false;
    std::vector<dint_block_ref> blocks;   // in_off: the docs part's offset inside the list
    std::vector<uint64_t> docs_end, freqs_end;
    std::vector<uint32_t> gaps, freqs;    // what Coder::decode returns for the docs / freqs parts, blocks back to back
};

int dint_list_cache_create(const dint_dict* docs_c, const dint_dict* freqs_dict, const uint8_t* list, size_t list_bytes,
                           dint_list_cache** out) {
    if (!docs_c || !list || !out || list_bytes < 2) return DINT_ERR_ARG;
    *out = nullptr;
    if (freqs_dict && (freqs_dict->device != docs_c->device || freqs_dict->kind != docs_c->kind)) return DINT_ERR_ARG;
    dint_block_ref* blocks = nullptr;
    size_t n_blocks = 0;
    uint64_t total = 0;
    const uint64_t zero = 0;
    int st = dint_index_posting_lists(list, list_bytes, &zero, 1, &blocks, &n_blocks, &total);
    if (st != DINT_OK) return st;
    auto* c = new (std::nothrow) dint_list_cache();
    if (!c) {
        dint_free(blocks);
        return DINT_ERR_NOMEM;
    }
    c->n_blocks = n_blocks;
    c->list_bytes = list_bytes;
    c->with_freqs = freqs_dict != nullptr;
    c->blocks.assign(blocks, blocks + n_blocks);
    dint_free(blocks);
    c->docs_end.assign(n_blocks, 0);
    c->freqs_end.assign(n_blocks, 0);
    c->gaps.assign(total, 0);
    if (freqs_dict) c->freqs.assign(total, 0);
    dint_dict* dd = const_cast<dint_dict*>(docs_c);
    dint_dict* fdd = const_cast<dint_dict*>(freqs_dict);
    // (the docs dictionary's workspace and both dictionaries' schedule memory: one call at a time on either)
    std::unique_lock<std::mutex> lock(dd->host_mutex, std::defer_lock), flock;
    if (fdd && fdd != dd) {
        flock = std::unique_lock<std::mutex>(fdd->host_mutex, std::defer_lock);
        std::lock(lock, flock);
    } else {
        lock.lock();
    }
    // device: [index, padded][blocks][units][funits][docs ends][freqs ends][spans][fspans][tails + count][docs out][freqs out]
    const size_t padded = list_bytes + 16;
    const size_t b_index = up256(padded), b_blocks = up256(n_blocks * sizeof(dint_block_ref)), b_units = up256(n_blocks * sizeof(dint_unit)),
                 b_u64 = up256(n_blocks * 8), b_u32 = up256((n_blocks + 1) * 4), b_out = up256(size_t(total) * 4);
    const size_t dev_bytes = b_index + b_blocks + 2 * b_units + 2 * b_u64 + 3 * b_u32 + 2 * b_out;
    const size_t pin_bytes = std::max(b_index + b_blocks, 2 * b_u64 + 2 * b_out);
    auto fail = [&](int code) {
        delete c;
        return code;
    };
    st = host_workspace(dd, pin_bytes, dev_bytes);
    if (st != DINT_OK) return fail(st);
    hipStream_t s = dd->host_stream;
    uint8_t* p = dd->d_host;
    uint8_t* const d_index = p;
    p += b_index;
    auto* d_blocks = reinterpret_cast<dint_block_ref*>(p);
    p += b_blocks;
    auto* d_units = reinterpret_cast<dint_unit*>(p);
    p += b_units;
    auto* d_funits = reinterpret_cast<dint_unit*>(p);
    p += b_units;
    auto* d_dends = reinterpret_cast<uint64_t*>(p);
    p += b_u64;
    auto* d_fends = reinterpret_cast<uint64_t*>(p);
    p += b_u64;
    auto* d_spans = reinterpret_cast<uint32_t*>(p);
    p += b_u32;
    auto* d_fspans = reinterpret_cast<uint32_t*>(p);
    p += b_u32;
    auto* d_tails = reinterpret_cast<uint32_t*>(p);
    p += b_u32;
    auto* d_gaps = reinterpret_cast<uint32_t*>(p);
    p += b_out;
    auto* d_freqs = reinterpret_cast<uint32_t*>(p);
    std::memcpy(dd->h_pin, list, list_bytes);
    std::memset(dd->h_pin + list_bytes, 0, b_index - list_bytes);
    std::memcpy(dd->h_pin + b_index, c->blocks.data(), n_blocks * sizeof(dint_block_ref));
    auto hip = [&](hipError_t e, const char* what) { return hip_ok(e, what) ? DINT_OK : DINT_ERR_HIP; };
    if ((st = hip(hipMemcpyAsync(dd->d_host, dd->h_pin, b_index + b_blocks, hipMemcpyHostToDevice, s), "hipMemcpyAsync(list)")) != DINT_OK) return fail(st);
    if ((st = hip(hipMemsetAsync(d_dends, 0, 2 * b_u64, s), "hipMemsetAsync")) != DINT_OK) return fail(st);
    if ((st = hip(hipMemsetAsync(d_tails + n_blocks, 0, 4, s), "hipMemsetAsync")) != DINT_OK) return fail(st);
    const uint32_t tb = 256, grid = uint32_t((n_blocks + tb - 1) / tb);
    const uint32_t tgrid = uint32_t((n_blocks + kTailLanes - 1) / kTailLanes);  // (sized for "every block is short": at most one is)
    hipLaunchKernelGGL(collect_tails_kernel, dim3(grid), dim3(tb), 0, s, d_blocks, uint64_t(n_blocks), d_tails, d_tails + n_blocks);
    hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, d_blocks, static_cast<const uint64_t*>(nullptr), uint64_t(n_blocks),
                       uint64_t(padded), d_units, d_spans, static_cast<uint32_t*>(nullptr));
    // docs parts as the Coder returns them (gaps), where they end; then the freqs parts from there
    dd->host_sched.valid = false;
    st = launch_decode(docs_c, d_index, padded, d_units, n_blocks, d_gaps, total, d_dends, s, 1, d_spans, 0, nullptr, nullptr, &dd->host_sched);
    if (st != DINT_OK) return fail(st);
    hipLaunchKernelGGL(interpolative_tails_kernel, dim3(tgrid), dim3(64), kTailLdsBytes, s, d_index, uint64_t(padded), d_blocks,
                       static_cast<const uint64_t*>(nullptr), d_tails, d_tails + n_blocks, d_gaps, uint64_t(total), d_dends, 0u);
    if (freqs_dict) {
        hipLaunchKernelGGL(blocks_to_units_kernel, dim3(grid), dim3(tb), 0, s, d_blocks, d_dends, uint64_t(n_blocks), uint64_t(padded), d_funits,
                           d_fspans, static_cast<uint32_t*>(nullptr));
        fdd->host_sched.valid = false;  // (the freqs dictionary's own schedule memory)
        st = launch_decode(freqs_dict, d_index, padded, d_funits, n_blocks, d_freqs, total, d_fends, s, 1, d_fspans, 0, nullptr, nullptr,
                           &fdd->host_sched);
        if (st != DINT_OK) return fail(st);
        hipLaunchKernelGGL(interpolative_tails_kernel, dim3(tgrid), dim3(64), kTailLdsBytes, s, d_index, uint64_t(padded), d_blocks, d_dends, d_tails,
                           d_tails + n_blocks, d_freqs, uint64_t(total), d_fends, 0u);
    }
    if ((st = hip(hipGetLastError(), "launch")) != DINT_OK) return fail(st);
    uint8_t* hp = dd->h_pin;
    if ((st = hip(hipMemcpyAsync(hp, d_dends, 2 * b_u64, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(ends)")) != DINT_OK) return fail(st);
    if ((st = hip(hipMemcpyAsync(hp + 2 * b_u64, d_gaps, (freqs_dict ? 2 : 1) * b_out, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(out)")) != DINT_OK)
        return fail(st);
    if ((st = hip(hipStreamSynchronize(s), "hipStreamSynchronize")) != DINT_OK) return fail(st);
    std::memcpy(c->docs_end.data(), hp, n_blocks * 8);
    std::memcpy(c->freqs_end.data(), hp + b_u64, n_blocks * 8);
    std::memcpy(c->gaps.data(), hp + 2 * b_u64, size_t(total) * 4);
    if (freqs_dict) std::memcpy(c->freqs.data(), hp + 2 * b_u64 + b_out, size_t(total) * 4);
    *out = c;
    return DINT_OK;
}

void dint_list_cache_destroy(dint_list_cache* c) { delete c; }

int dint_list_cache_decode(const dint_list_cache* c, size_t in_offset, uint32_t* out, size_t n, size_t* consumed) {
    if (!c || (!out && n)) return DINT_ERR_ARG;
    if (consumed) *consumed = 0;
    // the block whose docs part starts at in_offset, else the one whose freqs part does (where its docs part ended)
    auto it = std::lower_bound(c->blocks.begin(), c->blocks.end(), in_offset,
                               [](const dint_block_ref& b, size_t off) { return b.in_off < off; });
    size_t b = size_t(it - c->blocks.begin());
    if (b < c->n_blocks && c->blocks[b].in_off == in_offset) {
        if (n != c->blocks[b].n) return DINT_ERR_ARG;
        if (c->docs_end[b] < in_offset) return DINT_ERR_FORMAT;  // (a block the decode did not reach: its end was never written)
        std::memcpy(out, c->gaps.data() + c->blocks[b].out_off, n * 4);
        if (consumed) *consumed = size_t(c->docs_end[b] - in_offset);
        return DINT_OK;
    }
    if (!c->with_freqs || b == 0) return DINT_ERR_ARG;
    b -= 1;  // the last block that starts before in_offset
    if (c->docs_end[b] != in_offset || n != c->blocks[b].n) return DINT_ERR_ARG;
    if (c->freqs_end[b] < in_offset) return DINT_ERR_FORMAT;
    std::memcpy(out, c->freqs.data() + c->blocks[b].out_off, n * 4);
    if (consumed) *consumed = size_t(c->freqs_end[b] - in_offset);
    return DINT_OK;
}

int dint_debug_alloc_count(uint64_t* count) {
    if (!count) return DINT_ERR_ARG;
    *count = g_alloc_count.load();
    return DINT_OK;
}

#ifdef DINT_PROFILE
}  // extern "C"
#include "dint_profile_host.inc"
extern "C" {
#endif

// Test hook (not part of the decode ABI): inclusive wave prefix sum of 64 host words.
int dint_debug_wave_scan(const uint32_t* in64, uint32_t* out64) {
    if (!in64 || !out64) return DINT_ERR_ARG;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(counted_malloc(&d_in, 256));
    HIP_TRY(counted_malloc(&d_out, 256));
    HIP_TRY(hipMemcpy(d_in, in64, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_wave_scan_kernel, dim3(1), dim3(64), 0, nullptr, d_in, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out64, d_out, 256, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return DINT_OK;
}

}  // extern "C"
