// Writer for the vroom encoded stream (+ the unit-table sidecar).
//
// Stream layout (reference vroom_env/jobs.hpp:74-95, vroom_env/encode.cpp:133-191):
//   { vbyte(n) vbyte(universe) payload }*   with universe = sum of the gaps (u32 wrap)
// Lists are encoded in parallel and committed in order, like the reference's
// semiasync_queue (include/ds2i/semiasync_queue.hpp:57-85), but with plain
// threads over contiguous groups of lists.
#pragma once
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "binary_collection.hpp"
#include "dint_hip.h"
#include "encoders.hpp"
#include "vbyte.hpp"

namespace dint {

struct vroom_output {
    std::vector<uint8_t> bytes;
    std::vector<dint_unit> units;
    uint64_t total_ints = 0;
};

template <typename Fn>
void parallel_for(size_t n_tasks, int threads, Fn&& fn) {
    if (threads <= 1 || n_tasks <= 1) {
        for (size_t t = 0; t != n_tasks; ++t) fn(t);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int w = 0; w != threads; ++w) {
        pool.emplace_back([&] {
            for (size_t t; (t = next.fetch_add(1)) < n_tasks;) fn(t);
        });
    }
    for (auto& th : pool) th.join();
}

// The engine: list i has len_of(i) integers and gaps_of(i, scratch) returns them (a pointer into the caller's
// memory, or into `scratch` after filling it). Zero-length lists are skipped, as binary_collection does
// (include/ds2i/binary_collection.hpp:138).
template <typename Encoder, typename Builder, typename LenOf, typename GapsOf>
vroom_output encode_vroom_lists(Builder& builder, uint64_t n_lists, LenOf&& len_of, GapsOf&& gaps_of, uint32_t unit_ints,
                                int threads) {
    struct task {
        uint64_t first_list, last_list;
        std::vector<uint8_t> bytes;
        std::vector<dint_unit> units;  // offsets relative to the task
        uint64_t ints = 0;
    };
    std::vector<task> tasks;
    {
        const uint64_t ints_per_task = 1u << 20;
        uint64_t acc = 0, first = 0;
        for (uint64_t i = 0; i != n_lists; ++i) {
            acc += len_of(i);
            if (acc >= ints_per_task || i + 1 == n_lists) {
                tasks.push_back({first, i + 1, {}, {}, 0});
                first = i + 1;
                acc = 0;
            }
        }
    }
    parallel_for(tasks.size(), threads, [&](size_t t) {
        auto& tk = tasks[t];
        std::vector<sync_point> syncs;
        std::vector<uint32_t> scratch;
        for (uint64_t i = tk.first_list; i != tk.last_list; ++i) {
            uint32_t n = uint32_t(len_of(i));
            if (n == 0) continue;
            uint32_t const* in = gaps_of(i, scratch);
            uint32_t universe = 0;
            for (uint32_t k = 0; k != n; ++k) universe += in[k];
            list_header::write(n, universe, tk.bytes);
            uint64_t payload = tk.bytes.size();
            syncs.clear();
            Encoder::encode(builder, in, universe, n, tk.bytes, unit_ints ? &syncs : nullptr, unit_ints);
            uint64_t prev_byte = 0, prev_int = 0;
            for (size_t s = 0; s <= syncs.size(); ++s) {
                uint64_t next_int = s == syncs.size() ? n : syncs[s].int_off;
                tk.units.push_back({payload + prev_byte, tk.ints + prev_int, uint32_t(next_int - prev_int), uint32_t(i)});
                if (s != syncs.size()) {
                    prev_byte = syncs[s].byte_off;
                    prev_int = syncs[s].int_off;
                }
            }
            tk.ints += n;
        }
    });
    vroom_output out;
    size_t total_bytes = 0, total_units = 0;
    for (auto& tk : tasks) {
        total_bytes += tk.bytes.size();
        total_units += tk.units.size();
    }
    out.bytes.reserve(total_bytes);
    out.units.reserve(total_units);
    for (auto& tk : tasks) {
        uint64_t byte_base = out.bytes.size();
        for (auto u : tk.units) {
            u.in_off += byte_base;
            u.out_off += out.total_ints;
            out.units.push_back(u);
        }
        out.bytes.insert(out.bytes.end(), tk.bytes.begin(), tk.bytes.end());
        out.total_ints += tk.ints;
        std::vector<uint8_t>().swap(tk.bytes);
        std::vector<dint_unit>().swap(tk.units);
    }
    return out;
}

// gaps: all lists back to back; lens[i] = length of list i.
template <typename Encoder, typename Builder>
vroom_output encode_vroom(Builder& builder, uint32_t const* gaps, uint32_t const* lens, uint64_t n_lists,
                          uint32_t unit_ints, int threads) {
    std::vector<uint64_t> starts(n_lists + 1, 0);
    for (uint64_t i = 0; i != n_lists; ++i) starts[i + 1] = starts[i] + lens[i];
    return encode_vroom_lists<Encoder>(
        builder, n_lists, [&](uint64_t i) { return lens[i]; },
        [&](uint64_t i, std::vector<uint32_t>&) { return gaps + starts[i]; }, unit_ints, threads);
}

// The vroom `encode` program's loop (vroom_env/encode.cpp:133-191, jobs.hpp:74-95) over a collection file:
// docs = true: a .docs file — record 0 (the number of documents) is skipped, docIDs become d-gaps minus one;
// docs = false: a .freqs file, every value minus one.
template <typename Encoder, typename Builder>
vroom_output encode_vroom_collection(Builder& builder, binary_collection const& input, bool docs, uint32_t unit_ints,
                                     int threads) {
    auto lists = input.sequences();
    if (docs && !lists.empty()) lists.erase(lists.begin());
    return encode_vroom_lists<Encoder>(
        builder, lists.size(), [&](uint64_t i) { return lists[i].size(); },
        [&](uint64_t i, std::vector<uint32_t>& scratch) {
            scratch.resize(lists[i].size());
            list_to_gaps(lists[i], docs, scratch.data());
            return static_cast<uint32_t const*>(scratch.data());
        },
        unit_ints, threads);
}

}  // namespace dint
