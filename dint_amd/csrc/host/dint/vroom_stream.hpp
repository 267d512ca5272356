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

// gaps: all lists back to back; lens[i] = length of list i (zero-length lists are
// skipped, as binary_collection does — include/ds2i/binary_collection.hpp:138).
template <typename Encoder, typename Builder>
vroom_output encode_vroom(Builder& builder, uint32_t const* gaps, uint32_t const* lens, uint64_t n_lists,
                          uint32_t unit_ints, int threads) {
    struct task {
        uint64_t first_list, last_list, first_int;
        std::vector<uint8_t> bytes;
        std::vector<dint_unit> units;  // offsets relative to the task
        uint64_t ints = 0;
    };
    std::vector<task> tasks;
    {
        const uint64_t ints_per_task = 1u << 20;
        uint64_t pos = 0, acc = 0, first = 0, first_int = 0;
        for (uint64_t i = 0; i != n_lists; ++i) {
            acc += lens[i];
            pos += lens[i];
            if (acc >= ints_per_task || i + 1 == n_lists) {
                tasks.push_back({first, i + 1, first_int, {}, {}, 0});
                first = i + 1;
                first_int = pos;
                acc = 0;
            }
        }
    }
    parallel_for(tasks.size(), threads, [&](size_t t) {
        auto& tk = tasks[t];
        uint64_t pos = tk.first_int;
        std::vector<sync_point> syncs;
        for (uint64_t i = tk.first_list; i != tk.last_list; ++i) {
            uint32_t n = lens[i];
            if (n == 0) continue;
            uint32_t const* in = gaps + pos;
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
            pos += n;
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

}  // namespace dint
