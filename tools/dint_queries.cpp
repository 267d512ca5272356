// dint_queries — the reference's `queries` tool (src/queries.cpp:15-153) for the DINT index types, on the device path.
//
//   dint_queries <index_type> <query_type> <index_filename> [--batch] [--runs R] < query_log
//   index_type: single_rect_dint | single_packed_dint | multi_packed_dint        (include/index_types.hpp:73-79)
//   query_type: and | and_freq, several separated by ':' (src/queries.cpp:93-96); the ranked and OR queries are out of scope
//   index_filename: what dint_create_freq_index wrote (dint/index_file.hpp)
//   query_log on stdin: one query per line, term ids separated by blanks (include/ds2i/queries.hpp:15-27)
//
// Like op_perftest (src/queries.cpp:15-61): every query on its own, `runs` passes of which the first is not timed, the total
// of the result counts on stdout, then one stats line with the reference's keys — type, query, avg, q50, q90, q95 (µs). The
// device answers a BATCH per call much faster than a query per call: --batch times the whole log as one call
// and adds "batch_us_per_query" to the line.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <iostream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "dint/index_file.hpp"
#include "dint_hip.h"
#include "tool_common.hpp"

static void dint_ok(int st, const char* what) {
    if (st != DINT_OK) throw std::runtime_error(std::string(what) + ": " + dint_strerror(st) + " " + dint_last_hip_error());
}
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << argv[0] << " <index_type> <query_type> <index_filename> [--batch] [--runs R] < query_log" << std::endl;
        return 1;
    }
    try {
        std::string type = argv[1], query_type = argv[2];
        const char* index_filename = argv[3];
        bool batch = false;
        size_t runs = 10 + 1;  // src/queries.cpp:13
        for (int i = 4; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "--batch") batch = true;
            else if (a == "--runs" && i + 1 < argc) runs = size_t(std::max(2, std::atoi(argv[++i])));
            else throw std::runtime_error("unknown parameter");
        }
        const int kind = tool::kind_of_type(type);
        if (kind < 0) {
            std::cerr << "ERROR: Unknown type " << type << std::endl;  // src/queries.cpp:147-149
            return 0;
        }
        // read_query (queries.hpp:15-27)
        std::vector<std::vector<uint32_t>> queries;
        for (std::string line; std::getline(std::cin, line);) {
            std::istringstream iline(line);
            std::vector<uint32_t> q;
            for (uint32_t t; iline >> t;) q.push_back(t);
            queries.push_back(q);
        }
        std::cerr << "Loading index from " << index_filename << std::endl;
        tool::mapped_file m(index_filename);
        const dint::index_file_view v = dint::view_index_file(m.data, m.bytes);
        if (int(v.header.kind) != kind) throw std::runtime_error("the index file holds another index type");
        const size_t n_lists = size_t(v.header.n_lists);
        for (auto const& q : queries)
            for (uint32_t t : q)
                if (t >= n_lists) throw std::runtime_error("query term " + std::to_string(t) + " is not a list of this index");

        std::string device_name = "unknown";
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, 0) == hipSuccess) device_name = prop.gcnArchName;  // e.g. gfx950:sramecc+:xnack-
        }
        dint_dict *docs_dict = nullptr, *freqs_dict = nullptr;
        dint_ok(dint_dict_create(kind, v.docs_dict, size_t(v.header.docs_dict_bytes), 0, &docs_dict), "dint_dict_create(docs)");
        dint_ok(dint_dict_create(kind, v.freqs_dict, size_t(v.header.freqs_dict_bytes), 0, &freqs_dict), "dint_dict_create(freqs)");
        dint_block_ref* blocks = nullptr;
        size_t n_blocks = 0;
        uint64_t postings = 0;
        dint_ok(dint_index_posting_lists(v.index, size_t(v.header.index_bytes), v.offsets, n_lists, &blocks, &n_blocks, &postings),
                "dint_index_posting_lists");
        uint8_t* d_index = nullptr;
        const size_t index_bytes = size_t(v.header.index_bytes) + 16;  // (the kernels fetch whole words)
        if (hipMalloc(&d_index, index_bytes) != hipSuccess || hipMemset(d_index, 0, index_bytes) != hipSuccess ||
            hipMemcpy(d_index, v.index, size_t(v.header.index_bytes), hipMemcpyHostToDevice) != hipSuccess)
            throw std::runtime_error("could not place the index on the device");
        dint_query_index* qi = nullptr;
        dint_ok(dint_query_index_create(docs_dict, d_index, index_bytes, blocks, n_blocks, n_lists, &qi), "dint_query_index_create");

        std::vector<std::string> types;
        for (size_t a = 0; a <= query_type.size();) {
            size_t b = query_type.find(':', a);
            if (b == std::string::npos) b = query_type.size();
            types.push_back(query_type.substr(a, b - a));
            a = b + 1;
        }
        for (auto const& t : types) {
            if (t != "and" && t != "and_freq") {
                std::cerr << "Unsupported query type: " << t << std::endl;  // src/queries.cpp:108-110
                continue;
            }
            const bool with_freqs = t == "and_freq";
            std::vector<double> query_times;
            uint64_t total = 0, total_one_run = 0;
            for (size_t run = 0; run != runs; ++run) {  // op_perftest
                for (auto const& q : queries) {
                    const uint64_t offs[2] = {0, q.size()};
                    uint64_t results = 0, fsum = 0, fblocks = 0;
                    const double tick = now_us();
                    if (with_freqs) dint_ok(dint_and_queries_freqs(qi, freqs_dict, q.data(), offs, 1, &results, &fsum, &fblocks, nullptr), "dint_and_queries_freqs");
                    else dint_ok(dint_and_queries(qi, q.data(), offs, 1, &results, nullptr), "dint_and_queries");
                    total += results;
                    if (run == 0) total_one_run += results;
                    if (run != 0) query_times.push_back(now_us() - tick);  // first run is not timed
                }
            }
            std::cout << total << std::endl;
            double batch_us = -1;
            if (batch && !queries.empty()) {
                std::vector<uint32_t> terms;
                std::vector<uint64_t> offs(1, 0), counts(queries.size(), 0), fsums(queries.size(), 0);
                for (auto const& q : queries) {
                    terms.insert(terms.end(), q.begin(), q.end());
                    offs.push_back(terms.size());
                }
                double best = 1e300;
                for (size_t run = 0; run != std::min<size_t>(runs, 4); ++run) {
                    uint64_t fblocks = 0;
                    const double tick = now_us();
                    if (with_freqs) dint_ok(dint_and_queries_freqs(qi, freqs_dict, terms.data(), offs.data(), queries.size(), counts.data(), fsums.data(), &fblocks, nullptr), "dint_and_queries_freqs");
                    else dint_ok(dint_and_queries(qi, terms.data(), offs.data(), queries.size(), counts.data(), nullptr), "dint_and_queries");
                    if (run != 0) best = std::min(best, now_us() - tick);
                }
                batch_us = best / double(queries.size());
                // the batch call and the one-query calls answer the same log
                uint64_t batch_total = 0;
                for (uint64_t c : counts) batch_total += c;
                if (batch_total != total_one_run)
                    throw std::runtime_error("the batch call counted " + std::to_string(batch_total) + " results, the one-query calls " +
                                             std::to_string(total_one_run));
            }
            if (query_times.empty()) continue;
            std::sort(query_times.begin(), query_times.end());
            const double avg = std::accumulate(query_times.begin(), query_times.end(), double()) / double(query_times.size());
            const double q50 = query_times[query_times.size() / 2], q90 = query_times[90 * query_times.size() / 100],
                         q95 = query_times[95 * query_times.size() / 100];
            std::cout << "{\"type\": \"" << type << "\", \"query\": \"" << t << "\", \"avg\": " << avg << ", \"q50\": " << q50
                      << ", \"q90\": " << q90 << ", \"q95\": " << q95;
            if (batch_us >= 0) std::cout << ", \"batch_us_per_query\": " << batch_us;
            std::cout << ", \"device\": \"" << device_name << "\"}" << std::endl;
        }
        dint_query_index_destroy(qi);
        dint_free(blocks);
        (void)hipFree(d_index);
        dint_dict_destroy(docs_dict);
        dint_dict_destroy(freqs_dict);
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
