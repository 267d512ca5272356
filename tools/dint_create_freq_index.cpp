// dint_create_freq_index — the reference's `create_freq_index` tool (src/create_freq_index.cpp:112-153) for the three
// DINT index types.
//
//   dint_create_freq_index <index_type> <collection_basename> [<output_filename>] [--greedy] [--threads N]
//   index_type: single_rect_dint | single_packed_dint | multi_packed_dint        (include/index_types.hpp:73-79)
//
// build_model (dict_freq_index.hpp:52-66): the docs and the freqs dictionary are loaded from
// ./dict.<file>.<builder type>.DSF-65536-16 when those files exist, else built from the collection's block statistics and
// stored there; then every posting list of <basename>.docs / .freqs becomes a dict_posting_list
// (dict_posting_list.hpp:10-56, the reference's bytes) and the index is written to <output_filename> in this repo's
// container (dint/index_file.hpp — the reference's succinct::mapper::freeze format is not reproducible here).
// One stats line on stdout with the reference's keys where they apply (type, worker_threads, construction_time) plus sizes.
// The reference's `--check` re-decodes the index on the CPU (verify_collection.hpp); this repo has no CPU decoder in the
// product — check an index on the device: tests/test_gpu_index.py, dint_decode_posting_blocks.
#include <chrono>
#include <cstdlib>
#include <iostream>

#include "dint/index_file.hpp"
#include "tool_common.hpp"

static tool::blob* build_or_load_dict(int kind, std::string const& file_name, bool docs, int threads, tool::mapped_file& input,
                                      std::vector<uint8_t>& storage) {
    const std::string dictionary_file = tool::dictionary_file_name(file_name, kind);
    if (tool::file_exists(dictionary_file)) {  // builder.load_from_file
        tool::mapped_file f(dictionary_file);
        storage.assign(static_cast<uint8_t const*>(f.data), static_cast<uint8_t const*>(f.data) + f.bytes);
        return nullptr;
    }
    auto dict = new tool::blob;
    tool::host_ok(dinth_build_dictionary_collection(kind, input.words(), input.n_words(), docs ? 1 : 0, 0, threads, &dict->h),
                  "dinth_build_dictionary_collection");
    try {
        tool::write_file(dictionary_file, dict->data(), dict->size());
    } catch (std::exception const&) {
        std::cerr << "cannot write dictionary to file" << std::endl;  // dict_freq_index.hpp:157-159
    }
    storage.assign(static_cast<uint8_t const*>(dict->data()), static_cast<uint8_t const*>(dict->data()) + dict->size());
    return dict;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << argv[0] << ":\n\t<index_type> <collection_basename> [<output_filename>] [--greedy] [--threads N]"
                  << std::endl;
        return 1;
    }
    try {
        std::string type = argv[1], basename = argv[2];
        char const* output_filename = nullptr;
        int greedy = 0, threads = tool::default_threads();
        for (int i = 3; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "--greedy") greedy = 1;
            else if (a == "--threads" && i + 1 < argc) threads = std::max(1, std::atoi(argv[++i]));
            else if (a == "--check") std::cerr << "--check: not available on the CPU (see the header of this tool)" << std::endl;
            else if (!output_filename && a.rfind("--", 0) != 0) output_filename = argv[i];
            else throw std::runtime_error("unknown parameter");
        }
        int kind = tool::kind_of_type(type);
        if (kind < 0) {
            std::cerr << "ERROR: Unknown type " << type << std::endl;
            return 0;
        }
        auto tick = std::chrono::steady_clock::now();
        tool::mapped_file docs(basename + ".docs"), freqs(basename + ".freqs");
        std::vector<uint8_t> docs_dict, freqs_dict;
        std::cerr << "building or loading dictionary for docs..." << std::endl;
        delete build_or_load_dict(kind, basename + ".docs", true, threads, docs, docs_dict);
        std::cerr << "building or loading dictionary for freqs..." << std::endl;
        delete build_or_load_dict(kind, basename + ".freqs", false, threads, freqs, freqs_dict);

        tool::blob index, offsets;
        uint64_t num_docs = 0;
        tool::host_ok(dinth_build_index_collection(kind, greedy, docs_dict.data(), docs_dict.size(), freqs_dict.data(), freqs_dict.size(),
                                                   docs.words(), docs.n_words(), freqs.words(), freqs.n_words(), threads, &index.h,
                                                   &offsets.h, &num_docs),
                      "dinth_build_index_collection");
        const uint64_t n_lists = offsets.size() / 8 - 1;
        const double elapsed_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - tick).count();
        std::cerr << type << " collection built in " << elapsed_secs << " seconds" << std::endl;
        // the postings actually indexed: the lengths the lists' own headers carry (a .docs file may hold empty records, which
        // the reader skips — their length words are no postings — or a truncated tail: binary_collection.hpp:131-146)
        uint64_t postings = 0;
        {
            auto const* offs = static_cast<uint64_t const*>(offsets.data());
            auto const* bytes = static_cast<uint8_t const*>(index.data());
            for (uint64_t i = 0; i != n_lists; ++i) {
                uint64_t n = 0;
                unsigned shift = 0;
                for (uint8_t const* p = bytes + offs[i];; ++p, shift += 7) {  // TightVariableByte: the LAST byte has bit 7 set
                    n += uint64_t(*p & 127u) << shift;
                    if (*p & 128u) break;
                }
                postings += n;
            }
        }
        std::cout << "{\"type\": \"" << type << "\", \"worker_threads\": " << threads << ", \"construction_time\": " << elapsed_secs
                  << ", \"sequences\": " << n_lists << ", \"postings\": " << postings << ", \"num_docs\": " << num_docs
                  << ", \"lists_bytes\": " << index.size() << ", \"docs_dict_bytes\": " << docs_dict.size()
                  << ", \"freqs_dict_bytes\": " << freqs_dict.size()
                  << ", \"bits_per_posting\": " << (postings ? double(index.size()) * 8.0 / double(postings) : 0.0) << "}" << std::endl;
        if (output_filename)
            dint::write_index_file(output_filename, uint32_t(kind), uint32_t(greedy), num_docs, n_lists,
                                   static_cast<uint64_t const*>(offsets.data()), docs_dict.data(), docs_dict.size(), freqs_dict.data(),
                                   freqs_dict.size(), index.data(), index.size());
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
