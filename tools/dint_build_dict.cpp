// dint_build_dict — the dictionary half of the reference's index construction as a tool of its own:
// dict_freq_index::builder::build_model (reference include/dint/dict_freq_index.hpp:52-66, :139-161).
//
//   dint_build_dict <type> <collection_basename> [--sample N] [--threads N] [--force] [--only docs|freqs]
//   type: single_rect_dint | single_packed_dint | multi_packed_dint
//
// For <basename>.docs and <basename>.freqs: "build or load" — the dictionary file
//   ./dict.<collection file name>.<rectangular|single_packed|multi_packed>.DSF-65536-16      (:141-147)
// is left alone when it exists (the reference loads it; --force rebuilds), else built from the block statistics of
// every list of the file (block_statistics.hpp:45-108 / :201-279; --sample N: of the first lists holding at most N
// integers) by decreasing static frequencies (dictionary_builders.hpp:55-75) and written in the builder's own format
// (`builder::write`: what `vroom_env/encode --dict` and `decode --dict` read). One JSON line per dictionary on stdout.
#include <cstdlib>
#include <iostream>

#include "tool_common.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "Usage " << argv[0] << ":\n\t<type> <collection_basename> [--sample N] [--threads N] [--force] [--only docs|freqs]"
                  << std::endl;
        return 1;
    }
    try {
        std::string type = argv[1], basename = argv[2], only;
        uint64_t sample = 0;
        int threads = tool::default_threads();
        bool force = false;
        for (int i = 3; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "--sample" && i + 1 < argc) sample = std::strtoull(argv[++i], nullptr, 10);
            else if (a == "--threads" && i + 1 < argc) threads = std::max(1, std::atoi(argv[++i]));
            else if (a == "--only" && i + 1 < argc) only = argv[++i];
            else if (a == "--force") force = true;
            else throw std::runtime_error("unknown parameter");
        }
        int kind = tool::kind_of_type(type);
        if (kind < 0) {
            std::cerr << "ERROR: Unknown type " << type << std::endl;  // create_freq_index.cpp:148-150
            return 0;
        }
        for (int dt = 0; dt != 2; ++dt) {
            const bool docs = dt == 0;
            if (!only.empty() && only != (docs ? "docs" : "freqs")) continue;
            const std::string file_name = basename + (docs ? ".docs" : ".freqs");  // extension(dt), util.hpp:63-65
            const std::string dictionary_file = tool::dictionary_file_name(file_name, kind);
            std::cerr << "building or loading dictionary for " << (docs ? "docs" : "freqs") << "..." << std::endl;
            bool built = false;
            size_t bytes = 0;
            if (tool::file_exists(dictionary_file) && !force) {
                bytes = tool::mapped_file(dictionary_file).bytes;
            } else {
                tool::mapped_file input(file_name);
                tool::blob dict;
                tool::host_ok(dinth_build_dictionary_collection(kind, input.words(), input.n_words(), docs ? 1 : 0, sample, threads, &dict.h),
                              "dinth_build_dictionary_collection");
                tool::write_file(dictionary_file, dict.data(), dict.size());  // try_store_to_file
                bytes = dict.size();
                built = true;
            }
            std::cerr << "DONE" << std::endl;
            std::cout << "{\"collection\": \"" << file_name << "\", \"type\": \"" << type << "\", \"dictionary\": \"" << dictionary_file
                      << "\", \"bytes\": \"" << bytes << "\", \"built\": \"" << (built ? "true" : "false") << "\"}" << std::endl;
        }
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
