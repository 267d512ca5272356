// dint_encode — the vroom `encode` tool (reference vroom_env/encode.cpp:283-329) for the three DINT types.
//
//   dint_encode <type> <collection_name> [--dict <dictionary_filename>] [--out <output_filename>]
//   type: single_rect_dint | single_packed_dint | multi_packed_dint
//   collection_name: a ds2i `.docs` or `.freqs` file (include/ds2i/binary_collection.hpp)
//
// Same arguments, same stdout JSON keys (encode.cpp:49-58), the same bytes in <output_filename>. Not in the reference's
// tool: --units <file> writes the decoder's unit table (the sidecar of SURVEY H3: dint_unit records cut every --unit-ints
// integers), --greedy selects single_greedy_dint, --threads N (default DS2I_THREADS, else the machine's).
#include <cstdlib>
#include <iostream>

#include "tool_common.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "Usage " << argv[0] << ":\n"
                  << "\t<type> <collection_name> [--dict <dictionary_filename>] [--out <output_filename>]"
                  << " [--units <unit_table_filename>] [--unit-ints N] [--greedy] [--threads N]" << std::endl;
        return 1;
    }
    try {
        std::string type = argv[1];
        std::string collection_name = argv[2];
        char const* dictionary_filename = nullptr;
        char const* output_filename = nullptr;
        char const* units_filename = nullptr;
        uint32_t unit_ints = 16384;
        int greedy = 0, threads = tool::default_threads();
        for (int i = 3; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "--dict" && i + 1 < argc) dictionary_filename = argv[++i];
            else if (a == "--out" && i + 1 < argc) output_filename = argv[++i];
            else if (a == "--units" && i + 1 < argc) units_filename = argv[++i];
            else if (a == "--unit-ints" && i + 1 < argc) unit_ints = uint32_t(std::atoi(argv[++i]));
            else if (a == "--threads" && i + 1 < argc) threads = std::max(1, std::atoi(argv[++i]));
            else if (a == "--greedy") greedy = 1;
            else throw std::runtime_error("unknown parameter");
        }
        int kind = tool::kind_of_type(type);
        if (kind < 0) {
            std::cerr << "ERROR: unknown type '" << type << "'" << std::endl;
            return 0;  // the reference logs and returns 0 (encode.cpp:324-328)
        }
        if (!dictionary_filename) throw std::runtime_error("dictionary_filename must be specified");
        bool docs;
        std::string ext = tool::extension_of(collection_name);
        if (ext == ".freqs") docs = false;
        else if (ext == ".docs") docs = true;
        else throw std::runtime_error("unsupported file format");

        tool::mapped_file input(collection_name), dict(dictionary_filename);
        std::cerr << (docs ? "encoding docs..." : "encoding freqs...") << std::endl;
        tool::blob enc, units;
        uint64_t num_processed_lists = 0, num_total_ints = 0;
        tool::host_ok(dinth_encode_collection(kind, greedy, dict.data, dict.bytes, input.words(), input.n_words(), docs ? 1 : 0,
                                              unit_ints, threads, &enc.h, units_filename ? &units.h : nullptr,
                                              &num_processed_lists, &num_total_ints),
                      "dinth_encode_collection");

        // print_statistics, encode.cpp:37-59
        const double GiB_space = double(enc.size()) / 1073741824.0;
        const double bpi_space = num_total_ints ? double(enc.size()) * 8.0 / double(num_total_ints) : 0.0;
        std::cerr << "encoded " << num_processed_lists << " lists\nencoded " << num_total_ints << " integers\n"
                  << GiB_space << " [GiB]\nbits x integer: " << bpi_space << std::endl;
        std::cout << "{\"filename\": \"" << collection_name << "\", \"num_sequences\": \"" << num_processed_lists
                  << "\", \"num_integers\": \"" << num_total_ints << "\", \"type\": \"" << type << "\", \"GiB\": \"" << GiB_space
                  << "\", \"bpi\": \"" << bpi_space << "\"}" << std::endl;
        if (output_filename) {  // save_if, encode.cpp:26-35
            std::cerr << "writing encoded data..." << std::endl;
            tool::write_file(output_filename, enc.data(), enc.size());
            std::cerr << "DONE" << std::endl;
        }
        if (units_filename) tool::write_file(units_filename, units.data(), units.size());
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
