// dint_decode — the vroom `decode` tool (reference vroom_env/decode.cpp) on the device path.
//
//   dint_decode <type> <encoded_data_filename> --dict <dictionary_filename> [--unit-ints N] [--runs R]
//   type: single_rect_dint | single_packed_dint | multi_packed_dint
//
// Same inputs as the reference tool; prints the same JSON keys (vroom_env/statistics.hpp:26-34) plus
// GPU fields. Where the reference times each list's decode call on one CPU core, this tool indexes the
// stream on the host (untimed, like the reference's header parsing), keeps the stream resident in HBM
// and times the decode kernel over the whole file with HIP events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "dint_hip.h"

static std::vector<uint8_t> read_file(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("Error opening file ") + path);
    std::vector<uint8_t> bytes(size_t(f.tellg()));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(bytes.data()), std::streamsize(bytes.size()));
    return bytes;
}

#define HIP_OK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static void dint_ok(int st, const char* what) {
    if (st != DINT_OK) throw std::runtime_error(std::string(what) + ": " + dint_strerror(st) + " " + dint_last_hip_error());
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "Usage " << argv[0] << ":\n\t<type> <encoded_data_filename> --dict <dictionary_filename>"
                  << " [--unit-ints N] [--runs R] [--check <file of the expected integers, u32 little-endian>]" << std::endl;
        return 1;
    }
    try {
        std::string type = argv[1];
        const char* encoded = argv[2];
        const char* dict_file = nullptr;
        uint32_t unit_ints = 8192;
        int runs = 5;
        const char* check_file = nullptr;
        for (int i = 3; i < argc; ++i) {
            std::string arg = argv[i];
            if (arg == "--dict" && i + 1 < argc) dict_file = argv[++i];
            else if (arg == "--unit-ints" && i + 1 < argc) unit_ints = uint32_t(std::atoi(argv[++i]));
            else if (arg == "--runs" && i + 1 < argc) runs = std::max(1, std::atoi(argv[++i]));
            else if (arg == "--check" && i + 1 < argc) check_file = argv[++i];
            else throw std::runtime_error("unknown parameter");
        }
        int kind;
        if (type == "single_rect_dint") kind = DINT_DICT_RECTANGULAR;
        else if (type == "single_packed_dint") kind = DINT_DICT_SINGLE_PACKED;
        else if (type == "multi_packed_dint") kind = DINT_DICT_MULTI_PACKED;
        else {
            std::cerr << "ERROR: unknown type '" << type << "'" << std::endl;
            return 0;  // the reference logs and returns 0 (decode.cpp:258)
        }
        if (!dict_file) throw std::runtime_error("dictionary_filename must be specified");

        std::vector<uint8_t> dict_bytes = read_file(dict_file), enc = read_file(encoded);
        dint_dict* dict = nullptr;
        dint_ok(dint_dict_create(kind, dict_bytes.data(), dict_bytes.size(), 0, &dict), "dint_dict_create");
        dint_unit* units = nullptr;
        size_t n_units = 0;
        uint64_t total_ints = 0, n_lists = 0;
        dint_ok(dint_index_stream(dict, enc.data(), enc.size(), unit_ints, &units, &n_units, &total_ints, &n_lists),
                "dint_index_stream");

        uint8_t* d_enc = nullptr;
        dint_unit* d_units = nullptr;
        uint32_t* d_out = nullptr;
        const size_t enc_bytes = std::max<size_t>(enc.size(), 8);
        HIP_OK(hipMalloc(&d_enc, enc_bytes));
        HIP_OK(hipMemset(d_enc, 0, enc_bytes));
        HIP_OK(hipMemcpy(d_enc, enc.data(), enc.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&d_units, std::max<size_t>(1, n_units) * sizeof(dint_unit)));
        HIP_OK(hipMemcpy(d_units, units, n_units * sizeof(dint_unit), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&d_out, std::max<uint64_t>(1, total_ints) * 4));

        std::vector<float> ms(size_t(runs), 0.f);
        for (int r = -1; r < runs; ++r) {  // one warm-up
            dint_ok(dint_decode_units(dict, d_enc, enc_bytes, d_units, n_units, d_out, total_ints, nullptr, nullptr),
                    "dint_decode_units");
            HIP_OK(hipDeviceSynchronize());
            if (r >= 0) dint_ok(dint_last_kernel_ms(dict, &ms[size_t(r)]), "dint_last_kernel_ms");
        }
        std::sort(ms.begin(), ms.end());
        const double elapsed = double(ms[ms.size() / 2]) * 1e-3;  // median
        const double ns_x_int = total_ints ? elapsed * 1e9 / double(total_ints) : 0.0;
        const uint64_t ints_x_sec = ns_x_int > 0 ? uint64_t(1e9 / ns_x_int) : 0;
        dint_dict_info info;
        dint_ok(dint_dict_info_get(dict, &info), "dint_dict_info_get");

        // (not in the reference's tool: the decoded integers against a file of the expected ones)
        int bit_exact = -1;
        if (check_file) {
            std::vector<uint8_t> want = read_file(check_file);
            std::vector<uint32_t> got(total_ints);
            HIP_OK(hipMemcpy(got.data(), d_out, total_ints * 4, hipMemcpyDeviceToHost));
            bit_exact = want.size() == total_ints * 4 && std::memcmp(want.data(), got.data(), want.size()) == 0 ? 1 : 0;
        }
        std::cerr << "elapsed time " << elapsed << " [sec]\n" << ns_x_int << " [ns] x int\n" << ints_x_sec
                  << " ints x [sec]" << std::endl;
        std::cout << "{\"filename\": \"" << encoded << "\", \"num_sequences\": \"" << n_lists
                  << "\", \"num_integers\": \"" << total_ints << "\", \"type\": \"" << type
                  << "\", \"tot_elapsed_time\": \"" << elapsed << "\", \"ns_x_int\": \"" << ns_x_int
                  << "\", \"ints_x_sec\": \"" << ints_x_sec << "\", \"device\": \"gfx950\", \"units\": \"" << n_units
                  << "\", \"hot_codewords_in_lds\": \"" << info.hot_entries << "\", \"runs\": \"" << runs << "\""
                  << (bit_exact < 0 ? "" : bit_exact ? ", \"bit_exact\": \"true\"" : ", \"bit_exact\": \"false\"") << "}" << std::endl;
        dint_free(units);
        (void)hipFree(d_enc);
        (void)hipFree(d_units);
        (void)hipFree(d_out);
        dint_dict_destroy(dict);
        if (bit_exact == 0) return 2;
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
