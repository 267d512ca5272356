// The reference's Coder call shape, end to end: builder.load -> prepare_for_encoding ->
// Coder::encode (CPU) -> builder.build(dict) (device) -> Coder::decode (device) == input,
// and the returned pointer is one past the consumed bytes. Mirrors the structure of
// vroom_env/check_encoded_data.cpp:76-113. usage: coder_roundtrip <kind> <dictfile> <gapsfile u32>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "dint/coders.hpp"

template <typename T>
std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<T> v(size_t(f.tellg()) / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), std::streamsize(v.size() * sizeof(T)));
    return v;
}

template <typename Coder, typename Dictionary>
int run(std::vector<uint8_t> const& dict_file, std::vector<uint32_t> const& gaps) {
    typename Dictionary::builder builder;
    builder.load(dict_file);
    builder.prepare_for_encoding();
    Dictionary dict;
    builder.build(dict);
    size_t lens[] = {1, 2, 15, 16, 17, 255, 256, 257, 1000, 5000, gaps.size()};
    for (size_t n : lens) {
        if (n > gaps.size()) continue;
        std::vector<uint8_t> enc;
        uint32_t universe = 0;
        for (size_t i = 0; i != n; ++i) universe += gaps[i];
        Coder::encode(builder, gaps.data(), universe, uint32_t(n), enc);
        size_t produced = enc.size();
        enc.resize(produced + 64, 0xEE);  // bytes after the list, as in a stream
        std::vector<uint32_t> out(n + 4, 0xABABABABu);
        uint8_t const* end = Coder::decode(dict, enc.data(), enc.data() + enc.size(), out.data(), universe, n);
        if (size_t(end - enc.data()) != produced) {
            std::cerr << "n=" << n << ": consumed " << (end - enc.data()) << " of " << produced << " bytes\n";
            return 1;
        }
        for (size_t i = 0; i != n; ++i)
            if (out[i] != gaps[i]) {
                std::cerr << "n=" << n << ": mismatch at " << i << "\n";
                return 1;
            }
        if (out[n] != 0xABABABABu) {
            std::cerr << "n=" << n << ": wrote past the output\n";
            return 1;
        }
    }
    std::cout << "ok\n";
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    try {
        auto dict_file = slurp<uint8_t>(argv[2]);
        auto gaps = slurp<uint32_t>(argv[3]);
        switch (std::atoi(argv[1])) {
            case 0: return run<dint::single_opt_dint_device, dint::single_dictionary_rectangular_type>(dict_file, gaps);
            case 1: return run<dint::single_opt_dint_device, dint::single_dictionary_packed_type>(dict_file, gaps);
            case 2: return run<dint::multi_opt_dint_device, dint::multi_dictionary_packed_type>(dict_file, gaps);
            case 3: return run<dint::single_greedy_dint_device, dint::single_dictionary_packed_type>(dict_file, gaps);
        }
    } catch (std::exception const& e) {
        std::cerr << "exception: " << e.what() << "\n";
        return 1;
    }
    return 2;
}
