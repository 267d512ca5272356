// Shared by the host-side command-line tools (dint_encode, dint_build_dict, dint_create_freq_index): read-only file
// mappings, the reference's type names and file-name rules. g++ only; the tools call the C ABI of include/dint_host.h.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>

#include "dint_host.h"

namespace tool {

struct mapped_file {
    explicit mapped_file(std::string const& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("Error opening file " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) throw std::runtime_error("Error opening file " + path);
        bytes = size_t(st.st_size);
        if (bytes) {
            data = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            if (data == MAP_FAILED) throw std::runtime_error("Error mapping file " + path);
        }
    }
    mapped_file(mapped_file const&) = delete;
    ~mapped_file() {
        if (data && data != MAP_FAILED) ::munmap(data, bytes);
        if (fd >= 0) ::close(fd);
    }
    uint32_t const* words() const { return static_cast<uint32_t const*>(data); }
    size_t n_words() const { return bytes / 4; }
    int fd = -1;
    void* data = nullptr;
    size_t bytes = 0;
};

inline bool file_exists(std::string const& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}

inline void write_file(std::string const& path, void const* p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f.write(static_cast<char const*>(p), std::streamsize(n));
    if (!f) throw std::runtime_error("cannot write " + path);
}

// boost::filesystem::path(p).filename() / .extension() of the reference's tools
inline std::string filename_of(std::string const& path) {
    auto slash = path.find_last_of('/');
    return slash == std::string::npos ? path : path.substr(slash + 1);
}
inline std::string extension_of(std::string const& path) {
    std::string name = filename_of(path);
    auto dot = name.find_last_of('.');
    return dot == std::string::npos ? std::string() : name.substr(dot);
}

// type string -> dictionary kind (vroom_env/encode.cpp:312-320, include/index_types.hpp:73-79); -1: unknown
inline int kind_of_type(std::string const& type) {
    if (type == "single_rect_dint") return DINT_DICT_RECTANGULAR;
    if (type == "single_packed_dint") return DINT_DICT_SINGLE_PACKED;
    if (type == "multi_packed_dint") return DINT_DICT_MULTI_PACKED;
    return -1;
}
// Dictionary::builder::type() (rectangular_dictionary.hpp:151-153, single_dictionary.hpp:194-196, multi_dictionary.hpp:251-253)
inline char const* builder_type(int kind) {
    return kind == DINT_DICT_RECTANGULAR ? "rectangular" : kind == DINT_DICT_SINGLE_PACKED ? "single_packed" : "multi_packed";
}
// "./dict." + filename + "." + d_type::type() + "." + dictionary_builder::type()   (dict_freq_index.hpp:141-147,
// dictionary_builders.hpp:45-48: "DSF-65536-16")
inline std::string dictionary_file_name(std::string const& collection_file, int kind) {
    return "./dict." + filename_of(collection_file) + "." + builder_type(kind) + ".DSF-65536-16";
}

inline int default_threads() {  // configuration.hpp:33: DS2I_THREADS, else the hardware's
    if (char const* e = std::getenv("DS2I_THREADS")) {
        int t = std::atoi(e);
        if (t > 0) return t;
    }
    unsigned hc = std::thread::hardware_concurrency();
    return hc ? int(hc) : 1;
}

inline void host_ok(int status, char const* what) {
    if (status != 0) throw std::runtime_error(std::string(what) + ": " + dinth_last_error());
}

struct blob {  // owns a dinth_blob
    dinth_blob* h = nullptr;
    blob() = default;
    blob(blob const&) = delete;
    ~blob() { dinth_blob_free(h); }
    void const* data() const { return dinth_blob_data(h); }
    size_t size() const { return dinth_blob_size(h); }
};

}  // namespace tool
