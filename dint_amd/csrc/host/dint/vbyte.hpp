// Variable-byte integers and the vroom per-list header.
//
// Format (reference vroom_env/codecs.hpp:26-107, twin include/ds2i/block_codecs.hpp:87-101):
// 7 payload bits per byte, least-significant group first, and the LAST byte of
// a value has bit 7 set (the opposite of LEB128). A list header is vbyte(n)
// followed by vbyte(universe) (vroom_env/codecs.hpp:110-124).
#pragma once
#include <cstdint>
#include <vector>

namespace dint {

struct vbyte {
    static void append(uint32_t val, std::vector<uint8_t>& out) {
        while (val >= 128) {
            out.push_back(uint8_t(val & 127));
            val >>= 7;
        }
        out.push_back(uint8_t(val | 128));
    }

    // Reads one value; returns the pointer past it. `end` bounds the read; on a
    // truncated value returns nullptr.
    static uint8_t const* read(uint8_t const* in, uint8_t const* end, uint32_t* val) {
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
};

struct list_header {
    static void write(uint32_t n, uint32_t universe, std::vector<uint8_t>& out) {
        vbyte::append(n, out);
        vbyte::append(universe, out);
    }
    static uint8_t const* read(uint8_t const* in, uint8_t const* end, uint32_t* n,
                               uint32_t* universe) {
        in = vbyte::read(in, end, n);
        if (!in) return nullptr;
        return vbyte::read(in, end, universe);
    }
};

}  // namespace dint
