#!/usr/bin/env python3
"""Generates tests/golden/kat_vectors.json: hand-assembled known-answer vectors.

The reference (jermp/dint) cannot be built in this image and ships no DINT test
vectors, so these are NOT reference outputs. They are built from the byte
formats alone (include/dint/{single,rectangular,multi}_
dictionary.hpp `write`, vroom_env/dint_codecs.hpp decode loops): a tiny
dictionary whose entries are listed below, codeword streams assembled slot by
slot, and the expected integers written out by substituting each codeword with
its definition — no decoder (ours or anyone's) is run to produce `expect`.

This script imports nothing from the repo. Run it to regenerate the JSON.
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

# ---- the hand-made dictionary ------------------------------------------------------------
# codeword -> integers. 0/1 are the exception markers, 2..6 the zero runs.
RUNS = {2: 256, 3: 128, 4: 64, 5: 32, 6: 16}
ENTRIES = {
    7: [1],
    8: [2, 3],
    9: [4, 5, 6, 7],
    10: list(range(8, 16)),
    11: list(range(16, 32)),
    12: [0],
    13: [65535, 70000],
    14: [9, 9, 9, 9],
    15: [0, 0],
    16: [1, 0, 1, 0, 1, 0, 1, 0],
    17: [4294967295],
    18: [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3],
}
FIRST, LAST = 7, 18


def u32s(vals):
    return struct.pack("<%dI" % len(vals), *vals)


def packed_offsets_and_table(entries_in_order):
    """single_dictionary layout: table starts with 16 zeros, offsets = (size-1)<<24 | offset."""
    table = [0] * 16
    offsets = []
    for e in entries_in_order:
        offsets.append(((len(e) - 1) << 24) | len(table))
        table += e
    return offsets, table


def single_packed_file():
    reserved = [0, 0] + [((RUNS[i] - 1) << 24) for i in range(2, 7)]
    offs, table = packed_offsets_and_table([ENTRIES[i] for i in range(FIRST, LAST + 1)])
    offsets = reserved + offs
    m_size = len(offsets)
    return u32s([m_size, len(offsets), len(table)]) + u32s(offsets) + u32s(table)


def rectangular_file():
    rows = []
    for i in range(0, LAST + 1):
        if i < 2:
            row, size = [0] * 16, 1
        elif i < 7:
            row, size = [0] * 16, RUNS[i]
        else:
            e = ENTRIES[i]
            row, size = e + [0] * (16 - len(e)), len(e)
        rows += row + [size]
    return u32s([LAST + 1]) + u32s(rows)


# multi: dictionary d holds the same entries shifted by a per-dictionary constant so that
# a wrong dictionary choice shows. Entry i of dictionary d = ENTRIES[i] with +d on every value.
def multi_entry(d, i):
    return [(v + d) & 0xFFFFFFFF for v in ENTRIES[i]]


def multi_packed_file():
    table = [0] * 16
    start_offsets, offsets = [], []
    total = 7
    for d in range(6):
        start_offsets.append(len(offsets))
        offsets += [0, 0] + [((RUNS[i] - 1) << 24) for i in range(2, 7)]
        for i in range(FIRST, LAST + 1):
            e = multi_entry(d, i)
            offsets.append(((len(e) - 1) << 24) | len(table))
            table += e
            total += 1
    return (u32s([total, len(start_offsets), len(offsets), len(table)]) + u32s(start_offsets) + u32s(offsets)
            + u32s(table))


# ---- stream assembly ----------------------------------------------------------------------
def slot16(v):
    return struct.pack("<H", v)


class Stream:
    """A codeword stream under construction together with what it must decode to."""

    def __init__(self, width=16, dict_id=0):
        self.bytes = b""
        self.expect = []
        self.width = width
        self.dict_id = dict_id

    def code(self, i):
        self.bytes += slot16(i) if self.width == 16 else bytes([i])
        if i in RUNS:
            self.expect += [0] * RUNS[i]
        else:
            self.expect += multi_entry(self.dict_id, i) if self.dict_id else list(ENTRIES[i])
        return self

    def exc(self, value):
        marker = 1 if value > 65535 else 0
        self.bytes += slot16(marker) if self.width == 16 else bytes([marker])
        self.bytes += struct.pack("<I", value) if marker else struct.pack("<H", value)
        self.expect.append(value)
        return self

    def exc32(self, value):  # force the 4-byte form even for a small value
        self.bytes += slot16(1) if self.width == 16 else bytes([1])
        self.bytes += struct.pack("<I", value)
        self.expect.append(value)
        return self


def single_cases():
    cases = []

    def add(name, st, prefix=b""):
        cases.append({"name": name, "prefix": prefix.hex(), "stream": st.bytes.hex(), "n": len(st.expect),
                      "expect": st.expect})

    add("one_codeword_size1", Stream().code(7))
    add("one_exception16", Stream().exc(12345))
    add("every_size_class", Stream().code(7).code(8).code(9).code(10).code(11))
    add("n15", Stream().code(10).code(9).code(8).code(7))
    add("n16_one_entry", Stream().code(11))
    add("n17", Stream().code(11).code(7))
    add("exc16_values", Stream().exc(0).exc(1).exc(2).exc(65535))
    add("exc32_values", Stream().exc(65536).exc(0x00010000 + 1).exc(1 << 16).exc(4294967295).exc32(0).exc32(1))
    add("exception_last", Stream().code(9).code(8).exc(77777))
    add("exception_first", Stream().exc(5).code(9))
    add("entry16_ends_at_n", Stream().code(7).code(18))
    add("runs_each", Stream().code(6).code(5).code(4).code(3).code(2))
    add("zeros_300", Stream().code(2).code(5).code(15).code(15).code(15).code(15).code(15).code(15))
    add("run_between_data", Stream().code(9).code(6).code(13).code(2).code(17))
    add("payload_halves_look_like_markers", Stream().exc32(0x00010000).exc32(0x00000001).exc32(0x00010001).code(7))
    add("odd_address", Stream().code(9).exc(300).code(10).exc(99999).code(6), prefix=b"\xAA")
    add("odd_address_3", Stream().code(8).code(18).exc(65536), prefix=b"\xAA\xBB\xCC")
    # 255 one-integer codewords, then a 32-bit exception whose header is slot 255 (the last of a
    # 256-slot tile) and whose payload opens the next tile; then data
    st = Stream()
    for _ in range(255):
        st.code(7)
    st.exc(4000000000)
    for _ in range(40):
        st.code(9)
    add("exception_straddles_tile", st)
    # exception headers on every lane's last slot (slot 4l+3) for a stretch, 16- and 32-bit mixed
    st = Stream()
    for l in range(70):
        st.code(7).code(8).code(7)
        if l % 2:
            st.exc(70000 + l)
        else:
            st.exc(l)
    add("exceptions_on_lane_boundaries", st)
    # dense exceptions: every value an exception (worst case for the slot classifier)
    st = Stream()
    for i in range(700):
        st.exc((i * 2654435761) & 0xFFFFFFFF if i % 3 else i & 1)
    add("all_exceptions", st)
    # long mixed stream across several tiles and batches (runs make > 2048 outputs per tile)
    st = Stream()
    for i in range(300):
        st.code(2 + i % 5)
        st.code(7 + i % 12)
    add("many_runs", st)
    st = Stream()
    for i in range(2000):
        st.code(7 + (i * 7) % 12)
        if i % 97 == 0:
            st.exc(i * 1000)
    add("long_mixed", st)
    return cases


def multi_cases():
    """multi_packed: n integers in blocks of 256 (tail n % 256); each block = selector byte + slots."""
    cases = []

    def block(selector, build):
        narrow = selector >= 6
        st = Stream(width=8 if narrow else 16, dict_id=selector - 6 if narrow else selector)
        build(st)
        return bytes([selector]) + st.bytes, st.expect

    def add(name, blocks):
        data, expect = b"", []
        for i, (b, e) in enumerate(blocks):
            assert len(e) == 256 or i == len(blocks) - 1, (name, len(e))
            data += b
            expect += e
        cases.append({"name": name, "prefix": "", "stream": data.hex(), "n": len(expect), "expect": expect})

    def fill256(st):
        # 256 integers: 16 + 16 + 64 (run) + ... mix of everything
        st.code(11).code(18).code(4)                    # 16 + 16 + 64 = 96
        for _ in range(10):
            st.code(10)                                 # + 80 = 176
        for _ in range(10):
            st.code(9)                                  # + 40 = 216
        st.exc(300).exc(70000)                          # + 2 = 218
        for _ in range(19):
            st.code(8)                                  # + 38 = 256

    for sel in range(12):
        add("one_block_selector_%d" % sel, [block(sel, fill256)])
    add("tail_only", [block(3, lambda st: st.code(9).exc(65535).code(7))])
    add("tail_only_narrow", [block(9, lambda st: st.code(9).exc(65536).exc(2).code(7))])
    add("blocks_all_selectors_then_tail",
        [block(sel, fill256) for sel in range(12)] + [block(7, lambda st: st.code(18).code(6).exc(1))])
    add("narrow_exceptions", [block(6, lambda st: [st.exc(v) for v in (0, 1, 255, 256, 65535, 65536, 4294967295)])])
    return cases


def main():
    out = {
        "about": "hand-assembled known-answer vectors; see make_kats.py. NOT reference outputs.",
        "dict_entries": {str(k): v for k, v in ENTRIES.items()},
        "single_packed_dict": single_packed_file().hex(),
        "rectangular_dict": rectangular_file().hex(),
        "multi_packed_dict": multi_packed_file().hex(),
        "single_cases": single_cases(),
        "multi_cases": multi_cases(),
    }
    path = os.path.join(HERE, "kat_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["single_cases"]), "single cases,",
          len(out["multi_cases"]), "multi cases")


if __name__ == "__main__":
    main()
