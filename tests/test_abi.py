"""The C-ABI libraries load and export every symbol the public headers declare (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dinth?_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("dint_hip.h", "libdint_hip.so"), ("dint_host.h", "libdint_host.so")])
def test_every_declared_symbol_is_exported(header, lib):
    handle = C.CDLL(os.path.join(ROOT, "dint_amd", lib))
    names = declared_functions(header)
    assert len(names) >= 10
    for name in names:
        assert hasattr(handle, name), f"{lib} does not export {name}"


def test_python_binding_lists_the_same_symbols():
    from dint_amd import device

    assert sorted(device.ABI_SYMBOLS) == declared_functions("dint_hip.h")
    assert device.abi_version() == 6


def test_options_are_an_api_not_the_environment():
    """dint_set_option / dint_get_option: process-wide switches with defaults and ranges; no GPU needed. The library
    reads no environment variable (the strings of its binary hold no getenv target of ours)."""
    from dint_amd import device

    assert set(device.OPTIONS) == {"bundles", "index_concurrent", "query_lean_pages", "query_tail_pages", "query_fused_pages",
                                   "index_inline_tails", "chunk_split", "index_pair", "query_fused_copy", "query_batch_fused", "split_units", "refine_units"}
    device.reset_options()
    defaults = {k: device.get_option(k) for k in device.OPTIONS}
    assert defaults == {"bundles": 1, "index_concurrent": 1, "query_lean_pages": -1, "query_tail_pages": 4,
                        "query_fused_pages": 2, "index_inline_tails": 1, "chunk_split": -1, "index_pair": 1, "query_fused_copy": 1, "query_batch_fused": 1,
                        "split_units": 1, "refine_units": 1}
    with device.options(query_fused_pages=0, query_tail_pages=16):
        assert device.get_option("query_fused_pages") == 0 and device.get_option("query_tail_pages") == 16
    assert {k: device.get_option(k) for k in device.OPTIONS} == defaults
    lib = device._lib
    assert lib.dint_set_option(device.OPTIONS["bundles"], C.c_longlong(-1)) == -1
    assert lib.dint_set_option(99, C.c_longlong(0)) == -1 and lib.dint_get_option(0, None) == -1
    blob = open(os.path.join(ROOT, "dint_amd", "libdint_hip.so"), "rb").read()
    assert b"DINT_QUERY_" not in blob and b"DINT_NO_BUNDLES" not in blob and b"DINT_INDEX_CONCURRENT" not in blob


def test_errors_are_status_codes_not_exceptions():
    from dint_amd import device
    from kat import DICT_FILES

    lib = device._lib
    assert lib.dint_strerror(0) == b"ok"
    assert lib.dint_dict_create(1, None, 0, 0, None) == -1                    # DINT_ERR_ARG
    h = C.c_void_p()
    junk = (C.c_char * 8)(*b"\x01\x02\x03\x04\x05\x06\x07\x08")
    assert lib.dint_dict_create(1, junk, 8, 0, C.byref(h)) == -2              # DINT_ERR_FORMAT
    assert lib.dint_dict_create(7, junk, 8, 0, C.byref(h)) == -1              # bad kind
    if device.device_count() == 0:
        buf = (C.c_char * len(DICT_FILES[1])).from_buffer_copy(DICT_FILES[1])
        # a well-formed dictionary but no GPU in this process: the decode path has no CPU fallback
        assert lib.dint_dict_create(1, buf, len(DICT_FILES[1]), 0, C.byref(h)) == -4   # DINT_ERR_NO_DEVICE
        with pytest.raises(device.DintError):
            device.Dictionary(1, DICT_FILES[1])


def test_unit_struct_layout_matches_the_header():
    from dint_amd import host

    assert host.UNIT_DTYPE.itemsize == 24
    assert [host.UNIT_DTYPE.fields[n][1] for n in ("in_off", "out_off", "n", "list")] == [0, 8, 16, 20]


def test_dictionary_files_no_builder_writes_are_rejected():
    """Outside the decoders' domain (tests/fuzz_streams.py): an entry of more than 16 integers, reserved codewords that are not
    builder::init()'s, a table without its 16 leading zeros, a context without its reserved codewords. The reference decodes
    such a file to whatever its 64-byte copy finds behind the tables or the previous list left in the buffer
    (single_dictionary.hpp:230-238, rectangular_dictionary.hpp:206-213); here: DINT_ERR_FORMAT, before any device is touched."""
    import struct

    import numpy as np

    import fuzz_streams as F
    from dint_amd import device

    lib = device._lib
    no_gpu = device.device_count() == 0

    def status(kind, file):
        h = C.c_void_p()
        buf = (C.c_char * len(file)).from_buffer_copy(file)
        st = lib.dint_dict_create(kind, buf, len(file), 0, C.byref(h))
        if st == 0:
            lib.dint_dict_destroy(h)
        return st

    ok = -4 if no_gpu else 0  # a well-formed file gets as far as the device
    r = np.random.default_rng(1)
    good = {k: F.make_dictionary(r, k, 40, size_profile="any").file for k in (F.RECT, F.SINGLE, F.MULTI)}
    for k, f in good.items():
        assert status(k, f) == ok
    # single packed: header 3 words, then offsets
    f = bytearray(good[F.SINGLE])
    bad = bytearray(f); struct.pack_into("<I", bad, 12 + 4 * 20, (16 << 24) | 16); assert status(F.SINGLE, bytes(bad)) == -2  # 17 integers
    bad = bytearray(f); struct.pack_into("<I", bad, 12 + 4 * 3, (127 << 24) | 4); assert status(F.SINGLE, bytes(bad)) == -2   # a run off offset 0
    bad = bytearray(f); struct.pack_into("<I", bad, 12 + 4 * 0, 1 << 24); assert status(F.SINGLE, bytes(bad)) == -2           # a marker of 2 integers
    bad = bytearray(f); struct.pack_into("<I", bad, 12 + 4 * 40 + 4 * 5, 9); assert status(F.SINGLE, bytes(bad)) == -2        # table[5] != 0
    bad = bytearray(f); struct.pack_into("<I", bad, 12 + 4 * 30, (3 << 24) | 0xFFFFF0); assert status(F.SINGLE, bytes(bad)) == -2  # past the table
    assert status(F.SINGLE, struct.pack("<3I", 3, 3, 16) + bytes(4 * 3 + 64)) == -2                                        # fewer than 7 codewords
    # rectangular: 1 word, then rows of 17
    f = good[F.RECT]
    bad = bytearray(f); struct.pack_into("<I", bad, 4 + 68 * 9 + 64, 17); assert status(F.RECT, bytes(bad)) == -2
    bad = bytearray(f); struct.pack_into("<I", bad, 4 + 68 * 9 + 64, 0); assert status(F.RECT, bytes(bad)) == -2
    bad = bytearray(f); struct.pack_into("<I", bad, 4 + 68 * 4 + 64, 32); assert status(F.RECT, bytes(bad)) == -2           # run 64 resized
    bad = bytearray(f); struct.pack_into("<I", bad, 4 + 68 * 6 + 8, 1); assert status(F.RECT, bytes(bad)) == -2             # a run row with a non-zero word
    assert status(F.RECT, struct.pack("<I", 0)) == ok                                                                     # no rows: init()'s alone
    # multi: a context without its reserved codewords
    f = bytearray(good[F.MULTI])
    n_off = struct.unpack_from("<I", f, 8)[0]
    bad = bytearray(f); struct.pack_into("<I", bad, 16 + 4 * 5, n_off - 3); assert status(F.MULTI, bytes(bad)) == -2
