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
    assert device.abi_version() == 5


def test_options_are_an_api_not_the_environment():
    """dint_set_option / dint_get_option: process-wide switches with defaults and ranges; no GPU needed. The library
    reads no environment variable (the strings of its binary hold no getenv target of ours)."""
    from dint_amd import device

    assert set(device.OPTIONS) == {"bundles", "index_concurrent", "query_lean_pages", "query_tail_pages", "query_fused_pages",
                                   "index_inline_tails", "chunk_split", "index_pair", "query_fused_copy", "query_batch_fused"}
    device.reset_options()
    defaults = {k: device.get_option(k) for k in device.OPTIONS}
    assert defaults == {"bundles": 1, "index_concurrent": 1, "query_lean_pages": -1, "query_tail_pages": 4,
                        "query_fused_pages": 2, "index_inline_tails": 1, "chunk_split": -1, "index_pair": 1, "query_fused_copy": 1, "query_batch_fused": 1}
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
