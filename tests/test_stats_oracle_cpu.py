"""The oracle's restatement of dictionary construction (oracle/dint_oracle_stats.c: selector::get, adjusted::collect, the
saving filter, freq_length_sorter, the DSF cut — statistics_collectors.hpp:21-118, block_statistics.hpp:82-106,
dictionary_builders.hpp:15-75) against hand-computed cases, the reference-pinned hash vectors, and the host library's
construction; and dint/constants.hpp against the reference's own configuration header."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from dint_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_selector_by_hand():
    """code = ceil_log2(ceil_log2(max + 1)), 0 for max <= 1 (statistics_collectors.hpp:21-40, util.hpp:67-70):
    max 2..3 -> bits 2 -> 1; 4..15 -> bits 3..4 -> 2; 16..255 -> 5..8 -> 3; 256..65535 -> 9..16 -> 4; beyond -> 5 —
    up to 2^32 - 2: the reference adds the 1 in uint32_t (:23, :36), so at 2^32 - 1 the sum wraps to 0, ceil_log2(0) is 0 in
    a release build and the block is context 0. KAT at 0xFFFFFFFF: oracle and product both restate the wrap."""
    want = {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 15: 2, 16: 3, 255: 3, 256: 4, 65535: 4, 65536: 5, 2**32 - 2: 5, 2**32 - 1: 0}
    for x, code in want.items():
        block = np.zeros(256, dtype=np.uint32)
        block[37] = x
        assert oracle.selector_get(block) == code, x
    assert oracle.selector_get(np.array([7, 300, 2], dtype=np.uint32)) == 4
    # the product's own selector is the same function
    assert all(host.block_selector(np.array([x], dtype=np.uint32)) == c for x, c in want.items())


def test_hash_is_the_references():
    with open(os.path.join(ROOT, "tests", "golden", "murmur_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    for v in vectors:
        assert "%016x" % oracle.hash_u32s(np.array(v["words"], dtype=np.uint32)) == v["hash"]
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_hash.so")
    if os.path.exists(ref):
        lib = C.CDLL(ref)
        lib.ref_hash_u32s.restype = C.c_uint64
        lib.ref_hash_u32s.argtypes = [C.c_void_p, C.c_ulong]
        r = np.random.default_rng(5)
        for n in (1, 2, 4, 8, 16):
            w = r.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            assert oracle.hash_u32s(w) == lib.ref_hash_u32s(w.ctypes.data, n)


def test_counts_of_a_tiny_collection_by_hand():
    """two lists, 8 + 1 integers: the aligned 8-, 4-, 2-, 1-grams of each list on its own; no 16-gram fits; a new
    n-gram starts at frequency 1 (block_type(), statistics_collectors.hpp:9)."""
    gaps = np.array([5, 5, 5, 5, 1, 2, 1, 2, 9], dtype=np.uint32)
    st = oracle.Stats(False, gaps)
    st.collect_lists([8, 1])
    got = {st.ngram(e): int(e["freq"]) for e in st.entries(0)}
    want = {(5, 5, 5, 5, 1, 2, 1, 2): 1, (5, 5, 5, 5): 1, (1, 2, 1, 2): 1, (5, 5): 2, (1, 2): 2, (5,): 4, (1,): 2, (2,): 2, (9,): 1}
    assert got == want and st.total_integers == 9
    multi = oracle.Stats(True, gaps)
    multi.collect_lists([8, 1])
    assert all(multi.entries(c).size == 0 for c in range(6)) and multi.total_integers == 9  # no whole 256-block


def test_multi_collect_by_hand():
    """one list of 600 integers: two whole blocks counted (the 88-integer rest is not), each in the map of its own
    selector; inside a block every aligned 16/8/4/2/1-gram."""
    gaps = np.zeros(600, dtype=np.uint32)
    gaps[:256] = 1          # block 0: max 1 -> context 0
    gaps[256:512] = 3       # block 1: max 3 -> context 1
    gaps[300] = 2
    gaps[512:] = 1000       # the tail: never counted
    st = oracle.Stats(True, gaps)
    st.collect(0, 600)
    c0 = {st.ngram(e): int(e["freq"]) for e in st.entries(0)}
    assert c0 == {(1,) * 16: 16, (1,) * 8: 32, (1,) * 4: 64, (1,) * 2: 128, (1,): 256}
    c1 = {st.ngram(e): int(e["freq"]) for e in st.entries(1)}
    assert c1[(3,)] == 255 and c1[(2,)] == 1 and c1[(3, 3)] == 127 and c1[(2, 3)] == 1 and c1[(3,) * 16] == 15
    assert sum(c1.values()) == 16 + 32 + 64 + 128 + 256
    assert all(st.entries(c).size == 0 for c in (2, 3, 4, 5))


def _filter_passes(freq, length, total):
    saving = float(np.uint32(freq)) * (48.0 * length - 16.0) / total  # dictionary_builders.hpp:15-28
    return saving > 0.0001 / 1000 or length == 1


def test_selection_order_and_filter_by_hand():
    r = np.random.default_rng(3)
    gaps = r.integers(0, 4, 4096, dtype=np.uint64).astype(np.uint32)
    st = oracle.Stats(False, gaps)
    st.collect(0, gaps.size)
    picked, passed = st.select(0)
    ents = st.entries(0)
    assert passed == sum(_filter_passes(int(e["freq"]), int(e["len"]), gaps.size) for e in ents)
    keys = [(-int(e["freq"]), -int(e["len"]), st.ngram(e)) for e in picked]
    assert keys == sorted(keys) and len(picked) == min(passed, 65536)


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED])
@pytest.mark.parametrize("corpus_name", ["small_corpus", "sparse_corpus"])
def test_host_construction_appends_what_the_oracle_selects(request, kind, corpus_name):
    """the host library's dictionary == the oracle's selection (counts, filter, order, cut) packed by the host's
    packer: the file is byte-identical, so counts and selection agree entry for entry."""
    coll = request.getfixturevalue(corpus_name).coll
    multi = kind == host.MULTI_PACKED
    st = oracle.Stats(multi, coll.gaps)
    st.collect_lists(coll.lens)
    chosen = []
    for c in range(st.contexts):
        picked, _ = st.select(c)
        part = np.zeros(len(picked), dtype=host.NGRAM_DTYPE)
        part["pos"], part["freq"], part["len"], part["ctx"] = picked["pos"], picked["freq"], picked["len"], c
        chosen.append(part)
    chosen = np.concatenate(chosen)
    assert host.pack_dictionary(kind, np.ascontiguousarray(coll.gaps, dtype=np.uint32), chosen) == host.build_dictionary(kind, coll)


def test_constants_are_the_references():
    """tests/golden/ref_constants.json holds the values of the reference's dint_configuration.hpp as g++ compiled them
    (tests/golden/make_ref_constants.py); dint/constants.hpp is static_asserted against the header itself when the
    reference tree is present (oracle/ref_constants_check.cpp, `make -C oracle ref`)."""
    with open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")) as f:
        ref = json.load(f)["constants"]
    k = host.constants()
    assert k["exceptions"] == ref["EXCEPTIONS"] and k["num_selectors"] == ref["num_selectors"]
    assert k["max_entry_size"] == ref["max_entry_size"] and k["num_entries"] == ref["num_entries"] == 1 << ref["log2_num_entries"]
    assert k["target_sizes"] == [ref["target_sizes[%d]" % i] for i in range(ref["num_target_sizes"])]
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libref_constants.so")
    if os.path.exists(lib_path):
        assert C.CDLL(lib_path).ref_constants_check() == 0
