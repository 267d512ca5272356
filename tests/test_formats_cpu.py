"""Byte formats of the host side: vbyte header, murmur key function, dictionary files."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import oracle
from dint_amd import host
from kat import DICT_FILES, KAT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vbyte(v):
    out = []
    while v >= 128:
        out.append(v & 127)
        v >>= 7
    out.append(v | 128)
    return bytes(out)


@pytest.mark.parametrize("n,u", [(1, 0), (127, 128), (128, 16383), (16384, 2 ** 21), (2 ** 28 - 1, 2 ** 28),
                                 (50_000_000, 2 ** 32 - 1)])
def test_header_is_two_tight_vbytes(n, u):
    """vroom_env/codecs.hpp:26-107: 7 bits per byte, LSB group first, LAST byte has bit 7 set."""
    buf = np.frombuffer(_vbyte(n) + _vbyte(u) + b"\x00" * 4, dtype=np.uint8)
    assert oracle.header_read(buf, 0) == (n, u, len(_vbyte(n)) + len(_vbyte(u)))


def test_encoder_writes_the_same_header(small_corpus):
    enc, units = small_corpus.encoded(host.SINGLE_PACKED)
    bounds = small_corpus.coll.list_bounds()
    first = np.r_[True, units["list"][1:] != units["list"][:-1]]
    for u in units[first][:200]:
        i = int(u["list"])
        gaps = small_corpus.coll.gaps[int(bounds[i]):int(bounds[i + 1])]
        hdr = _vbyte(len(gaps)) + _vbyte(int(gaps.sum(dtype=np.uint64)) & 0xFFFFFFFF)
        start = int(u["in_off"]) - len(hdr)
        assert bytes(enc[start:int(u["in_off"])]) == hdr


def test_murmur_matches_golden_vectors():
    """tests/golden/murmur_vectors.json was produced by the REFERENCE's own hash_utils.hpp
    (compiled stand-alone into oracle/_ref, see tests/golden/make_murmur_vectors.py)."""
    import json

    with open(os.path.join(ROOT, "tests", "golden", "murmur_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    assert len(vectors) >= 64
    for v in vectors:
        assert host.hash_u32s(np.array(v["words"], dtype=np.uint32)) == int(v["hash"], 16)


def test_murmur_matches_reference_build_when_present():
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libref_hash.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    lib = C.CDLL(lib_path)
    lib.ref_hash_u32s.restype = C.c_uint64
    lib.ref_hash_u32s.argtypes = [C.c_void_p, C.c_ulong]
    r = np.random.default_rng(7)
    for n in list(range(1, 18)) + [32, 64, 256]:
        for _ in range(20):
            w = r.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
            assert host.hash_u32s(w) == lib.ref_hash_u32s(w.ctypes.data, n)


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED])
def test_dictionary_file_layout(small_corpus, kind):
    d = small_corpus.dict_file(kind)
    if kind == host.RECTANGULAR:
        (m_size,) = struct.unpack_from("<I", d, 0)
        assert len(d) == 4 + m_size * 17 * 4 and m_size <= 65536
        rows = np.frombuffer(d, dtype=np.uint32, offset=4).reshape(m_size, 17)
        assert rows[:2, 16].tolist() == [1, 1] and rows[2:7, 16].tolist() == [256, 128, 64, 32, 16]
        assert set(np.unique(rows[7:, 16]).tolist()) <= {1, 2, 4, 8, 16}
    elif kind == host.SINGLE_PACKED:
        m_size, n_off, n_tab = struct.unpack_from("<3I", d, 0)
        assert len(d) == 12 + 4 * (n_off + n_tab) and m_size == n_off <= 65536
        offs = np.frombuffer(d, dtype=np.uint32, count=n_off, offset=12)
        table = np.frombuffer(d, dtype=np.uint32, count=n_tab, offset=12 + 4 * n_off)
        assert not table[:16].any()                      # 16 leading zeros for the runs
        assert ((offs[2:7] >> 24) + 1).tolist() == [256, 128, 64, 32, 16] and not (offs[2:7] & 0xFFFFFF).any()
        sizes = (offs[7:] >> 24) + 1
        assert set(np.unique(sizes).tolist()) <= {1, 2, 4, 8, 16}
        assert ((offs[7:] & 0xFFFFFF) + sizes <= n_tab).all()
    else:
        m_size, n_start, n_off, n_tab = struct.unpack_from("<4I", d, 0)
        assert n_start == 6 and len(d) == 16 + 4 * (n_start + n_off + n_tab)
        start = np.frombuffer(d, dtype=np.uint32, count=6, offset=16)
        assert start[0] == 0 and (np.diff(start.astype(np.int64)) >= 7).all()


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED])
def test_packed_and_rect_hold_the_same_entries(small_corpus, kind):
    """Same statistics -> same entries in the same order, whatever the container."""
    n = host.dict_num_entries(kind, small_corpus.dict_file(kind))
    other = host.SINGLE_PACKED if kind == host.RECTANGULAR else host.RECTANGULAR
    assert n == host.dict_num_entries(other, small_corpus.dict_file(other))
    for i in list(range(7, 60)) + list(range(n - 40, n)):
        sa, wa = host.dict_entry(kind, small_corpus.dict_file(kind), i)
        sb, wb = host.dict_entry(other, small_corpus.dict_file(other), i)
        assert sa == sb and np.array_equal(wa[:sa], wb[:sb])


def test_kat_dictionaries_parse_with_the_host_loader():
    entries = {int(k): v for k, v in KAT["dict_entries"].items()}
    for kind in (0, 1, 2):
        for i, e in entries.items():
            size, words = host.dict_entry(kind, DICT_FILES[kind], i)
            assert size == len(e) and words[:size].tolist() == e


def _python_ngram_entries(coll, multi):
    """adjusted::collect (statistics_collectors.hpp:90-118) with a dict: aligned 16/8/4/2/1-grams of every list
    (multi: of every whole 256-block, under the block's context :21-40) -> NGRAM_DTYPE entries."""
    import math

    seen = {}
    bounds = coll.list_bounds()
    for i in range(len(coll.lens)):
        lo, n = int(bounds[i]), int(coll.lens[i])
        chunks = [(lo + at, min(256, n - at)) for at in range(0, n, 256)]
        for start, c in chunks:
            if multi and c != 256:
                continue
            block = coll.gaps[start:start + c]
            ctx = 0
            if multi:
                mx = int(block.max())
                ctx = 0 if mx <= 1 else math.ceil(math.log2(math.ceil(math.log2(mx + 1))))
            for ln in (16, 8, 4, 2, 1):
                for at in range(0, c - ln + 1, ln):
                    key = (ctx, block[at:at + ln].tobytes())
                    e = seen.get(key)
                    if e is None:
                        seen[key] = [start + at, 1, ln, ctx]
                    else:
                        e[1] += 1
    out = np.zeros(len(seen), dtype=host.NGRAM_DTYPE)
    for k, (pos, freq, ln, ctx) in enumerate(seen.values()):
        out[k] = (pos, freq, ln, ctx, 0)
    return out


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED])
def test_dictionary_from_ngram_counts_is_the_same_file(kind):
    """The selection half of the construction (dinth_build_dictionary_from_ngrams) over counts made by a plain
    Python counter == the whole construction (dinth_build_dictionary): what the device's counting kernel feeds."""
    coll = host.synth_collection(60_000, universe=50_000, seed=21)
    entries = _python_ngram_entries(coll, kind == host.MULTI_PACKED)
    got = host.build_dictionary_from_ngrams(kind, coll.gaps, coll.num_postings, entries)
    assert got == host.build_dictionary(kind, coll)
    with pytest.raises(Exception):
        bad = entries.copy()
        bad["pos"][0] = coll.num_postings  # points past the integers
        host.build_dictionary_from_ngrams(kind, coll.gaps, coll.num_postings, bad)


def _select_in_python(entries, gaps, total_ints, tie_key):
    """The selection (filter, order, first 65536 per context) with a caller-chosen order among n-grams of equal count and
    length — what dint_select_ngrams does on the device with the deterministic order."""
    keep = [e for e in entries if e["len"] == 1 or float(e["freq"]) * (48.0 * int(e["len"]) - 16.0) / total_ints > 1e-7]
    data = lambda e: tuple(int(x) for x in gaps[int(e["pos"]):int(e["pos"]) + int(e["len"])])
    keep.sort(key=lambda e: (int(e["ctx"]), -int(e["freq"]), -int(e["len"]), tie_key(data(e))))
    out, per_ctx = [], {}
    for e in keep:
        c = int(e["ctx"])
        per_ctx[c] = per_ctx.get(c, 0) + 1
        if per_ctx[c] <= 65536:
            out.append(e)
    return np.array(out, dtype=host.NGRAM_DTYPE)


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED])
def test_packing_selected_ngrams_is_the_same_file(kind):
    """dinth_pack_dictionary over the selection in dictionary order (the device's dint_select_ngrams; here in Python) ==
    the whole host construction, byte for byte."""
    coll = host.synth_collection(60_000, universe=50_000, seed=21)
    entries = _python_ngram_entries(coll, kind == host.MULTI_PACKED)
    chosen = _select_in_python(entries, coll.gaps, coll.num_postings, tie_key=lambda d: d)
    assert host.pack_dictionary(kind, coll.gaps, chosen) == host.build_dictionary(kind, coll)
    with pytest.raises(Exception):
        bad = chosen.copy()
        bad["len"][3] = 3   # not a power of two
        host.pack_dictionary(kind, coll.gaps, bad)


def test_tie_order_does_not_move_the_compression_ratio():
    """the reference's order among n-grams of equal count depends on libstdc++ (unordered_map iteration,
    std::sort), so dictionaries can only be compared by what they achieve: on a sample of the shape of SURVEY Appendix B
    (clustered gaps, a 5 M-document universe) dictionaries built with the ties in the deterministic order, in the
    reverse order and in a shuffled order compress the sample to within 1 % of each other."""
    coll = host.synth_collection(300_000, universe=5_000_000, seed=1459)
    entries = _python_ngram_entries(coll, False)
    r = np.random.default_rng(3)
    salt = {}

    def shuffled(d):
        return salt.setdefault(d, float(r.random()))

    bpi = {}
    for name, key in (("deterministic", lambda d: d), ("reversed", lambda d: tuple(-x for x in d)), ("shuffled", shuffled)):
        chosen = _select_in_python(entries, coll.gaps, coll.num_postings, tie_key=key)
        dict_file = host.pack_dictionary(host.SINGLE_PACKED, coll.gaps, chosen)
        enc, _ = host.encode_vroom(host.SINGLE_PACKED, dict_file, coll, unit_ints=8192)
        bpi[name] = enc.size * 8 / coll.num_postings
    assert bpi["deterministic"] == pytest.approx(
        host.encode_vroom(host.SINGLE_PACKED, host.build_dictionary(host.SINGLE_PACKED, coll), coll, unit_ints=8192)[0].size * 8 / coll.num_postings)
    for name in ("reversed", "shuffled"):
        assert abs(bpi[name] - bpi["deterministic"]) / bpi["deterministic"] < 0.01, bpi
