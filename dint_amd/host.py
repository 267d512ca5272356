"""ctypes binding of the offline CPU half (include/dint_host.h).

Synthetic collections, DSF dictionary construction and the vroom encoder: the
producers of the decode path's inputs. In the reference these are CPU C++ too
(vroom_env/encode.cpp, include/dint/dictionary_builders.hpp). Nothing in this
module decodes.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

RECTANGULAR, SINGLE_PACKED, MULTI_PACKED = 0, 1, 2
KIND_BY_TYPE = {
    # reference type strings, vroom_env/decode.cpp:236-244
    "single_rect_dint": RECTANGULAR,
    "single_packed_dint": SINGLE_PACKED,
    "multi_packed_dint": MULTI_PACKED,
}

UNIT_DTYPE = np.dtype(
    [("in_off", "<u8"), ("out_off", "<u8"), ("n", "<u4"), ("list", "<u4")], align=False
)
assert UNIT_DTYPE.itemsize == 24

_HERE = os.path.dirname(os.path.abspath(__file__))
# DINT_HOST_LIB: another build of the same library (the sanitizer build, `make -C dint_amd/csrc host-asan`; README "Sanitizers")
_LIB_PATH = os.environ.get("DINT_HOST_LIB") or os.path.join(_HERE, "libdint_host.so")


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("universe", C.c_uint32),
        ("min_len", C.c_uint32),
        ("max_len", C.c_uint32),
        ("reserved", C.c_uint32),
        ("alpha", C.c_double),
        ("stay_cluster", C.c_double),
        ("stay_sparse", C.c_double),
        ("p_cluster_min", C.c_double),
        ("p_cluster_max", C.c_double),
    ]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C dint_amd/csrc`) first"
        )
    lib = C.CDLL(_LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    lib.dinth_blob_data.restype = vp
    lib.dinth_blob_data.argtypes = [vp]
    lib.dinth_blob_size.restype = C.c_size_t
    lib.dinth_blob_size.argtypes = [vp]
    lib.dinth_blob_free.restype = None
    lib.dinth_blob_free.argtypes = [vp]
    lib.dinth_last_error.restype = C.c_char_p
    lib.dinth_synth_defaults.restype = None
    lib.dinth_synth_defaults.argtypes = [C.POINTER(SynthParams)]
    lib.dinth_synth_lengths.argtypes = [C.POINTER(SynthParams), u64, C.POINTER(vp)]
    lib.dinth_synth_gaps.argtypes = [C.POINTER(SynthParams), vp, u64, u64, vp, i32]
    lib.dinth_build_dictionary.argtypes = [i32, vp, vp, u64, u64, i32, C.POINTER(vp)]
    lib.dinth_build_dictionary_from_ngrams.argtypes = [i32, vp, u64, u64, vp, u64, C.POINTER(vp)]
    lib.dinth_pack_dictionary.argtypes = [i32, vp, u64, vp, u64, C.POINTER(vp)]
    lib.dinth_encode_vroom.argtypes = [i32, i32, vp, C.c_size_t, vp, vp, u64, u32, i32,
                                       C.POINTER(vp), C.POINTER(vp)]
    lib.dinth_build_index.argtypes = [i32, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, u64, i32,
                                      C.POINTER(vp), C.POINTER(vp)]
    lib.dinth_build_index_coder.argtypes = [i32, i32, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, u64, i32,
                                            C.POINTER(vp), C.POINTER(vp)]
    lib.dinth_encode_collection.argtypes = [i32, i32, vp, C.c_size_t, vp, C.c_size_t, i32, u32, i32, C.POINTER(vp), C.POINTER(vp),
                                            C.POINTER(u64), C.POINTER(u64)]
    lib.dinth_build_dictionary_collection.argtypes = [i32, vp, C.c_size_t, i32, u64, i32, C.POINTER(vp)]
    lib.dinth_build_index_collection.argtypes = [i32, i32, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, i32,
                                                 C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    lib.dinth_hash_u32s.restype = u64
    lib.dinth_hash_u32s.argtypes = [vp, C.c_size_t]
    lib.dinth_constants.argtypes = [vp, i32]
    lib.dinth_block_selector.restype = u32
    lib.dinth_block_selector.argtypes = [vp, C.c_size_t]
    lib.dinth_dict_entry.argtypes = [i32, vp, C.c_size_t, u32, u32, C.POINTER(u32), vp]
    lib.dinth_dict_num_entries.argtypes = [i32, vp, C.c_size_t, u32, C.POINTER(u32)]
    return lib


_lib = _load()


class HostError(RuntimeError):
    pass


def _check(status: int) -> None:
    if status != 0:
        raise HostError(f"dint host call failed ({status}): {_lib.dinth_last_error().decode()}")


def _take_blob(handle, dtype) -> np.ndarray:
    """Copy a dinth_blob into a numpy array and free the blob."""
    try:
        size = _lib.dinth_blob_size(handle)
        arr = np.empty(size // np.dtype(dtype).itemsize, dtype=dtype)
        if size:
            C.memmove(arr.ctypes.data, _lib.dinth_blob_data(handle), size)
        return arr
    finally:
        _lib.dinth_blob_free(handle)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def default_threads() -> int:
    return max(1, len(os.sched_getaffinity(0)))


def synth_params(**overrides) -> SynthParams:
    p = SynthParams()
    _lib.dinth_synth_defaults(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown synthetic parameter {k!r}")
        setattr(p, k, v)
    return p


def synth_lengths(params: SynthParams, target_postings: int) -> np.ndarray:
    h = C.c_void_p()
    _check(_lib.dinth_synth_lengths(C.byref(params), target_postings, C.byref(h)))
    return _take_blob(h, np.uint32)


def synth_gaps(params: SynthParams, lens: np.ndarray, first_list_id: int = 0,
               threads: int | None = None) -> np.ndarray:
    lens = _u32(lens)
    gaps = np.empty(int(lens.sum(dtype=np.uint64)), dtype=np.uint32)
    _check(_lib.dinth_synth_gaps(C.byref(params), lens.ctypes.data, len(lens), first_list_id,
                                 gaps.ctypes.data, threads or default_threads()))
    return gaps


@dataclass
class Collection:
    """Posting lists as d-gaps minus one (what the encoders consume), back to back."""
    gaps: np.ndarray  # u32
    lens: np.ndarray  # u32, one per list

    @property
    def num_postings(self) -> int:
        return int(self.gaps.size)

    def list_bounds(self) -> np.ndarray:
        b = np.zeros(len(self.lens) + 1, dtype=np.uint64)
        np.cumsum(self.lens, dtype=np.uint64, out=b[1:])
        return b


def synth_collection(target_postings: int, threads: int | None = None, **params) -> Collection:
    p = synth_params(**params)
    lens = synth_lengths(p, target_postings)
    return Collection(synth_gaps(p, lens, 0, threads), lens)


def readme_test_collection(seed: int = 1) -> Collection:
    """Stand-in for the reference's test collection (test/test_data/test_collection.docs is not in the checkout;
    README.md:53 gives its shape): 10 000 documents, 113 306 posting lists, 3 327 520 postings; list lengths
    Zipf-like (a few lists hold every other document, most hold a handful), every list a sorted set of distinct
    docIDs < 10 000, lists in no particular order. Seeded and exact: the same collection everywhere."""
    docs, n_lists, postings = 10_000, 113_306, 3_327_520
    rank = np.arange(1, n_lists + 1, dtype=np.float64)

    def lengths(c):
        return np.clip(np.rint(c / rank ** 0.93), 1, docs // 2).astype(np.int64)

    lo, hi = 1.0, 1e7
    for _ in range(80):  # the scale whose lengths sum to just under the target
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if lengths(mid).sum() <= postings else (lo, mid)
    lens = lengths(lo)
    short = postings - int(lens.sum())
    assert 0 <= short < n_lists
    lens[np.nonzero(lens < docs // 2)[0][:short]] += 1  # the remainder, one posting each, to the longest lists with room
    assert int(lens.sum()) == postings and int(lens.max()) >= 4096
    rng = np.random.default_rng(seed)
    lens = lens[rng.permutation(n_lists)]
    gaps = np.empty(postings, dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    for k in np.unique(lens):  # all the lists of one length at a time
        idx = np.nonzero(lens == k)[0]
        if k > 64:
            rows = np.stack([np.sort(rng.permutation(docs)[:k]) for _ in idx])
        else:
            rows = np.sort(rng.integers(0, docs, size=(idx.size, k)), axis=1)
            while True:  # redraw the rows that hold a docID twice
                dup = np.nonzero((rows[:, 1:] == rows[:, :-1]).any(axis=1))[0] if k > 1 else np.zeros(0, dtype=np.int64)
                if dup.size == 0:
                    break
                rows[dup] = np.sort(rng.integers(0, docs, size=(dup.size, k)), axis=1)
        g = rows.astype(np.int64)
        g[:, 1:] = g[:, 1:] - g[:, :-1] - 1
        pos = (starts[idx][:, None] + np.arange(k)[None, :]).ravel()
        gaps[pos] = g.ravel().astype(np.uint32)
    return Collection(gaps, lens.astype(np.uint32))


def docids_to_gaps(docids: np.ndarray) -> np.ndarray:
    """gap[i] = doc[i] - doc[i-1] - 1 with doc[-1] = -1 (vroom_env/jobs.hpp:74-84)."""
    d = np.asarray(docids, dtype=np.uint32)
    g = np.empty_like(d)
    if d.size:
        g[0] = d[0]
        g[1:] = d[1:] - d[:-1] - np.uint32(1)
    return g


def build_dictionary(kind: int, coll: Collection, max_sample_ints: int = 0,
                     threads: int | None = None) -> bytes:
    """DSF-65536-16 dictionary file image for `kind`, built from (a prefix sample of) coll."""
    h = C.c_void_p()
    gaps, lens = _u32(coll.gaps), _u32(coll.lens)
    _check(_lib.dinth_build_dictionary(kind, gaps.ctypes.data, lens.ctypes.data, len(lens),
                                       max_sample_ints, threads or default_threads(), C.byref(h)))
    return _take_blob(h, np.uint8).tobytes()


NGRAM_DTYPE = np.dtype([("pos", "<u8"), ("freq", "<u4"), ("len", "u1"), ("ctx", "u1"), ("pad", "<u2")])  # dinth_ngram


def build_dictionary_from_ngrams(kind: int, gaps: np.ndarray, total_ints: int, entries: np.ndarray) -> bytes:
    """The selection half of build_dictionary (filter, frequency sort, DSF, packing) over n-gram counts made
    elsewhere — device.count_ngrams — as NGRAM_DTYPE entries pointing into `gaps`."""
    h = C.c_void_p()
    gaps = _u32(gaps)
    entries = np.ascontiguousarray(entries, dtype=NGRAM_DTYPE)
    _check(_lib.dinth_build_dictionary_from_ngrams(kind, gaps.ctypes.data, gaps.size, total_ints, entries.ctypes.data,
                                                   entries.size, C.byref(h)))
    return _take_blob(h, np.uint8).tobytes()


def pack_dictionary(kind: int, gaps: np.ndarray, entries: np.ndarray) -> bytes:
    """Packing only: `entries` (NGRAM_DTYPE, pointing into `gaps`) are the dictionary's n-grams already selected and in
    dictionary order — device.select_ngrams — appended as they come."""
    h = C.c_void_p()
    gaps = _u32(gaps)
    entries = np.ascontiguousarray(entries, dtype=NGRAM_DTYPE)
    _check(_lib.dinth_pack_dictionary(kind, gaps.ctypes.data, gaps.size, entries.ctypes.data, entries.size, C.byref(h)))
    return _take_blob(h, np.uint8).tobytes()


def encode_vroom(kind: int, dict_file: bytes, coll: Collection, unit_ints: int = 4096,
                 greedy: bool = False, threads: int | None = None):
    """-> (encoded stream u8[], unit table UNIT_DTYPE[])"""
    enc, units = C.c_void_p(), C.c_void_p()
    gaps, lens = _u32(coll.gaps), _u32(coll.lens)
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    _check(_lib.dinth_encode_vroom(kind, int(greedy), C.addressof(buf), len(dict_file),
                                   gaps.ctypes.data, lens.ctypes.data, len(lens), unit_ints,
                                   threads or default_threads(), C.byref(enc), C.byref(units)))
    return _take_blob(enc, np.uint8), _take_blob(units, UNIT_DTYPE)


def gaps_to_docids(coll: Collection) -> np.ndarray:
    """Invert docids_to_gaps per list: docid[i] = sum(gap[0..i]) + i."""
    g = coll.gaps.astype(np.uint64)
    csum = np.cumsum(g + 1) - 1
    bounds = coll.list_bounds()
    starts = bounds[:-1][coll.lens > 0].astype(np.int64)
    base = np.zeros_like(csum)
    prev = np.r_[0, csum[starts[1:] - 1] + 1] if len(starts) else np.zeros(0, dtype=np.uint64)
    base[starts] = prev
    base = np.maximum.accumulate(base)
    return (csum - base).astype(np.uint32)


def synth_freqs(n: int, seed: int = 1) -> np.ndarray:
    """Term frequencies >= 1, mostly 1-3 with a geometric tail (synthetic)."""
    r = np.random.default_rng(seed)
    return r.geometric(0.55, n).astype(np.uint32)


def build_index(kind: int, docs_dict: bytes, freqs_dict: bytes, docids: np.ndarray, freqs: np.ndarray,
                lens: np.ndarray, threads: int | None = None, greedy: bool = False):
    """In-index layout (dict_posting_list per list). -> (index bytes u8[], list offsets u64[n_lists + 1]).
    greedy: greedy_dint_single_dict_block instead of the optimal parse (single-dictionary kinds)."""
    idx, offs = C.c_void_p(), C.c_void_p()
    d, f, l = _u32(docids), _u32(freqs), _u32(lens)
    db = (C.c_char * len(docs_dict)).from_buffer_copy(docs_dict)
    fb = (C.c_char * len(freqs_dict)).from_buffer_copy(freqs_dict)
    _check(_lib.dinth_build_index_coder(kind, int(greedy), C.addressof(db), len(docs_dict), C.addressof(fb), len(freqs_dict),
                                        d.ctypes.data, f.ctypes.data, l.ctypes.data, len(l),
                                        threads or default_threads(), C.byref(idx), C.byref(offs)))
    return _take_blob(idx, np.uint8), _take_blob(offs, np.uint64)


# ---- ds2i collection files (reference include/ds2i/binary_collection.hpp; README.md:43-51) ----

def collection_words(lists, num_docs: int | None = None) -> np.ndarray:
    """The u32 words of a collection file: records `len, v[len]`; num_docs given -> a .docs file (record 0 = `1, num_docs`)."""
    parts = [] if num_docs is None else [np.array([1, num_docs], dtype=np.uint32)]
    for v in lists:
        v = _u32(v)
        parts.append(np.array([v.size], dtype=np.uint32))
        parts.append(v)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)


def write_collection(basename: str, docid_lists, freq_lists, num_docs: int) -> None:
    """<basename>.docs and <basename>.freqs as the reference's tools read them."""
    collection_words(docid_lists, num_docs).tofile(basename + ".docs")
    collection_words(freq_lists).tofile(basename + ".freqs")


def encode_collection(kind: int, dict_file: bytes, words: np.ndarray, docs: bool, unit_ints: int = 4096, greedy: bool = False,
                      threads: int | None = None):
    """The vroom `encode` program over a collection file's words -> (stream u8[], units, lists, integers)."""
    enc, units = C.c_void_p(), C.c_void_p()
    w = _u32(words)
    n_lists, n_ints = C.c_uint64(), C.c_uint64()
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    _check(_lib.dinth_encode_collection(kind, int(greedy), C.addressof(buf), len(dict_file), w.ctypes.data, w.size, int(docs),
                                        unit_ints, threads or default_threads(), C.byref(enc), C.byref(units),
                                        C.byref(n_lists), C.byref(n_ints)))
    return _take_blob(enc, np.uint8), _take_blob(units, UNIT_DTYPE), n_lists.value, n_ints.value


def build_dictionary_collection(kind: int, words: np.ndarray, docs: bool, max_sample_ints: int = 0,
                                threads: int | None = None) -> bytes:
    """DSF-65536-16 over the statistics of a collection file's lists -> the dictionary file image."""
    h = C.c_void_p()
    w = _u32(words)
    _check(_lib.dinth_build_dictionary_collection(kind, w.ctypes.data, w.size, int(docs), max_sample_ints,
                                                  threads or default_threads(), C.byref(h)))
    return _take_blob(h, np.uint8).tobytes()


def build_index_collection(kind: int, docs_dict: bytes, freqs_dict: bytes, docs_words: np.ndarray, freqs_words: np.ndarray,
                           greedy: bool = False, threads: int | None = None):
    """dict_freq_index::builder over a .docs / .freqs pair -> (index bytes, list offsets, num_docs)."""
    idx, offs = C.c_void_p(), C.c_void_p()
    d, f = _u32(docs_words), _u32(freqs_words)
    num_docs = C.c_uint64()
    db = (C.c_char * len(docs_dict)).from_buffer_copy(docs_dict)
    fb = (C.c_char * len(freqs_dict)).from_buffer_copy(freqs_dict)
    _check(_lib.dinth_build_index_collection(kind, int(greedy), C.addressof(db), len(docs_dict), C.addressof(fb), len(freqs_dict),
                                             d.ctypes.data, d.size, f.ctypes.data, f.size, threads or default_threads(),
                                             C.byref(idx), C.byref(offs), C.byref(num_docs)))
    return _take_blob(idx, np.uint8), _take_blob(offs, np.uint64), num_docs.value


INDEX_FILE_HEADER = np.dtype([("magic", "S8"), ("kind", "<u4"), ("coder", "<u4"), ("num_docs", "<u8"), ("n_lists", "<u8"),
                              ("docs_dict_bytes", "<u8"), ("freqs_dict_bytes", "<u8"), ("index_bytes", "<u8")])


def read_index_file(path: str) -> dict:
    """The container `dint_create_freq_index` writes (dint/index_file.hpp)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < INDEX_FILE_HEADER.itemsize:
        raise ValueError("index file truncated")
    h = raw[:INDEX_FILE_HEADER.itemsize].view(INDEX_FILE_HEADER)[0]
    if h["magic"] != b"DINTIDX1":
        raise ValueError("not a DINT index file")
    pad8 = lambda n: (int(n) + 7) & ~7
    p = INDEX_FILE_HEADER.itemsize
    n = int(h["n_lists"])
    sizes = [8 * (n + 1), pad8(h["docs_dict_bytes"]), pad8(h["freqs_dict_bytes"]), int(h["index_bytes"])]
    if sum(sizes) > raw.size - p:  # (Python integers: no wrap)
        raise ValueError("index file truncated")
    offsets = raw[p:p + 8 * (n + 1)].view("<u8").copy()
    p += sizes[0]
    docs_dict = raw[p:p + int(h["docs_dict_bytes"])].tobytes()
    p += sizes[1]
    freqs_dict = raw[p:p + int(h["freqs_dict_bytes"])].tobytes()
    p += sizes[2]
    index = raw[p:p + int(h["index_bytes"])].copy()
    if int(offsets[-1]) != int(h["index_bytes"]) or (np.diff(offsets.astype(np.int64)) < 0).any():
        raise ValueError("index file: list offsets decrease or do not end at the index's size")
    return {"kind": int(h["kind"]), "coder": int(h["coder"]), "num_docs": int(h["num_docs"]), "offsets": offsets,
            "docs_dict": docs_dict, "freqs_dict": freqs_dict, "index": index}


def hash_u32s(words) -> int:
    w = _u32(words)
    return int(_lib.dinth_hash_u32s(w.ctypes.data, w.size))


def constants() -> dict:
    """The compile-time constants of this build (dint/constants.hpp; reference dint_configuration.hpp:6,20,24-28)."""
    v = np.zeros(16, dtype=np.uint32)
    n = _lib.dinth_constants(v.ctypes.data, v.size)
    assert n == 12
    return {"exceptions": int(v[0]), "num_selectors": int(v[1]), "max_entry_size": int(v[2]), "num_entries": int(v[3]),
            "target_sizes": [int(x) for x in v[5:5 + int(v[4])]], "block_size": int(v[10]), "reserved": int(v[11])}


def block_selector(values) -> int:
    """The context of a block of gaps (selector::get, reference statistics_collectors.hpp:21-40)."""
    w = _u32(values)
    return int(_lib.dinth_block_selector(w.ctypes.data, w.size))


def dict_num_entries(kind: int, dict_file: bytes, d: int = 0) -> int:
    n = C.c_uint32()
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    _check(_lib.dinth_dict_num_entries(kind, C.addressof(buf), len(dict_file), d, C.byref(n)))
    return n.value


def dict_entry(kind: int, dict_file: bytes, index: int, d: int = 0):
    """-> (size, first 16 payload words)"""
    size = C.c_uint32()
    words = np.zeros(16, dtype=np.uint32)
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    _check(_lib.dinth_dict_entry(kind, C.addressof(buf), len(dict_file), d, index, C.byref(size),
                                 words.ctypes.data))
    return size.value, words
