"""ctypes binding of oracle/liboracle.so (see oracle/dint_oracle.h).

TEST INFRASTRUCTURE. Importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (dint_amd/) never imports it.
PARITY UNPINNED: the reference cannot be built here (see dint_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DINT_ORACLE_LIB: another build of the same library (the sanitizer build, `make -C oracle asan`; README "Sanitizers")
_LIB = os.environ.get("DINT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

RECT, SINGLE_PACKED, MULTI_PACKED = 0, 1, 2


class _Bytes(C.Structure):  # oracle_bytes
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t), ("cap", C.c_size_t)]


def build(quiet: bool = True) -> None:
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    vp = C.c_void_p
    lib.oracle_dict_load.restype = vp
    lib.oracle_dict_load.argtypes = [C.c_int, vp, C.c_size_t]
    lib.oracle_dict_free.restype = None
    lib.oracle_dict_free.argtypes = [vp]
    lib.oracle_dict_copy.restype = C.c_uint32
    lib.oracle_dict_copy.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.oracle_decode_list.restype = vp
    lib.oracle_decode_list.argtypes = [vp, vp, vp, C.c_size_t]
    lib.oracle_header_read.restype = vp
    lib.oracle_header_read.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.oracle_decode_stream.restype = C.c_uint64
    lib.oracle_decode_stream.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.oracle_interpolative_decode.restype = vp
    lib.oracle_interpolative_decode.argtypes = [vp, vp, C.c_uint32, C.c_size_t]
    lib.oracle_posting_list_decode.restype = C.c_uint32
    lib.oracle_posting_list_decode.argtypes = [vp, vp, vp, vp, vp]
    lib.oracle_and_query.restype = C.c_uint64
    lib.oracle_and_query.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_size_t]
    lib.oracle_and_query_freqs.restype = C.c_uint64
    lib.oracle_and_query_freqs.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.oracle_and_queries_parallel.restype = C.c_double
    lib.oracle_and_queries_parallel.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    lib.oracle_time_stream.restype = C.c_double
    lib.oracle_time_stream.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.c_double,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.oracle_time_stream_parallel.restype = C.c_double
    lib.oracle_time_stream_parallel.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, C.c_double,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.oracle_selector_get.restype = C.c_uint32
    lib.oracle_selector_get.argtypes = [vp, C.c_size_t]
    lib.oracle_hash_u32s.restype = C.c_uint64
    lib.oracle_hash_u32s.argtypes = [vp, C.c_size_t]
    lib.oracle_stats_create.restype = vp
    lib.oracle_stats_create.argtypes = [C.c_int, vp]
    lib.oracle_stats_free.restype = None
    lib.oracle_stats_free.argtypes = [vp]
    lib.oracle_stats_collect.restype = C.c_int
    lib.oracle_stats_collect.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.oracle_stats_total.restype = C.c_uint64
    lib.oracle_stats_total.argtypes = [vp]
    lib.oracle_stats_distinct.restype = C.c_uint64
    lib.oracle_stats_distinct.argtypes = [vp, C.c_uint32]
    lib.oracle_stats_entries.restype = C.c_uint64
    lib.oracle_stats_entries.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    lib.oracle_stats_select.restype = C.c_uint64
    lib.oracle_stats_select.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    bp = C.POINTER(_Bytes)
    lib.oracle_bytes_free.restype = None
    lib.oracle_bytes_free.argtypes = [bp]
    lib.oracle_builder_load.restype = vp
    lib.oracle_builder_load.argtypes = [C.c_int, vp, C.c_size_t]
    lib.oracle_builder_free.restype = None
    lib.oracle_builder_free.argtypes = [vp]
    lib.oracle_builder_lookup.restype = C.c_uint32
    lib.oracle_builder_lookup.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    lib.oracle_encode_list.argtypes = [vp, C.c_int, vp, C.c_uint32, bp]
    lib.oracle_vbyte_encode.argtypes = [C.c_uint32, bp]
    lib.oracle_interpolative_encode.argtypes = [vp, C.c_uint32, C.c_size_t, bp]
    lib.oracle_block_encode.argtypes = [vp, C.c_int, vp, C.c_uint32, C.c_uint32, bp]
    lib.oracle_posting_list_write.argtypes = [vp, vp, C.c_int, C.c_uint32, vp, vp, bp]
    lib.oracle_encode_collection.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, bp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.oracle_pack_dictionary.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, bp]
    return lib


_lib = _load()


class OracleDict:
    def __init__(self, kind: int, file_bytes: bytes):
        self.kind = kind
        self._buf = (C.c_char * len(file_bytes)).from_buffer_copy(file_bytes)
        self._h = _lib.oracle_dict_load(kind, C.addressof(self._buf), len(file_bytes))
        if not self._h:
            raise ValueError("oracle: malformed dictionary file")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.oracle_dict_free(h)

    def copy(self, index: int, dict_id: int = 0):
        """Dictionary::copy — (logical size, the 16 words it writes)."""
        out = np.zeros(16, dtype=np.uint32)
        size = _lib.oracle_dict_copy(self._h, dict_id, index, out.ctypes.data)
        return int(size), out

    def decode_list(self, enc: np.ndarray, offset: int, n: int):
        """Decoder::decode(dict, in, out, universe, n) -> (out[0:n], bytes consumed)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        out = np.zeros(n + 256 + 16, dtype=np.uint32)  # zeroed, with the overflow area
        base = enc.ctypes.data + offset
        end = _lib.oracle_decode_list(self._h, base, out.ctypes.data, n)
        return out[:n].copy(), int(end - base)

    def decode_stream(self, enc: np.ndarray, total_ints: int | None = None):
        """vroom_env/decode.cpp loop -> (all lists' integers back to back, number of lists)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        if total_ints is None:
            total_ints = int(_lib.oracle_decode_stream(self._h, enc.ctypes.data, enc.size, None, 0, None))
        out = np.empty(total_ints, dtype=np.uint32)
        lists = C.c_uint64()
        got = _lib.oracle_decode_stream(self._h, enc.ctypes.data, enc.size, out.ctypes.data, out.size,
                                        C.byref(lists))
        if got != total_ints:
            raise ValueError(f"oracle: decoded {got} integers, expected {total_ints}")
        return out, lists.value

    def time_stream(self, enc: np.ndarray, max_lists: int = 0, max_seconds: float = 0.0):
        """Reference benchmark loop -> (summed decode seconds, ints, lists)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        ints, lists = C.c_uint64(), C.c_uint64()
        sec = _lib.oracle_time_stream(self._h, enc.ctypes.data, enc.size, max_lists, max_seconds,
                                      C.byref(ints), C.byref(lists))
        if sec < 0:
            raise MemoryError("oracle: could not allocate the 50M-int decode buffer")
        return sec, ints.value, lists.value


    def time_stream_parallel(self, enc: np.ndarray, range_starts, seconds: float):
        """The all-cores leg: one pthread per range of lists (byte offsets of the ranges' first headers), each with a
        persistent buffer, looping for `seconds` -> (wall seconds first start to last end, ints, lists)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        starts = np.ascontiguousarray(range_starts, dtype=np.uint64)
        ints, lists = C.c_uint64(), C.c_uint64()
        wall = _lib.oracle_time_stream_parallel(self._h, enc.ctypes.data, enc.size, starts.ctypes.data, starts.size,
                                                seconds, C.byref(ints), C.byref(lists))
        if wall < 0:
            raise RuntimeError("oracle: the parallel timing run failed (bad ranges, or out of memory / threads)")
        return wall, ints.value, lists.value


def header_read(enc: np.ndarray, offset: int):
    """header::read -> (n, universe, offset of the payload)."""
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    n, u = C.c_uint32(), C.c_uint32()
    base = enc.ctypes.data
    end = _lib.oracle_header_read(base + offset, C.byref(n), C.byref(u))
    return n.value, u.value, int(end - base)


def interpolative_decode(buf: np.ndarray, offset: int, sum_of_values: int, n: int):
    """interpolative_block::decode -> (values, bytes consumed)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    padded = np.concatenate([buf, np.zeros(8, dtype=np.uint8)])  # the bit reader fetches whole u32 words
    out = np.zeros(n + 1, dtype=np.uint32)
    base = padded.ctypes.data + offset
    end = _lib.oracle_interpolative_decode(base, out.ctypes.data, sum_of_values & 0xFFFFFFFF, n)
    return out[:n].copy(), int(end - base)


def posting_list_decode(docs_dict: OracleDict, freqs_dict: OracleDict, index: np.ndarray, offset: int):
    """document_enumerator walked front to back -> (docids, freqs)."""
    index = np.ascontiguousarray(index, dtype=np.uint8)
    padded = np.concatenate([index, np.zeros(16, dtype=np.uint8)])
    base = padded.ctypes.data + offset
    n = _lib.oracle_posting_list_decode(docs_dict._h, freqs_dict._h, base, None, None)
    docids, freqs = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
    _lib.oracle_posting_list_decode(docs_dict._h, freqs_dict._h, base, docids.ctypes.data, freqs.ctypes.data)
    return docids, freqs


def and_query(docs_dict: OracleDict, index: np.ndarray, list_offsets: np.ndarray, num_docs: int, terms) -> int:
    """and_query<false>: number of documents containing every term (block-max skipping enumerators)."""
    index = np.ascontiguousarray(index, dtype=np.uint8)
    padded = np.concatenate([index, np.zeros(16, dtype=np.uint8)])
    offs = np.ascontiguousarray(list_offsets, dtype=np.uint64)
    t = np.ascontiguousarray(terms, dtype=np.uint32)
    return int(_lib.oracle_and_query(docs_dict._h, padded.ctypes.data, offs.ctypes.data, num_docs, t.ctypes.data, t.size))


class OracleIndex:
    """An index padded once for the enumerators' word-wise reads; and_query per call without copies."""

    def __init__(self, docs_dict: OracleDict, index: np.ndarray, list_offsets: np.ndarray, num_docs: int):
        self.docs_dict = docs_dict
        self._padded = np.concatenate([np.ascontiguousarray(index, dtype=np.uint8), np.zeros(16, dtype=np.uint8)])
        self._offs = np.ascontiguousarray(list_offsets, dtype=np.uint64)
        self.num_docs = num_docs

    def and_query(self, terms) -> int:
        t = np.ascontiguousarray(terms, dtype=np.uint32)
        return int(_lib.oracle_and_query(self.docs_dict._h, self._padded.ctypes.data, self._offs.ctypes.data,
                                         self.num_docs, t.ctypes.data, t.size))

    def and_queries_parallel(self, queries, threads: int, passes: int = 1):
        """The whole log on `threads` pthreads inside liboracle (query q on thread q % threads) -> (match counts, wall
        seconds of one pass)."""
        terms = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint32) for q in queries]) if len(queries) else
                                     np.zeros(0, dtype=np.uint32))
        offs = np.zeros(len(queries) + 1, dtype=np.uint64)
        np.cumsum([len(q) for q in queries], out=offs[1:])
        counts = np.zeros(len(queries), dtype=np.uint64)
        wall = _lib.oracle_and_queries_parallel(self.docs_dict._h, self._padded.ctypes.data, self._offs.ctypes.data, self.num_docs,
                                                terms.ctypes.data, offs.ctypes.data, len(queries), threads, passes,
                                                counts.ctypes.data)
        if wall < 0:
            raise RuntimeError("oracle_and_queries_parallel failed")
        return counts, wall

    def and_query_freqs(self, freqs_dict: OracleDict, terms):
        """and_query<true> -> (matches, sum of the freq() of every term at every match, freqs blocks decoded)."""
        t = np.ascontiguousarray(terms, dtype=np.uint32)
        fsum, fblocks = C.c_uint64(), C.c_uint64()
        n = _lib.oracle_and_query_freqs(self.docs_dict._h, freqs_dict._h, self._padded.ctypes.data, self._offs.ctypes.data,
                                        self.num_docs, t.ctypes.data, t.size, C.byref(fsum), C.byref(fblocks))
        return int(n), int(fsum.value), int(fblocks.value)


# ---- dictionary construction statistics (dint_oracle_stats.c) ----

NGRAM_DTYPE = np.dtype([("pos", "<u8"), ("freq", "<u8"), ("len", "<u4"), ("context", "<u4")])  # oracle_ngram


def selector_get(values) -> int:
    """selector::get (statistics_collectors.hpp:21-40): the context of a block."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    return int(_lib.oracle_selector_get(v.ctypes.data, v.size))


def hash_u32s(values) -> int:
    """hash_bytes64 over u32 words (hash_utils.hpp:77-80)."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    return int(_lib.oracle_hash_u32s(v.ctypes.data, v.size))


class Stats:
    """block_statistics (multi=False) / block_multi_statistics (multi=True) over a collection's gaps: collect() once per
    list, then entries(context) = every distinct n-gram with its count, select(context) = what DSF appends, in order."""

    def __init__(self, multi: bool, gaps: np.ndarray):
        self.gaps = np.ascontiguousarray(gaps, dtype=np.uint32)
        self.contexts = 6 if multi else 1
        self._h = _lib.oracle_stats_create(int(bool(multi)), self.gaps.ctypes.data)
        if not self._h:
            raise MemoryError("oracle_stats_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.oracle_stats_free(h)

    def collect(self, first: int, n: int) -> None:
        assert 0 <= first and first + n <= self.gaps.size
        if not _lib.oracle_stats_collect(self._h, first, n):
            raise MemoryError("oracle_stats_collect")

    def collect_lists(self, lens) -> None:
        first = 0
        for n in lens:
            self.collect(first, int(n))
            first += int(n)

    @property
    def total_integers(self) -> int:
        return int(_lib.oracle_stats_total(self._h))

    def entries(self, context: int = 0) -> np.ndarray:
        n = int(_lib.oracle_stats_distinct(self._h, context))
        out = np.empty(n, dtype=NGRAM_DTYPE)
        got = int(_lib.oracle_stats_entries(self._h, context, out.ctypes.data, n))
        assert got == n
        return out

    def select(self, context: int = 0):
        """-> (the entries DSF appends for this context, in dictionary order; how many passed the filter)."""
        out = np.empty(65536, dtype=NGRAM_DTYPE)
        passed = int(_lib.oracle_stats_select(self._h, context, out.ctypes.data, out.size))
        return out[: min(passed, 65536)].copy(), passed

    def ngram(self, e) -> tuple:
        return tuple(int(x) for x in self.gaps[int(e["pos"]): int(e["pos"]) + int(e["len"])])


# ---- the encode side (dint_oracle_encode.c) and the dictionary packing (f2) ----

def _take(b: _Bytes, ok: int, what: str) -> np.ndarray:
    try:
        if not ok:
            raise ValueError(f"oracle: {what} failed")
        out = np.empty(b.size, dtype=np.uint8)
        if b.size:
            C.memmove(out.ctypes.data, b.data, b.size)
        return out
    finally:
        _lib.oracle_bytes_free(C.byref(b))


class OracleBuilder:
    """Dictionary::builder after load(file) + prepare_for_encoding(), and the encoders that run against it."""

    def __init__(self, kind: int, file_bytes: bytes):
        self.kind = kind
        buf = (C.c_char * len(file_bytes)).from_buffer_copy(file_bytes)
        self._h = _lib.oracle_builder_load(kind, C.addressof(buf), len(file_bytes))
        if not self._h:
            raise ValueError("oracle: malformed dictionary file")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.oracle_builder_free(h)

    def lookup(self, values, dictionary_id: int = 0, b: int = 16) -> int:
        v = np.ascontiguousarray(values, dtype=np.uint32)
        return int(_lib.oracle_builder_lookup(self._h, dictionary_id, v.ctypes.data, v.size, b))

    def encode_list(self, gaps, greedy: bool = False) -> np.ndarray:
        """Encoder::encode(builder, in, universe, n, out) of the vroom environment -> the list's payload bytes."""
        g = np.ascontiguousarray(gaps, dtype=np.uint32)
        b = _Bytes()
        return _take(b, _lib.oracle_encode_list(self._h, int(greedy), g.ctypes.data, g.size, C.byref(b)), "encode_list")

    def block_encode(self, values, sum_of_values: int, greedy: bool = False) -> np.ndarray:
        """Coder::encode(builder, in, sum_of_values, n, out) of the index, one block."""
        v = np.ascontiguousarray(values, dtype=np.uint32)
        b = _Bytes()
        return _take(b, _lib.oracle_block_encode(self._h, int(greedy), v.ctypes.data, sum_of_values & 0xFFFFFFFF, v.size,
                                                 C.byref(b)), "block_encode")

    def encode_collection(self, words, docs: bool, greedy: bool = False):
        """The vroom `encode` program over a collection file's u32 words -> (stream bytes, lists, integers)."""
        w = np.ascontiguousarray(words, dtype=np.uint32)
        b = _Bytes()
        lists, ints = C.c_uint64(), C.c_uint64()
        ok = _lib.oracle_encode_collection(self._h, int(greedy), w.ctypes.data, w.size, int(docs), C.byref(b),
                                           C.byref(lists), C.byref(ints))
        return _take(b, ok, "encode_collection"), lists.value, ints.value


def posting_list_write(docs_builder: OracleBuilder, freqs_builder: OracleBuilder, docids, freqs, greedy: bool = False) -> np.ndarray:
    """dict_posting_list::write -> the list's bytes."""
    d = np.ascontiguousarray(docids, dtype=np.uint32)
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    assert d.size == f.size and d.size > 0
    b = _Bytes()
    return _take(b, _lib.oracle_posting_list_write(docs_builder._h, freqs_builder._h, int(greedy), d.size, d.ctypes.data,
                                                   f.ctypes.data, C.byref(b)), "posting_list_write")


def interpolative_encode(values, sum_of_values: int) -> np.ndarray:
    v = np.ascontiguousarray(values, dtype=np.uint32)
    b = _Bytes()
    return _take(b, _lib.oracle_interpolative_encode(v.ctypes.data, sum_of_values & 0xFFFFFFFF, v.size, C.byref(b)),
                 "interpolative_encode")


def vbyte_encode(val: int) -> bytes:
    b = _Bytes()
    return _take(b, _lib.oracle_vbyte_encode(val & 0xFFFFFFFF, C.byref(b)), "vbyte_encode").tobytes()


def pack_dictionary(kind: int, entries, contexts=None) -> bytes:
    """builder::init / append / build / write over a selection: entries = sequences of integers in dictionary order,
    contexts[k] = the dictionary entry k goes to (multi) -> the dictionary file."""
    lens = np.array([len(e) for e in entries], dtype=np.uint32)
    words = np.ascontiguousarray(np.concatenate([np.asarray(e, dtype=np.uint32) for e in entries]) if len(entries) else
                                 np.zeros(0, dtype=np.uint32))
    ctx = None if contexts is None else np.ascontiguousarray(contexts, dtype=np.uint32)
    b = _Bytes()
    ok = _lib.oracle_pack_dictionary(kind, words.ctypes.data, lens.ctypes.data, None if ctx is None else ctx.ctypes.data,
                                     lens.size, C.byref(b))
    return _take(b, ok, "pack_dictionary").tobytes()
