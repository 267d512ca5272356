"""ctypes binding of the device decode path (include/dint_hip.h).

torch is used for what it is good at here — owning device memory and streams;
every decode goes through the C ABI into the hand-written HIP kernels. There is
no CPU fallback: if libdint_hip.so is missing the import fails, and a decode
without a GPU fails with DINT_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:
    # torch ships its own HIP runtime under the same soname as the system one. Whichever is loaded first serves
    # the whole process; with the system's loaded first (by libdint_hip.so), torch later finds "No HIP GPUs".
    # So torch — which owns device memory and streams in this layer anyway — is imported before the library.
    import torch  # noqa: F401
except ImportError:  # (the C ABI itself does not need torch)
    pass

from .host import UNIT_DTYPE, KIND_BY_TYPE, RECTANGULAR, SINGLE_PACKED, MULTI_PACKED  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DINT_HIP_LIB") or os.path.join(_HERE, "libdint_hip.so")

#: every symbol include/dint_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "dint_abi_version", "dint_set_option", "dint_get_option", "dint_option_name", "dint_reset_options", "dint_strerror", "dint_last_hip_error", "dint_device_count",
    "dint_dict_create", "dint_dict_destroy", "dint_dict_info_get", "dint_index_stream", "dint_free",
    "dint_decode_units", "dint_unit_table_create", "dint_unit_table_destroy", "dint_decode_unit_table", "dint_unit_table_rank_outputs", "dint_probe_placement", "dint_decode_list_host", "dint_last_kernel_ms", "dint_recent_kernel_ms",
    "dint_stream_stats_get",
    "dint_decode_block_host", "dint_index_posting_lists", "dint_decode_posting_blocks",
    "dint_list_cache_create", "dint_list_cache_decode", "dint_list_cache_destroy",
    "dint_block_table_create", "dint_block_table_destroy", "dint_block_table_learn", "dint_block_table_ready", "dint_block_table_info_get", "dint_decode_block_table",
    "dint_query_index_create", "dint_query_index_destroy", "dint_and_queries", "dint_and_queries_freqs", "dint_count_ngrams", "dint_select_ngrams", "dint_last_kernel_clock_mhz",
)

#: dint_block_ref (include/dint_hip.h)
BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("n", "<u4"), ("base", "<u4"), ("max", "<u4"),
                        ("list", "<u4")], align=False)
assert BLOCK_DTYPE.itemsize == 32


class DictInfo(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("device", C.c_int32), ("num_dicts", C.c_uint32),
        ("entries", C.c_uint32), ("hot_entries", C.c_uint32), ("lds_bytes", C.c_uint32),
        ("table_words", C.c_uint32), ("compute_units", C.c_uint32),
    ]


class BlockTableInfo(C.Structure):  # dint_block_table_info
    _fields_ = [("n_blocks", C.c_uint64), ("n_short_blocks", C.c_uint64)] + [(k, C.c_uint32) for k in (
        "complete_decodes", "spans_exact", "freqs_units_ready", "docs_schedule", "freqs_schedule", "docs_queue_items",
        "freqs_queue_items", "short_block_tickets")]


class StreamStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "lists", "ints", "payload_bytes", "codewords", "run_codewords", "exceptions16", "exceptions32",
        "hot_codewords", "hot_ints", "wide_blocks", "narrow_blocks")]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing — the HIP extension has not been built "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    lib = C.CDLL(_LIB_PATH)
    vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64
    lib.dint_abi_version.restype = C.c_int
    lib.dint_strerror.restype = C.c_char_p
    lib.dint_strerror.argtypes = [C.c_int]
    lib.dint_set_option.argtypes = [C.c_int, C.c_longlong]
    lib.dint_get_option.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
    lib.dint_option_name.restype = C.c_char_p
    lib.dint_option_name.argtypes = [C.c_int]
    lib.dint_last_hip_error.restype = C.c_char_p
    lib.dint_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.dint_dict_create.argtypes = [C.c_int, vp, sz, C.c_int, C.POINTER(vp)]
    lib.dint_dict_destroy.restype = None
    lib.dint_dict_destroy.argtypes = [vp]
    lib.dint_dict_info_get.argtypes = [vp, C.POINTER(DictInfo)]
    lib.dint_index_stream.argtypes = [vp, vp, sz, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(u64),
                                      C.POINTER(u64)]
    lib.dint_free.restype = None
    lib.dint_free.argtypes = [vp]
    lib.dint_decode_units.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.dint_unit_table_create.argtypes = [vp, vp, sz, vp, sz, sz, vp, C.POINTER(vp)]
    lib.dint_unit_table_destroy.restype = None
    lib.dint_unit_table_destroy.argtypes = [vp]
    lib.dint_decode_unit_table.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.dint_unit_table_rank_outputs.argtypes = [vp, vp, C.POINTER(vp), sz, sz, vp, C.POINTER(C.c_float), C.POINTER(sz)]
    lib.dint_probe_placement.argtypes = [vp, C.POINTER(vp), sz, sz, vp, sz, C.POINTER(vp), sz, sz, C.c_uint64, vp, C.POINTER(C.c_float)]
    lib.dint_decode_list_host.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    lib.dint_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.dint_last_kernel_clock_mhz.argtypes = [vp, C.POINTER(C.c_float)]
    lib.dint_recent_kernel_ms.argtypes = [vp, vp, sz, C.POINTER(sz)]
    lib.dint_stream_stats_get.argtypes = [vp, vp, sz, C.POINTER(StreamStats)]
    lib.dint_decode_block_host.argtypes = [vp, vp, sz, vp, u32, sz, C.POINTER(sz)]
    lib.dint_index_posting_lists.argtypes = [vp, sz, vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(u64)]
    lib.dint_list_cache_create.argtypes = [vp, vp, vp, sz, C.POINTER(vp)]
    lib.dint_list_cache_decode.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]
    lib.dint_list_cache_destroy.restype = None
    lib.dint_list_cache_destroy.argtypes = [vp]
    lib.dint_debug_alloc_count.argtypes = [C.POINTER(u64)]
    lib.dint_decode_posting_blocks.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, sz, vp]
    lib.dint_block_table_create.argtypes = [vp, vp, sz, sz, C.POINTER(vp)]
    lib.dint_block_table_destroy.restype = None
    lib.dint_block_table_destroy.argtypes = [vp]
    lib.dint_decode_block_table.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp]
    lib.dint_block_table_learn.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.dint_block_table_ready.argtypes = [vp, C.c_int]
    lib.dint_block_table_info_get.argtypes = [vp, C.POINTER(BlockTableInfo)]
    lib.dint_query_index_create.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(vp)]
    lib.dint_query_index_destroy.restype = None
    lib.dint_query_index_destroy.argtypes = [vp]
    lib.dint_and_queries.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.dint_and_queries_freqs.argtypes = [vp, vp, vp, vp, sz, vp, vp, C.POINTER(u64), vp]
    lib.dint_count_ngrams.argtypes = [C.c_int, C.c_int, vp, u64, vp, u64, C.c_uint32, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_float)]
    lib.dint_select_ngrams.argtypes = [C.c_int, vp, u64, u64, vp, sz, C.c_uint32, C.POINTER(sz)]
    lib.dint_debug_wave_scan.argtypes = [vp, vp]
    return lib


_lib = _load()


MAX_UNIT_INTS = 1 << 28  # DINT_MAX_UNIT_INTS (include/dint_hip.h)


class DintError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        detail = _lib.dint_last_hip_error().decode()
        super().__init__(f"{where}: {_lib.dint_strerror(status).decode()} ({status})"
                         + (f" — {detail}" if detail and status == -3 else ""))


def _check(status: int, where: str) -> None:
    if status != 0:
        raise DintError(status, where)


def abi_version() -> int:
    return _lib.dint_abi_version()


#: dint_option (include/dint_hip.h): name -> number, from the library itself
OPTIONS = {}
_i = 0
while _lib.dint_option_name(_i):
    OPTIONS[_lib.dint_option_name(_i).decode()] = _i
    _i += 1


def set_option(name: str, value: int) -> None:
    """dint_set_option: a process-wide switch for tests and measurements ("bundles", "index_concurrent",
    "query_lean_pages", "query_tail_pages", "query_fused_pages", ... : device.OPTIONS lists them)."""
    _check(_lib.dint_set_option(OPTIONS[name], int(value)), f"dint_set_option({name})")


def get_option(name: str) -> int:
    v = C.c_longlong()
    _check(_lib.dint_get_option(OPTIONS[name], C.byref(v)), f"dint_get_option({name})")
    return int(v.value)


def reset_options() -> None:
    _check(_lib.dint_reset_options(), "dint_reset_options")


class options:
    """with device.options(query_fused_pages=0): ...  — the options set inside, restored on the way out."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def device_count() -> int:
    n = C.c_int()
    _check(_lib.dint_device_count(C.byref(n)), "dint_device_count")
    return n.value


class Dictionary:
    """Device-resident dictionary (reference: Dictionary::builder::load + build(dict))."""

    def __init__(self, kind: int, file_bytes: bytes, device: int = 0):
        self.kind = kind
        self._h = C.c_void_p()
        buf = (C.c_char * len(file_bytes)).from_buffer_copy(file_bytes)
        _check(_lib.dint_dict_create(kind, C.addressof(buf), len(file_bytes), device, C.byref(self._h)),
               "dint_dict_create")
        self.device = device

    def close(self) -> None:
        h, self._h = self._h, C.c_void_p()
        if h and _lib is not None:  # (None at interpreter shutdown)
            _lib.dint_dict_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> DictInfo:
        info = DictInfo()
        _check(_lib.dint_dict_info_get(self._h, C.byref(info)), "dint_dict_info_get")
        return info

    # -- H3: sidecar ---------------------------------------------------------
    def index_stream(self, enc: np.ndarray, unit_ints: int = 4096):
        """Host pre-pass over a vroom stream -> (unit table, total ints, number of lists)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        units, n_units = C.c_void_p(), C.c_size_t()
        total, lists = C.c_uint64(), C.c_uint64()
        _check(_lib.dint_index_stream(self._h, enc.ctypes.data, enc.size, unit_ints, C.byref(units),
                                      C.byref(n_units), C.byref(total), C.byref(lists)), "dint_index_stream")
        try:
            arr = np.empty(n_units.value, dtype=UNIT_DTYPE)
            if n_units.value:
                C.memmove(arr.ctypes.data, units, arr.nbytes)
        finally:
            _lib.dint_free(units)
        return arr, total.value, lists.value

    # -- decode ----------------------------------------------------------------
    def decode_units(self, enc_dev, units_dev, n_units: int, out_dev, end_off_dev=None, stream=None) -> None:
        """Asynchronous batched decode. All tensors are torch CUDA tensors on this
        dictionary's device: enc_dev uint8, units_dev the raw bytes of a UNIT_DTYPE
        table, out_dev int32/uint32 storage, end_off_dev (optional) int64[n_units]."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(enc_dev.device).cuda_stream
        _check(_lib.dint_decode_units(
            self._h, enc_dev.data_ptr(), enc_dev.numel() * enc_dev.element_size(), units_dev.data_ptr(), n_units,
            out_dev.data_ptr(), out_dev.numel(), end_off_dev.data_ptr() if end_off_dev is not None else None,
            stream), "dint_decode_units")

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(_lib.dint_last_kernel_ms(self._h, C.byref(ms)), "dint_last_kernel_ms")
        return ms.value

    def last_kernel_clock_mhz(self) -> float:
        """The shader clock the most recent decode kernel ran at (its first wave's cycle count / its duration)."""
        mhz = C.c_float()
        _check(_lib.dint_last_kernel_clock_mhz(self._h, C.byref(mhz)), "dint_last_kernel_clock_mhz")
        return mhz.value

    def recent_kernel_ms(self, max_n: int = 64) -> np.ndarray:
        """Kernel times (ms) of the most recent launches, oldest first (from their own event pairs)."""
        out = np.zeros(max_n, dtype=np.float32)
        n = C.c_size_t()
        _check(_lib.dint_recent_kernel_ms(self._h, out.ctypes.data, max_n, C.byref(n)), "dint_recent_kernel_ms")
        return out[: n.value].copy()

    def stream_stats(self, enc: np.ndarray) -> StreamStats:
        """Host pre-pass: what the stream is made of (codewords, exceptions, on-chip share)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        st = StreamStats()
        _check(_lib.dint_stream_stats_get(self._h, enc.ctypes.data, enc.size, C.byref(st)), "dint_stream_stats_get")
        return st

    def decode_list(self, enc: np.ndarray, offset: int, n: int):
        """The reference's Decoder::decode(dict, in, out, universe, n) call shape on host
        memory -> (out[0:n], bytes consumed). One wavefront wide: for tests and small lists."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        out = np.empty(n, dtype=np.uint32)
        consumed = C.c_size_t()
        _check(_lib.dint_decode_list_host(self._h, enc.ctypes.data + offset, enc.size - offset, out.ctypes.data, n,
                                          C.byref(consumed)), "dint_decode_list_host")
        return out, consumed.value


class UnitTable:
    """A unit table prepared once for repeated decodes of one resident stream (dint_unit_table): the bundle schedule —
    a property of the stream and its sidecar — is computed here, a decode is then one kernel launch. Borrows the
    dictionary and the two device tensors (kept alive here)."""

    def __init__(self, dictionary: "Dictionary", enc_dev, units_dev, n_units: int, out_capacity: int, stream=None):
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(enc_dev.device).cuda_stream
        self._dict, self._enc, self._units = dictionary, enc_dev, units_dev
        self._h = C.c_void_p()
        _check(_lib.dint_unit_table_create(dictionary._h, enc_dev.data_ptr(), enc_dev.numel() * enc_dev.element_size(),
                                           units_dev.data_ptr(), n_units, out_capacity, stream, C.byref(self._h)),
               "dint_unit_table_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dint_unit_table_destroy(h)

    __del__ = close

    def decode(self, out_dev, end_off_dev=None, stream=None) -> None:
        """Enqueue the decode of every unit (asynchronous)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(out_dev.device).cuda_stream
        _check(_lib.dint_decode_unit_table(self._dict._h, self._h, out_dev.data_ptr(), out_dev.numel(),
                                           end_off_dev.data_ptr() if end_off_dev is not None else None, stream),
               "dint_decode_unit_table")

    def rank_outputs(self, outs, stream=None):
        """dint_unit_table_rank_outputs: decode into every candidate output tensor, -> (kernel ms of each, index of the
        fastest). The kernel's time depends on where the output lies relative to the stream."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(outs[0].device).cuda_stream
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        ms = (C.c_float * len(outs))()
        best = C.c_size_t(0)
        _check(_lib.dint_unit_table_rank_outputs(self._dict._h, self._h, ptrs, len(outs), min(o.numel() for o in outs), stream, ms,
                                                 C.byref(best)), "dint_unit_table_rank_outputs")
        return [float(x) for x in ms], int(best.value)


def probe_placement(dictionary: "Dictionary", encs, units_dev, n_units: int, outs, sample_ints: int = 0, stream=None) -> np.ndarray:
    """dint_probe_placement: kernel ms of a sampled decode on every (stream copy, output buffer) pair -> float array
    [len(encs), len(outs)]. `encs`: CUDA tensors holding the same encoded bytes at different addresses."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream(outs[0].device).cuda_stream
    e = (C.c_void_p * len(encs))(*[t.data_ptr() for t in encs])
    o = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    ms = (C.c_float * (len(encs) * len(outs)))()
    _check(_lib.dint_probe_placement(dictionary._h, e, len(encs), min(t.numel() for t in encs), units_dev.data_ptr(), n_units, o,
                                     len(outs), min(t.numel() for t in outs), int(sample_ints), stream, ms), "dint_probe_placement")
    return np.array(list(ms), dtype=np.float64).reshape(len(encs), len(outs))


def decode_block(dictionary: "Dictionary", buf: np.ndarray, offset: int, sum_of_values: int, n: int):
    """The reference's in-index block Coder call, Coder::decode(dict, in, out, sum_of_values, n), on host
    memory -> (out[0:n], bytes consumed)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint32)
    consumed = C.c_size_t()
    _check(_lib.dint_decode_block_host(dictionary._h, buf.ctypes.data + offset, buf.size - offset, out.ctypes.data,
                                       sum_of_values & 0xFFFFFFFF, n, C.byref(consumed)), "dint_decode_block_host")
    return out, consumed.value


class ListCache:
    """One posting list (host bytes, dict_posting_list layout) decoded once on the device; `decode(offset, n)` is then the
    block Coder's call for the docs or freqs part that starts `offset` bytes into the list, served from host memory."""

    def __init__(self, docs_dict: "Dictionary", freqs_dict, list_bytes: np.ndarray):
        self._list = np.ascontiguousarray(list_bytes, dtype=np.uint8)
        self._h = C.c_void_p()
        _check(_lib.dint_list_cache_create(docs_dict._h, freqs_dict._h if freqs_dict is not None else None,
                                           self._list.ctypes.data, self._list.size, C.byref(self._h)), "dint_list_cache_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dint_list_cache_destroy(h)

    __del__ = close

    def decode(self, offset: int, n: int):
        out = np.empty(n, dtype=np.uint32)
        consumed = C.c_size_t()
        _check(_lib.dint_list_cache_decode(self._h, offset, out.ctypes.data, n, C.byref(consumed)), "dint_list_cache_decode")
        return out, consumed.value


def alloc_count() -> int:
    """Device / pinned allocations the library has made so far in this process (test hook)."""
    n = C.c_uint64()
    _check(_lib.dint_debug_alloc_count(C.byref(n)), "dint_debug_alloc_count")
    return n.value


def index_posting_lists(index: np.ndarray, list_offsets: np.ndarray):
    """Host: flatten the block directories of the posting lists starting at list_offsets[:-1]
    (dict_posting_list layout) -> (block table BLOCK_DTYPE[], total postings)."""
    index = np.ascontiguousarray(index, dtype=np.uint8)
    offs = np.ascontiguousarray(list_offsets, dtype=np.uint64)
    n_lists = max(0, len(offs) - 1)
    blocks, n_blocks, total = C.c_void_p(), C.c_size_t(), C.c_uint64()
    _check(_lib.dint_index_posting_lists(index.ctypes.data, index.size, offs.ctypes.data, n_lists, C.byref(blocks),
                                         C.byref(n_blocks), C.byref(total)), "dint_index_posting_lists")
    try:
        arr = np.empty(n_blocks.value, dtype=BLOCK_DTYPE)
        if n_blocks.value:
            C.memmove(arr.ctypes.data, blocks, arr.nbytes)
    finally:
        _lib.dint_free(blocks)
    return arr, total.value


def decode_posting_lists(docs_dict: "Dictionary", freqs_dict, index: np.ndarray, blocks: np.ndarray, total: int):
    """Upload an index + block table, decode every posting on the device, download.
    -> (docids u32[], freqs u32[] or None)"""
    import torch

    dev = torch.device("cuda", docs_dict.device)
    padded = np.concatenate([np.ascontiguousarray(index, dtype=np.uint8), np.zeros(16, dtype=np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    blocks_dev = torch.from_numpy(np.ascontiguousarray(blocks).view(np.uint8).copy()).to(dev)
    docids_dev = torch.empty(max(1, total), dtype=torch.int32, device=dev)
    freqs_dev = torch.empty(max(1, total), dtype=torch.int32, device=dev) if freqs_dict is not None else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    _check(_lib.dint_decode_posting_blocks(
        docs_dict._h, freqs_dict._h if freqs_dict is not None else None, index_dev.data_ptr(), padded.size,
        blocks_dev.data_ptr(), len(blocks), docids_dev.data_ptr(),
        freqs_dev.data_ptr() if freqs_dev is not None else None, total, stream), "dint_decode_posting_blocks")
    torch.cuda.synchronize(dev)
    docids = docids_dev.cpu().numpy().view(np.uint32)[:total]
    freqs = freqs_dev.cpu().numpy().view(np.uint32)[:total] if freqs_dev is not None else None
    return docids, freqs


class BlockTable:
    """A block table prepared once for asynchronous in-index decodes (dint_block_table)."""

    def __init__(self, docs_dict: "Dictionary", blocks: np.ndarray, index_bytes: int):
        self._h = C.c_void_p()
        self._blocks = np.ascontiguousarray(blocks)
        _check(_lib.dint_block_table_create(docs_dict._h, self._blocks.ctypes.data, len(self._blocks), index_bytes,
                                            C.byref(self._h)), "dint_block_table_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dint_block_table_destroy(h)

    __del__ = close

    def learn(self, docs_dict, freqs_dict, index_dev, index_bytes, stream=None):
        """The sizing pass at set-up (dint_block_table_learn): after it the first decode is already the one launch."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(index_dev.device).cuda_stream
        _check(_lib.dint_block_table_learn(self._h, docs_dict._h, freqs_dict._h if freqs_dict is not None else None,
                                           index_dev.data_ptr(), index_bytes, stream), "dint_block_table_learn")

    def ready(self, with_freqs: bool = True) -> bool:
        return bool(_lib.dint_block_table_ready(self._h, int(with_freqs)))

    def info(self) -> dict:
        i = BlockTableInfo()
        _check(_lib.dint_block_table_info_get(self._h, C.byref(i)), "dint_block_table_info_get")
        return {k: int(getattr(i, k)) for k, _ in BlockTableInfo._fields_}

    def decode(self, docs_dict, freqs_dict, index_dev, index_bytes, docids_dev, freqs_dev, stream=None):
        """Enqueue the decode of every block (asynchronous); tensors are CUDA tensors on the dictionaries' device."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(index_dev.device).cuda_stream
        _check(_lib.dint_decode_block_table(
            docs_dict._h, freqs_dict._h if freqs_dict is not None else None, index_dev.data_ptr(), index_bytes, self._h,
            docids_dev.data_ptr(), freqs_dev.data_ptr() if freqs_dev is not None else None, docids_dev.numel(), stream),
            "dint_decode_block_table")


class QueryIndex:
    """An index resident on the device, ready for conjunctive queries: the reference's
    `index` + `and_query<false>` pair (include/ds2i/queries.hpp:34-84), a batch per call."""

    def __init__(self, docs_dict: "Dictionary", index: np.ndarray, list_offsets: np.ndarray):
        import torch

        self.docs_dict = docs_dict
        self.blocks, self.total = index_posting_lists(index, list_offsets)
        self.n_lists = max(0, len(list_offsets) - 1)
        dev = torch.device("cuda", docs_dict.device)
        padded = np.concatenate([np.ascontiguousarray(index, dtype=np.uint8), np.zeros(16, dtype=np.uint8)])
        self._index_dev = torch.from_numpy(padded).to(dev)
        self._h = C.c_void_p()
        _check(_lib.dint_query_index_create(docs_dict._h, self._index_dev.data_ptr(), padded.size,
                                            self.blocks.ctypes.data, len(self.blocks), self.n_lists,
                                            C.byref(self._h)), "dint_query_index_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:  # (None at interpreter shutdown)
            _lib.dint_query_index_destroy(h)

    __del__ = close

    def and_queries(self, queries) -> np.ndarray:
        """queries: sequence of term-id sequences -> u64 result counts, one per query."""
        import torch

        lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
        offs = np.zeros(len(queries) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        terms = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint32) for q in queries])
                                     if len(queries) else np.zeros(0, np.uint32), dtype=np.uint32)
        counts = np.zeros(len(queries), dtype=np.uint64)
        stream = torch.cuda.current_stream(torch.device("cuda", self.docs_dict.device)).cuda_stream
        _check(_lib.dint_and_queries(self._h, terms.ctypes.data, offs.ctypes.data, len(queries),
                                     counts.ctypes.data, stream), "dint_and_queries")
        return counts

    def and_queries_packed(self, terms: np.ndarray, offsets: np.ndarray, counts: np.ndarray, stream: int = 0) -> None:
        """The bare call: queries already packed (`terms` u32, `offsets` u64[n + 1], `counts` u64[n] out) — what a
        C++ caller hands over; and_queries() packs Python lists first."""
        _check(_lib.dint_and_queries(self._h, terms.ctypes.data, offsets.ctypes.data, counts.size, counts.ctypes.data, stream),
               "dint_and_queries")

    def and_queries_with_freqs(self, freqs_dict: "Dictionary", queries):
        """`and_query<true>` for a batch -> (counts, sums of the freqs read at the matches, freqs blocks decoded)."""
        return and_queries_with_freqs(self, freqs_dict, queries)


def _pack_queries(queries):
    lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
    offs = np.zeros(len(queries) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    terms = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint32) for q in queries])
                                 if len(queries) else np.zeros(0, np.uint32), dtype=np.uint32)
    return terms, offs


def and_queries_with_freqs(qi: "QueryIndex", freqs_dict: "Dictionary", queries):
    """and_query<true> for a batch: -> (counts u64[], freq sums u64[], freqs blocks decoded)."""
    import torch

    terms, offs = _pack_queries(queries)
    counts = np.zeros(len(queries), dtype=np.uint64)
    sums = np.zeros(len(queries), dtype=np.uint64)
    nblocks = C.c_uint64()
    stream = torch.cuda.current_stream(torch.device("cuda", qi.docs_dict.device)).cuda_stream
    _check(_lib.dint_and_queries_freqs(qi._h, freqs_dict._h, terms.ctypes.data, offs.ctypes.data, len(queries),
                                       counts.ctypes.data, sums.ctypes.data, C.byref(nblocks), stream), "dint_and_queries_freqs")
    return counts, sums, nblocks.value


NGRAM_DTYPE = np.dtype([("pos", "<u8"), ("freq", "<u4"), ("len", "u1"), ("ctx", "u1"), ("pad", "<u2")])  # dint_ngram


def count_ngrams(gaps_dev, list_starts: np.ndarray, multi: bool, device: int = 0, top_k: int = 0):
    """Block statistics on the device (dint_count_ngrams): gaps_dev — a torch CUDA tensor of u32 d-gaps, lists back
    to back; list_starts — their n + 1 offsets; top_k > 0: only the entries that can make the first top_k of their
    context. -> (entries NGRAM_DTYPE[], kernel ms)."""
    starts = np.ascontiguousarray(list_starts, dtype=np.uint64)
    out, n, ms = C.c_void_p(), C.c_size_t(), C.c_float()
    _check(_lib.dint_count_ngrams(device, int(bool(multi)), gaps_dev.data_ptr(), gaps_dev.numel(), starts.ctypes.data,
                                  len(starts) - 1, top_k, C.byref(out), C.byref(n), C.byref(ms)), "dint_count_ngrams")
    try:
        arr = np.empty(n.value, dtype=NGRAM_DTYPE)
        if n.value:
            C.memmove(arr.ctypes.data, out, arr.nbytes)
    finally:
        _lib.dint_free(out)
    return arr, ms.value


def select_ngrams(gaps_dev, entries: np.ndarray, total_ints: int, device: int = 0, top_k: int = 65536) -> np.ndarray:
    """The selection on the device (dint_select_ngrams): of count_ngrams' entries, the ones the filter keeps, every context's
    in dictionary order, the first top_k of each."""
    e = np.ascontiguousarray(entries, dtype=NGRAM_DTYPE).copy()
    n = C.c_size_t()
    _check(_lib.dint_select_ngrams(device, gaps_dev.data_ptr(), gaps_dev.numel(), total_ints, e.ctypes.data, e.size, top_k, C.byref(n)),
           "dint_select_ngrams")
    return e[: n.value]


def build_dictionary(kind: int, coll, max_sample_ints: int = 0, device: int = 0):
    """host.build_dictionary with counting AND selection on the device: the sampled lists' n-grams counted by
    dint_count_ngrams, filtered / sorted / cut to the first 65536 of every context by dint_select_ngrams, packed by the host
    library — byte-identical to the host-only path. -> (dictionary file, counting kernel ms)."""
    import torch

    from . import host

    n_lists, ints = 0, 0
    for n_lists in range(len(coll.lens) + 1):  # the same prefix sample as dinth_build_dictionary
        if n_lists == len(coll.lens):
            break
        if max_sample_ints and n_lists and ints + int(coll.lens[n_lists]) > max_sample_ints:
            break
        ints += int(coll.lens[n_lists])
    starts = np.zeros(n_lists + 1, dtype=np.uint64)
    np.cumsum(coll.lens[:n_lists], out=starts[1:])
    gaps = np.ascontiguousarray(coll.gaps[:ints], dtype=np.uint32)
    gaps_dev = torch.from_numpy(gaps.view(np.int32)).to(torch.device("cuda", device))
    entries, ms = count_ngrams(gaps_dev, starts, kind == host.MULTI_PACKED, device, top_k=65536)  # DSF-65536-16
    chosen = select_ngrams(gaps_dev, entries, ints, device, top_k=65536)
    return host.pack_dictionary(kind, gaps, chosen), ms


def units_to_device(units: np.ndarray, device):
    """Upload a UNIT_DTYPE table as raw bytes."""
    import torch

    raw = np.ascontiguousarray(units).view(np.uint8)
    return torch.from_numpy(raw.copy()).to(device)


def decode_stream(dictionary: Dictionary, enc: np.ndarray, units: np.ndarray, total_ints: int):
    """Upload, decode every unit, download. -> (integers, end offsets, kernel ms)"""
    import torch

    dev = torch.device("cuda", dictionary.device)
    enc_dev = torch.from_numpy(np.ascontiguousarray(enc, dtype=np.uint8)).to(dev)
    units_dev = units_to_device(units, dev)
    out_dev = torch.empty(max(1, total_ints), dtype=torch.int32, device=dev)
    end_dev = torch.zeros(max(1, len(units)), dtype=torch.int64, device=dev)
    dictionary.decode_units(enc_dev, units_dev, len(units), out_dev, end_dev)
    torch.cuda.synchronize(dev)
    ms = dictionary.last_kernel_ms() if len(units) else 0.0
    out = out_dev.cpu().numpy().view(np.uint32)[:total_ints]
    return out, end_dev.cpu().numpy().view(np.uint64)[: len(units)], ms


def debug_wave_scan(values) -> np.ndarray:
    v = np.ascontiguousarray(values, dtype=np.uint32)
    assert v.size == 64
    out = np.empty(64, dtype=np.uint32)
    _check(_lib.dint_debug_wave_scan(v.ctypes.data, out.ctypes.data), "dint_debug_wave_scan")
    return out
