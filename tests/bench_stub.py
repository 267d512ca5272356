"""TEST-ONLY stand-in for dint_amd.device, loaded by bench.py when DINT_BENCH_STUB=bench_stub: lets the
multi-process path of bench.py (self-launch, rendezvous, list-range sharding, reductions, the JSON line) run on
CPU ranks over gloo. It decodes nothing — the product has no CPU decode — so the bench line it produces is
meaningless as a measurement."""
import os
import time
import types

import numpy as np
import torch


class Dictionary:
    def __init__(self, kind, file_bytes, device=0):
        self.kind, self.device = kind, device
        # (tests of bench.py's launcher: a rank that dies, a rank that never answers)
        if os.environ.get("BENCH_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("bench_stub: this rank was told to fail")
        if os.environ.get("BENCH_STUB_HANG_RANK") == os.environ.get("RANK", "0"):
            time.sleep(3600)

    def info(self):
        return types.SimpleNamespace(hot_entries=0, lds_bytes=0)

    def decode_units(self, enc_dev, units_dev, n_units, out_dev, end_off_dev=None, stream=None):
        if end_off_dev is not None:
            end_off_dev.zero_()

    def recent_kernel_ms(self, max_n=64):
        return np.ones(max_n, dtype=np.float32)

    def stream_stats(self, enc):
        keys = ("lists", "ints", "payload_bytes", "codewords", "run_codewords", "exceptions16", "exceptions32",
                "hot_codewords", "hot_ints", "wide_blocks", "narrow_blocks")
        return types.SimpleNamespace(**{k: 0 for k in keys})


def units_to_device(units, device):
    return torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).copy())
