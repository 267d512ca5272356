"""Static partition of posting lists across GPUs.

Lists are independent (vroom_env/decode.cpp:139-150 carries no state from one
list to the next), so the multi-GPU path is a contiguous list-range partition
balanced by the number of postings — not by the number of lists: lengths are
heavy-tailed. The dictionary is replicated; no data moves between ranks.
"""
from __future__ import annotations

import numpy as np

from .host import UNIT_DTYPE


def partition_lists(lens: np.ndarray, world: int):
    """-> [(first_list, end_list)] * world, contiguous, covering every list, with
    per-rank posting counts as equal as list boundaries allow."""
    lens = np.asarray(lens, dtype=np.uint64)
    if world < 1:
        raise ValueError("world must be >= 1")
    cum = np.cumsum(lens, dtype=np.uint64)
    total = int(cum[-1]) if len(cum) else 0
    bounds = [0]
    for k in range(1, world):
        target = total * k // world
        # first list whose cumulative end exceeds the target goes to the next rank
        # if the target is closer to its start than to its end
        i = int(np.searchsorted(cum, target, side="right"))
        if i < len(cum):
            start = int(cum[i - 1]) if i else 0
            if target - start > int(cum[i]) - target:
                i += 1
        bounds.append(max(bounds[-1], min(i, len(lens))))
    bounds.append(len(lens))
    return [(bounds[k], bounds[k + 1]) for k in range(world)]


def partition_units(units: np.ndarray, world: int):
    """Split a unit table (one stream, all lists) into per-rank tables along list
    boundaries; each rank's table is rebased so that its byte and integer offsets
    start at the rank's first list. -> [(units_k, byte_lo, byte_hi, int_lo, int_hi)]"""
    assert units.dtype == UNIT_DTYPE
    if len(units) == 0:
        return [(units.copy(), 0, 0, 0, 0) for _ in range(world)]
    first = np.r_[True, units["list"][1:] != units["list"][:-1]]
    list_starts = np.flatnonzero(first)
    list_ints = np.add.reduceat(units["n"].astype(np.uint64), list_starts)
    out = []
    for lo, hi in partition_lists(list_ints, world):
        u_lo = int(list_starts[lo]) if lo < len(list_starts) else len(units)
        u_hi = int(list_starts[hi]) if hi < len(list_starts) else len(units)
        part = units[u_lo:u_hi].copy()
        if len(part):
            byte_lo, int_lo = int(part["in_off"][0]), int(part["out_off"][0])
            int_hi = int(part["out_off"][-1]) + int(part["n"][-1])
            byte_hi = int(units["in_off"][u_hi]) if u_hi < len(units) else None
            part["in_off"] -= np.uint64(byte_lo)
            part["out_off"] -= np.uint64(int_lo)
        else:
            byte_lo = byte_hi = int_lo = int_hi = 0
        out.append((part, byte_lo, byte_hi, int_lo, int_hi))
    return out
