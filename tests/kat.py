"""Loader for tests/golden/kat_vectors.json (hand-assembled known-answer vectors)."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_vectors.json")
with open(_PATH) as f:
    KAT = json.load(f)

DICT_FILES = {
    0: bytes.fromhex(KAT["rectangular_dict"]),
    1: bytes.fromhex(KAT["single_packed_dict"]),
    2: bytes.fromhex(KAT["multi_packed_dict"]),
}


def cases(which):
    """-> [(name, buffer u8[], payload offset, n, expected u32[])]"""
    out = []
    for c in KAT[which]:
        prefix = bytes.fromhex(c["prefix"])
        buf = np.frombuffer(prefix + bytes.fromhex(c["stream"]), dtype=np.uint8).copy()
        out.append((c["name"], buf, len(prefix), c["n"], np.array(c["expect"], dtype=np.uint32)))
    return out
