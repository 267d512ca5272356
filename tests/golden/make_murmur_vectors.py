#!/usr/bin/env python3
"""Generates tests/golden/murmur_vectors.json from the REFERENCE's own hash function.

Needs oracle/_ref/libref_hash.so, i.e. /root/reference must be present: `make -C oracle ref`
compiles /root/reference/include/dint/hash_utils.hpp where it lies (it is the one reference source
on this path that builds without any absent dependency). The vectors (inputs + outputs, data only)
travel; the reference does not.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_hash.so"))
lib.ref_hash_u32s.restype = C.c_uint64
lib.ref_hash_u32s.argtypes = [C.c_void_p, C.c_ulong]

r = np.random.default_rng(20240601)
vectors = []
for n in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 64, 128, 256]:
    for variant in range(5):
        if variant == 0:
            w = np.zeros(n, dtype=np.uint32)           # the run keys of prepare_for_encoding
        elif variant == 1:
            w = np.arange(n, dtype=np.uint32)
        else:
            w = r.integers(0, 2 ** (4 + 9 * variant), n, dtype=np.uint64).astype(np.uint32)
        vectors.append({"words": w.tolist(), "hash": "%016x" % lib.ref_hash_u32s(w.ctypes.data, n)})
with open(os.path.join(HERE, "murmur_vectors.json"), "w") as f:
    json.dump({"about": "MurmurHash64A(seed 0) of u32 words, computed by /root/reference/include/dint/hash_utils.hpp",
               "vectors": vectors}, f, separators=(",", ":"))
print("wrote", len(vectors), "vectors")
