#!/usr/bin/env python3
"""Writes tests/golden/encoder_vectors.json: for the hand-assembled KAT dictionaries (kat_vectors.json) and the KATs' integer
sequences, the bytes the ORACLE's encoder restatement emits (oracle/dint_oracle_encode.c: single_opt_dint, single_greedy_dint,
multi_opt_dint — vroom_env/dint_codecs.hpp:110-518) and the bytes of dict_posting_list::write for a few small lists.
NOT reference outputs (the reference cannot be built here: oracle/dint_oracle.h): a regression pin — the oracle and the
product must both keep emitting exactly these bytes (tests/test_encoder_oracle_cpu.py). usage: python tests/golden/make_encoder_vectors.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import kat, oracle

out = {"about": "oracle encoder outputs over the KAT dictionaries; see make_encoder_vectors.py. NOT reference outputs.", "lists": [], "posting_lists": []}
for which, kind in (("single_cases", 0), ("single_cases", 1), ("multi_cases", 2)):
    b = oracle.OracleBuilder(kind, kat.DICT_FILES[kind])
    for name, _buf, _off, n, expect in kat.cases(which):
        for greedy in ((False, True) if kind != 2 else (False,)):
            payload = b.encode_list(expect, greedy=greedy).tobytes()
            # (the gaps are the KAT's expected integers: kat_vectors.json, by case name; long payloads by their hash)
            rec = {"kind": kind, "which": which, "case": name, "greedy": greedy, "bytes": len(payload)}
            rec.update({"payload": payload.hex()} if len(payload) <= 512 else {"payload_sha256": hashlib.sha256(payload).hexdigest()})
            out["lists"].append(rec)
r = np.random.default_rng(17)
for kind in (1, 2):
    b = oracle.OracleBuilder(kind, kat.DICT_FILES[kind])
    for n in (1, 5, 256, 300):
        gaps = r.integers(0, 4, n).astype(np.uint32)
        gaps[r.integers(0, n, max(1, n // 50))] = r.integers(0, 100000, max(1, n // 50)).astype(np.uint32)
        docs = (np.cumsum(gaps.astype(np.uint64) + 1) - 1).astype(np.uint32)
        freqs = r.integers(1, 5, n).astype(np.uint32)
        out["posting_lists"].append({"kind": kind, "docids": [int(x) for x in docs], "freqs": [int(x) for x in freqs],
                                     "bytes": oracle.posting_list_write(b, b, docs, freqs).tobytes().hex()})
json.dump(out, open(os.path.join(HERE, "encoder_vectors.json"), "w"))
print(len(out["lists"]), "lists,", len(out["posting_lists"]), "posting lists")
