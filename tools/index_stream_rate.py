#!/usr/bin/env python3
"""Development aid (GPU box: the call needs a dictionary handle): seconds per GB of dint_index_stream — the host pre-pass a
reference-produced vroom file needs before its first decode (the stream carries no sync points) — per dictionary type.
usage: tools/index_stream_rate.py [postings] [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
res = {"postings": int(coll.num_postings), "note": "one host thread (the pre-pass is a sequential parse: a slot's meaning depends on its predecessors)"}
for typ in ("single_packed_dint", "multi_packed_dint"):
    kind = host.KIND_BY_TYPE[typ]
    dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
    enc, _ = host.encode_vroom(kind, dict_file, coll, unit_ints=16384)
    d = device.Dictionary(kind, dict_file)
    best = 1e9
    for unit_ints in (16384, 256):
        for _ in range(2):
            t = time.perf_counter()
            units, total, lists = d.index_stream(enc, unit_ints)
            best_u = time.perf_counter() - t
            best = min(best, best_u)
        res[f"{typ}/unit_ints_{unit_ints}"] = {"stream_bytes": int(enc.size), "seconds": round(best_u, 3),
                                              "seconds_per_GB": round(best_u / (enc.size / 1e9), 3),
                                              "G_ints_per_s": round(total / best_u / 1e9, 3), "units": int(len(units))}
    print(typ, json.dumps({k: v for k, v in res.items() if k.startswith(typ)}), flush=True)
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
