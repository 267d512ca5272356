#!/usr/bin/env python3
"""Block statistics on the device next to the host: dictionary construction from a sample of a
synthetic collection, counting by dint_count_ngrams (device) vs dinth_build_dictionary (host threads); the two
dictionary files must be byte-identical. usage: tools/stats_bench.py [sample ints] [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import device, host

sample = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
coll = host.synth_collection(sample, universe=25_000_000, seed=12345)
res = {"sample_ints": int(coll.num_postings), "lists": int(len(coll.lens)), "host_threads": host.default_threads()}
for typ in ("single_packed_dint", "multi_packed_dint"):
    kind = host.KIND_BY_TYPE[typ]
    device.build_dictionary(kind, coll)  # warm-up (allocations, first launches)
    t0 = time.perf_counter(); dev_file, ms = device.build_dictionary(kind, coll); t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); host_file = host.build_dictionary(kind, coll); t_host = time.perf_counter() - t0
    ngrams = sum(coll.num_postings // k for k in (1, 2, 4, 8, 16))
    res[typ] = {"identical_files": dev_file == host_file, "count_kernel_ms": round(ms, 3),
                "G_ngrams_per_s": round(ngrams / ms / 1e6, 2),
                "device_path_s (upload, count, compact, select + sort on the device, copy back, host packing)": round(t_dev, 3),
                "host_path_s": round(t_host, 3)}
    print(typ, json.dumps(res[typ]), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
