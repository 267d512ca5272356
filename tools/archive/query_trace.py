#!/usr/bin/env python3
"""Development aid (GPU box): N single-query calls and nothing else, for rocprofv3 --kernel-trace --stats
(what the device does for one query, launch by launch). usage: tools/query_trace.py [postings] [queries] [batch]
(batch: the queries in ONE call, ten times, instead)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from dint_amd import device, host
from queries import reference_queries
postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kind = host.SINGLE_PACKED
coll = host.synth_collection(postings, seed=11)
docids = host.gaps_to_docids(coll)
freqs = np.ones(coll.num_postings, dtype=np.uint32)
dd = host.build_dictionary(kind, coll, max_sample_ints=50_000_000)
fd = host.build_dictionary(kind, host.Collection(freqs[:1000] - 1, np.array([1000], dtype=np.uint32)))
idx, offs = host.build_index(kind, dd, fd, docids, freqs, coll.lens)
qi = device.QueryIndex(device.Dictionary(kind, dd), idx, offs)
qs = reference_queries(len(coll.lens))[:nq]
packed = [(np.ascontiguousarray(q, dtype=np.uint32), np.array([0, len(q)], dtype=np.uint64), np.zeros(1, dtype=np.uint64)) for q in qs]
stream = torch.cuda.current_stream().cuda_stream
if len(sys.argv) > 3 and sys.argv[3] == "batch":
    qi.and_queries(qs)
    t0 = time.perf_counter()
    for _ in range(10):
        qi.and_queries(qs)
    print(f"{(time.perf_counter() - t0) / 10 / len(qs) * 1e6:.2f} us per query in a batch of {len(qs)}")
    sys.exit(0)
for t, o, c in packed:
    qi.and_queries_packed(t, o, c, stream)
t0 = time.perf_counter()
for t, o, c in packed:
    qi.and_queries_packed(t, o, c, stream)
print(f"{(time.perf_counter() - t0) / len(packed) * 1e6:.1f} us per query, {sum(len(q) for q in qs) / len(qs):.2f} terms per query")
