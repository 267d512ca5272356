#!/usr/bin/env python3
"""Development aid (GPU box): where the one-launch query form (query_fused_body) spends its time — thread 0's shader clock
at the phase boundaries of every single-query call of the reference's log, from a -DDINT_PROFILE build
(tools/build_variants.sh "qtrace:-DDINT_PROFILE -Itools/variants"). usage: DINT_HIP_LIB=dint_amd/variants/qtrace.so tools/query_trace.py [postings]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from dint_amd import device, host
from queries import reference_queries

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
kind = host.KIND_BY_TYPE["single_packed_dint"]
coll = host.synth_collection(postings, seed=11)
docids = host.gaps_to_docids(coll)
freqs = np.ones(coll.num_postings, dtype=np.uint32)
dd = host.build_dictionary(kind, coll, max_sample_ints=50_000_000)
fd = host.build_dictionary(kind, host.Collection(freqs[:1000] - 1, np.array([1000], dtype=np.uint32)))
idx, offs = host.build_index(kind, dd, fd, docids, freqs, coll.lens)
qi = device.QueryIndex(device.Dictionary(kind, dd), idx, offs)
qs = reference_queries(len(coll.lens))
lib = device._lib
trace = (C.c_ulonglong * 32)()
stream = torch.cuda.current_stream().cuda_stream
rows = []
for rep in range(2):
    for q in qs:
        t = np.ascontiguousarray(q, dtype=np.uint32)
        o = np.array([0, t.size], dtype=np.uint64); c = np.zeros(1, dtype=np.uint64)
        lib.dint_debug_read_query_trace(trace)  # clear
        qi.and_queries_packed(t, o, c, stream)
        lib.dint_debug_read_query_trace(trace)
        v = np.array(trace[:], dtype=np.int64)
        if rep == 1 and v[0] != 0:
            rows.append((len(set(q)), v.copy(), sorted(int(coll.lens[x]) for x in set(q))))
MHZ = float(os.environ.get("QTRACE_MHZ", "2350"))  # s_memtime ticks per microsecond: the shader clock (bench.py's shader_mhz: 2340-2366 under load; approximate here)
print(f"{len(rows)} one-launch queries traced")
for terms in sorted(set(r[0] for r in rows)):
    sel = [r[1] for r in rows if r[0] == terms]
    if terms < 2 or 5 + 4 * (terms - 1) > 31: continue
    n_steps = terms
    tot = np.mean([(v[5 + 4 * (n_steps - 1)] - v[0]) for v in sel]) / MHZ
    print(f"-- {terms} distinct terms: {len(sel)} queries, kernel body {tot:.2f} us (thread 0, entry to the last tail)")
    print(f"   set-up (inputs fetched, image, class table, barrier): {np.mean([v[1] - v[0] for v in sel]) / MHZ:.2f} us")
    for s in range(n_steps):
        b = 2 + 4 * s
        f = lambda i, j: np.mean([v[j] - v[i] for v in sel]) / MHZ
        prev_end = 1 if s == 0 else 5 + 4 * (s - 1)
        print(f"   step {s}: step record {f(prev_end, b):.2f}  pages decoded {f(b, b + 1):.2f}  barrier {f(b + 1, b + 2):.2f}  tail {f(b + 2, b + 3):.2f} us")

# the slowest: which lists, which phase
def body(r):
    v = r[1]; last = max(i for i in range(32) if v[i] != 0)
    return (v[last] - v[0]) / MHZ
print("-- the slowest one-launch queries: list lengths | us from entry at every mark (set-up end, then per step: record read, pages decoded, barrier, tail)")
for r in sorted(rows, key=body, reverse=True)[:10]:
    v = r[1]; marks = [round((v[i] - v[0]) / MHZ, 1) for i in range(1, 32) if v[i] != 0]
    print(f"   {body(r):6.1f} us  lists {r[2]}  marks {marks}")
