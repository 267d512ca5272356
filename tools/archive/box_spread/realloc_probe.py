#!/usr/bin/env python3
"""Development aid (GPU box): the decode kernel's time against WHICH memory the driver hands out for the output and the
stream buffer — the same launch into N freshly allocated output buffers (the earlier ones stay allocated), then, into
the fastest of them, from M freshly uploaded copies of the stream. usage: realloc_probe.py [postings] [N] [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
M = int(sys.argv[3]) if len(sys.argv) > 3 else 6
torch.cuda.init(); dev = torch.device("cuda:0")
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=16384)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
units_dev = device.units_to_device(units, dev)
def run(e, o):
    ms = []
    for _ in range(5):
        d.decode_units(e, units_dev, len(units), o); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    return float(np.median(ms[1:]))
enc_dev = torch.from_numpy(enc).to(dev)
outs, t_out = [], []
for i in range(N):
    o = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
    outs.append(o); t_out.append(run(enc_dev, o))
    print(f"output buffer {i:2d} at {o.data_ptr():#x}: {t_out[-1]:.4f} ms", flush=True)
best = int(np.argmin(t_out)); o = outs[best]
print(f"-> output buffer {best}; again: {run(enc_dev, o):.4f} ms")
encs = [enc_dev]
for j in range(M):
    e = torch.from_numpy(enc).to(dev); encs.append(e)
    print(f"stream copy {j:2d} at {e.data_ptr():#x} into output buffer {best}: {run(e, o):.4f} ms", flush=True)
print(f"first stream buffer into output buffer {best} again: {run(enc_dev, o):.4f} ms; into output buffer 0: {run(enc_dev, outs[0]):.4f} ms")
print("bit-exact:", bool(np.array_equal(o.cpu().numpy().view(np.uint32), coll.gaps)))
