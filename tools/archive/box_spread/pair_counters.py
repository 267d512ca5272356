#!/usr/bin/env python3
"""Development aid (GPU box, under rocprofv3 --pmc): the decode kernel into a SLOW and into a FAST output buffer of one
process, alternately — which counters move with the placement? Finds the two among 8 fresh output buffers (untimed by the
profiler's reader: the last 8 decode launches of the process are slow, fast, slow, fast, ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
coll = host.synth_collection(1_000_000_000, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=16384)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
units_dev = device.units_to_device(units, dev)
def run(e, o):
    d.decode_units(e, units_dev, len(units), o); torch.cuda.synchronize(); return d.last_kernel_ms()
enc_dev = torch.from_numpy(enc).to(dev)
outs = [torch.empty(coll.num_postings, dtype=torch.int32, device=dev) for _ in range(8)]
t = [min(run(enc_dev, o) for _ in range(3)) for o in outs]
slow, fast = int(np.argmax(t)), int(np.argmin(t))
print("candidates ms:", [round(x, 3) for x in t], "slow", slow, "fast", fast, flush=True)
for _ in range(4):
    print(f"slow {run(enc_dev, outs[slow]):.4f}  fast {run(enc_dev, outs[fast]):.4f}", flush=True)
