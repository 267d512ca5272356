#!/usr/bin/env python3
"""Development aid (GPU box): (1) what a plain fill reaches over regions of 0.5 .. 32 GB of one allocation — is the memory
interleaved finely enough that a 4 GB buffer sees the whole HBM? (2) the decode kernel's time against the DISTANCE between
its stream and its output inside one 72 GB allocation. usage: distance_probe.py [postings]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
torch.cuda.init(); dev = torch.device("cuda:0")
GB = 1 << 30
pool = torch.empty(72 * GB, dtype=torch.uint8, device=dev)
print(f"pool at {pool.data_ptr():#x}")
def fill_rate(off_gb, size_gb):
    v = pool[int(off_gb * GB): int((off_gb + size_gb) * GB)].view(torch.int32)
    v.fill_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): v.fill_(2)
    e1.record(); torch.cuda.synchronize()
    return 5 * size_gb * GB / (e0.elapsed_time(e1) * 1e-3) / 1e12
for size in (0.5, 1, 2, 4, 8, 16, 32):
    print(f"fill over {size:4} GB at offset 0: {fill_rate(0, size):.2f} TB/s;  at offset 36 GB: {fill_rate(36, size):.2f} TB/s", flush=True)
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=16384)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
units_dev = device.units_to_device(units, dev)
enc_t = torch.from_numpy(enc)
def run(enc_off_gb, out_off_gb):
    e = pool[int(enc_off_gb * GB): int(enc_off_gb * GB) + enc.size]; e.copy_(enc_t)
    o = pool[int(out_off_gb * GB): int(out_off_gb * GB) + 4 * coll.num_postings].view(torch.int32)
    ms = []
    for _ in range(5):
        d.decode_units(e, units_dev, len(units), o); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    return float(np.median(ms[1:]))
for dist in (1, 2, 4, 6, 8, 10, 12, 16, 24, 32, 48, 64):
    print(f"stream at 0, output {dist:2d} GB above it: {run(0, dist):.4f} ms   |   stream at 68 GB, output {dist:2d} GB below it: {run(68, 68 - dist - 4 if 68 - dist - 4 >= 0 else 0):.4f} ms", flush=True)
