#!/usr/bin/env python3
"""Development aid (GPU box): the decode kernel's time launch by launch over a long back-to-back run — does it
hold its speed? usage: tools/sustained.py [postings] [launches] [idle seconds between bursts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import device, host
postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
kind = host.SINGLE_PACKED
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=8192)
d = device.Dictionary(kind, dict_file)
dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
times, clocks = [], []
t0 = time.perf_counter()
for i in range(launches):
    d.decode_units(enc_dev, units_dev, len(units), out_dev)
    if i % 32 == 31:  # (64 event slots: read them back in time)
        times += list(d.recent_kernel_ms(32))
        clocks.append(d.last_kernel_clock_mhz())
torch.cuda.synchronize()
wall = time.perf_counter() - t0
t = np.array(times)
print(f"{launches} launches in {wall:.2f} s wall; kernel ms by group of 32 launches (and the shader clock of each group's last launch):")
for g in range(len(t) // 32):
    print(f"  launches {32 * g:5d}-{32 * g + 31:5d} (t = {t[:32 * g].sum() / 1e3:6.2f} s): median {np.median(t[32 * g:32 * g + 32]):.4f} ms, {clocks[g]:.0f} MHz")
