#!/usr/bin/env python3
"""Development aid (GPU box): what dint_unit_table_create costs — the schedule kernels and, for a multi-dictionary table of
units of several blocks, the walk that finds the blocks (DINT_OPT_REFINE_UNITS) — beside what a decode of the table takes.
usage: tools/table_create_cost.py [postings=1e9] [unit_ints=256,16384,65536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256,16384,65536").split(",")]
kind = host.MULTI_PACKED
coll = host.synth_collection(N, universe=25_000_000, seed=12345)
df = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
dev = torch.device("cuda:0")
d = device.Dictionary(kind, df)
for unit_ints in sizes:
    enc, units = host.encode_vroom(kind, df, coll, unit_ints=unit_ints)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
    for refine in (1, 0):
        device.set_option("refine_units", refine)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            table = device.UnitTable(d, enc_dev, units_dev, len(units), coll.num_postings)
            ts.append((time.perf_counter() - t) * 1e3)
            if rep != 2:
                table.close()
        ms = []
        for rep in range(5):
            table.decode(out_dev)
            torch.cuda.synchronize()
            ms.append(d.last_kernel_ms())
        table.close()
        print(f"{N} postings, units of {unit_ints} ({len(units)}), refine_units {refine}: create {min(ts):.1f} ms (first {ts[0]:.1f}), decode {sorted(ms)[2]:.3f} ms", flush=True)
    device.reset_options()
    del enc_dev, units_dev, out_dev
