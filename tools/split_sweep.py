#!/usr/bin/env python3
"""Development aid (GPU box): the bundle path's ticket size (dint_set_option chunk_split: 1/2^n of a 64-unit chunk per ticket)
against the kernel's time, one process, one prepared unit table, rounds interleaved.
usage: tools/split_sweep.py [--postings 4e8] [--type multi_packed_dint] [--unit-ints 256] [--splits -1,0,2,3,4]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

ap = argparse.ArgumentParser()
ap.add_argument("--postings", type=float, default=4e8)
ap.add_argument("--type", default="multi_packed_dint")
ap.add_argument("--unit-ints", type=int, default=256)
ap.add_argument("--splits", default="-1,0,2,3,4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()
kind = host.KIND_BY_TYPE[args.type]
t = time.time()
coll = host.synth_collection(int(args.postings), universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=args.unit_ints)
print(f"set-up {time.time() - t:.1f}s: {coll.num_postings} postings, {enc.size} B, {len(units)} units", flush=True)
dev = torch.device("cuda:0")
d = device.Dictionary(kind, dict_file)
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
table = device.UnitTable(d, enc_dev, units_dev, len(units), coll.num_postings)
splits = [int(x) for x in args.splits.split(",")]
ms = {s: [] for s in splits}
for rnd in range(args.rounds + 1):
    for s in splits:
        device.set_option("chunk_split", s)
        for _ in range(args.reps):
            table.decode(out_dev, None)
            torch.cuda.synchronize()
            if rnd:
                ms[s].append(d.last_kernel_ms())
    if rnd == 0:
        assert np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)
device.reset_options()
for s in splits:
    v = np.array(ms[s])
    print(f"chunk_split {s:2d}: min {v.min():.4f} median {np.median(v):.4f} max {v.max():.4f} ms   {coll.num_postings / np.median(v) / 1e6:.1f} G ints/s")
