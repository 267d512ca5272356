#!/usr/bin/env python3
"""Development aid (GPU box): the decode kernel's time as a function of the sidecar's granularity (unit_ints): one
encoded stream, re-indexed by dint_index_stream at every size, decoded through a prepared unit table.
usage: tools/unit_sweep.py [postings] [type] [sizes,comma,separated] [replicate]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
typ = sys.argv[2] if len(sys.argv) > 2 else "single_packed_dint"
sizes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4096, 8192, 16384, 32768, 65536, 131072]
R = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kind = host.KIND_BY_TYPE[typ]
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, _ = host.encode_vroom(kind, dict_file, coll, unit_ints=8192)
d = device.Dictionary(kind, dict_file)
dev = torch.device("cuda:0")
enc_dev = torch.empty(enc.size * R, dtype=torch.uint8, device=dev)
one = torch.from_numpy(enc).to(dev)
for r in range(R):
    enc_dev[r * enc.size:(r + 1) * enc.size].copy_(one)
out_dev = torch.empty(coll.num_postings * R, dtype=torch.int32, device=dev)
for ui in sizes:
    units, total, _ = d.index_stream(enc, ui)
    ua = np.tile(units, R)
    for r in range(R):
        sl = slice(r * len(units), (r + 1) * len(units))
        ua["in_off"][sl] += np.uint64(r * enc.size)
        ua["out_off"][sl] += np.uint64(r * coll.num_postings)
    units_dev = device.units_to_device(ua, dev)
    table = device.UnitTable(d, enc_dev, units_dev, len(ua), out_dev.numel())
    out_dev.zero_()
    for _ in range(3):
        table.decode(out_dev)
    torch.cuda.synchronize()
    for _ in range(8):
        table.decode(out_dev)
    torch.cuda.synchronize()
    ms = d.recent_kernel_ms(8)
    ok = all(np.array_equal(out_dev[r * coll.num_postings:(r + 1) * coll.num_postings].cpu().numpy().view(np.uint32), coll.gaps) for r in range(R))
    print(f"unit_ints {ui:7d}: {len(ua):8d} units, kernel {ms.mean():.4f} ms (min {ms.min():.4f}), "
          f"{coll.num_postings * R / ms.mean() / 1e6:.1f} G ints/s, bit-exact {ok}", flush=True)
    table.close()
