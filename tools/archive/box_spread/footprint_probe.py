#!/usr/bin/env python3
"""Development aid (GPU box): does the 15 % spread between boxes of the pool follow the FOOTPRINT of the
bench's 5e9-integer launch? The same launch — five replicas of the 1e9-posting stream — timed with the replicas
(a) reading and writing their own regions (the bench: 2.8 GB in, 20 GB out), (b) all writing ONE 4 GB output region
(identical values), (c) also all reading ONE copy of the stream. Same process, same box, same number of integers.
usage: tools/footprint_probe.py [postings]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dint_amd import device, host

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
R, kind, dev = 5, host.SINGLE_PACKED, torch.device("cuda", 0)
threads = host.default_threads()
p = host.synth_params(universe=25_000_000, seed=12345)
lens = host.synth_lengths(p, postings)
coll = host.Collection(host.synth_gaps(p, lens, first_list_id=0, threads=threads), lens)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000, threads=threads)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=8192, threads=threads)
d = device.Dictionary(kind, dict_file, device=0)
enc_dev = torch.empty(enc.size * R, dtype=torch.uint8, device=dev)
one = torch.from_numpy(enc).to(dev)
for r in range(R):
    enc_dev[r * enc.size:(r + 1) * enc.size].copy_(one)
del one
n_ints = coll.num_postings * R
out_dev = torch.empty(n_ints, dtype=torch.int32, device=dev)


def table(own_in, own_out):
    u = np.tile(units, R)
    for r in range(R):
        sl = slice(r * len(units), (r + 1) * len(units))
        if own_in:
            u["in_off"][sl] += np.uint64(r * enc.size)
        if own_out:
            u["out_off"][sl] += np.uint64(r * coll.num_postings)
    return device.units_to_device(u, dev), len(u)


res = {"ints_per_launch": n_ints}
for name, own_in, own_out in (("own_regions", True, True), ("one_output_region", True, False), ("one_region_each", False, False),
                              ("own_regions_again", True, True)):
    units_dev, n_units = table(own_in, own_out)
    for _ in range(5):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, None)
    torch.cuda.synchronize()
    for _ in range(20):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, None)
    torch.cuda.synchronize()
    ms = np.asarray(d.recent_kernel_ms(20))
    res[name] = {"kernel_ms_median": round(float(np.median(ms)), 4), "min": round(float(ms.min()), 4)}
    del units_dev
got = out_dev[:coll.num_postings].cpu().numpy().view(np.uint32)
res["bit_exact_first_replica"] = bool(np.array_equal(got, coll.gaps))
res["shader_mhz"] = round(float(d.last_kernel_clock_mhz()), 1)
print(json.dumps(res))
