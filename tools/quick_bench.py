"""Scratch: decode rate of the current kernel on a synthetic collection (development aid)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
unit_ints = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
t = time.time()
min_len = int(os.environ.get("MIN_LEN", "1"))
coll = host.synth_collection(postings, universe=25_000_000, seed=12345, min_len=min_len)
print("generated", coll.num_postings, "postings in", len(coll.lens), "lists", round(time.time() - t, 1), "s", flush=True)
t = time.time()
dict_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
print("dictionary", len(dict_file), "B", round(time.time() - t, 1), "s", flush=True)
t = time.time()
enc, units = host.encode_vroom(host.SINGLE_PACKED, dict_file, coll, unit_ints=unit_ints)
print("encoded", enc.size, "B bpi", round(enc.size * 8 / coll.num_postings, 3), "units", len(units), round(time.time() - t, 1), "s", flush=True)
d = device.Dictionary(host.SINGLE_PACKED, dict_file)
info = d.info()
print("hot entries", info.hot_entries, "lds bytes", info.lds_bytes, "CUs", info.compute_units)
dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
def runs(tag, n=5):
    for i in range(n):
        d.decode_units(enc_dev, units_dev, len(units), out_dev)
        torch.cuda.synchronize()
        ms = d.last_kernel_ms()
        gb = (coll.num_postings * 4 + enc.size) / 1e9
        print(f"{tag} run {i}: {ms:.3f} ms  {coll.num_postings / ms / 1e6:.2f} G ints/s  {gb / ms * 1e3:.1f} GB/s algorithmic", flush=True)
runs("decode")
ok = np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)
print("bit-exact vs encoder input:", ok)
