"""Development aid: kernel time vs the byte offset of the stream (and of the output) inside one allocation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 800_000_000
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=8192)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
units_dev = device.units_to_device(units, dev)
PAD = 64 << 20
big_enc = torch.empty(enc.size + PAD, dtype=torch.uint8, device=dev)
big_out = torch.empty(coll.num_postings + PAD // 4, dtype=torch.int32, device=dev)
enc_t = torch.from_numpy(enc)
def run(eo, oo):
    e = big_enc[eo:eo + enc.size]; e.copy_(enc_t)
    o = big_out[oo // 4: oo // 4 + coll.num_postings]
    ms = []
    for _ in range(5):
        d.decode_units(e, units_dev, len(units), o); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    return float(np.median(ms[1:]))
print("base", run(0, 0), flush=True)
for eo in (256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 16 << 20, 33 << 20):
    print(f"enc offset {eo:>9d}: {run(eo, 0):.4f} ms", flush=True)
for oo in (4096, 65536, 1 << 20, 2 << 20, 16 << 20, 33 << 20):
    print(f"out offset {oo:>9d}: {run(0, oo):.4f} ms", flush=True)
