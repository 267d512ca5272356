"""Development aid: does the kernel time depend on where the allocator happens to put things?
One process, one collection; re-create the dictionary / the output buffer / the stream buffer in turn
and time the same decode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 800_000_000
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=8192)
def run(d, enc_dev, units_dev, out_dev, tag):
    ms = []
    for _ in range(6):
        d.decode_units(enc_dev, units_dev, len(units), out_dev); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    print(f"{tag:28s} kernel ms {np.median(ms[1:]):.4f}  enc {enc_dev.data_ptr():#x} out {out_dev.data_ptr():#x}", flush=True)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
enc_dev = torch.from_numpy(enc).to(dev); units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
run(d, enc_dev, units_dev, out_dev, "initial")
keep = []
for i in range(4):
    keep.append(d); d = device.Dictionary(host.SINGLE_PACKED, d_file)   # new tables, old ones stay allocated
    run(d, enc_dev, units_dev, out_dev, f"new dictionary #{i}")
for i in range(4):
    keep.append(out_dev); out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
    run(d, enc_dev, units_dev, out_dev, f"new output buffer #{i}")
for i in range(3):
    keep.append(enc_dev); enc_dev = torch.from_numpy(enc).to(dev)
    run(d, enc_dev, units_dev, out_dev, f"new stream buffer #{i}")
