import os, sys, time
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
coll = host.synth_collection(200_000_000, universe=25_000_000, seed=12345)
kind = host.MULTI_PACKED
d_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
for ui in (256, 1024, 2048, 8192, 32768):
    enc, units = host.encode_vroom(kind, d_file, coll, unit_ints=ui)
    d = device.Dictionary(kind, d_file)
    enc_dev = torch.from_numpy(enc).to(dev); units_dev = device.units_to_device(units, dev)
    out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
    ms = []
    for i in range(6):
        d.decode_units(enc_dev, units_dev, len(units), out_dev); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    ok = np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)
    m = float(np.median(ms[1:]))
    print(f"multi unit_ints {ui:6d}: units {len(units):8d}  {m:.3f} ms  {coll.num_postings / m / 1e6:.1f} G ints/s  ok {ok}", flush=True)
