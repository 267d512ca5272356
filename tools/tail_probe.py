"""Development aid: how long do the waves of the decode kernel wait for the last ones? (build with -DDINT_EXP_FINISH)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ui = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=ui)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
enc_dev = torch.from_numpy(enc).to(dev); units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
for _ in range(4):
    d.decode_units(enc_dev, units_dev, len(units), out_dev); torch.cuda.synchronize()
    ms = d.last_kernel_ms()
    buf = (C.c_ulonglong * 8192)()
    assert device._lib.dint_debug_read_finish(buf) == 0
    t = np.array(buf[:4096], dtype=np.float64)
    t = (t - t.min()) / 100.0  # us (100 MHz)
    end = t.max()
    print(f"kernel {ms:.3f} ms; wave finish times relative to the first finisher: mean {t.mean():.1f} us, p50 {np.median(t):.1f}, p90 {np.percentile(t, 90):.1f}, max {end:.1f} us; idle before the end: mean {(end - t).mean():.1f} us = {(end - t).mean() / (ms * 1e3) * 100:.1f} % of the kernel", flush=True)
