#!/usr/bin/env python3
"""Development aid (GPU box): does the decode kernel's time change over MINUTES inside one process? One collection, one
launch every few seconds for a while, the kernel time of each printed. usage: timeline.py [seconds] [period] [postings]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
period = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
P = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000_000
torch.cuda.init(); dev = torch.device("cuda:0")
# BALLAST_GB: that much device memory allocated (and held) before anything else — does the decode's time depend on WHICH
# memory its buffers get? (tools/box_spread/realloc_probe.py: the first ~8 GB a process is handed are slow to write to)
ballast = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(int(os.environ.get("BALLAST_GB", "0")))]
coll = host.synth_collection(P, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=16384)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
enc_dev = torch.from_numpy(enc).to(dev); units_dev = device.units_to_device(units, dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
t0 = time.time()
line = []
while time.time() - t0 < seconds:
    ms = []
    for _ in range(4):
        d.decode_units(enc_dev, units_dev, len(units), out_dev); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    line.append(f"{time.time() - t0:5.0f}s {min(ms[1:]):.3f}")
    time.sleep(period)
print(f"pid {os.getpid()} ballast {len(ballast)} GB out at {out_dev.data_ptr():#x}: kernel ms (best of 3) over time: " + " | ".join(line), flush=True)
