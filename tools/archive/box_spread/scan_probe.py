#!/usr/bin/env python3
"""Development aid (GPU box): the decode kernel's time against the offset of its output inside one 80 GB allocation, 1 GB
steps, for three positions of the stream — is there a rule (a period, a boundary) to where a pair of buffers is slow?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init(); dev = torch.device("cuda:0")
GB = 1 << 30
POOL = int(os.environ.get("POOL_GB", "80")); STEP = int(os.environ.get("STEP_GB", "1"))
pool = torch.empty(POOL * GB, dtype=torch.uint8, device=dev)
print(f"pool at {pool.data_ptr():#x} ({pool.data_ptr() / GB % 64:.2f} GB into its 64 GB-aligned window)")
coll = host.synth_collection(1_000_000_000, universe=25_000_000, seed=12345)
d_file = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d_file, coll, unit_ints=16384)
d = device.Dictionary(host.SINGLE_PACKED, d_file)
units_dev = device.units_to_device(units, dev)
enc_t = torch.from_numpy(enc)
def run(e, o):
    ms = []
    for _ in range(3):
        d.decode_units(e, units_dev, len(units), o); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    return min(ms[1:])
for enc_off in [int(x) for x in os.environ.get("STREAM_AT", "0,5,40").split(",")]:
    e = pool[enc_off * GB: enc_off * GB + enc.size]; e.copy_(enc_t)
    row = []
    for out_off in range(0, POOL - 4, STEP):
        if abs(out_off - enc_off) < 1 or (out_off < enc_off < out_off + 4): row.append("  -  "); continue
        o = pool[out_off * GB: out_off * GB + 4 * coll.num_postings].view(torch.int32)
        row.append(f"{run(e, o):.2f}")
    print(f"stream at +{enc_off:3d} GB; output at +0, +{STEP}, ... GB: " + " ".join(row), flush=True)
# Is it the decode kernel, or any kernel that reads one buffer and writes another? A plain copy of 4 GB from the first stream
# position's stretch into the slowest and into the fastest output position found for it.
enc_off = int(os.environ.get("STREAM_AT", "0,5,40").split(",")[0])
e = pool[enc_off * GB: enc_off * GB + enc.size]; e.copy_(enc_t)
times = {}
for out_off in range(0, POOL - 4, STEP):
    if abs(out_off - enc_off) < 5: continue
    o = pool[out_off * GB: out_off * GB + 4 * coll.num_postings].view(torch.int32)
    times[out_off] = run(e, o)
slow, fast = max(times, key=times.get), min(times, key=times.get)
src = pool[enc_off * GB: (enc_off + 4) * GB]
def copy_rate(dst_off):
    dst = pool[dst_off * GB: (dst_off + 4) * GB]
    dst.copy_(src); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    return 10 * 8 * GB / (e0.elapsed_time(e1) * 1e-3) / 1e12
print(f"decode: output at +{slow} GB {times[slow]:.3f} ms (slow), at +{fast} GB {times[fast]:.3f} ms (fast); a plain 4 GB copy from +{enc_off} GB "
      f"(read + write): into +{slow} GB {copy_rate(slow):.2f} TB/s, into +{fast} GB {copy_rate(fast):.2f} TB/s, again {copy_rate(slow):.2f} / {copy_rate(fast):.2f}")
