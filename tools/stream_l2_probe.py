#!/usr/bin/env python3
"""Development aid (GPU box): what would the decode kernel gain if the compressed stream never came from HBM?

A slice of K consecutive units of an encoded collection is decoded R times into R distinct output regions, (A) from R
physical copies of the slice's bytes (every stream byte read once, from HBM: the ordinary situation) and (B) every
replica from copy 0 (the slice's bytes stay in L2). Same units, same integers, same stores; only where the stream's
reads are served from differs. usage: tools/stream_l2_probe.py [--postings 2e8] [--slice-mb 1.0] [--total 1e9] [name=lib.so ...]
"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--postings", type=float, default=2e8)
ap.add_argument("--slice-mb", type=float, default=1.0)
ap.add_argument("--total", type=float, default=1e9)
ap.add_argument("--unit-ints", type=int, default=16384)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("libs", nargs="*")
args = ap.parse_args()
if not args.libs:
    args.libs = ["product=" + os.path.join(ROOT, "dint_amd", "libdint_hip.so")]

kind = host.SINGLE_PACKED
t = time.time()
coll = host.synth_collection(int(args.postings), universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=args.unit_ints)
print(f"set-up {time.time() - t:.1f}s: {coll.num_postings} postings, {enc.size} B, {len(units)} units", flush=True)

# a slice of consecutive units from the middle of the stream, about slice_mb of stream bytes
k0 = len(units) // 2
b0 = int(units["in_off"][k0])
k1 = k0
while k1 + 1 < len(units) and int(units["in_off"][k1 + 1]) - b0 < args.slice_mb * 1e6:
    k1 += 1
b1 = int(units["in_off"][k1])  # the slice: units [k0, k1), bytes [b0, b1)
sl = units[k0:k1].copy()
o0 = int(sl["out_off"][0])
n_slice = int(sl["out_off"][-1]) + int(sl["n"][-1]) - o0
pad = 4096  # bytes behind a copy (a tile's loads may read a little past a unit's end)
stride = (b1 - b0 + pad + 255) // 256 * 256
R = max(1, int(args.total // n_slice))
print(f"slice: {k1 - k0} units, {b1 - b0} stream bytes, {n_slice} integers; {R} replicas = {R * n_slice} integers", flush=True)
expect = coll.gaps[o0:o0 + n_slice]

enc_rep = np.zeros(R * stride + 64, dtype=np.uint8)
for r in range(R):
    enc_rep[r * stride:r * stride + (b1 - b0) + pad] = enc[b0:b1 + pad] if b1 + pad <= enc.size else np.pad(enc[b0:], (0, b1 + pad - enc.size))[: b1 - b0 + pad]


def table(shared):
    u = np.empty(R * len(sl), dtype=host.UNIT_DTYPE)
    for r in range(R):
        v = u[r * len(sl):(r + 1) * len(sl)]
        v[:] = sl
        v["in_off"] = sl["in_off"] - b0 + (0 if shared else r * stride)
        v["out_off"] = sl["out_off"] - o0 + r * n_slice
    return u


dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc_rep).to(dev)
out_dev = torch.empty(R * n_slice, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
vp, sz = C.c_void_p, C.c_size_t
for spec in args.libs:
    name, path = spec.split("=", 1)
    lib = C.CDLL(os.path.abspath(path))
    lib.dint_dict_create.argtypes = [C.c_int, vp, sz, C.c_int, C.POINTER(vp)]
    lib.dint_decode_units.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.dint_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    h = vp()
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    assert lib.dint_dict_create(kind, C.addressof(buf), len(dict_file), 0, C.byref(h)) == 0, name
    for label, shared in (("A stream from HBM (R copies)", False), ("B stream L2-resident (copy 0)", True)):
        u = table(shared)
        units_dev = torch.from_numpy(np.ascontiguousarray(u).view(np.uint8).copy()).to(dev)
        ts = []
        out_dev.zero_()
        for i in range(args.reps + 3):
            st = lib.dint_decode_units(h, enc_dev.data_ptr(), enc_rep.size, units_dev.data_ptr(), len(u), out_dev.data_ptr(),
                                       R * n_slice, None, stream)
            assert st == 0, st
            torch.cuda.synchronize(dev)
            ms = C.c_float()
            lib.dint_last_kernel_ms(h, C.byref(ms))
            if i >= 3:
                ts.append(ms.value)
        got = out_dev.cpu().numpy().view(np.uint32)
        ok = all(np.array_equal(got[r * n_slice:(r + 1) * n_slice], expect) for r in (0, R // 2, R - 1))
        if hasattr(lib, "dint_debug_read_profile"):  # a -DDINT_PROFILE build: where the waves' cycles went (last launch)
            prof = (C.c_ulonglong * 16)()
            if lib.dint_debug_read_profile(prof) == 0 and sum(prof):
                names = {0: "outside", 1: "classify (tile-top wait)", 2: "sizes + scans", 3: "metas + heads of next tile", 4: "tables",
                         5: "rotate + far prefetch", 7: "tails land", 8: "wait point", 9: "expand + stores", 10: "slow stores",
                         11: "prologue", 12: "bundle front end", 13: "epilogue", 14: "queue ticket"}
                for i in range(16):
                    if prof[i]:
                        print(f"      {i:2d} {names.get(i, '?'):28s} {prof[i] / (R * n_slice) * 900:8.0f} cycles per 900 ints")
        v = np.array(ts)
        print(f"{name:12s} {label:32s} median {np.median(v):.4f} ms  min {v.min():.4f}  {R * n_slice / np.median(v) / 1e6:.1f} G ints/s  bit-exact {ok}",
              flush=True)
