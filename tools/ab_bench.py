#!/usr/bin/env python3
"""Development aid (GPU box): A/B timing of several builds of libdint_hip.so in ONE process on the same
collection, launches interleaved round by round (cross-process variance would otherwise look like a kernel
property). Every build's output is compared bit for bit with the encoder's input.

usage: tools/ab_bench.py [--postings 1e9] [--type single_packed_dint] [--unit-ints 8192] [--rounds 6] [--reps 5]
                         name=path.so [name=path.so ...]
"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--postings", type=float, default=1e9)
ap.add_argument("--type", default="single_packed_dint")
ap.add_argument("--unit-ints", type=int, default=8192)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--json", default=None)
ap.add_argument("--table", action="store_true", help="decode through a prepared unit table (dint_unit_table_create / dint_decode_unit_table)")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()

kind = host.KIND_BY_TYPE[args.type]
t = time.time()
coll = host.synth_collection(int(args.postings), universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=args.unit_ints)
print(f"set-up {time.time() - t:.1f}s: {coll.num_postings} postings, {enc.size} B, {len(units)} units", flush=True)
dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).copy()).to(dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
vp, sz = C.c_void_p, C.c_size_t

builds = []
tables = {}
for spec in args.libs:
    name, path = spec.split("=", 1)
    lib = C.CDLL(os.path.abspath(path))
    lib.dint_dict_create.argtypes = [C.c_int, vp, sz, C.c_int, C.POINTER(vp)]
    lib.dint_decode_units.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.dint_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    h = vp()
    buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
    assert lib.dint_dict_create(kind, C.addressof(buf), len(dict_file), 0, C.byref(h)) == 0, name
    tab = vp()
    if args.table:
        lib.dint_unit_table_create.argtypes = [vp, vp, sz, vp, sz, sz, vp, C.POINTER(vp)]
        lib.dint_decode_unit_table.argtypes = [vp, vp, vp, sz, vp, vp]
        assert lib.dint_unit_table_create(h, enc_dev.data_ptr(), enc.size, units_dev.data_ptr(), len(units), coll.num_postings, stream, C.byref(tab)) == 0
    tables[name] = tab
    builds.append((name, lib, h))


def launch(lib, h, name=None):
    if args.table:
        st = lib.dint_decode_unit_table(h, tables[name], out_dev.data_ptr(), coll.num_postings, end_dev.data_ptr(), stream)
    else:
        st = lib.dint_decode_units(h, enc_dev.data_ptr(), enc.size, units_dev.data_ptr(), len(units), out_dev.data_ptr(),
                                   coll.num_postings, end_dev.data_ptr(), stream)
    assert st == 0, st
    torch.cuda.synchronize(dev)
    ms = C.c_float()
    assert lib.dint_last_kernel_ms(h, C.byref(ms)) == 0
    return ms.value


times = {name: [] for name, _, _ in builds}
for name, lib, h in builds:  # warm-up + correctness
    out_dev.zero_()
    for _ in range(3):
        launch(lib, h, name)
    ok = bool(np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps))
    print(f"{name}: bit-exact {ok}", flush=True)
    assert ok or os.environ.get("AB_NO_ASSERT"), name
for r in range(args.rounds):
    for name, lib, h in builds:
        for _ in range(args.reps):
            times[name].append(launch(lib, h, name))
algo = 4 * coll.num_postings + enc.size  # (headers included: a slight over-count, the same for every build)
res = {}
for name, _, _ in builds:
    v = np.array(times[name])
    res[name] = dict(min=float(v.min()), median=float(np.median(v)), max=float(v.max()),
                     gints=coll.num_postings / np.median(v) / 1e6, frac=algo / (np.median(v) * 1e-3) / 8e12)
    print(f"{name:16s} min {v.min():.4f}  median {np.median(v):.4f}  max {v.max():.4f} ms   "
          f"{res[name]['gints']:.1f} G ints/s  frac {res[name]['frac']:.4f}", flush=True)
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
