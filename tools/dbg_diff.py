#!/usr/bin/env python3
"""Development aid (GPU box): where a build's output differs from the encoder's input.
usage: tools/dbg_diff.py lib.so [postings] [type] [unit_ints]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host
lib_path = sys.argv[1]
postings = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
typ = sys.argv[3] if len(sys.argv) > 3 else "single_packed_dint"
unit_ints = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
kind = host.KIND_BY_TYPE[typ]
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=unit_ints)
dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).copy()).to(dev)
out_dev = torch.full((coll.num_postings,), -1, dtype=torch.int32, device=dev)
vp, sz = C.c_void_p, C.c_size_t
lib = C.CDLL(os.path.abspath(lib_path))
lib.dint_dict_create.argtypes = [C.c_int, vp, sz, C.c_int, C.POINTER(vp)]
lib.dint_decode_units.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
h = vp()
buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
assert lib.dint_dict_create(kind, C.addressof(buf), len(dict_file), 0, C.byref(h)) == 0
stream = torch.cuda.current_stream(dev).cuda_stream
for it in range(int(os.environ.get("LAUNCHES", "3"))):
    out_dev.fill_(-1)
    assert lib.dint_decode_units(h, enc_dev.data_ptr(), enc.size, units_dev.data_ptr(), len(units), out_dev.data_ptr(),
                                 coll.num_postings, None, stream) == 0
    torch.cuda.synchronize(dev)
    got = out_dev.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != coll.gaps)[0]
    print(f"launch {it}: {bad.size} of {got.size} differ")
    if bad.size:
        uo = units["out_off"].astype(np.int64)
        un = units["n"].astype(np.int64)
        which = np.searchsorted(uo, bad, side="right") - 1
        uniq, first = np.unique(which, return_index=True)
        print(f"  units affected: {uniq.size} of {len(units)}")
        for u, f in list(zip(uniq, first))[:12]:
            i = bad[f]
            in_unit = bad[which == u] - uo[u]
            print(f"  unit {u}: n={un[u]} first bad at {i - uo[u]} (last {in_unit.max()}, {in_unit.size} bad); "
                  f"want {coll.gaps[i:i+6]} got {got[i:i+6]}")
            if os.environ.get("BAD_INDICES"):
                print("     bad offsets:", list(in_unit[:80]), " unit in_off", int(units["in_off"][u]), "out_off", int(uo[u]))
