#!/usr/bin/env python3
"""Development aid (GPU box): per-section wave cycles of the decode kernel, from a -DDINT_PROFILE build
(tools/variants/dint_profile.hpp). usage: tools/section_profile.py lib.so [postings] [type] [unit_ints]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host
lib_path = sys.argv[1]
postings = int(float(sys.argv[2])) if len(sys.argv) > 2 else 400_000_000
typ = sys.argv[3] if len(sys.argv) > 3 else "single_packed_dint"
unit_ints = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
kind = host.KIND_BY_TYPE[typ]
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
dict_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=unit_ints)
dev = torch.device("cuda:0")
enc_dev = torch.from_numpy(enc).to(dev)
units_dev = torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).copy()).to(dev)
out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
vp, sz = C.c_void_p, C.c_size_t
lib = C.CDLL(os.path.abspath(lib_path))
lib.dint_dict_create.argtypes = [C.c_int, vp, sz, C.c_int, C.POINTER(vp)]
lib.dint_decode_units.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
lib.dint_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
h = vp()
buf = (C.c_char * len(dict_file)).from_buffer_copy(dict_file)
assert lib.dint_dict_create(kind, C.addressof(buf), len(dict_file), 0, C.byref(h)) == 0
prof = (C.c_ulonglong * 16)()
WAVES = int(os.environ.get("WAVES", "4096"))
stream = torch.cuda.current_stream(dev).cuda_stream
names = {0: "outside (queue, exit)", 1: "classify", 2: "sizes + scans", 3: "metas + heads of the next tile", 4: "flag/delta/rank tables",
         5: "rotate + far prefetch", 7: "tails land", 8: "wait point + heads land + tails requested", 9: "expand + stores", 10: "slow stores", 11: "segment prologue",
         12: "bundle front end", 13: "epilogue", 14: "queue ticket", 15: "rows of the next tile"}
if "multi" in typ or unit_ints <= 256:  # bundle_process's own marks (bundles.inc)
    names.update({1: "bundle: classify", 3: "bundle: metas + literals", 2: "bundle: sizes + scans + cells", 11: "bundle: cells, tails asked", 15: "bundle: next bundle mapped + asked"})
for it in range(4):
    assert lib.dint_decode_units(h, enc_dev.data_ptr(), enc.size, units_dev.data_ptr(), len(units), out_dev.data_ptr(),
                                 coll.num_postings, None, stream) == 0
    torch.cuda.synchronize(dev)
    ms = C.c_float(); lib.dint_last_kernel_ms(h, C.byref(ms))
    assert lib.dint_debug_read_profile(prof) == 0
    if it < 2: continue
    tot = sum(prof)
    print(f"launch {it}: kernel {ms.value:.3f} ms, {tot / WAVES / 1e3:.0f}k cycles per wave "
          f"(= {tot / WAVES / ms.value / 1e3:.0f} MHz if every wave lived the whole kernel)")
    for i in range(16):
        if prof[i]:
            print(f"  {i:2d} {names.get(i, '?'):34s} {100.0 * prof[i] / tot:5.1f} %   {prof[i] / coll.num_postings * 900:8.0f} cycles per 900 ints")
print("bit-exact:", bool(np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)))
