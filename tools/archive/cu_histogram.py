#!/usr/bin/env python3
"""Development aid (GPU box): how evenly does the dynamic work queue spread the decode over the chip? From a
-DDINT_PROFILE build (tools/variants/dint_profile.hpp): per compute unit the waves that ran there, their lifetime,
the cycles they spent decoding, the work items they drew. usage: tools/cu_histogram.py lib.so [postings] [unit_ints]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host
lib_path = sys.argv[1]
postings = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
unit_ints = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
kind = host.KIND_BY_TYPE["single_packed_dint"]
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
stream = torch.cuda.current_stream(dev).cuda_stream
cu = (C.c_ulonglong * 8192)()
for it in range(3):
    assert lib.dint_decode_units(h, enc_dev.data_ptr(), enc.size, units_dev.data_ptr(), len(units), out_dev.data_ptr(),
                                 coll.num_postings, None, stream) == 0
    torch.cuda.synchronize(dev)
    ms = C.c_float(); lib.dint_last_kernel_ms(h, C.byref(ms))
    assert lib.dint_debug_read_cu_profile(cu) == 0
a = np.array(cu[:], dtype=np.float64).reshape(8, 256, 4)
used = a[:, :, 0] > 0
print(f"{coll.num_postings} integers, {len(units)} units of <= {unit_ints}, kernel {ms.value:.3f} ms (profile build), "
      f"{int(used.sum())} compute units ran waves")
print("per XCC: CUs  waves  items  decoding cycles per CU (min / mean / max, relative to the chip's mean)  lifetime (same)")
busy_mean, life_mean = a[:, :, 2][used].mean(), a[:, :, 1][used].mean()
for x in range(8):
    u = used[x]
    b, l = a[x, :, 2][u] / busy_mean, a[x, :, 1][u] / life_mean
    print(f"  xcc {x}: {int(u.sum()):3d} {int(a[x, :, 0].sum()):6d} {int(a[x, :, 3].sum()):7d}   "
          f"{b.min():.3f} / {b.mean():.3f} / {b.max():.3f}      {l.min():.3f} / {l.mean():.3f} / {l.max():.3f}")
items = a[:, :, 3][used]
busy = a[:, :, 2][used] / busy_mean
print(f"work items per CU: min {int(items.min())}  mean {items.mean():.1f}  max {int(items.max())}")
hist, edges = np.histogram(busy, bins=10)
print("decoding cycles per CU / chip mean, histogram:")
for c, e0, e1 in zip(hist, edges[:-1], edges[1:]):
    print(f"  {e0:.3f} - {e1:.3f}: {c:3d} " + "#" * int(c))
idle = 1.0 - a[:, :, 2][used] / a[:, :, 1][used]
print(f"share of a wave's lifetime outside the decode sections (queue draws, start-up, waiting to exit): "
      f"min {idle.min():.3f} mean {idle.mean():.3f} max {idle.max():.3f}")
print("bit-exact:", bool(np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)))
