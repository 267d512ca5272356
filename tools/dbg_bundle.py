import sys, os
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import torch; torch.cuda.init()
from dint_amd import host, device
for kind in (host.SINGLE_PACKED, host.RECTANGULAR):
  for seed,post in ((7,400_000),(3,50_000),(11,2_000_000)):
    coll = host.synth_collection(post, universe=200_000, seed=seed)
    d = host.build_dictionary(kind, coll)
    for ui in (1024, 64, 8192):
        enc, units = host.encode_vroom(kind, d, coll, unit_ints=ui)
        dd = device.Dictionary(kind, d)
        u2, total, nl = dd.index_stream(enc, ui)
        out, ends, ms = device.decode_stream(dd, enc, u2, total)
        ok = np.array_equal(out, coll.gaps)
        exp_ends = np.r_[u2["in_off"][1:], 0]
        print(kind, seed, post, ui, "units", len(u2), "ok", ok, flush=True)
        if not ok:
            bad = np.nonzero(out != coll.gaps)[0]
            print("  first bad", bad[:5], "of", len(bad)); 
            # which unit
            uo = u2["out_off"]; k = np.searchsorted(uo, bad[0], side="right")-1
            print("  unit", k, u2[k], "next", u2[k+1] if k+1<len(u2) else None)
            print("  units around:", u2[max(0,k-1):k+8])
            lo=int(bad[0]); print("  got ", out[lo-2:lo+24]); print("  want", coll.gaps[lo-2:lo+24])
            sys.exit(1)
print("all ok")
