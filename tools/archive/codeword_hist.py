"""Development aid: codeword frequency coverage of a synthetic collection vs hot-set size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dint_amd import host

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
d = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.SINGLE_PACKED, d, coll, unit_ints=0)
print("bpi", enc.size * 8 / coll.num_postings, "lists", len(units))
n_off = np.frombuffer(d[4:8], dtype=np.uint32)[0]
offs = np.frombuffer(d[12:12 + 4 * n_off], dtype=np.uint32)
sizes = (offs >> 24) + 1
hist = np.zeros(65536, dtype=np.int64)
ints_by = np.zeros(65536, dtype=np.int64)
exc = 0
for u in units[:: max(1, len(units) // 400)]:
    p = int(u['in_off']); n = int(u['n']); i = 0
    while i < n:
        idx = int(enc[p]) | (int(enc[p + 1]) << 8)
        if idx >= 2:
            hist[idx] += 1; s = int(sizes[idx]); ints_by[idx] += min(s, n - i); i += s; p += 2
        elif idx == 1:
            exc += 1; i += 1; p += 6
        else:
            exc += 1; i += 1; p += 4
tot = hist.sum()
print("codewords", tot, "exceptions", exc, "ints/codeword", ints_by.sum() / tot)
c = np.cumsum(hist) / tot
ci = np.cumsum(ints_by) / ints_by.sum()
for K in (7, 256, 1024, 2048, 4096, 7268, 10000, 16384, 24000, 32768, 50000, 65536):
    print(K, "codeword cov %.4f" % c[K - 1], "ints cov %.4f" % ci[K - 1])
print("run codewords share", hist[2:7].sum() / tot, "ints share", ints_by[2:7].sum() / ints_by.sum())
tab_n = np.frombuffer(d[8:12], dtype=np.uint32)[0]
table = np.frombuffer(d[12 + 4 * n_off:12 + 4 * n_off + 4 * tab_n], dtype=np.uint32)
o = offs & 0xFFFFFF
maxv = np.array([table[o[i]:o[i] + sizes[i]].max() if sizes[i] <= 16 else 0 for i in range(n_off)])
for K in (7268, 16384, 32768, 65536):
    m = maxv[7:K]
    print("K", K, "entries max<256: %.3f  <65536: %.3f" % ((m < 256).mean(), (m < 65536).mean()), "avg size", sizes[7:K].mean())
sz = sizes[7:]
print("size dist", {s: int((sz == s).sum()) for s in (1, 2, 4, 8, 16)})
for s in (1, 2, 4, 8, 16):
    print("size", s, "codeword share %.3f" % (hist[7:][sz == s].sum() / tot))

# ---- what would a usage-ranked hot set buy? -------------------------------------------------
use = hist.copy(); use[:7] = 0
cost = 4 + 4 * sizes.astype(np.int64)          # meta word + payload words
order = np.argsort(-(use[:n_off] / cost))       # best codewords per LDS byte first
cum_bytes = np.cumsum(cost[order]); cum_cov = np.cumsum(use[order]) / tot
for kb in (16, 32, 64, 96, 120, 140):
    i = np.searchsorted(cum_bytes, kb * 1024)
    print("usage-ranked hot set %3d KB: %5d entries, codeword coverage %.4f" % (kb, i, cum_cov[min(i, len(cum_cov) - 1)]))
idx_cost = np.cumsum(cost[7:n_off]); idx_cov = np.cumsum(use[7:n_off]) / tot
for kb in (16, 32, 64, 96, 120, 140):
    i = np.searchsorted(idx_cost, kb * 1024)
    print("index-ranked hot set %3d KB: %5d entries, codeword coverage %.4f" % (kb, i, idx_cov[min(i, len(idx_cov) - 1)]))
