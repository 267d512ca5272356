#!/usr/bin/env python3
"""Development aid (GPU box): does a sidecar of UNEVEN units pay — long units first (a unit's fixed costs: the queue
ticket, the segment's prologue and epilogue, are per unit), short ones at the end of every queue shard's range (the launch
lasts as long as its slowest wave: the tail is a unit long)? One stream indexed at 4096 integers; the units of a list merged
into longer ones (they are consecutive in the stream and in the output) by pattern; decoded through a prepared unit table,
all patterns in one process on one pair of buffers.
usage: tools/unit_guided_sweep.py [postings=2e9] [replicate=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = host.SINGLE_PACKED
p = host.synth_params(universe=25_000_000, seed=12345)
lens = host.synth_lengths(p, postings)
PIECE = 500_000_000
cum = np.cumsum(lens, dtype=np.uint64)
cuts = [0] + [int(np.searchsorted(cum, k * PIECE)) for k in range(1, max(1, postings // PIECE))] + [len(lens)]
cuts = sorted(set(cuts))
dict_file = None
enc_parts, unit_parts, gaps_parts = [], [], []
eb = ints = lists = 0
for a, b in zip(cuts[:-1], cuts[1:]):
    c = host.Collection(host.synth_gaps(p, lens[a:b], first_list_id=a), lens[a:b])
    if dict_file is None:
        dict_file = host.build_dictionary(kind, c, max_sample_ints=20_000_000)
    e, u = host.encode_vroom(kind, dict_file, c, unit_ints=4096)
    u = u.copy(); u["in_off"] += np.uint64(eb); u["out_off"] += np.uint64(ints); u["list"] += np.uint32(lists)
    enc_parts.append(e); unit_parts.append(u); gaps_parts.append(c.gaps)
    eb += e.size; ints += c.num_postings; lists += len(c.lens)
enc = np.concatenate(enc_parts); fine = np.concatenate(unit_parts); gaps = np.concatenate(gaps_parts)
del enc_parts, unit_parts, gaps_parts
print(f"{ints} postings, {enc.size} B, {len(fine)} units of <= 4096", flush=True)


def merge(units, target):
    """consecutive units of one list joined until they hold `target[i]` integers (target: per fine unit, the size wanted where
    it starts)"""
    out = []
    cur = None
    for i in range(len(units)):
        u = units[i]
        if cur is not None and cur["list"] == u["list"] and int(cur["n"]) + int(u["n"]) <= cur_target:
            cur["n"] += u["n"]
        else:
            if cur is not None: out.append(cur)
            cur = u.copy(); cur_target = int(target[i])
    if cur is not None: out.append(cur)
    return np.array(out, dtype=units.dtype)


def merge_fast(units, target):
    # vectorised over runs: greedy join inside a list by cumulative integers (a unit starts a group when the running sum since
    # the group's start would pass the group's target)
    n = units["n"].astype(np.int64); lst = units["list"]
    keep = np.ones(len(units), dtype=bool)
    acc = 0; tgt = 0; prev_list = -1
    nn = n.tolist(); ll = lst.tolist(); tt = target.tolist()
    for i in range(len(nn)):
        if ll[i] == prev_list and acc + nn[i] <= tgt:
            keep[i] = False; acc += nn[i]
        else:
            acc = nn[i]; tgt = tt[i]; prev_list = ll[i]
    idx = np.flatnonzero(keep)
    out = units[idx].copy()
    sums = np.add.reduceat(n, idx)
    out["n"] = sums.astype(np.uint32)
    return out


pos = fine["out_off"].astype(np.float64) / ints          # where a unit lies in the collection, 0..1
def pattern(big, small, tail_frac, parts):
    inside = (pos * parts) % 1.0                          # ... inside its part (a queue shard's range, roughly)
    return np.where(inside < 1.0 - tail_frac, big, small).astype(np.int64)

patterns = [("uniform 16384", np.full(len(fine), 16384)), ("uniform 32768", np.full(len(fine), 32768)),
            ("uniform 65536", np.full(len(fine), 65536)),
            ("65536 then 4096 for the last 10% of each of 8 parts", pattern(65536, 4096, 0.10, 8)),
            ("65536 then 8192 for the last 15% of each of 8 parts", pattern(65536, 8192, 0.15, 8)),
            ("131072 then 8192 for the last 15% of each of 8 parts", pattern(131072, 8192, 0.15, 8)),
            ("65536 then 4096 for the last 5% of the whole", pattern(65536, 4096, 0.05, 1)),
            ("uniform 16384 again", np.full(len(fine), 16384))]
d = device.Dictionary(kind, dict_file)
dev = torch.device("cuda:0")
one = torch.from_numpy(enc).to(dev)
enc_dev = torch.empty(enc.size * R, dtype=torch.uint8, device=dev)
for r in range(R): enc_dev[r * enc.size:(r + 1) * enc.size].copy_(one)
del one
out_dev = torch.empty(ints * R, dtype=torch.int32, device=dev)
for name, tgt in patterns:
    units = merge_fast(fine, tgt)
    ua = np.tile(units, R)
    for r in range(R):
        sl = slice(r * len(units), (r + 1) * len(units))
        ua["in_off"][sl] += np.uint64(r * enc.size); ua["out_off"][sl] += np.uint64(r * ints)
    units_dev = device.units_to_device(ua, dev)
    table = device.UnitTable(d, enc_dev, units_dev, len(ua), out_dev.numel())
    out_dev.zero_()
    for _ in range(3): table.decode(out_dev)
    torch.cuda.synchronize()
    for _ in range(10): table.decode(out_dev)
    torch.cuda.synchronize()
    ms = d.recent_kernel_ms(10)
    ok = all(np.array_equal(out_dev[r * ints:(r + 1) * ints].cpu().numpy().view(np.uint32), gaps) for r in range(min(R, 1)))
    print(f"{name:55s}: {len(ua):8d} units, kernel {ms.mean():.4f} ms (min {ms.min():.4f}), {ints * R / ms.mean() / 1e6:.1f} G ints/s, bit-exact {ok}", flush=True)
    table.close()
    del units_dev
