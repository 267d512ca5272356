#!/usr/bin/env python3
"""Development aid (GPU box): one list of a real collection through the host-pointer call (one unit, one wave) — where does the
decode differ, and what kind of codeword sits there? usage: tools/one_list_diff.py [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
if len(sys.argv) > 1:
    os.environ["DINT_HIP_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np
from dint_amd import device, host
import pydecode
coll = host.synth_collection(3_000_000, universe=25_000_000, seed=12345)
kind = host.SINGLE_PACKED
dict_file = host.build_dictionary(kind, coll)
d = device.Dictionary(kind, dict_file)
info = d.info()
print("hot entries", info.hot_entries, "of", info.entries, flush=True)
pd = pydecode.parse_single_packed(dict_file)
bounds = np.concatenate([[0], np.cumsum(coll.lens)]).astype(np.int64)
order = np.argsort(-coll.lens.astype(np.int64))
for li in order[:4]:
    n = int(coll.lens[li])
    gaps = np.ascontiguousarray(coll.gaps[bounds[li]:bounds[li] + n])
    one = host.Collection(gaps, np.array([n], dtype=np.uint32))
    enc, units = host.encode_vroom(kind, dict_file, one, unit_ints=0)
    # header: two vbytes
    p = 0
    for _ in range(2):
        while enc[p] < 128: p += 1
        p += 1
    got, consumed = d.decode_list(enc, p, n)
    bad = np.nonzero(got != gaps)[0]
    print(f"list {li}: n {n}, payload {enc.size - p} B: {bad.size} wrong" + (f", first at {bad[0]}: want {gaps[bad[0]:bad[0]+8]} got {got[bad[0]:bad[0]+8]}" if bad.size else ""))
    if bad.size:
        # walk the codewords up to the first bad position
        slots = np.frombuffer(enc[p:].tobytes() + b"\0\0", dtype="<u2")
        pos, i = 0, 0
        trail = []
        while pos <= bad[0] + 16 and i < slots.size:
            s = int(slots[i])
            if s == 0: size, step, what = 1, 2, "exc16"
            elif s == 1: size, step, what = 1, 3, "exc32"
            else:
                size = len(pd(0, s))
                step, what = 1, ("hot" if s < info.hot_entries else "COLD")
            trail.append((i, i // 256, (i % 256) // 4, s, what, size, pos))
            pos += size if size else 1
            i += step
        # every codeword of the first tiles: right or wrong
        pos, i, rows = 0, 0, []
        while i < min(slots.size, 3 * 256):
            sv = int(slots[i])
            if sv == 0: size, step, what = 1, 2, "exc16"
            elif sv == 1: size, step, what = 1, 3, "exc32"
            else: size, step, what = len(pd(0, sv)), 1, ("hot" if sv < info.hot_entries else "COLD")
            okc = bool(np.array_equal(got[pos:pos + size], gaps[pos:pos + size]))
            rows.append((i, i // 256, (i % 256) // 4, i % 4, sv, what, size, pos, "ok" if okc else "WRONG " + str(got[pos:pos + min(size, 4)]) + " want " + str(gaps[pos:pos + min(size, 4)])))
            pos += size
            i += step
        for r in rows[:40] + rows[250:262]:
            print("   slot %d (tile %d lane %d k %d) value %d %s size %d -> out %d: %s" % r)
        # which tiles are wrong (tile = 256 slots), in two more decodes of the same list
        tile_of_out = np.zeros(n, dtype=np.int32)
        pos, i = 0, 0
        while i < slots.size and pos < n:
            sv = int(slots[i])
            size, step = (1, 2) if sv == 0 else (1, 3) if sv == 1 else (len(pd(0, sv)), 1)
            tile_of_out[pos:pos + size] = i // 256
            pos += size
            i += step
        feats = {}
        pos, i = 0, 0
        while i < slots.size and pos < n:
            sv = int(slots[i]); tl = i // 256
            f = feats.setdefault(tl, dict(exc16=0, exc32=0, cold=0, cold_long=0, cold16=0, runs=0))
            if sv == 0: size, step = 1, 2; f["exc16"] += 1
            elif sv == 1: size, step = 1, 3; f["exc32"] += 1
            else:
                size, step = len(pd(0, sv)), 1
                if 2 <= sv <= 6: f["runs"] += 1
                if sv >= info.hot_entries:
                    f["cold"] += 1
                    if size > 6: f["cold_long"] += 1
                    if size > 14: f["cold16"] += 1
            pos += size; i += step
        g2, _ = d.decode_list(enc, p, n)
        wt = set(np.unique(tile_of_out[np.nonzero(g2 != gaps)[0]]).tolist())
        import collections
        for key in ("exc16", "exc32", "cold", "cold_long", "cold16", "runs"):
            r = collections.Counter(min(feats[t][key], 3) for t in feats if t not in wt)
            w = collections.Counter(min(feats[t][key], 3) for t in feats if t in wt)
            print("   ", key, "right tiles:", dict(sorted(r.items())), " wrong tiles:", dict(sorted(w.items())))
        for rep in range(1):
            g2, _ = d.decode_list(enc, p, n)
            wrong_tiles = np.unique(tile_of_out[np.nonzero(g2 != gaps)[0]])
            print("   decode", rep, "wrong tiles:", len(wrong_tiles), "of", int(tile_of_out.max()) + 1, "first ones", wrong_tiles[:40])
        import collections
        cnt = collections.Counter((r[5], r[8][:2]) for r in rows)
        print(cnt)
        break
