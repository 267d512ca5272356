#!/usr/bin/env python3
"""In-index decode at index scale (round 6; include/dint/dict_posting_list.hpp:284-318 is per block, scale-free): every
posting of indexes of 1e8 / 1e9 / 5e9 postings -> docIDs (+ freqs) through a prepared, taught block table (ONE launch a
decode), docs + freqs and docs only, single_packed_dint and multi_packed_dint; per size the time, postings/s and the
fraction of 8 TB/s (algorithmic bytes: 4 or 8 per posting written + the index bytes read), then a fixed + marginal fit
across the sizes. The index is generated, built and uploaded in pieces of 1e9 postings (host memory: one piece at a time);
the expected docIDs / freqs stay on the device and the bit-exact check compares there.
usage: tools/inindex_scale.py out.json [sizes, e.g. 1e8,1e9,5e9] [types, e.g. single_packed_dint,multi_packed_dint]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

out_path = sys.argv[1]
sizes = [int(float(x)) for x in (sys.argv[2] if len(sys.argv) > 2 else "1e8,1e9,5e9").split(",")]
types = (sys.argv[3] if len(sys.argv) > 3 else "single_packed_dint,multi_packed_dint").split(",")
PIECE = int(float(os.environ.get("PIECE", "5e8")))
dev = torch.device("cuda:0")
res = {"sizes": sizes, "types": types, "rows": []}


def build(typ, postings):
    """-> (index bytes on the host, offsets, blocks, total, dictionaries, expected docids / freqs on the device)"""
    kind = host.KIND_BY_TYPE[typ]
    p = host.synth_params(universe=25_000_000, seed=777)
    lens = host.synth_lengths(p, postings)
    cum = np.cumsum(lens, dtype=np.uint64)
    n_pieces = max(1, int(round(postings / PIECE)))
    cuts = sorted(set([0] + [int(np.searchsorted(cum, postings * k // n_pieces, side="left")) + 1 for k in range(1, n_pieces)] + [len(lens)]))
    idx_parts, off_parts, exp_d, exp_f = [], [], [], []
    dd = fd = None
    byte0 = 0
    for i in range(len(cuts) - 1):
        a, b = cuts[i], cuts[i + 1]
        c = host.Collection(host.synth_gaps(p, lens[a:b], first_list_id=a), lens[a:b])
        docids = host.gaps_to_docids(c)
        freqs = np.random.default_rng([5, a]).geometric(0.55, c.num_postings).astype(np.uint32)
        if dd is None:
            dd = host.build_dictionary(kind, c, max_sample_ints=20_000_000)
            fd = host.build_dictionary(kind, host.Collection(freqs - 1, c.lens), max_sample_ints=20_000_000)
        idx, offs = host.build_index(kind, dd, fd, docids, freqs, c.lens)
        idx_parts.append(idx)
        off_parts.append(offs[:-1].astype(np.uint64) + np.uint64(byte0))
        byte0 += idx.size
        exp_d.append(torch.from_numpy(docids.view(np.int32)).to(dev))
        exp_f.append(torch.from_numpy(freqs.view(np.int32)).to(dev))
        del c, docids, freqs
    index = np.concatenate(idx_parts) if len(idx_parts) > 1 else idx_parts[0]
    offsets = np.concatenate(off_parts + [np.array([byte0], dtype=np.uint64)])
    return kind, index, offsets, dd, fd, (torch.cat(exp_d) if len(exp_d) > 1 else exp_d[0]), (torch.cat(exp_f) if len(exp_f) > 1 else exp_f[0])


for typ in types:
    for postings in sizes:
        t0 = time.time()
        kind, index, offsets, dd, fd, exp_d, exp_f = build(typ, postings)
        blocks, total = device.index_posting_lists(index, offsets)
        D, F = device.Dictionary(kind, dd), device.Dictionary(kind, fd)
        padded = np.concatenate([index, np.zeros(16, dtype=np.uint8)])
        index_dev = torch.from_numpy(padded).to(dev)
        table = device.BlockTable(D, blocks, padded.size)
        t1 = time.perf_counter()
        table.learn(D, F, index_dev, padded.size)
        torch.cuda.synchronize()
        learn_ms = (time.perf_counter() - t1) * 1e3
        docs_dev, freqs_dev = torch.empty(total, dtype=torch.int32, device=dev), torch.empty(total, dtype=torch.int32, device=dev)
        row = {"type": typ, "postings": int(total), "lists": int(len(offsets) - 1), "blocks": int(len(blocks)),
               "short_blocks": int((blocks["n"] < 256).sum()), "index_bytes": int(index.size),
               "bits_per_posting": round(index.size * 8 / total, 3), "learn_ms": round(learn_ms, 2), "ready": bool(table.ready(True)),
               "table": table.info(), "set_up_s": round(time.time() - t0, 1)}
        for label, fdev in (("docs_and_freqs", freqs_dev), ("docs_only", None)):
            ms = []
            for i in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                table.decode(D, F if fdev is not None else None, index_dev, padded.size, docs_dev, fdev)
                e1.record(); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            m = float(np.median(ms[2:]))
            algo = total * (8 if fdev is not None else 4) + index.size * (1.0 if fdev is not None else 0.5)
            row[label] = {"ms": round(m, 4), "ms_all": [round(x, 4) for x in ms], "G_postings_per_s": round(total / m / 1e6, 1),
                          "algorithmic_GBps": round(algo / m / 1e6, 1), "frac_of_8TBps": round(algo / m / 1e6 / 8000, 4)}
        ok = True
        for a in range(0, total, 1 << 28):
            b = min(total, a + (1 << 28))
            ok = ok and bool(torch.equal(docs_dev[a:b], exp_d[a:b])) and bool(torch.equal(freqs_dev[a:b], exp_f[a:b]))
        row["bit_exact"] = ok
        if not ok:  # where: the first differing posting of each output, how many differ, the block it lies in
            for name, got, want in (("docids", docs_dev, exp_d), ("freqs", freqs_dev, exp_f)):
                bad = 0; first = None
                for a in range(0, total, 1 << 28):
                    b = min(total, a + (1 << 28))
                    ne = torch.nonzero(got[a:b] != want[a:b])
                    bad += int(ne.numel())
                    if first is None and ne.numel():
                        first = a + int(ne[0])
                blk = int(np.searchsorted(blocks["out_off"], first, side="right") - 1) if first is not None else None
                row[f"mismatch_{name}"] = {"count": bad, "first": first, "block": blk,
                                           "block_in_off": int(blocks["in_off"][blk]) if blk is not None else None,
                                           "block_n": int(blocks["n"][blk]) if blk is not None else None}
        res["rows"].append(row)
        print(json.dumps(row), flush=True)
        del table, index_dev, docs_dev, freqs_dev, exp_d, exp_f, D, F
        torch.cuda.empty_cache()
        json.dump(res, open(out_path, "w"), indent=1)
# fixed + marginal: ms = fixed + postings / rate, least squares over the sizes, per type and decode form
for typ in types:
    rows = [r for r in res["rows"] if r["type"] == typ]
    if len(rows) < 2:
        continue
    for label in ("docs_and_freqs", "docs_only"):
        x = np.array([r["postings"] for r in rows], dtype=np.float64)
        y = np.array([r[label]["ms"] for r in rows], dtype=np.float64)
        A = np.stack([np.ones_like(x), x], axis=1)
        (fixed, slope), *_ = np.linalg.lstsq(A, y, rcond=None)
        per_posting_bytes = np.mean([(r["postings"] * (8 if label == "docs_and_freqs" else 4) + r["index_bytes"] * (1.0 if label == "docs_and_freqs" else 0.5)) / r["postings"] for r in rows])
        res.setdefault("fit", {})[f"{typ}/{label}"] = {"fixed_ms": round(float(fixed), 4), "marginal_ms_per_1e9": round(float(slope) * 1e9, 4),
                                                      "marginal_G_postings_per_s": round(1e-6 / float(slope), 1) if slope > 0 else None,
                                                      "marginal_frac_of_8TBps": round(per_posting_bytes / (float(slope) * 1e-3) / 8e12, 4) if slope > 0 else None}
print(json.dumps(res.get("fit", {})), flush=True)
json.dump(res, open(out_path, "w"), indent=1)
