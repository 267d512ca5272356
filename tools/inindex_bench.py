#!/usr/bin/env python3
"""In-index decode: every posting of an index in the dict_posting_list layout
-> docIDs and freqs on the device, through a prepared block table (dint_block_table), timed with HIP events
around the enqueued launches. usage: tools/inindex_bench.py [postings] [out.json] [option=value ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
for kv in sys.argv[3:]:  # library options, e.g. index_pair=1 index_inline_tails=0 (dint_set_option)
    k, v = kv.split("=")
    device.set_option(k, int(v))
TRIALS = int(os.environ.get("PLACEMENT_TRIALS", "4"))
dev = torch.device("cuda:0")
sub = host.synth_collection(postings, universe=25_000_000, seed=777)
docids = host.gaps_to_docids(sub); freqs = host.synth_freqs(sub.num_postings, 5)
res = {"postings": int(sub.num_postings), "lists": int(np.count_nonzero(sub.lens))}
for typ in ("single_packed_dint", "multi_packed_dint"):
    kind = host.KIND_BY_TYPE[typ]
    dd = host.build_dictionary(kind, sub, max_sample_ints=20_000_000)
    fd = host.build_dictionary(kind, host.Collection(freqs - 1, sub.lens), max_sample_ints=20_000_000)
    idx, offs = host.build_index(kind, dd, fd, docids, freqs, sub.lens)
    blocks, total = device.index_posting_lists(idx, offs)
    D, F = device.Dictionary(kind, dd), device.Dictionary(kind, fd)
    padded = np.concatenate([idx, np.zeros(16, dtype=np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    table = device.BlockTable(D, blocks, padded.size)
    # the sizing pass at set-up (dint_block_table_learn), then the FIRST decode a caller makes, timed on the stream: already
    # the one launch. (An untaught table: its first two decodes are the learning ones — `untaught_first_three_ms`.)
    first = {}
    probe_d, probe_f = torch.empty(total, dtype=torch.int32, device=dev), torch.empty(total, dtype=torch.int32, device=dev)
    def once(tab):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tab.decode(D, F, index_dev, padded.size, probe_d, probe_f); e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1), 4)
    untaught = device.BlockTable(D, blocks, padded.size)
    first["untaught_first_three_ms"] = [once(untaught) for _ in range(3)]
    del untaught
    t0 = time.perf_counter(); table.learn(D, F, index_dev, padded.size); first["learn_wall_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    first["ready_after_learn"] = bool(table.ready(True))
    first["table"] = table.info()
    first["taught_first_decode_ms"] = once(table)
    first["taught_next_decodes_ms"] = [once(table) for _ in range(3)]
    r = {"first_decode": first, "bits_per_posting": round(idx.size * 8 / total, 3), "blocks": int(len(blocks)), "short_blocks": int((blocks["n"] < 256).sum())}
    # placement: the decode's time depends on where the driver puts the buffers it writes, relative to what it
    # reads — a few candidate pairs of output buffers, the fastest stays (bench.py --placement-trials does the same)
    def timed(dd_, ff_):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); table.decode(D, F, index_dev, padded.size, dd_, ff_); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    cands = [(torch.empty(total, dtype=torch.int32, device=dev), torch.empty(total, dtype=torch.int32, device=dev)) for _ in range(TRIALS)]
    trial_ms = []
    for dd_, ff_ in cands:
        for _ in range(3): timed(dd_, ff_)  # (the table's first decodes learn the spans and build the schedules)
        trial_ms.append(round(min(timed(dd_, ff_) for _ in range(3)), 4))
    docs_dev, freqs_dev = cands[int(np.argmin(trial_ms))]
    del cands
    r["placement_trial_ms"] = trial_ms
    for label, fdev in (("docs_and_freqs", freqs_dev), ("docs_only", None)):
        ms = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            table.decode(D, F if fdev is not None else None, index_dev, padded.size, docs_dev, fdev)
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        m = float(np.median(ms[2:]))
        algo = total * (8 if fdev is not None else 4) + idx.size * (1.0 if fdev is not None else 0.5)
        r[label] = {"ms": round(m, 4), "ms_all": [round(x, 4) for x in ms], "G_postings_per_s": round(total / m / 1e6, 1),
                    "algorithmic_GBps": round(algo / m / 1e6, 1), "frac_of_8TBps": round(algo / m / 1e6 / 8000, 4)}
    r["bit_exact"] = bool(np.array_equal(docs_dev.cpu().numpy().view(np.uint32), docids)) and \
        bool(np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), freqs))
    res[typ] = r
    print(typ, json.dumps(r), flush=True)
    del table
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
