#!/usr/bin/env python3
"""AND-query timing (BASELINE config 5): the reference's op_perftest shape
(src/queries.cpp:15-61 — every query run on its own, first pass untimed, avg/q50/q90/q95 in µs)
plus the batch rate the device path is built for, with the oracle's and_query timed beside it.

    python tests/query_timing.py [--postings 1e8] [--type single_packed_dint] [--runs 3]

Lives under tests/ because it times the CPU oracle next to the device path (the oracle is test
infrastructure: nothing outside tests/, smoke() and bench.py's cpu_baseline may touch it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--postings", type=float, default=1e8)
    ap.add_argument("--type", default="single_packed_dint")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--cpu-queries", type=int, default=500)
    ap.add_argument("--forms", action="store_true", help="also time the single-query calls with each launch form forced")
    args = ap.parse_args()
    args.forms = [("round_per_launch_no_fusing", {"query_fused_pages": 0}),
                  ("whole_query_in_one_launch_up_to_1_page", {"query_fused_pages": 1}),
                  ("whole_query_in_one_launch_up_to_4_pages", {"query_fused_pages": 4}),
                  ("one_launch_inputs_copied_on_the_stream", {"query_fused_copy": 0}),
                  ("three_launches_per_decode", {"query_lean_pages": 0}),
                  ("one_launch_per_decode", {"query_tail_pages": 0}),
                  ("round_tail_up_to_4_pages", {"query_tail_pages": 4}),
                  ("round_tail_up_to_16_pages", {"query_tail_pages": 16}),
                  ("round_tail_up_to_64_pages", {"query_tail_pages": 64})] if args.forms else []

    import torch
    from dint_amd import device, host
    from queries import heavy_queries, reference_queries
    import oracle

    kind = host.KIND_BY_TYPE[args.type]
    coll = host.synth_collection(int(args.postings), seed=11)
    docids = host.gaps_to_docids(coll)
    freqs = np.ones(coll.num_postings, dtype=np.uint32)
    dd = host.build_dictionary(kind, coll, max_sample_ints=50_000_000)
    fd = host.build_dictionary(kind, host.Collection(freqs[:1000] - 1, np.array([1000], dtype=np.uint32)))
    idx, offs = host.build_index(kind, dd, fd, docids, freqs, coll.lens)
    n_lists = len(coll.lens)
    workloads = {
        "reference_log_mod_lists": reference_queries(n_lists),
        "longest_lists": heavy_queries(coll.lens, 500, pool=256, max_terms=5),
    }
    ddev = device.Dictionary(kind, dd)
    qi = device.QueryIndex(ddev, idx, offs)
    od = oracle.OracleDict(kind, dd)
    oi = oracle.OracleIndex(od, idx, offs, int(docids.max()) + 1)
    out = {"postings": coll.num_postings, "lists": n_lists, "blocks": int(len(qi.blocks)), "type": args.type,
           "index_bytes": int(idx.size)}
    for name, qs in workloads.items():
        counts = qi.and_queries(qs)  # warm-up (and the first, untimed pass)
        pages = [min(-(-int(coll.lens[t]) // 256) for t in q) for q in qs if len(q)]
        # the batch call as a C++ caller makes it: the log parsed (packed) beforehand, like the single calls below and the
        # oracle's parallel leg; `gpu_batch_from_python_lists` is the same through and_queries(), which packs the lists first
        b_terms = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint32) for q in qs]), dtype=np.uint32)
        b_offs = np.zeros(len(qs) + 1, dtype=np.uint64)
        np.cumsum([len(q) for q in qs], out=b_offs[1:])
        b_counts = np.zeros(len(qs), dtype=np.uint64)
        b_stream = torch.cuda.current_stream().cuda_stream
        qi.and_queries_packed(b_terms, b_offs, b_counts, b_stream)
        assert np.array_equal(b_counts, counts)
        t_batch, t_lists = [], []
        for _ in range(max(args.runs, 5)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            qi.and_queries_packed(b_terms, b_offs, b_counts, b_stream)
            t_batch.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            qi.and_queries(qs)
            t_lists.append(time.perf_counter() - t0)
        batch_forms = {}
        for form, env in ([("three_launches_per_decode", {"query_lean_pages": 0, "query_batch_fused": 0}),
                           ("one_launch_per_decode", {"query_lean_pages": 1000000000, "query_batch_fused": 0}),
                           ("a_workgroup_per_query", {"query_batch_fused": 1})] if args.forms else []):
            for k, v in env.items():
                device.set_option(k, v)
            qi.and_queries_packed(b_terms, b_offs, b_counts, b_stream)
            assert np.array_equal(b_counts, counts)
            ts = []
            for _ in range(args.runs):
                t0 = time.perf_counter()
                qi.and_queries_packed(b_terms, b_offs, b_counts, b_stream)
                ts.append(time.perf_counter() - t0)
            batch_forms[form] = min(ts) * 1e6 / len(qs)
            device.reset_options()
        # one query per call, the reference's op_perftest shape: the queries are parsed (packed) beforehand, the
        # timed region is the call
        packed = []
        for q in qs:
            t = np.ascontiguousarray(q, dtype=np.uint32)
            packed.append((t, np.array([0, t.size], dtype=np.uint64), np.zeros(1, dtype=np.uint64)))
        stream = torch.cuda.current_stream().cuda_stream
        for t, o, c in packed:
            qi.and_queries_packed(t, o, c, stream)
        def singles():
            us = []
            for (t, o, c), want in zip(packed, counts):
                t0 = time.perf_counter()
                qi.and_queries_packed(t, o, c, stream)
                us.append((time.perf_counter() - t0) * 1e6)
                assert int(c[0]) == int(want)
            return np.sort(np.array(us))
        single = singles()
        # the same calls with the launch forms forced (dint_hip.hip: lean_pages, tail_pages) — in this process, on
        # this box: boxes differ by more than the forms do
        forms = {}
        for form, env in args.forms:
            for k, v in env.items():
                device.set_option(k, v)
            singles()
            forms[form] = float(singles().mean())
            device.reset_options()
        cpu_q = qs[:args.cpu_queries]
        cpu = []
        for q in cpu_q:
            t0 = time.perf_counter()
            oi.and_query(q)
            cpu.append((time.perf_counter() - t0) * 1e6)
        cpu_counts = np.array([oi.and_query(q) for q in cpu_q[:50]], dtype=np.uint64)
        assert np.array_equal(cpu_counts, counts[:len(cpu_counts)])
        cpu = np.sort(np.array(cpu))
        # ... and on every CPU the container may use (its cgroup quota: 16 of the box's 256): the same log answered by that
        # many pthreads INSIDE liboracle (query q on thread q % threads), three passes, wall time of a pass / queries
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q_, period_ = f.read().split()[:2]
            quota = None if q_ == "max" else float(q_) / float(period_)
        except (OSError, ValueError):
            quota = None
        n_thr = len(os.sched_getaffinity(0)) if quota is None else max(1, min(len(os.sched_getaffinity(0)), int(quota)))
        par_counts, _ = oi.and_queries_parallel(cpu_q, n_thr, 1)  # (warm: pages touched, threads created once before)
        assert np.array_equal(par_counts[:len(cpu_counts)], cpu_counts)
        par_counts, wall = oi.and_queries_parallel(cpu_q, n_thr, 3)
        one_counts, wall_one = oi.and_queries_parallel(cpu_q, 1, 3)
        assert np.array_equal(par_counts, one_counts)
        cpu_all = {"us_per_query": wall * 1e6 / max(1, len(cpu_q)), "threads": n_thr,
                   "us_per_query_one_thread_same_call": wall_one * 1e6 / max(1, len(cpu_q)),
                   "note": "oracle_and_queries_parallel: pthreads inside liboracle, query q on thread q % threads; wall time of one "
                           "pass of the log / queries (round 4 drove the oracle from a Python thread pool and measured the interpreter)"}
        pct = lambda a, p: float(a[min(len(a) - 1, int(p * len(a) / 100))])
        out[name] = {
            "queries": len(qs), "results": int(counts.sum()),
            "gpu_batch_us_per_query": min(t_batch) * 1e6 / len(qs),
            "gpu_batch_from_python_lists_us_per_query": min(t_lists) * 1e6 / len(qs),
            "gpu_single": {"avg": float(single.mean()), "q50": pct(single, 50), "q90": pct(single, 90), "q95": pct(single, 95)},
            "gpu_single_avg_by_form": forms,
            "gpu_batch_us_per_query_by_form": batch_forms,
            "candidate_pages_q50_q90": [float(np.percentile(pages, 50)), float(np.percentile(pages, 90))],
            "cpu_oracle": {"avg": float(cpu.mean()), "q50": pct(cpu, 50), "q90": pct(cpu, 90), "q95": pct(cpu, 95), "cores": 1},
            "cpu_oracle_all_threads": cpu_all,
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
