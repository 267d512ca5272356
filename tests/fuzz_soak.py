#!/usr/bin/env python3
"""Test infrastructure (GPU box; lives in tests/ because it checks against the oracle): the differential fuzzer of tests/test_gpu_fuzz.py over seeds of its OWN — the committed plan's
dictionary and stream shapes, fresh random numbers — for as long as asked. Every case runs the same checks as the pinned
ones (integers, docIDs, freqs, end offsets, canaries, prepared tables, AND queries against the generator's substitution
and the oracle); the first failure stops the run and names the seed, which then reproduces with
`tests/fuzz_soak.py --rounds 1 --first-round R`.

--random-options: every case under a random legal setting of the library's switches (dint_set_option: bundles, chunk_split,
index_pair, index_inline_tails, the query forms, ...) instead of the defaults.

usage: tests/fuzz_soak.py [--seconds 300] [--first-round 1] [--rounds 1000000] [--random-options]"""
import argparse, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]  # (like tests/conftest.py: oracle/oracle.py, the checker)
import numpy as np
import test_gpu_fuzz as T
import fuzz_streams as F
from dint_amd import device

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--first-round", type=int, default=1)
ap.add_argument("--rounds", type=int, default=1_000_000)
ap.add_argument("--random-options", action="store_true")
args = ap.parse_args()
CHOICES = {"bundles": [0, 1], "index_concurrent": [0, 1], "query_lean_pages": [-1, 1, 3], "query_tail_pages": [0, 1, 4, 16],
           "query_fused_pages": [0, 2, 8], "index_inline_tails": [0, 1], "chunk_split": [-1, 0, 2, 4], "index_pair": [0, 1],
           "query_fused_copy": [0, 1], "query_batch_fused": [0, 1]}
t0, done = time.time(), {"vroom": 0, "index": 0, "lists": 0}
for rnd in range(args.first_round, args.first_round + args.rounds):
    shift = 1_000_000 * rnd
    cases = [("vroom", (c[0] + shift,) + tuple(c[1:])) for c in F.plan(*T.GOLDEN["vroom_plan"])]
    cases += [("index", (c[0] + shift,) + tuple(c[1:])) for c in F.index_plan(*T.GOLDEN["index_plan"])]
    for what, case in cases:
        setting = {}
        if args.random_options:
            r = np.random.default_rng(case[0])
            setting = {k: int(r.choice(v)) for k, v in CHOICES.items() if k in device.OPTIONS}
            for k, v in setting.items():
                device.set_option(k, v)
        try:
            (T.test_vroom_case if what == "vroom" else T.test_index_case)(device, case)
        except Exception:
            traceback.print_exc()
            print(f"FAILED: round {rnd}, {what} case seed {case[0]}, options {setting}", flush=True)
            sys.exit(1)
        finally:
            device.reset_options()
        done[what] += 1
        done["lists"] += case[-1]
        if time.time() - t0 > args.seconds:
            break
    print(f"round {rnd} done: {done} in {time.time() - t0:.0f}s", flush=True)
    if time.time() - t0 > args.seconds:
        break
print(f"no difference: {done['vroom']} vroom cases, {done['index']} index cases, {done['lists']} lists, rounds {args.first_round}..{rnd}")
