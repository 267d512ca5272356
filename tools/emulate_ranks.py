#!/usr/bin/env python3
"""BASELINE config 4 on a one-GPU box: the W shards of a W-rank job decoded one after the other, each by a fresh
`bench.py --as-rank K/W` process on the same GPU, and what those W ranks would aggregate to if they ran side by side:
sum of the integers / the slowest shard's time per step. EMULATED — NOT A SCALING MEASUREMENT: one GPU stands in for W,
nothing runs concurrently, no RCCL call is made (the job's only collectives reduce 24 bytes at the end).

usage: tools/emulate_ranks.py [--world 8] [--ranks 0,3,7] [-- <bench.py arguments: --workload clueweb --steps 10 ...>]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    rest = []
    if "--" in argv:
        i = argv.index("--")
        argv, rest = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default=None, help="comma-separated subset (default: all)")
    ap.add_argument("--out", default=None, help="write the summary JSON here too")
    a = ap.parse_args(argv)
    ranks = [int(x) for x in a.ranks.split(",")] if a.ranks else list(range(a.world))
    rest = rest or ["--workload", "clueweb", "--type", "single_packed_dint", "--steps", "10", "--warmup", "3", "--cpu-seconds", "0"]
    lines = []
    for k in ranks:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--as-rank", f"{k}/{a.world}"] + rest,
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit(f"shard {k}/{a.world} failed")
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        lines.append(d)
        print(f"[emulate] shard {k}/{a.world}: {d['config']['ints_per_gpu_per_step']} ints, {d['ms_per_step']} ms/step, "
              f"{d['value'] / 1e3:.1f} G ints/s, roofline {d['roofline']['frac']}, bit_exact {d['bit_exact']}", file=sys.stderr, flush=True)
    ints = sum(d["config"]["ints_per_gpu_per_step"] for d in lines)
    slowest = max(d["ms_per_step"] for d in lines)
    scale = a.world / len(ranks)  # a subset stands for the whole job (shards are balanced by postings)
    summary = {
        "emulated": True,
        "note": "EMULATED, NOT A SCALING MEASUREMENT: each shard decoded alone on ONE GPU by bench.py --as-rank; aggregate = "
                "sum of integers / slowest shard's ms per step" + ("" if len(ranks) == a.world else f", scaled from {len(ranks)} of {a.world} shards"),
        "world": a.world, "ranks_run": ranks,
        "bench_args": rest,
        "ints_per_step_all_ranks": int(ints * scale),
        "slowest_shard_ms_per_step": slowest,
        "predicted_aggregate_M_ints_per_s": round(ints * scale / (slowest * 1e-3) / 1e6, 1),
        "bit_exact_all": all(d["bit_exact"] for d in lines),
        "shards": [{"rank": d["emulated_rank"], "ints": d["config"]["ints_per_gpu_per_step"], "lists": d["config"]["lists_per_gpu"],
                    "ms_per_step": d["ms_per_step"], "M_ints_per_s": d["value"], "roofline_frac": d["roofline"]["frac"],
                    "kernel_ms": d["roofline"]["kernel_ms"], "bits_per_int": d["config"]["bits_per_int"],
                    "bit_exact": d["bit_exact"]} for d in lines],
    }
    text = json.dumps(summary)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
