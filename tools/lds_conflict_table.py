#!/usr/bin/env python3
"""LDS bank conflicts of decode_single_kernel by instruction class (development aid, GPU box).

The kernel's LDS instructions fall into classes whose ADDRESSES can be made conflict-free without changing what the
kernel does next (tools/variants/lds_linear.patch, -DDINT_EXP_LDS=<bits>: the results are wrong by construction, the
instruction stream and the control flow are the product's): the expansion's u16 gathers (16), its delta reads (8), the
flag atomics (2), the general tiles' delta stores (4). The builds are cumulative — 16, 16+8, 16+8+2, 16+8+2+4 — so the
drop in SQ_LDS_BANK_CONFLICT from one to the next is that class's conflict cycles; what is left belongs to the classes
whose addresses steer the decode (hot-metadata reads, classification rows, staging-cell writes, rank bases).

usage, from a `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- python3
tools/ab_bench.py --rounds 1 --reps R name=lib ...` directory:  tools/lds_conflict_table.py <dir> R name [name ...]"""
import csv, glob, os, sys

d, reps, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "decode_single_kernel" in r["Kernel_Name"]]
by_dispatch = {}
for r in rows:
    by_dispatch.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
order = [by_dispatch[k] for k in sorted(by_dispatch)]
warm = 3
assert len(order) == len(names) * (warm + reps), (len(order), len(names), reps)
timed = order[len(names) * warm:]  # ab_bench: every build's warm-up launches first, then one round of `reps` launches per build
print(f"{'build':10s} {'LDS insts':>12s} {'IDX_ACTIVE':>12s} {'BANK_CONFLICT':>14s} {'conflict / active':>18s} {'conflict cycles less than the build before':>44s}")
prev = None
for i, n in enumerate(names):
    mine = timed[i * reps:(i + 1) * reps]
    g = lambda k: sum(m.get(k, 0.0) for m in mine) / len(mine)
    c, a, ins = g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"), g("SQ_INSTS_LDS")
    less = "" if prev is None else f"{prev - c:14.4g} ({100 * (prev - c) / base_c:5.1f} % of the product's)"
    if prev is None:
        base_c = c
    print(f"{n:10s} {ins:12.4g} {a:12.4g} {c:14.4g} {c / a:18.3f} {less:>44s}")
    prev = c
