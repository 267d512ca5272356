#!/usr/bin/env python3
"""SQ counters of the decode kernel, per launch and per decoded integer, from the sq1/ sq2/ passes of
tools/profile_round.sh (a bench step is one launch of the decode kernel).
usage: tools/pmc_sq_summary.py <out dir> <ints per step>"""
import collections, csv, glob, json, os, sys
out, ints_per_step = sys.argv[1], float(sys.argv[2])
bench = json.load(open(os.path.join(out, "bench.json")))
ints = ints_per_step
agg = collections.defaultdict(lambda: [0.0, 0])
kernel = None
durations = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "sq*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if not row["Kernel_Name"].startswith(("dint_dev::decode_single_kernel", "dint_dev::decode_multi_kernel", "dint_dev::decode_multi_bundles_kernel")):
            continue
        kernel = row["Kernel_Name"].split("(")[0] if kernel is None or "bundles" in row["Kernel_Name"] else kernel
        a = agg[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        durations[row["Counter_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
# (a step may be several dispatches — the bundles kernel over the table, the same kernel over the cut units, the general one for
# what is left: a counter's total over everything, per STEP = per dispatch that lasts at least half as long as the longest one)
for counter, ds in durations.items():
    agg[counter][1] = sum(1 for x in ds if 2 * x >= max(ds))
print(f"kernel {kernel}: {ints:.4g} integers per launch ({bench['metric']})")
for k in sorted(agg):
    v, n = agg[k]
    print(f"{k:24s} per launch {v / n:14.6g}   per integer {v / n / ints:9.4f}   ({n} launches)")
g = lambda k: agg[k][0] / max(1, agg[k][1])
if "SQ_WAVE_CYCLES" in agg:
    print(f"VALU instructions per integer {g('SQ_INSTS_VALU') / ints:.3f}; SALU {g('SQ_INSTS_SALU') / ints:.3f}; LDS {g('SQ_INSTS_LDS') / ints:.3f}; "
          f"waves waiting {100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.1f} % of their cycles")
