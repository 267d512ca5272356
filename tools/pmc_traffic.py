#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a bench.py command into the JSON that
`bench.py --traffic-file` reads.

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <type> <ints_per_launch> <out.json> [kernel-substring]

rocprofv3 reports both counters in units of 1024 bytes per dispatch. On gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): the guide's correction is x2. This
kernel reads the stream 8 B per lane (consecutive lanes: coalesced) plus scattered 16-B gathers served mostly by
L2, for which the guide gives no calibration — so both the raw and the x2 figure are reported and the total uses
the corrected one (an upper bound). WRITE_SIZE is exact for 16-B-per-lane stores, which is what the kernel issues.
"""
import csv, glob, hashlib, json, os, sys

fetch_dir, write_dir, typ, ints, out_path = sys.argv[1], sys.argv[2], sys.argv[3], int(float(sys.argv[4])), sys.argv[5]
needle = sys.argv[6] if len(sys.argv) > 6 else "decode_"


def mean_counter(d, name):
    """per STEP: a step may be more than one dispatch (the multi-dictionary table: the bundles kernel over the table, the same
    kernel again over the cut units, the general kernel for what is left) — everything that matches, over the number of steps =
    the dispatches that last at least half as long as the longest one (a step's main launch)"""
    total, durations = 0.0, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == name:
                total += float(row["Counter_Value"])
                durations.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    steps = sum(1 for x in durations if 2 * x >= max(durations))
    return total / steps, steps


fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
out = {"type": typ, "ints_per_launch": ints, "write_gb": round(write_kb * 1024 / 1e9, 3),
       "fetch_gb_raw": round(fetch_kb * 1024 / 1e9, 3), "fetch_gb_corrected": round(2 * fetch_kb * 1024 / 1e9, 3),
       "launches_averaged": [nf, nw],
       # the build the passes ran: bench.py attaches the file to a line only when its own library is this one
       "lib_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dint_amd",
                                                     "libdint_hip.so"), "rb").read()).hexdigest()[:16],
       "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same bench command ({os.path.basename(os.path.dirname(os.path.abspath(fetch_dir)))}); "
               "FETCH_SIZE x2 per the guide's gfx950 correction (upper bound), WRITE_SIZE exact"}
with open(out_path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
