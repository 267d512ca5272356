#!/usr/bin/env python3
"""Development aid: the launches of a profiled run side by side — from a `rocprofv3 --kernel-trace --output-format csv`
directory, the last `--calls` groups of launches (a group = launches less than `--gap-us` apart), each launch with its
start and end relative to the group's first start. Answers "did the three launches of an in-index decode overlap?".
usage: tools/kernel_timeline.py <rocprof dir> [--calls 2] [--gap-us 150] [--match decode_,interpolative,finalize]"""
import argparse, csv, glob

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--calls", type=int, default=2)
ap.add_argument("--gap-us", type=float, default=150.0)
ap.add_argument("--match", default="")
args = ap.parse_args()
rows = []
for f in glob.glob(args.dir + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
want = [m for m in args.match.split(",") if m]
rows = [r for r in rows if not want or any(m in r["Kernel_Name"] for m in want)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > args.gap_us * 1e3:
        groups.append(cur)
        cur = []
    cur.append(r)
    last_end = e if last_end is None else max(last_end, e)
if cur:
    groups.append(cur)
print(f"{len(rows)} launches in {len(groups)} groups; the last {args.calls}:")
for g in groups[-args.calls:]:
    t0 = int(g[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in g)
    print(f"-- group of {len(g)} launches, {(t1 - t0) / 1e3:.1f} us from first start to last end")
    for r in g:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0][-48:]
        print(f"   {s / 1e3:9.1f} .. {e / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s} lds {r.get('LDS_Block_Size', '?'):>7s}  {name}")
