#!/usr/bin/env python3
"""Development aid: the basic blocks of a kernel's assembly between two line numbers — label, VALU / all instruction
counts, the section mark in force, how the block ends — for walking the common path of a tile by hand.
usage: tools/isa_blocks.py file.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
a, b = int(sys.argv[2]), int(sys.argv[3])
blocks, cur = [], None
mark = "?"
def new(label, i):
    global cur
    cur = {"label": label, "line": i, "valu": 0, "all": 0, "lds": 0, "vmem": 0, "end": "", "mark": mark}
    blocks.append(cur)
new("(entry)", a)
for i in range(a, b):
    line = lines[i]
    m = re.search(r"; MARK (\w+)", line)
    if m:
        mark = m.group(1)
        if cur["all"] == 0: cur["mark"] = mark
        continue
    s = line.strip()
    if not s or s[0] == ";": continue
    if re.match(r"^\.?[A-Za-z_][\w.]*:", s):
        new(s.split(":")[0], i)
        continue
    t = s.split()[0]
    if cur["end"]:  # an instruction behind a branch: a new (fall-through) block
        new("(ft)", i)
    cur["all"] += 1
    if t.startswith("v_"): cur["valu"] += 1
    elif t.startswith("ds_"): cur["lds"] += 1
    elif t.startswith("buffer_") or t.startswith("global_") or t.startswith("flat_") or t.startswith("scratch_"): cur["vmem"] += 1
    if t.startswith("s_cbranch") or t == "s_branch" or t.startswith("s_setpc"):
        cur["end"] = s
for blk in blocks:
    print(f"{blk['line']:6d} {blk['label']:14s} {blk['mark']:14s} valu {blk['valu']:3d} lds {blk['lds']:2d} vmem {blk['vmem']:2d} all {blk['all']:3d}  {blk['end']}")
