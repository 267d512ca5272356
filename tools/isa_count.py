#!/usr/bin/env python3
"""Development aid: static instruction counts of the decode kernel per source section.
Compiles dint_hip.hip with -DDINT_MARKS (the MARK() comments survive into the assembly) and counts the
instructions between marks, by issue class. usage: tools/isa_count.py [single|multi|<kernel name, e.g. decode_multi_bundles_kernel>] [--dump SECTION]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernel = {"multi": "decode_multi_kernel", "single": "decode_single_kernel"}.get((sys.argv[1:2] or ["single"])[0], (sys.argv[1:2] or ["single"])[0])
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-DDINT_MARKS", *os.environ.get("ISA_FLAGS", "").split(),
                f"-I{ROOT}/include", f"-I{ROOT}/dint_amd/csrc/hip", "-save-temps", "-c",
                f"{ROOT}/dint_amd/csrc/hip/dint_hip.hip", "-o", "/tmp/isa/x.o"], check=True, cwd="/tmp/isa")
text = open("/tmp/isa/dint_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
body = text[text.index(f"{kernel}E"):]
body = body[:body.index("s_endpgm")]
cur, counts, order = "prologue", {}, ["prologue"]
for line in body.split("\n"):
    m = re.search(r"; MARK (\w+)", line)
    if m:
        cur = m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    t = line.strip().split()[0] if line.strip() else ""
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    if dump == cur:
        print(line)
    k = ("valu" if t.startswith("v_") else "branch" if t.startswith("s_cbranch") or t.startswith("s_branch") else
         "wait" if t.startswith("s_waitcnt") or t.startswith("s_nop") else "salu" if t.startswith("s_") else
         "lds" if t.startswith("ds_") else "vmem")
    counts.setdefault(cur, {}).setdefault(k, 0)
    counts[cur][k] += 1
tot = {}
print(f"{'section':16s} {'valu':>5s} {'salu':>5s} {'branch':>6s} {'wait':>5s} {'lds':>4s} {'vmem':>5s} {'all':>5s}")
for c in order:
    d = counts.get(c, {})
    row = [d.get(k, 0) for k in ("valu", "salu", "branch", "wait", "lds", "vmem")]
    for k, v in zip(("valu", "salu", "branch", "wait", "lds", "vmem"), row):
        tot[k] = tot.get(k, 0) + v
    print(f"{c:16s} {row[0]:5d} {row[1]:5d} {row[2]:6d} {row[3]:5d} {row[4]:4d} {row[5]:5d} {sum(row):5d}")
row = [tot.get(k, 0) for k in ("valu", "salu", "branch", "wait", "lds", "vmem")]
print(f"{'total':16s} {row[0]:5d} {row[1]:5d} {row[2]:6d} {row[3]:5d} {row[4]:4d} {row[5]:5d} {sum(row):5d}")
m = re.search(kernel + r"E\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text)
for key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count"):
    mm = re.findall(r"\.name:\s+\S*" + kernel + r"\S*\n(?:.*\n){0,12}?\s+\." + key + r":\s+(\d+)", text)
    if mm:
        print(key, mm[0])
