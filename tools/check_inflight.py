#!/usr/bin/env python3
"""Build check for the kernels that issue loads in inline asm (round 4: decode_segment_lean; round 6: the chunked bundle loop's
bundle_raw_async — decode_multi_bundles_kernel, the in-index bundle kernels): between an asm-issued buffer load
and the next asm s_waitcnt, no compiler-generated instruction may read or write the load's destination registers (a
register copy or a spill there would take the value before it has arrived). Scans the assembly text linearly (the
blocks of the tile loop are laid out in program order); prints every suspect line. Exit code 1 if any.
usage: tools/check_inflight.py [kernel name substring]   (compiles dint_hip.hip itself)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernel = sys.argv[1] if len(sys.argv) > 1 else "decode_single_kernel"
os.makedirs("/tmp/isa_chk", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", f"-I{ROOT}/include",
                f"-I{ROOT}/dint_amd/csrc/hip", "-save-temps", "-c", f"{ROOT}/dint_amd/csrc/hip/dint_hip.hip", "-o", "/tmp/isa_chk/x.o"],
               check=True, cwd="/tmp/isa_chk")
text = open("/tmp/isa_chk/dint_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
body = text[text.index(f"{kernel}E"):]
body = body[:body.index("s_endpgm")]


def regs(tok):
    """v5 -> {5}; v[2:5] -> {2,3,4,5}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


inflight = {}   # reg -> line number of the asm load
in_asm = False
bad = 0
n_loads = n_waits = 0
for ln, line in enumerate(body.split("\n"), 1):
    s = line.strip()
    if s.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if s.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not s or s[0] in ";." or s.endswith(":"):
        continue
    code = s.split(";")[0]
    if in_asm:
        if code.startswith(("buffer_load", "global_load")):
            dst = code.split()[1].rstrip(",")
            for r in regs(dst):
                inflight[r] = ln
            n_loads += 1
        elif code.startswith("s_waitcnt") and "vmcnt" in code:
            n_waits += 1
            if "vmcnt(0)" in code:
                inflight.clear()
            else:
                # a counted wait retires the heads (the oldest loads): conservatively, only the registers of 16-byte loads
                # issued more than 2 loads ago — simpler: retire everything but the most recently issued 8-byte load
                last = max(inflight.values(), default=0)
                keep = {r: l for r, l in inflight.items() if l == last}
                inflight = keep if len(keep) <= 2 else {}
        continue
    touched = regs(code) & set(inflight)
    if touched and not code.startswith(("s_", "ds_")) or (touched and code.startswith("ds_")):
        print(f"line {ln}: {code}   <- touches in-flight v{sorted(touched)} (loaded at line {min(inflight[r] for r in touched)})")
        bad += 1
print(f"{kernel}: {n_loads} asm-issued loads, {n_waits} asm waits, {bad} suspect instruction(s)")
sys.exit(1 if bad else 0)
