#!/usr/bin/env python3
"""Build check for the kernels that issue loads in inline asm (round 4: decode_segment_lean; round 6: the chunked bundle loop's
bundle_raw_async — decode_multi_bundles_kernel, the in-index bundle and pair kernels): between an asm-issued buffer load
and the next asm s_waitcnt, no compiler-generated instruction may read or write the load's destination registers (a
register copy or a spill there would take the value before it has arrived). Scans the assembly text linearly (the
blocks of the tile loop are laid out in program order); prints every suspect line. Exit code 1 if any.
tests/test_kernel_build_cpu.py runs the same scan (and the register / spill ceilings) in the CPU suite.
usage: tools/check_inflight.py [kernel name substring]   (compiles dint_hip.hip itself)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_NAME = "dint_hip-hip-amdgcn-amd-amdhsa-gfx950.s"


def compile_to_asm(workdir, extra_flags=()):
    """dint_hip.hip -> gfx950 assembly text (hipcc -save-temps; ~25 s, no GPU)."""
    os.makedirs(workdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", f"-I{ROOT}/include",
                    f"-I{ROOT}/dint_amd/csrc/hip", *extra_flags, "-save-temps", "-c", f"{ROOT}/dint_amd/csrc/hip/dint_hip.hip",
                    "-o", os.path.join(workdir, "x.o")], check=True, cwd=workdir)
    return open(os.path.join(workdir, ASM_NAME)).read()


def kernel_body(text, kernel):
    body = text[text.index(f"{kernel}E"):]
    return body[:body.index("s_endpgm")]


def kernel_resources(text):
    """{mangled kernel name: {vgpr_count, vgpr_spill_count, sgpr_count, sgpr_spill_count, private_segment_fixed_size}}"""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n){0,40}?)\s+\.vgpr_spill_count:\s+(\d+)", text):
        blk = m.group(0)
        out[m.group(1)] = {k: int((re.search(r"\." + k + r":\s+(\d+)", blk) or [0, "-1"])[1])
                           for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return out


def regs(tok):
    """v5 -> {5}; v[2:5] -> {2,3,4,5}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def scan(body):
    """-> (asm-issued loads, asm waits, [(line number, instruction, registers touched, line of the load)])"""
    inflight = {}   # reg -> line number of the asm load
    in_asm = False
    bad = []
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
                    # a counted wait retires the oldest loads: conservatively, everything but the most recently issued 8-byte load
                    last = max(inflight.values(), default=0)
                    keep = {r: l for r, l in inflight.items() if l == last}
                    inflight = keep if len(keep) <= 2 else {}
            continue
        if code.startswith("s_waitcnt") and "vmcnt(0)" in code:  # the compiler's own wait for everything
            inflight.clear()
            continue
        touched = regs(code) & set(inflight)
        # (the compiler's own s_waitcnt / scalar instructions name no vector register; an LDS instruction that does is a use)
        if touched and (not code.startswith("s_")):
            bad.append((ln, code, sorted(touched), min(inflight[r] for r in touched)))
    return n_loads, n_waits, bad


def untracked_uses(body, reserved):
    """Instructions OUTSIDE inline asm that name one of the `reserved` vector registers (the chunk touch's v126 / v127, which
    amdgpu_num_vgpr(126) keeps away from the register allocator): there must be none."""
    out, in_asm = [], False
    for ln, line in enumerate(body.split("\n"), 1):
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif s and s[0] not in ";." and not s.endswith(":") and not in_asm and regs(s.split(";")[0]) & set(reserved):
            out.append((ln, s.split(";")[0]))
    return out


if __name__ == "__main__":
    kernel = sys.argv[1] if len(sys.argv) > 1 else "decode_multi_bundles_kernel"
    n_loads, n_waits, bad = scan(kernel_body(compile_to_asm("/tmp/isa_chk"), kernel))
    for ln, code, touched, at in bad:
        print(f"line {ln}: {code}   <- touches in-flight v{touched} (loaded at line {at})")
    print(f"{kernel}: {n_loads} asm-issued loads, {n_waits} asm waits, {len(bad)} suspect instruction(s)")
    sys.exit(1 if bad else 0)
