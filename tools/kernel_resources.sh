#!/bin/bash
# Development aid: registers and spills of every decode kernel (compiles dint_hip.hip to assembly; no GPU needed).
# usage: tools/kernel_resources.sh [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p /tmp/isa_res && cd /tmp/isa_res
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$R/include -I$R/dint_amd/csrc/hip "$@" -save-temps -c $R/dint_amd/csrc/hip/dint_hip.hip -o x.o || exit 1
python3 - <<'PY'
import re
t = open("/tmp/isa_res/dint_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n){0,40}?)\s+\.vgpr_spill_count:\s+(\d+)", t):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [0, "?"])[1]
    name = m.group(1)
    if "decode_" in name or "interpolative" in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} spilled {g('vgpr_spill_count'):>3s}  sgpr {g('sgpr_count'):>4s} spilled {g('sgpr_spill_count'):>3s}  scratch {g('private_segment_fixed_size'):>4s}")
PY
