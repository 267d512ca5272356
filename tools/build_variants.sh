#!/bin/bash
# Development aid: timing-experiment builds of libdint_hip.so (results of the EXP variants are wrong
# by construction; they only answer "what does this piece cost"). Output: dint_amd/variants/*.so
# usage: tools/build_variants.sh name:flags [name:flags ...]
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/dint_amd/variants; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -I$R/include -I$R/dint_amd/csrc/hip -shared"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc $F $flags -o $O/$name.so $R/dint_amd/csrc/hip/dint_hip.hip &
done
wait; ls -la $O
