#!/bin/bash
# usage: tools/isa.sh [kernel-name-substring ...]   - gfx950 assembly of csrc/kernels.hip into /tmp/isa/kernels.s
# (cross-compiles without a GPU) and tools/isa_stats.py for each substring given.
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only \
  "$R/quickchem_amd/csrc/kernels.hip" -o /tmp/isa/kernels.s 2>&1 | grep -v "hip-link" 
for k in "$@"; do python3 "$R/tools/isa_stats.py" /tmp/isa/kernels.s "$k"; done
