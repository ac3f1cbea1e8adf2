#!/bin/bash
# tools/cc_kernels.sh [pattern] - compile conv_kernels.hip for gfx950 with temporaries in /tmp and print registers / spills / scratch of
# the kernels whose mangled name matches the pattern (development aid; the product build is hyperpose_amd/build.py)
cd /root/repo/hyperpose_amd/csrc || exit 1
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-function -I../../include -c ${2:-conv_kernels.hip} -o /tmp/ck.o -save-temps=obj 2>&1 | grep -v "^$" | grep "error" -A4 | head -40
f=/tmp/$(basename ${2:-conv_kernels.hip} .hip)-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s*\.(vgpr_count|private_segment_fixed_size|name|vgpr_spill_count):" $f | paste - - - - | grep "${1:-.}"
