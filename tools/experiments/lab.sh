#!/bin/bash
# compile an experiment for gfx950 and print per-kernel register/LDS use:  tools/experiments/lab.sh convlab
set -e
D=$(cd "$(dirname "$0")" && pwd)
mkdir -p /tmp/lab && cd /tmp/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -save-temps -o "$D/$1" "$D/$1.hip" 2>&1 | grep -E "error|warning: v" || true
grep -E "^\s+\.(vgpr_count|sgpr_count|name|vgpr_spill_count|group_segment_fixed_size):" /tmp/lab/$1-hip-amdgcn-amd-amdhsa-gfx950.s \
  | paste - - - - - | awk '{print $2, $4, $6, $8, $10}'
