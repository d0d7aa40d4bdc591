#!/bin/bash
# fp32 step timelines, same box: production, then each variant library:  fp32_ab_libs.sh A.so B.so ...
bash tools/diag/timeline_fp32.sh prod
for v in "$@"; do
  nm=$(basename $v .so)
  bash tools/diag/with_lib.sh $v bash tools/diag/timeline_fp32.sh $nm
done
bash tools/diag/timeline_fp32.sh prod2
