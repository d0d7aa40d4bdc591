#!/bin/bash
# dense-hook fp32 test, first parameter's error under each Python A/B switch (split mode)
for v in NONE SPCL_UP2_BWD_FUSED SPCL_ACC_FILL SPCL_BN_ACC SPCL_WGRAD_TAILS SPCL_PREPACK SPCL_PACK_AT; do
  for r in 1 2; do
    echo -n "$v=0 run $r: "
    env $v=0 python tools/diag/dense_hook_errs.py 1 2>&1 | grep "_Up5.up.1.weight\|_Up_conv4.conv.4.bias" | awk '{printf "%s %s  ", $1, $NF} END {print ""}'
  done
done
