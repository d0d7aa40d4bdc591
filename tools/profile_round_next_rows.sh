#!/bin/bash
# rocprofv3 --kernel-trace summaries of what the bench does not run (round 6):  bash tools/profile_round_next_rows.sh r06
#   the dense decoder hook's step at Up_conv3 / Up_conv2 (row N3), the pre-train step on the product's own data path (row N2),
#   the validation pass (row N1)  ->  gpurun_out/<tag>_kernel_stats_{dense_up3,dense_up2,datapath,eval}.csv
TAG=${1:-rXX}
OUT=gpurun_out
export TMPDIR=/tmp
run() {  # name, script args...
  local name=$1; shift
  rm -rf $OUT/prof_n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n -- python3 "$@" > $OUT/${TAG}_next_$name.log 2>&1
  python3 tools/prof_summary.py $OUT/prof_n $OUT/${TAG}_kernel_stats_$name.csv 12 | head -14
  rm -rf $OUT/prof_n
}
run dense_up3 tools/diag/dense_step_time.py Up_conv3
run dense_up2 tools/diag/dense_step_time.py Up_conv2
run datapath tools/diag/pretrain_epoch_time.py 2 real
run eval tools/diag/eval_speed.py 8
