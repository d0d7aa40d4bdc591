#!/bin/bash
# Everything profiles/<tag>_* holds, in one GPU-box call:  bash tools/profile_round_all.sh r04
#   tools/profile_round.sh (pre-train: bench line, kernel stats, per-launch timeline, PMC traffic) + the three other workloads'
#   bench lines, the fine-tune / prostate per-launch timelines and kernel stats, the contrastive kernel stats, the SQ pipe
#   counters of the three workloads (tools/profile_round_sq.sh), the full GPU test log
TAG=${1:-rXX}
OUT=gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh $TAG
for wl in contrastive finetune prostate; do
  timeout 600 python bench.py --workload $wl > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err
done
for wl in finetune prostate; do
  rm -rf $OUT/prof_tl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tl -- python3 bench.py --workload $wl --no-cpu-baseline --no-roofline --steps 30 > /dev/null 2> $OUT/tl_err.txt
  python3 tools/step_timeline.py $OUT/prof_tl conv_pack_multi > $OUT/${TAG}_step_timeline_$wl.txt 2>&1
  [ $wl = finetune ] && python3 tools/prof_summary.py $OUT/prof_tl $OUT/${TAG}_kernel_stats_finetune.csv 60 > /dev/null 2>&1
  rm -rf $OUT/prof_tl
  tail -1 $OUT/${TAG}_step_timeline_$wl.txt
done
# the contrastive 4096 x 128 workload's kernels (rocprofv3 --kernel-trace --stats of its bench command)
rm -rf $OUT/prof_c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c -- python3 bench.py --workload contrastive --no-cpu-baseline --steps 30 > /dev/null 2> $OUT/${TAG}_kt_contrastive.err
python3 tools/prof_summary.py $OUT/prof_c $OUT/${TAG}_kernel_stats_contrastive.csv 20 > /dev/null 2>&1
rm -rf $OUT/prof_c
bash tools/diag/timeline_fp32.sh $TAG   # -> ${TAG}_step_timeline_fp32.txt (f32 storage, split-bf16 products)
bash tools/profile_round_sq.sh $TAG > $OUT/${TAG}_profile_sq.log 2>&1   # SQ pipe counters of the three workloads -> ${TAG}_pmc_sq*.json
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/${TAG}_gputest_full.log
cat $OUT/${TAG}_gputest_full.log | tail -2
