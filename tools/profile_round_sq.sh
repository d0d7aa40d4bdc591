#!/bin/bash
# SQ counters (matrix / vector / LDS pipe activity per kernel symbol) of the three workloads' eager steps:
#   bash tools/profile_round_sq.sh r06  ->  gpurun_out/<tag>_pmc_sq.json (pre-train), _contrastive, _finetune + the text tables
TAG=${1:-rXX}
OUT=gpurun_out
for wl in pretrain contrastive finetune; do
  sfx=""; [ $wl != pretrain ] && sfx=_$wl
  SQ_JSON=$OUT/${TAG}_pmc_sq$sfx.json bash tools/diag/pmc_step_sq.sh --workload $wl > $OUT/${TAG}_pmc_sq$sfx.txt 2>&1
done
rm -rf $OUT/pmc_step_sq
head -12 $OUT/${TAG}_pmc_sq.txt
