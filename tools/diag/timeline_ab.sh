#!/bin/bash
# per-launch timelines of the replayed step for two settings of ONE environment switch, same box:
#   bash tools/diag/timeline_ab.sh SPCL_CONV_STREAM 0 1 [grep pattern]   ->  gpurun_out/tl_<VAR>_<value>.txt
VAR=$1; A=$2; B=$3; PAT=${4:-conv}
OUT=gpurun_out
export TMPDIR=/tmp
for v in $A $B; do
  rm -rf $OUT/prof_tl
  export $VAR=$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tl -- python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 > /dev/null 2> $OUT/tl_err.txt
  python tools/step_timeline.py $OUT/prof_tl flip_pair_stage > $OUT/tl_${VAR}_$v.txt 2>&1
  rm -rf $OUT/prof_tl
  echo "== $VAR=$v"; grep -E "$PAT" $OUT/tl_${VAR}_$v.txt; tail -1 $OUT/tl_${VAR}_$v.txt
done
