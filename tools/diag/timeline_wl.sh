#!/bin/bash
# per-launch timelines of another workload's replayed step for two settings of ONE environment switch, same box:
#   bash tools/diag/timeline_wl.sh finetune SPCL_LAZY_HEAD 0 1 [grep pattern]   ->  gpurun_out/tl_<WL>_<VAR>_<value>.txt
WL=$1; VAR=$2; A=$3; B=$4; PAT=${5:-conv}
OUT=gpurun_out
export TMPDIR=/tmp
for v in $A $B; do
  rm -rf $OUT/prof_tl
  export $VAR=$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tl -- python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 30 > /dev/null 2> $OUT/tl_err.txt
  python tools/step_timeline.py $OUT/prof_tl conv_pack_multi > $OUT/tl_${WL}_${VAR}_$v.txt 2>&1
  rm -rf $OUT/prof_tl
  echo "== $VAR=$v"; grep -E "$PAT" $OUT/tl_${WL}_${VAR}_$v.txt; tail -1 $OUT/tl_${WL}_${VAR}_$v.txt
done
