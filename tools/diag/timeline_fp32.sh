#!/bin/bash
# per-launch timeline of one replayed fp32 step:  bash tools/diag/timeline_fp32.sh <tag>  -> gpurun_out/<tag>_step_timeline_fp32.txt
TAG=${1:-tl}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-roofline --steps 20 > /dev/null 2> $OUT/${TAG}_kt.err
python tools/step_timeline.py $OUT/prof_kt flip_pair_stage > $OUT/${TAG}_step_timeline_fp32.txt 2>&1
rm -rf $OUT/prof_kt
tail -1 $OUT/${TAG}_step_timeline_fp32.txt
