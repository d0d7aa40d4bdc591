#!/bin/bash
# per-launch timelines of the replayed step for prebuilt libraries, same box: timeline_lib.sh "<bench args>" A.so B.so ... -> gpurun_out/tl_<name>.txt
ARGS=$1; shift
LIB=self-paced-contrastive-learning_amd/lib/libspcl_hip.so
cp $LIB /tmp/libspcl_prod.so
trap 'cp /tmp/libspcl_prod.so '$LIB EXIT INT TERM
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  cp $v $LIB
  nm=$(basename $v .so)
  rm -rf $OUT/prof_tl
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tl -- python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 $ARGS > /dev/null 2> $OUT/tl_err.txt
  python3 tools/step_timeline.py $OUT/prof_tl flip_pair_stage > $OUT/tl_$nm.txt 2>&1
  rm -rf $OUT/prof_tl
  echo "== $nm"; tail -1 $OUT/tl_$nm.txt
done
