#!/bin/bash
# k-loop timing experiments: stamps with parts of the step removed (SPCL_FAST_DBG bits; results are wrong by design)
P=self-paced-contrastive-learning_amd
F="$P/csrc/conv_fast.hip"
# the production source and library are ALWAYS restored, also when the run is interrupted
BAK=$(mktemp /tmp/cf_orig.XXXXXX.hip)
cp "$F" "$BAK"
trap 'cp "$BAK" "$F"; rm -f "$BAK"; python $P/build.py > /dev/null 2>&1' EXIT INT TERM
for dbg in 0 1 2 3; do
  cp "$BAK" "$F"
  sed -i "s/^#define SPCL_FAST_STAMPS_BUILD 0/#define SPCL_FAST_STAMPS_BUILD 1/; s/^#define SPCL_FAST_DBG 0/#define SPCL_FAST_DBG $dbg/" $P/csrc/conv_fast.hip
  python $P/build.py > /dev/null 2>&1 || echo BUILD FAILED
  echo "== SPCL_FAST_DBG=$dbg"
  SPCL_FAST_STAMPS=1 python bench.py --no-cpu-baseline --no-extras --no-graph --steps 1 --warmup 1 2>&1 | grep "conv_fast stamps" | grep "<64,7,2,m1,4> 14x14\|<64,7,1,m1,4>" | cut -c1-215
done
