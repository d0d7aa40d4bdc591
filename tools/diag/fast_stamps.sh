#!/bin/bash
# phase stamps of the conv_fast kernels (debug build on the box; restores the production library afterwards)
P=self-paced-contrastive-learning_amd
F="$P/csrc/conv_fast.hip"
# the production source and library are ALWAYS restored, also when the run is interrupted
BAK=$(mktemp /tmp/cf_orig.XXXXXX.hip)
cp "$F" "$BAK"
trap 'cp "$BAK" "$F"; rm -f "$BAK"; python $P/build.py > /dev/null 2>&1' EXIT INT TERM
sed -i 's/^#define SPCL_FAST_STAMPS_BUILD 0/#define SPCL_FAST_STAMPS_BUILD 1/' $P/csrc/conv_fast.hip
python $P/build.py > /dev/null 2>&1 || echo BUILD FAILED
SPCL_FAST_STAMPS=1 python bench.py --no-cpu-baseline --no-extras --no-graph --steps 1 --warmup 1 2>&1 | grep "conv_fast stamps" | tail -18
