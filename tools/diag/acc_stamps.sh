#!/bin/bash
# phase stamps of the conv_fast kernels with and without the BatchNorm accumulator blocks (prebuilt stamp variant:
# tools/diag/mk_variant.sh stamps conv_fast.hip "-DSPCL_FAST_STAMPS_BUILD=1"); restores the production library
LIB=self-paced-contrastive-learning_amd/lib/libspcl_hip.so
cp $LIB /tmp/libspcl_prod.so
trap 'cp /tmp/libspcl_prod.so '$LIB EXIT INT TERM
cp tools/experiments/libspcl_stamps.so $LIB
for v in 0 1; do
  echo "== SPCL_BN_ACC=$v"
  SPCL_BN_ACC=$v SPCL_FAST_STAMPS=1 python bench.py --no-cpu-baseline --no-extras --no-roofline --no-graph --steps 1 --warmup 1 2>&1 | grep "conv_fast stamps" | grep "m1" | tail -5 | cut -c1-230
done
