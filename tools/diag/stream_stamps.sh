#!/bin/bash
# phase stamps of the persistent DMA-pipelined convolution (debug build, on the box): ticks per tile per wave
P=self-paced-contrastive-learning_amd
trap 'env -u SPCL_BUILD_DEFS python $P/build.py > /dev/null 2>&1' EXIT INT TERM
SPCL_BUILD_DEFS="-DSPCL_STREAM_STAMPS_BUILD=1" python $P/build.py > /dev/null 2>&1
for wpc in 0 4 6; do
  echo "== SPCL_CONV_STREAM_WPC=$wpc"
  SPCL_CONV_STREAM=1 SPCL_CONV_STREAM_WPC=$wpc SPCL_STREAM_STAMPS=1 python tools/bench_kernels.py fwd dgrad 2>&1 | grep -E "stamps\]" | awk '{k=$3" "$4" "$5; last[k]=$0} END{for(k in last) print last[k]}' | sort
done
