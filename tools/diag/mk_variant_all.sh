#!/bin/bash
# build a variant of the WHOLE library with extra -D flags, without touching the production objects:
#   mk_variant_all.sh <tag> "<extra -D flags>"   ->  tools/experiments/libspcl_<tag>.so  (git-ignored, travels with gpurun)
TAG=$1; DEFS=$2
P=self-paced-contrastive-learning_amd
O=/tmp/variant_all_$TAG; mkdir -p $O
pids=()
for f in $P/csrc/*.hip $P/csrc/*.cpp; do
  b=$(basename ${f%.*}); PRE=16; [ "$b" = "supcon" ] && PRE=14
  X="-x hip"; EXTRA="-mllvm -amdgpu-kernarg-preload-count=$PRE"; [ "${f##*.}" = "cpp" ] && { X=""; EXTRA=""; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $EXTRA -DSPCL_LAB=1 $DEFS $X -c $f -o $O/$b.o &
  pids+=($!)
  [ ${#pids[@]} -ge 6 ] && { wait ${pids[0]}; pids=("${pids[@]:1}"); }
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libspcl_$TAG.so $O/*.o && echo built tools/experiments/libspcl_$TAG.so
