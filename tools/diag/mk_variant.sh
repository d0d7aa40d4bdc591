#!/bin/bash
# build a variant of the library in the build container WITHOUT touching the production objects:
#   mk_variant.sh <tag> <file.hip> "<extra -D flags>"   ->  tools/experiments/libspcl_<tag>.so  (git-ignored, travels with gpurun)
TAG=$1; SRC=$2; DEFS=$3
P=self-paced-contrastive-learning_amd
PRE=16; [ "$SRC" = "supcon.hip" ] && PRE=14
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=$PRE -DSPCL_LAB=1 $DEFS -x hip -c $P/csrc/$SRC -o /tmp/variant_$TAG.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libspcl_$TAG.so /tmp/variant_$TAG.o $(ls $P/build/*.o | grep -v "/${SRC%.*}.o") && echo built tools/experiments/libspcl_$TAG.so
