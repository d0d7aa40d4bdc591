#!/bin/bash
# SLP packing on / off for the one-pass kernel, 1 / 2 / 4 waves per tile (experiment of round 3)
trap 'env -u SPCL_BUILD_DEFS -u SPCL_BUILD_ARCH -u SPCL_BUILD_NOSLP python self-paced-contrastive-learning_amd/build.py > /dev/null 2>&1' EXIT INT TERM
for slp in "" conv16_bwd.hip; do
  SPCL_BUILD_NOSLP=$slp python self-paced-contrastive-learning_amd/build.py --force > /dev/null 2>&1
  for nw in 1 2 4; do echo "NOSLP='$slp' NW=$nw"; SPCL_CONV16_NW=$nw timeout 300 python -m pytest tests/test_gpu_encoder.py -q -m gpu -k "conv16" 2>&1 | tail -1; QUICK=1 SPCL_CONV16_NW=$nw timeout 300 python tools/diag/conv16_phases.py 2>&1 | grep "dbg"; done
done
