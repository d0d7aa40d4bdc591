#!/bin/bash
# fp32 step, same box: production, lab weight gradient with 16 x 32 blocks for the three-image tiles (double / single buffered)
bash tools/diag/timeline_fp32.sh base
SPCL_WGRAD_SPLIT_MI1=1 bash tools/diag/with_lib.sh tools/experiments/libspcl_mi1.so bash tools/diag/timeline_fp32.sh mi1d
SPCL_WGRAD_SPLIT_MI1=1 SPCL_WGRAD_DBUF=0 bash tools/diag/with_lib.sh tools/experiments/libspcl_mi1.so bash tools/diag/timeline_fp32.sh mi1s
bash tools/diag/with_lib.sh tools/experiments/libspcl_nomw.so bash tools/diag/timeline_fp32.sh nopre
