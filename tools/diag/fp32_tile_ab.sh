#!/bin/bash
# fp32 step: production library, then the lab build with 7-row tiles at 224^2 (same box)
bash tools/diag/timeline_fp32.sh base
SPCL_CONV_TH7_MAXH=224 bash tools/diag/with_lib.sh tools/experiments/libspcl_lab.so bash tools/diag/timeline_fp32.sh th7
