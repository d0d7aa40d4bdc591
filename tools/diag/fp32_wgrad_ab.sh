#!/bin/bash
# fp32 step: production library, then the lab build's weight-gradient tiles of 8 rows, double and single buffered (same box)
bash tools/diag/timeline_fp32.sh base
SPCL_WGRAD_TH=8 bash tools/diag/with_lib.sh tools/experiments/libspcl_lab2.so bash tools/diag/timeline_fp32.sh th8d
SPCL_WGRAD_TH=8 SPCL_WGRAD_DBUF=0 bash tools/diag/with_lib.sh tools/experiments/libspcl_lab2.so bash tools/diag/timeline_fp32.sh th8s
SPCL_WGRAD_DBUF=0 bash tools/diag/with_lib.sh tools/experiments/libspcl_lab2.so bash tools/diag/timeline_fp32.sh th14s
