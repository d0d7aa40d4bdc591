#!/bin/bash
# fp32 step timelines, same box: the production library, then a variant:  fp32_ab_lib.sh <variant.so> [tagA tagB]
V=$1; A=${2:-prod}; B=${3:-var}
bash tools/diag/timeline_fp32.sh $A
bash tools/diag/with_lib.sh $V bash tools/diag/timeline_fp32.sh $B
bash tools/diag/timeline_fp32.sh ${A}2
