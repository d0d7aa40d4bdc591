#!/bin/bash
# NOTE (round 5): switches read by the LIBRARY (csrc lab_env / lab_flag) exist in LAB builds only -- build one with
#   bash tools/diag/mk_variant_all.sh lab ""   and run this script with it in place (tools/diag/ab_lib.sh swaps libraries);
# the Python-side switches (functional.py, unet.py) work with the shipped library.
# same-box A/B of an environment switch on another workload: ab_env_wl.sh WORKLOAD VAR v1 v2 [rounds]
WL=$1; VAR=$2; A=$3; B=$4; R=${5:-2}
for i in $(seq $R); do for v in $A $B; do
  echo -n "$WL $VAR=$v "
  env $VAR=$v python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
