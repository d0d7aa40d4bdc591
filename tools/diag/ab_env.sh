#!/bin/bash
# NOTE (round 5): switches read by the LIBRARY (csrc lab_env / lab_flag) exist in LAB builds only -- build one with
#   bash tools/diag/mk_variant_all.sh lab ""   and run this script with it in place (tools/diag/ab_lib.sh swaps libraries);
# the Python-side switches (functional.py, unet.py) work with the shipped library.
# same-box A/B of an environment switch: ab_env.sh VAR v1 v2 [rounds]  -> per run: mean ms per step and the MEDIAN of 100
# single-replay HIP-event times (robust against the occasional slow run on a shared box)
VAR=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for v in $A $B; do
  echo -n "$VAR=$v "
  env $VAR=$v python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('replay_us') or {}; print(d['ms_per_step'], 'median', r.get('median'), 'p10', r.get('p10'))"
done; done
