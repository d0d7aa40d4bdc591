#!/bin/bash
# NOTE (round 5): switches read by the LIBRARY (csrc lab_env / lab_flag) exist in LAB builds only -- build one with
#   bash tools/diag/mk_variant_all.sh lab ""   and run this script with it in place (tools/diag/ab_lib.sh swaps libraries);
# the Python-side switches (functional.py, unet.py) work with the shipped library.
# one-box sweep of environment switches on one workload: sweep_env.sh WORKLOAD "VAR=val" "VAR=val" ...  (baseline first / last)
WL=$1; shift
run() { env "$@" python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 80 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo -n "baseline "; run X=1
for kv in "$@"; do echo -n "$kv "; run $kv; done
echo -n "baseline "; run X=1
