#!/bin/bash
# one-box sweep of environment switches on one workload: sweep_env.sh WORKLOAD "VAR=val" "VAR=val" ...  (baseline first / last)
WL=$1; shift
run() { env "$@" python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 80 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo -n "baseline "; run X=1
for kv in "$@"; do echo -n "$kv "; run $kv; done
echo -n "baseline "; run X=1
