#!/bin/bash
# same-box A/B of an environment switch on another workload: ab_env_wl.sh WORKLOAD VAR v1 v2 [rounds]
WL=$1; VAR=$2; A=$3; B=$4; R=${5:-2}
for i in $(seq $R); do for v in $A $B; do
  echo -n "$WL $VAR=$v "
  env $VAR=$v python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
