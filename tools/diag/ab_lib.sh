#!/bin/bash
# same-box A/B of two prebuilt libraries (built in the build container: tools/diag/mk_variant.sh): ab_lib.sh A.so B.so [rounds] [bench args]
# the production library is put back at the end (and by the trap if interrupted)
A=$1; B=$2; R=${3:-3}; shift 3
LIB=self-paced-contrastive-learning_amd/lib/libspcl_hip.so
cp $LIB /tmp/libspcl_prod.so
trap 'cp /tmp/libspcl_prod.so '$LIB EXIT INT TERM
for i in $(seq $R); do for v in $A $B; do
  cp $v $LIB
  echo -n "$(basename $v) "
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --steps 100 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('replay_us') or {}; print(d['ms_per_step'], 'median', r.get('median'), 'p10', r.get('p10'))"
done; done
