#!/bin/bash
# same-box A/B/C... of prebuilt libraries: ab_libs3.sh ROUNDS lib1.so lib2.so ...   (production library restored at the end)
R=$1; shift
LIB=self-paced-contrastive-learning_amd/lib/libspcl_hip.so
cp $LIB /tmp/libspcl_prod.so
trap 'cp /tmp/libspcl_prod.so '$LIB EXIT INT TERM
for i in $(seq $R); do for v in "$@"; do
  cp $v $LIB
  echo -n "$(basename $v) "
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('replay_us') or {}; print(d['ms_per_step'], 'median', r.get('median'), 'p10', r.get('p10'))"
done; done
