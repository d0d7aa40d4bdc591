#!/bin/bash
# the fp32 pre-train step for 60 steps under both f32 product modes (same seeds): step time and the meters' means
for m in "" "--f32-exact"; do
  python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10 $m 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mode', '$m' or 'split', 'ms/step', d['ms_per_step'], 'meters', d.get('final_meters'))"
done
