#!/bin/bash
# NOTE (round 5): switches read by the LIBRARY (csrc lab_env / lab_flag) exist in LAB builds only -- build one with
#   bash tools/diag/mk_variant_all.sh lab ""   and run this script with it in place (tools/diag/ab_lib.sh swaps libraries);
# the Python-side switches (functional.py, unet.py) work with the shipped library.
# same-box comparison of the one-pass block-1 backward (csrc/conv16_bwd.hip) with 1 / 2 / 4 waves per tile against the two
# separate launches: median of 100 single-replay HIP-event times per run
for i in 1 2; do
  for cfg in "SPCL_CONV16_FUSED=0" "SPCL_CONV16_NW=1" "SPCL_CONV16_NW=2" "SPCL_CONV16_NW=4"; do
    echo -n "$cfg "
    env $cfg python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'median', d['replay_us']['median'], 'p10', d['replay_us']['p10'])"
  done
done
