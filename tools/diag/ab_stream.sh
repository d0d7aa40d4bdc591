#!/bin/bash
# NOTE (round 5): switches read by the LIBRARY (csrc lab_env / lab_flag) exist in LAB builds only -- build one with
#   bash tools/diag/mk_variant_all.sh lab ""   and run this script with it in place (tools/diag/ab_lib.sh swaps libraries);
# the Python-side switches (functional.py, unet.py) work with the shipped library.
# same-box A/B of the persistent DMA-pipelined convolution (csrc/conv_stream.hip): kernel micro-bench + whole step
OUT=gpurun_out
for v in 0 1; do
  echo "== SPCL_CONV_STREAM=$v"
  SPCL_CONV_STREAM=$v python tools/bench_kernels.py fwd dgrad 2>&1 | grep -E "^C(1b|2a|2b|3a)"
done
for i in 1 2; do for v in 0 1; do
  echo -n "SPCL_CONV_STREAM=$v "
  SPCL_CONV_STREAM=$v python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
