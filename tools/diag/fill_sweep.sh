for v in 256 512 1100; do
  echo "== SPCL_CONV_FAST_FILL_MAX=$v"
  SPCL_CONV_FAST_FILL_MAX=$v bash tools/diag/timeline.sh fill_$v
  grep "conv3x3_fast_kernel<64,7" gpurun_out/fill_${v}_step_timeline.txt | cut -c1-100
done
