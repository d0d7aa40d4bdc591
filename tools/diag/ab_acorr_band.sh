#!/bin/bash
P=self-paced-contrastive-learning_amd
F="$P/csrc/bn.hip"
BAK=$(mktemp /tmp/bn_orig.XXXXXX.hip); cp "$F" "$BAK"
trap 'cp "$BAK" "$F"; rm -f "$BAK"; python $P/build.py > /dev/null 2>&1' EXIT INT TERM
for band in 28 14 8 56; do
  cp "$BAK" "$F"; sed -i "s/constexpr int ACORR_BAND = 28, ACORR_MAXW = 256;/constexpr int ACORR_BAND = $band, ACORR_MAXW = 256;/" "$F"
  python $P/build.py > /dev/null 2>&1 || echo BUILD FAILED
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 > /dev/null 2>&1
  python tools/step_timeline.py gpurun_out/prof_kt flip_pair_stage > gpurun_out/tl_band.txt; rm -rf gpurun_out/prof_kt
  echo "band $band: $(grep -h 'autocorr' gpurun_out/tl_band.txt | head -1)"
done
