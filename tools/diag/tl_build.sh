#!/bin/bash
# per-launch timeline lines of the replayed pre-train step for several BUILDS (on-box rebuild): tl_build.sh PATTERN "<defs>" ...
PAT=$1; shift
trap 'env -u SPCL_BUILD_DEFS python self-paced-contrastive-learning_amd/build.py > /dev/null 2>&1' EXIT INT TERM
export TMPDIR=/tmp
for v in "$@"; do
  SPCL_BUILD_DEFS="$v" python self-paced-contrastive-learning_amd/build.py --force > /dev/null 2>&1
  rm -rf gpurun_out/prof_tl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 > /dev/null 2> gpurun_out/tl_err.txt
  echo "== defs='$v'"; python tools/step_timeline.py gpurun_out/prof_tl flip_pair_stage | grep -E "$PAT|launches"
  rm -rf gpurun_out/prof_tl
done
