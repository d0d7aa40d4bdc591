#!/bin/bash
# same-box A/B of the code-object target: ab_arch.sh gfx950 gfx950:xnack- [rounds]
A=$1; B=$2; R=${3:-2}
# whatever ends this script (normal exit, Ctrl-C, the box's timeout) leaves the PRODUCTION library behind: build.py's flag
# stamps make the plain build below recompile every object the experiment touched
trap 'env -u SPCL_BUILD_DEFS -u SPCL_BUILD_ARCH -u SPCL_BUILD_NOSLP python self-paced-contrastive-learning_amd/build.py > /dev/null 2>&1' EXIT INT TERM
for i in $(seq $R); do for v in "$A" "$B"; do
  SPCL_BUILD_ARCH="$v" python self-paced-contrastive-learning_amd/build.py --force > /dev/null 2>&1
  echo -n "arch='$v' "
  python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'median', d['replay_us']['median'], 'p10', d['replay_us']['p10'])"
done; done
