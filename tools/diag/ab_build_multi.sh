#!/bin/bash
# same-box comparison of several BUILDS on two workloads: ab_build_multi.sh "<defs 1>" "<defs 2>" ...   (on-box rebuild each)
trap 'env -u SPCL_BUILD_DEFS python self-paced-contrastive-learning_amd/build.py > /dev/null 2>&1' EXIT INT TERM
for r in 1 2; do for v in "$@"; do
  SPCL_BUILD_DEFS="$v" python self-paced-contrastive-learning_amd/build.py --force > /dev/null 2>&1
  echo -n "defs='$v' pretrain "
  python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], end=' ')"
  echo -n " finetune "
  python bench.py --workload finetune --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
