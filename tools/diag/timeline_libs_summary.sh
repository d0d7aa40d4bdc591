#!/bin/bash
# timeline_lib.sh for several prebuilt libraries + the per-launch table side by side for the lines matching a pattern
#   timeline_libs_summary.sh "<grep -E pattern>" A.so B.so ...
PAT=$1; shift
bash tools/diag/timeline_lib.sh "" "$@" > gpurun_out/tl_libs.txt 2>&1
grep -E "^==|launches" gpurun_out/tl_libs.txt
for v in "$@"; do nm=$(basename $v .so); echo "== $nm"; grep -E "$PAT" gpurun_out/tl_$nm.txt | cut -c1-96; done
