#!/bin/bash
# same-box A/B of compile-time variants of one csrc file: ab_macro.sh <file.hip> "<M1=V1 M2=V2>" "<M1=V3 ...>" ...
# (each set rewrites the `#define M V` defaults in the file, rebuilds the library, runs the default bench twice; the file is
#  restored at the end).  Box-to-box variance is +-3 %: only numbers from one call compare.
P=self-paced-contrastive-learning_amd
F="$P/csrc/$1"; shift
# the production source and library are ALWAYS restored, also when the run is interrupted
BAK=$(mktemp /tmp/ab_orig.XXXXXX.hip)
cp "$F" "$BAK"
trap 'cp "$BAK" "$F"; rm -f "$BAK"; python $P/build.py > /dev/null 2>&1' EXIT INT TERM
for round in 1 2; do
for set in "$@"; do
  cp "$BAK" "$F"
  for kv in $set; do m=${kv%%=*}; v=${kv##*=}; sed -i "s/^#define $m .*/#define $m $v/" "$F"; done
  python $P/build.py > /dev/null 2>&1 || echo "BUILD FAILED"
  echo -n "[$set] "
  for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --steps 50 2>/dev/null | grep -o "ms_per_step[^,]*" | tr "\n" " "; done; echo
done
done
