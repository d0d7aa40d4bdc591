#!/bin/bash
# The round's profile records, in one GPU-box call:  bash tools/profile_round.sh r03   ->  gpurun_out/<tag>_*
#   <tag>_bench.json                      python bench.py (the driver's command line)
#   <tag>_kernel_stats_bench_graph.csv    rocprofv3 --kernel-trace --stats of the same command (per symbol and grid)
#   <tag>_step_timeline.txt               per-launch timeline of one replayed step
#   <tag>_pmc_hbm_traffic.json            HBM bytes per launch: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), eager steps
# Copy what should be judged into profiles/.
TAG=${1:-rXX}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rm -rf $OUT/prof_kt $OUT/prof_f $OUT/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.err
python tools/prof_summary.py $OUT/prof_kt $OUT/${TAG}_kernel_stats_bench_graph.csv 60 > $OUT/${TAG}_kernel_stats_bench_graph.txt 2>&1
python tools/step_timeline.py $OUT/prof_kt flip_pair_stage > $OUT/${TAG}_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_f -- python bench.py --no-cpu-baseline --no-extras --no-roofline --no-graph --steps 3 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_w -- python bench.py --no-cpu-baseline --no-extras --no-roofline --no-graph --steps 3 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_w.err
python tools/pmc_traffic.py $OUT/prof_f $OUT/prof_w $OUT/${TAG}_pmc_hbm_traffic.json > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
rm -rf $OUT/prof_kt $OUT/prof_f $OUT/prof_w   # (raw traces are tens of MB; the summaries above are what is kept)
tail -3 $OUT/${TAG}_step_timeline.txt; head -3 $OUT/${TAG}_kernel_stats_bench_graph.txt; head -5 $OUT/${TAG}_pmc_hbm_traffic.txt
