#!/bin/bash
# SQ counters of every kernel of the eager pre-train step (or --workload W): one rocprofv3 --pmc pass per group, counters only
export TMPDIR=/tmp
OUT=gpurun_out/pmc_step_sq
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-roofline --no-graph --steps 3 --warmup 1 "$@" > $OUT/g$i.log 2>&1
done
python3 tools/pmc_sq_survey.py ${SQ_JSON:+--json $SQ_JSON} $OUT/g1 $OUT/g2
