#!/bin/bash
# SQ counters of the one-pass block-1 backward alone (tools/diag/conv16_phases.py QUICK=1), one rocprofv3 --pmc pass per group
# (counters only: no trace domains beside them).  usage: bash tools/diag/pmc_conv16.sh [kernel substring]
SUB=${1:-conv16_bwd}
export TMPDIR=/tmp QUICK=1
OUT=gpurun_out/pmc_conv16
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i --output-format csv -- python3 tools/diag/conv16_phases.py > $OUT/g$i.log 2>&1
  python3 tools/pmc_kernel.py $SUB $OUT/g$i
done
