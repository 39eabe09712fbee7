#!/bin/bash
# SQ / LDS / TCC counters of ONE op (tools/one_op.py <op>) in separate rocprofv3 --pmc passes (8 SQ slots per pass; FETCH_SIZE
# and WRITE_SIZE never share a pass).  usage (on the GPU box): bash tools/pmc_kernel.sh <op> <kernel substring> [out name]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OP=$1; SUB=$2; NAME=${3:-$1}
OUT=$R/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set -d $OUT/p$i -o p -- python $R/tools/one_op.py $OP 5 > $OUT/p$i.log 2>&1
  db=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db "$SUB"
done 2>&1 | tee $OUT/summary.txt
