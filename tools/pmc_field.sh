#!/bin/bash
# PMC passes over the field micro-benchmark (counters only: no trace domains alongside --pmc).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_field.py $BENCH_ARGS > $OUT.p$i.log 2>&1
  echo "pass $i done"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py ${KPAT:-field_kernel} $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 > $GRAFT_REPO_ROOT/gpurun_out/pmc_field_generic.json
