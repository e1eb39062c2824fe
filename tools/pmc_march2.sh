#!/bin/bash
# SQ / TCC counters of the marching kernels of single frames (tools/iter_profile.py), averaged per kernel name
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py ${SCENE:-dnerf} > $OUT.p$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py march_ $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
