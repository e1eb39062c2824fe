#!/bin/bash
# round 4: L1 (TCP) / L2 (TCC) / TA counters of the f16x2 field kernel on one 15 M-sample launch
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcc
mkdir -p $OUT $R/gpurun_out/r4_pmc
cd /tmp && export TMPDIR=/tmp
export PRECISION=f16x2
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_READ_sum" \
           "TCP_TCR_RDRET_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/tools/bench_field.py > $OUT.p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $R/tools/pmc_generic.py field_half_kernel $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 > $R/gpurun_out/r4_pmc/field_f16x2_cache.json
cat $R/gpurun_out/r4_pmc/field_f16x2_cache.json
