#!/bin/bash
# round 4: instruction-cache / fetch counters and memory-instruction levels of the f16x2 field kernel (one 15 M-sample launch)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmci
mkdir -p $OUT $R/gpurun_out/r4_pmc
cd /tmp && export TMPDIR=/tmp
export PRECISION=f16x2
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/tools/bench_field.py > $OUT.p$i.log 2>&1
  echo "pass $i done"
done
python3 $R/tools/pmc_generic.py field_half_kernel $OUT/p1 $OUT/p2 > $R/gpurun_out/r4_pmc/field_f16x2_icache.json
cat $R/gpurun_out/r4_pmc/field_f16x2_icache.json
