#!/bin/bash
# round 4: SQ counters of the f16x2 field kernel on one 15 M-sample launch, and of the marching kernels of single frames
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_pmc
export PRECISION=f16x2
BENCH_ARGS="" KPAT=field_half_kernel bash $R/tools/pmc_field.sh 2>&1 | tail -n 4
cp $R/gpurun_out/pmc_field_generic.json $R/gpurun_out/r4_pmc/field_f16x2_sq_counters.json
bash $R/tools/pmc_march2.sh > $R/gpurun_out/r4_pmc/march_pmc.json 2> $R/gpurun_out/r4_pmc/march_pmc.err || true
head -c 1500 $R/gpurun_out/r4_pmc/field_f16x2_sq_counters.json
