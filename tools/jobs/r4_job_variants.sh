#!/bin/bash
set -e
mkdir -p gpurun_out/r4_ab
for prec in f16x2 f16; do
  CED_NERF_LIB=$GRAFT_REPO_ROOT/build/ab/lib_nt1.so HALF_VARIANTS=0,1,2,3 PRECISION=$prec timeout -k 10 300 python tools/bench_field.py 2>&1 | grep "Gsamples"
done | tee gpurun_out/r4_ab/half_variants.txt
