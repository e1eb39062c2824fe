#!/bin/bash
# round 4: standalone field-kernel throughput of diagnostic libraries (tools/ab_build.sh): usage r4_job_ab_libs.sh <lib>...
set -e
mkdir -p gpurun_out/r4_ab
for lib in "$@"; do
  for prec in f16x2 f16; do
    echo "== $lib precision $prec"
    CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
  done
done | tee -a gpurun_out/r4_ab/ab_libs.txt
