#!/bin/bash
set -e
mkdir -p gpurun_out/r4_ab
for lib in build/ab/lib_base.so ced_nerf_amd/libcednerf_hip.so; do
  for prec in f32 f32+h16x2; do
    echo "== $lib precision $prec"
    CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
  done
done | tee gpurun_out/r4_ab/f32_prefetch.txt
