#!/bin/bash
set -e
mkdir -p gpurun_out/r4_xcd
for lib in ced_nerf_amd/libcednerf_hip.so build/ab/lib_rev.so; do
  for prec in f16x2 f16; do
    echo "== $lib precision $prec"
    CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib CED_OPTIONS=field_spread_tiles=2 PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
  done
done | tee gpurun_out/r4_xcd/slot_reverse.txt
