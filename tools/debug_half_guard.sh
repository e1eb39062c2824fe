#!/bin/bash
# correctness (tools/debug_half.py) + speed (tools/bench_field.py) of the half kernels for each guarded build
for g in 0 1 3 7; do
  echo "== guard $g"
  CED_NERF_LIB=$PWD/build/libguard$g.so timeout -k 10 300 python tools/debug_half.py 2>&1 | grep -E "^n=(2000000|15068622)" | cut -c1-110 || exit 1
  CED_NERF_LIB=$PWD/build/libguard$g.so PRECISION=f16x2 HALF_VARIANTS=0,1 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep precision | cut -c1-80 || exit 1
  CED_NERF_LIB=$PWD/build/libguard$g.so PRECISION=f16 HALF_VARIANTS=0,1 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep precision | cut -c1-80 || exit 1
done
