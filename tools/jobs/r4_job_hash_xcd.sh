#!/bin/bash
# round 4: XCD-contiguous block mapping of the standalone hash kernels (encode, table gradient, position gradient) on / off
set -e
mkdir -p gpurun_out/r4_ab
{
for m in 0 1 0 1; do
  echo "== CED_HASH_XCD_MAP=$m"
  CED_HASH_XCD_MAP=$m timeout -k 10 200 python tools/bench_hash_backward.py 2>&1 | grep "hash backward"
  CED_HASH_XCD_MAP=$m timeout -k 10 200 python tools/bench_hash_backward_levels.py 2>&1 | grep "all 16"
  CED_HASH_XCD_MAP=$m N_RAYS=262144 timeout -k 10 300 python tools/bench_train.py 2>&1 | grep train_step | cut -c1-80
done
} | tee gpurun_out/r4_ab/hash_xcd.txt
