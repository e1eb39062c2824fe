#!/bin/bash
# round 4, final kernels: all 48 timed frames against the oracle's f16x2 mode (strided), and the fuzz drivers in forced modes
set -e
mkdir -p gpurun_out/r4_final
for m in f16x2 f16; do
  echo -n "CED_FUZZ_MODE=$m: "
  CED_FUZZ_MODE=$m CED_FUZZ_SEEDS=40 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_configurations" 2>&1 | tail -n 1
done | tee gpurun_out/r4_final/fuzz_half_modes.txt
timeout -k 10 1000 python tools/oracle_mode_all_frames.py > gpurun_out/r4_final/oracle_mode_all_frames.txt 2>&1
tail -n 4 gpurun_out/r4_final/oracle_mode_all_frames.txt
