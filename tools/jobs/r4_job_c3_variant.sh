#!/bin/bash
# round 4: C3 / C4 (time-embedding kernels, 16 registers spilt at 768 threads) with 2 x 512 against 2 x 768 threads
set -e
mkdir -p gpurun_out/r4_ab
for hv in 0 1; do
  for cfg in "--scene hypernerf --width 536 --height 960" "--scene dynerf --width 1352 --height 1014"; do
    echo "== half_variant $hv $cfg"
    CED_HALF_VARIANT=$hv timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also= --oracle-mode-frames 0 $cfg 2>/dev/null | python tools/print_bench_line.py
  done
done | tee gpurun_out/r4_ab/c3_half_variant.txt
