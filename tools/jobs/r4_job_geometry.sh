#!/bin/bash
# round 4: launch geometry of the half kernels after the pair loads of dense levels (registers: tools/kernel_schedule.py)
set -e
mkdir -p gpurun_out/r4_ab
{
for prec in f16x2 f16; do
  HALF_VARIANTS=3,2,1,3,2 PRECISION=$prec timeout -k 10 300 python tools/bench_field.py 2>&1 | grep "Gsamples" | sed 's/; vs f32.*//'
done
for hv in 3 2 1 3 2; do
  echo "== C2 f16x2 half_variant $hv"
  CED_HALF_VARIANT=$hv timeout -k 10 300 python bench.py --no-cpu-baseline --also= --oracle-mode-frames 0 --no-single-frame 2>/dev/null | python tools/print_bench_line.py
done
for hv in 3 1 2 3 1; do
  for cfg in "--scene hypernerf --width 536 --height 960" "--scene dynerf --width 1352 --height 1014"; do
    echo "== half_variant $hv $cfg"
    CED_HALF_VARIANT=$hv timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also= --oracle-mode-frames 0 $cfg 2>/dev/null | python tools/print_bench_line.py
  done
done
} | tee gpurun_out/r4_ab/half_geometry.txt
