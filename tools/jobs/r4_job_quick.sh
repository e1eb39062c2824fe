#!/bin/bash
# round 4: quick check of a kernel change: standalone field throughput (f16x2, f16), bit-exactness of the half modes, bench line
set -e
mkdir -p gpurun_out/r4_quick
for prec in f16x2 f16; do
  PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
done
timeout -k 10 600 python tools/check_half_exact.py > gpurun_out/r4_quick/half_exact.txt 2>&1 || true
tail -n 1 gpurun_out/r4_quick/half_exact.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --also=f32+h16x2,f16 --oracle-mode-frames 0 > gpurun_out/r4_quick/bench.json 2> gpurun_out/r4_quick/bench.err
python tools/print_bench_line.py < gpurun_out/r4_quick/bench.json
