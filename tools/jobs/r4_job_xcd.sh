#!/bin/bash
# round 4: XCD-contiguous tile mapping of the half kernels (field_spread_tiles=2) against the round-robin one (=1)
set -e
mkdir -p gpurun_out/r4_xcd
for sp in 1 2; do
  for prec in f16x2 f16; do
    echo "== field_spread_tiles=$sp precision $prec"
    CED_OPTIONS=field_spread_tiles=$sp PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
  done
  CED_OPTIONS=field_spread_tiles=$sp timeout -k 10 300 python bench.py --no-cpu-baseline --also= --oracle-mode-frames 0 > gpurun_out/r4_xcd/bench_sp$sp.json 2> gpurun_out/r4_xcd/bench_sp$sp.err
  python tools/print_bench_line.py < gpurun_out/r4_xcd/bench_sp$sp.json
done | tee gpurun_out/r4_xcd/xcd_map.txt
