#!/bin/bash
# round 4: the split's remainder as one v_fma_mix{lo,hi}_f16: exhaustive probe, bit-exactness of the half modes, bench
set -e
mkdir -p gpurun_out/r4_mix
timeout -k 10 120 tools/probes/split_mix_exhaustive > gpurun_out/r4_mix/split_mix_exhaustive.txt
cat gpurun_out/r4_mix/split_mix_exhaustive.txt
timeout -k 10 600 python tools/check_half_exact.py > gpurun_out/r4_mix/half_exact.txt 2>&1
grep -v "differing sigma 0 geo 0 rgb 0" gpurun_out/r4_mix/half_exact.txt | tail -n 20
timeout -k 10 300 python bench.py --no-cpu-baseline --also= --oracle-mode-frames 0 > gpurun_out/r4_mix/bench.json 2> gpurun_out/r4_mix/bench.err
python tools/print_bench_line.py < gpurun_out/r4_mix/bench.json
