#!/bin/bash
# A/B of library variants on ONE box through bench.py (frames/s of the default workload), variants interleaved.
# usage: bash tools/ab_bench.sh "base v1 v2" [reps] [extra bench args]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_bench.txt
: > $OUT
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    if [ "$v" = "base" ]; then unset CED_NERF_LIB; else export CED_NERF_LIB=$R/build/variants/libcednerf_hip.$v.so; fi
    line=$(timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --also= --min-seconds 1.5 --no-single-frame $3 2>/dev/null | python3 $R/tools/print_bench_line.py)
    echo "rep $rep $v: $line" | tee -a $OUT
  done
done
