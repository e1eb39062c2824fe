#!/bin/bash
# round 4: calls in flight on the marching-heavy configurations (C3, C4), f16x2, one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_sweep
mkdir -p $OUT
export CED_BENCH_OTHER_TABLE=0
run() {
  echo "== $*"
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also= "$@" 2>/dev/null | python3 $R/tools/print_bench_line.py
}
{
for sc in "dynerf --width 1352 --height 1014" "hypernerf --width 536 --height 960"; do
  for l in 3 4 5 6; do run --scene $sc --frames-in-flight $l; done
done
} > $OUT/sweep_c3c4.txt 2>&1
cat $OUT/sweep_c3c4.txt
