#!/bin/bash
# round 4: calls in flight x frames per call in the new default arithmetic (f16x2), one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_sweep
mkdir -p $OUT
export CED_BENCH_OTHER_TABLE=0
run() {
  echo "== $*"
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also= "$@" 2>/dev/null | python3 $R/tools/print_bench_line.py
}
{
run --frames-in-flight 3 --frames-per-call 16
run --frames-in-flight 1 --frames-per-call 48
run --frames-in-flight 1 --frames-per-call 64
run --frames-in-flight 2 --frames-per-call 24
run --frames-in-flight 4 --frames-per-call 12
run --frames-in-flight 4 --frames-per-call 16
run --frames-in-flight 3 --frames-per-call 21
run --frames-in-flight 3 --frames-per-call 16
} > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
