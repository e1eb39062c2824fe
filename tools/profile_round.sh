#!/bin/bash
# Produces the per-round profile artefacts on the GPU box under gpurun_out/prof/ (copy what is to be judged into
# profiles/).  Counters are collected in their own passes (never together with trace domains).
# usage (on the GPU box): bash tools/profile_round.sh [tag]
set -e
TAG=${1:-r04_final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CED_BENCH_OTHER_TABLE=0      # profiled runs: only the timed table type's kernels
for prec in ${MODES:-f16x2 f32+h16x2}; do
  ARGS="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --also= --min-seconds 0 --mlp-precision $prec"
  # 1. the bench line itself (not under the profiler)
  timeout -k 10 300 python3 $ARGS > $OUT/${TAG}_${prec}_bench.json 2> $OUT/${TAG}_${prec}_bench.err
  echo "bench $prec done"
  # 2. per-kernel times (profiled runs skip the one-frame-alone epilogue: all field launches are of the timed kind)
  ARGS="$ARGS --no-single-frame"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$prec -o kt --output-format csv -- python3 $ARGS > $OUT/${TAG}_${prec}_bench_under_rocprof.json 2> $OUT/kt_$prec.err
  cp $(ls $OUT/kt_$prec/*kernel_stats.csv | head -1) $OUT/${TAG}_${prec}_kernel_stats.csv
  # the same two figures bench.py reports, from the profiler's own trace (warm-up step's 36 dispatches left out)
  python3 $R/tools/field_busy_from_trace.py $OUT/kt_$prec 36 > $OUT/${TAG}_${prec}_busy_from_trace.txt
  echo "kernel trace $prec done"
  # 3. HBM bytes: FETCH_SIZE and WRITE_SIZE in separate passes
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$prec -o f --output-format csv -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch_$prec.err
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$prec -o w --output-format csv -- python3 $ARGS > /dev/null 2> $OUT/pmc_write_$prec.err
  python3 $R/tools/pmc_summary.py $OUT/${TAG}_${prec}_pmc.json $OUT/pmc_fetch_$prec $OUT/pmc_write_$prec > /dev/null
  echo "pmc $prec done"
done
# the default bench.py line (default mode + the other modes + cpu baselines + oracle-mode parity)
unset CED_BENCH_OTHER_TABLE
timeout -k 10 600 python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
echo "default bench done"
