#!/bin/bash
# round 4: A/B of two libraries under bench.py conditions and standalone, alternating (same box): usage r4_job_ab_bench.sh <libA> <libB>
set -e
mkdir -p gpurun_out/r4_ab
{
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib"
    CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib PRECISION=f16x2 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples" | sed 's/; vs f32.*//'
    CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --also= --oracle-mode-frames 0 --no-single-frame 2>/dev/null | python tools/print_bench_line.py
  done
done
} | tee gpurun_out/r4_ab/ab_bench.txt
