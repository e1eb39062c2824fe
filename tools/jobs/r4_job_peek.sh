#!/bin/bash
# round 4: peek-ahead of the later-iteration walk (CED_FRAME_PEEK), one box: per-kernel frame timelines, then the pipelined bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_peek
mkdir -p $OUT
export PRECISION=f16x2
for sc in dnerf hypernerf dynerf; do
  echo "#### $sc"
  SCENE=$sc bash $R/tools/march_variants.sh peek0 base peek8 peek16 peek24 2>&1 | grep -E "^==|march_frame_kernel<[a-z]+, false, false>|frame:"
done > $OUT/timelines.txt 2>&1
cat $OUT/timelines.txt
export CED_BENCH_OTHER_TABLE=0
for sc in "dnerf 800 800" "hypernerf 536 960" "dynerf 1352 1014"; do
  set -- $sc
  for v in peek0 base; do
    if [ "$v" = "base" ]; then unset CED_NERF_LIB; else export CED_NERF_LIB=$R/build/variants/libcednerf_hip.$v.so; fi
    echo "== $1 $v: $(timeout -k 10 300 python3 $R/bench.py --scene $1 --width $2 --height $3 --no-cpu-baseline --no-single-frame --also= 2>/dev/null | python3 $R/tools/print_bench_line.py)"
  done
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
