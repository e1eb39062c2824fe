#!/bin/bash
# round 4: where a later marching iteration's time goes -- timing experiments on one box (variants with wrong results are
# marked): no regeneration stores; 64-thread workgroups; compositing with 8 / 16 samples per round trip
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_mexp
mkdir -p $OUT
export PRECISION=f16x2
for sc in dnerf hypernerf; do
  echo "#### $sc"
  SCENE=$sc bash $R/tools/march_variants.sh base noregen t64 ku8 ku16 2>&1 | grep -E "^==|march_frame_kernel|frame_composite|frame:"
done > $OUT/timelines.txt 2>&1
cat $OUT/timelines.txt
