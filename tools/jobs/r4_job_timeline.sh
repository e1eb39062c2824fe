#!/bin/bash
# round 4: every dispatch of ONE frame (800x800 C2; then C3, C4) in the default arithmetic, from a kernel trace
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PRECISION=${PRECISION:-f16x2}
for sc in dnerf hypernerf dynerf; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/kt_$sc -o tl --output-format csv -- python3 $R/tools/iter_profile.py $sc > $OUT/iter_$sc.txt 2> $OUT/iter_$sc.err
  python3 $R/tools/frame_timeline.py $OUT/kt_$sc > $OUT/timeline_$sc.txt
  tail -n 14 $OUT/timeline_$sc.txt
  tail -n 3 $OUT/iter_$sc.txt
done
