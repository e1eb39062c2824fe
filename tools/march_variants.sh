#!/bin/bash
# Per-kernel frame timeline for each library variant given (names of build/variants/libcednerf_hip.NAME.so; "base" = the shipped one)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/variants
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "base" ]; then unset CED_NERF_LIB; else export CED_NERF_LIB=$R/build/variants/libcednerf_hip.$v.so; fi
  rm -rf $OUT/kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/kt_$v -o kt --output-format csv -- python3 $R/tools/iter_profile.py ${SCENE:-dnerf} > $OUT/$v.log 2>&1
  echo "== $v: $(grep 'frame ' $OUT/$v.log | tail -1)"
  python3 $R/tools/frame_timeline.py $OUT/kt_$v | grep -E "march" | awk '{printf "%s ", $3} END {print ""}'
  python3 $R/tools/frame_timeline.py $OUT/kt_$v | tail -8
done
