# marching A/B: single-frame timelines (two-pass first iteration on / off; library variants) 
mkdir -p gpurun_out/r3f && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp
run() {  # scene two_pass libtag
  local sc=$1 tp=$2 tag=$3
  local lib=$GRAFT_REPO_ROOT/ced_nerf_amd/libcednerf_hip.so
  [ "$tag" != "base" ] && lib=$GRAFT_REPO_ROOT/build/variants/libcednerf_hip.$tag.so
  CED_NERF_LIB=$lib CED_OPTIONS=march_two_pass=$tp,march_sm=${CED_SM:-1} timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py $sc > $OUT/iter_${sc}_${tp}_$tag.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/frame_timeline.py $OUT/tr > $OUT/timeline_${sc}_${tp}_$tag.txt 2>&1
  echo "== $sc two_pass=$tp sm=${CED_SM:-1} lib=$tag"; grep "frame " $OUT/iter_${sc}_${tp}_$tag.log | tail -1; head -5 $OUT/timeline_${sc}_${tp}_$tag.txt | tail -3; grep -A8 "^frame:" $OUT/timeline_${sc}_${tp}_$tag.txt
  rm -rf $OUT/tr
}
for sc in ${SCENES:-dnerf dynerf hypernerf}; do
  for sm in 1 0; do CED_SM=$sm; run $sc 1 base; done
done
