# two-pass first iteration: parity (full-frame + fuzz tests), then the single-frame timeline with and without it
mkdir -p gpurun_out/r3e && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for sc in dnerf dynerf hypernerf; do for tp in 1 0; do
  CED_OPTIONS=march_two_pass=$tp timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr_${sc}_$tp -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py $sc > $OUT/iter_${sc}_$tp.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/frame_timeline.py $OUT/tr_${sc}_$tp > $OUT/timeline_${sc}_$tp.txt 2>&1
  echo "== $sc two_pass=$tp"; tail -1 $OUT/iter_${sc}_$tp.log; head -9 $OUT/timeline_${sc}_$tp.txt; grep -A12 "^frame:" $OUT/timeline_${sc}_$tp.txt
  rm -rf $OUT/tr_${sc}_$tp
done; done
