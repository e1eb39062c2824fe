mkdir -p gpurun_out/r3k && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3k
CED_OPTIONS=march_two_pass=1,march_sm=1 timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_parity.py -m gpu -x -q --capture=sys -k "fullframe or full_frame or fuzz or frames or C1 or C2 or C3 or C4 or render_image_test" > $OUT/tests_sm.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests_sm.log
tail -3 $OUT/tests_sm.log
bash tools/jobs/r3_job_march_quick.sh 2>&1 | grep -E "^==|march_|frame:|frame [0-9]"
