cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
timeout -k 10 600 python -m pytest tests -m gpu -x -q --capture=sys -k "sort_intersections or one_shot or render_image_parity or sampling or first_iteration_forms" > gpurun_out/r3n/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3n/tests.log
for sc in dynerf hypernerf; do PRECISION=f32 timeout -k 10 120 python tools/bench_render_image.py $sc 2>&1 | grep ms/frame; done
