# round 3: state of the tree on the GPU (full -m gpu suite, default bench line)
mkdir -p gpurun_out/r3c && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/r3c/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3c/tests.log
tail -5 gpurun_out/r3c/tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3c/bench_default.json
