mkdir -p gpurun_out/r3j && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3j
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
run() {
  echo "== $*"
  timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --also "" "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f single-frame %.3f ms' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac'], j.get('single_frame_latency_ms') or 0))
"
}
for fs in 1 0; do CED_OPTIONS=fold_schedule=$fs run --scene dnerf; CED_OPTIONS=fold_schedule=$fs run --scene dnerf --mlp-precision f32;  done 2>&1 | tee $OUT/bench_fold.txt
