mkdir -p gpurun_out/r3h && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3h
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
SCENES="dynerf hypernerf" TAGS=base bash tools/jobs/r3_job_march_quick.sh 2>&1 | grep -E "^==|march_|frame_prep|frame:"
run() {
  echo "== $1"; shift
  timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --also "" "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f single-frame %.3f ms' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac'], j.get('single_frame_latency_ms') or 0))
"
}
run "C4 dynerf 1352x1014 f32" --scene dynerf --width 1352 --height 1014 --mlp-precision f32
run "C4 dynerf 1352x1014" --scene dynerf --width 1352 --height 1014
run "C3 hypernerf f32" --scene hypernerf --width 536 --height 960 --mlp-precision f32
