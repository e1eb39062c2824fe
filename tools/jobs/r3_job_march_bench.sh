# marching changes: full GPU suite, then the pipelined bench on C2 / C3 / C4 with the first iteration in one and in two passes
mkdir -p gpurun_out/r3g && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
run() {
  echo "== $1 two_pass=$2"; shift
  local tp=$1; shift
  CED_OPTIONS=march_two_pass=$tp timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --also "" "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f single-frame %.3f ms' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac'], j.get('single_frame_latency_ms') or 0))
"
}
for tp in 0 1; do
run "C2 dnerf 800x800" $tp --scene dnerf
run "C2 dnerf 800x800 f32" $tp --scene dnerf --mlp-precision f32
run "C3 hypernerf 536x960" $tp --scene hypernerf --width 536 --height 960
run "C4 dynerf 1352x1014" $tp --scene dynerf --width 1352 --height 1014
run "C4 dynerf 1352x1014 f32" $tp --scene dynerf --width 1352 --height 1014 --mlp-precision f32
done 2>&1 | tee $OUT/bench_ab.txt
