cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
run() {
  echo "== $*"
  timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also "" "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f ms/frame %.3f' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac'], j['ms_per_frame']))
"
}
for fpc in 8 16 32; do run --scene dnerf --width 400 --height 400 --frames-per-call $fpc; done
for fpc in 8 16; do run --scene dnerf --frames-per-call $fpc; done
for fif in 2 4; do run --scene dnerf --frames-in-flight $fif; done
