cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --also "" --min-seconds 1.0 "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s  %.0f Mrays/s  frac %.3f' % (j['value'] / 1e9, j['rays_per_sec'] / 1e6, j['roofline']['frac']))
"
}
for fif in 2 3 4; do for fpc in 12 16 24; do run --frames-in-flight $fif --frames-per-call $fpc; done; done
run --scene dynerf --width 1352 --height 1014 --frames-per-call 8
run --scene dynerf --width 1352 --height 1014 --frames-per-call 12
run --scene hypernerf --width 536 --height 960 --frames-per-call 12
run --scene hypernerf --width 536 --height 960 --frames-per-call 20
run --scene hypernerf --width 536 --height 960 --frames-per-call 32
