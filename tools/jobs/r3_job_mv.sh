cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --also "" --min-seconds 1.5 "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s  %.0f Mrays/s  frac %.3f' % (j['value'] / 1e9, j['rays_per_sec'] / 1e6, j['roofline']['frac']))
"
}
for mv in 0 2 0 2; do export CED_MIXED_VARIANT=$mv; echo "mixed_variant=$mv"; run --scene hypernerf --width 536 --height 960; run --scene dynerf --width 1352 --height 1014; done
