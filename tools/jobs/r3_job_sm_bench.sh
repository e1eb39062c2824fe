cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame --also "" "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac']))
"
}
for opt in "march_two_pass=0" "march_two_pass=1,march_sm=0" "march_two_pass=1,march_sm=1"; do
  export CED_OPTIONS=$opt; echo "### $opt"
  run --scene dnerf
  run --scene dynerf --width 1352 --height 1014
done
