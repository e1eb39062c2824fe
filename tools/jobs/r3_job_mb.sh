cd $GRAFT_REPO_ROOT
for mb in 256 248 240 224 256; do
  echo -n "CED_FIELD_MAX_BLOCKS=$mb: "
  CED_FIELD_MAX_BLOCKS=$mb timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-single-frame --also "" --min-seconds 2.0 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s (windows median %.3f) frac %.3f' % (j['value'] / 1e9, j['windows']['median'] / 1e9, j['roofline']['frac']))
"
done
