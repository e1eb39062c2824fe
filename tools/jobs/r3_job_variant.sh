cd $GRAFT_REPO_ROOT
for v in 2 4 3 1 0; do
  echo -n "field_variant=$v: "
  CED_FIELD_VARIANT=$v timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --also "" --mlp-precision f32 --min-seconds 0.5 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s pipelined, single-frame %.3f ms, single-frame field frac %.3f' % (j['value'] / 1e9, j['single_frame_latency_ms'], j['roofline_single_frame']['frac']))
"
done
