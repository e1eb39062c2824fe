cd $GRAFT_REPO_ROOT
for st in 0 4 0 6 3 8; do
  echo -n "field_stagger=$st: "
  CED_OPTIONS=field_stagger=$st timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --also "" --min-seconds 1.0 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s pipelined, single-frame %.3f ms (p10 %.3f p90 %.3f), single-frame field frac %.3f' % (j['value'] / 1e9, j['single_frame_latency_ms'], j['single_frame_latency_stats_ms']['p10'], j['single_frame_latency_stats_ms']['p90'], j['roofline_single_frame']['frac']))
"
done
