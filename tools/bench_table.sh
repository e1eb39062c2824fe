#!/bin/bash
# The HIP rows of DESIGN.md section 7: bench.py on the four single-GPU configurations.
# usage (on the GPU box): bash tools/bench_table.sh > gpurun_out/bench_table.txt
R=${GRAFT_REPO_ROOT:-.}
run() {
  echo "== $1"
  shift
  timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); o = j.get('other_mlp_precisions', {})
        print('%s: %.0f Mrays/s %.3f Gsamples/s frac %.3f hbm %.3f' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, j['roofline']['frac'], j['roofline_hbm']['frac']))
        for k, v in o.items():
            print('%s: %.0f Mrays/s %.3f Gsamples/s' % (k, v['rays_per_sec'] / 1e6, v['value'] / 1e9))
"
}
# frames per call: 8 (bench.py default)
run "C1 dnerf 400x400" --scene dnerf --width 400 --height 400
run "C2 dnerf 800x800" --scene dnerf --width 800 --height 800
run "C3 hypernerf 536x960" --scene hypernerf --width 536 --height 960
run "C4 dynerf 1352x1014" --scene dynerf --width 1352 --height 1014
