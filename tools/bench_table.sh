#!/bin/bash
# The HIP rows of DESIGN.md section 7: bench.py on the four single-GPU configurations.
# usage (on the GPU box): bash tools/bench_table.sh > gpurun_out/bench_table.txt
#        PARITY=1 bash tools/bench_table.sh   also renders frame 0 with the CPU oracle (plain: whole frame; the timed mode's
#        own oracle mode: every 2nd pixel) and prints parity_vs_oracle / parity_vs_oracle_mode of the timed mode per config
R=${GRAFT_REPO_ROOT:-.}
if [ "${PARITY:-0}" = "1" ]; then EXTRA="--cpu-passes 1 --torch-stride 0"; else EXTRA="--no-cpu-baseline"; fi
run() {
  echo "== $1"
  shift
  timeout -k 10 900 python3 $R/bench.py --steps 10 --warmup 2 $EXTRA --no-single-frame "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); o = j.get('other_mlp_precisions', {})
        r = j['roofline']
        hl = ' '.join('%s %.3f' % (k, v.get('frac_kernel_time', v['frac_whole_pipeline'])) for k, v in r.get('hash_lookup_hbm_frac', {}).items())
        print('%s: %.0f Mrays/s %.3f Gsamples/s roofline(%s) %.3f  hash lookup / HBM: %s' % (j['mlp_precision'], j['rays_per_sec'] / 1e6, j['value'] / 1e9, r['bound'], r['frac'], hl))
        for k, v in o.items():
            print('%s: %.0f Mrays/s %.3f Gsamples/s' % (k, v['rays_per_sec'] / 1e6, v['value'] / 1e9))
        for key in ('parity_vs_oracle', 'parity_vs_oracle_mode'):
            if key in j:
                print('  %s: %s' % (key, json.dumps(j[key])))
"
}
run "C1 dnerf 400x400" --scene dnerf --width 400 --height 400
run "C2 dnerf 800x800" --scene dnerf --width 800 --height 800
run "C3 hypernerf 536x960" --scene hypernerf --width 536 --height 960
run "C4 dynerf 1352x1014" --scene dynerf --width 1352 --height 1014
