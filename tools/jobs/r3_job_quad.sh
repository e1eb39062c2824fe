mkdir -p gpurun_out/r3r && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3r
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
for cq in 1 0 1 0; do
  echo -n "composite_quad=$cq: "
  CED_OPTIONS=composite_quad=$cq timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --also "" --min-seconds 1.5 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s pipelined (windows median %.3f), single-frame %.3f ms' % (j['value'] / 1e9, j['windows']['median'] / 1e9, j['single_frame_latency_ms']))
"
done 2>&1 | tee $OUT/bench_quad.txt
cd /tmp && export TMPDIR=/tmp
for cq in 1 0; do
CED_OPTIONS=composite_quad=$cq timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/tr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py dnerf > /dev/null 2>&1
echo "composite_quad=$cq"; python3 $GRAFT_REPO_ROOT/tools/frame_timeline.py $GRAFT_REPO_ROOT/$OUT/tr | grep -A7 "^frame:"; rm -rf $GRAFT_REPO_ROOT/$OUT/tr
done
