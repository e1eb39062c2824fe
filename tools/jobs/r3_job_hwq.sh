cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q: "
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --also "" --min-seconds 1.5 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%.3f Gsamples/s, windows median %.3f' % (j['value'] / 1e9, j['windows']['median'] / 1e9))
"
done
echo "two ranks (gloo) with 16 queues:"
GPU_MAX_HW_QUEUES=16 CED_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --also "" --min-seconds 0.5 --width 400 --height 400 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'] / 1e9, j['gather_check']['ok'])
"
