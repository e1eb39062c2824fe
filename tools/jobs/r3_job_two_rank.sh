# the driver's launch line with 2 ranks on this box's one card (collectives over gloo: RCCL refuses two ranks on one GPU),
# full-size frames, both scalings in one line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3q
CED_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --also "" --min-seconds 1.0 > gpurun_out/r3q/bench_two_ranks_one_card.json 2> gpurun_out/r3q/bench_two_ranks.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r3q/bench_two_ranks_one_card.json") if l.startswith("{")][0])
print({k:d.get(k) for k in ("n_gpus","scaling","value","ms_per_step","rccl_ranks","rccl_backend")})
print(d.get("other_scaling")); print(d.get("gather_check")); print(d.get("comm"))
PY
