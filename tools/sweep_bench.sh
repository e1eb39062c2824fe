#!/bin/bash
# frames in flight x frames per call x field workgroup cap, on one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sweep_bench.txt
: > $OUT
IFS=";" read -ra CFGS <<< "${SWEEP:-3 3 128}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  line=$(CED_FIELD_MAX_BLOCKS=$3 timeout -k 10 200 python3 $R/bench.py --frames-in-flight $1 --frames-per-call $2 --steps 6 --warmup 2 --no-cpu-baseline --also= --min-seconds 1.0 --no-single-frame 2>/dev/null | python3 $R/tools/print_bench_line.py)
  echo "in_flight $1 per_call $2 blocks $3: $line" | tee -a $OUT
done
