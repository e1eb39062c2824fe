#!/bin/bash
# round 4 diagnostic: is the f16x2 / f16 field kernel sensitive to WHERE its gathers go?  Same instruction stream, the
# gathers of each level confined to 64 B (one line), 1 KB (L1-resident), 64 KB (L2-resident), or unconfined.
set -e
mkdir -p gpurun_out/r4_gather
for w in w64 w1k w64k wall; do
  for prec in f16x2 f16; do
    echo "== window $w precision $prec"
    CED_NERF_LIB=$GRAFT_REPO_ROOT/build/ab/lib_$w.so PRECISION=$prec timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
  done
done | tee gpurun_out/r4_gather/gather_windows.txt
