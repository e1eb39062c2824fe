#!/bin/bash
# A/B of library variants on ONE box: standalone field kernel (tools/bench_field.py, 15 M samples), variants interleaved.
# usage: bash tools/ab_field.sh "base v1 v2" [reps]     (names of build/variants/libcednerf_hip.NAME.so; base = shipped)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_field.txt
: > $OUT
for rep in $(seq 1 ${2:-2}); do
  for prec in ${PRECS:-f32 f16x2}; do
    for v in $1; do
      if [ "$v" = "base" ]; then unset CED_NERF_LIB; else export CED_NERF_LIB=$R/build/variants/libcednerf_hip.$v.so; fi
      echo "rep $rep $prec $v: $(PRECISION=$prec timeout -k 10 120 python3 $R/tools/bench_field.py 2>/dev/null | grep -E 'Gsamples' | tail -1)" | tee -a $OUT
    done
  done
done
