#!/bin/bash
# The v_mfma_f32_16x16x32_f16 / packed-op_sel hazard, reproduced at kernel level:
#   build (no GPU needed):  tools/k32_experiments.sh build
#       slpfloor = the half kernels WITH the SLP vectoriser and round 1's floor-based hash coordinates
#                  (hipcc then emits v_pk_mul_f32 ... op_sel:[0,1] beside the MFMAs: irreproducible)
#       k16      = the same MFMA blocks as two v_mfma_f32_16x16x16_f16 (round 1's shipping form)
#   run on the GPU box:     tools/k32_experiments.sh run > gpurun_out/k32/log.txt
#       one pass of tools/debug_half.py (two launches of up to 15 M samples, compared) per variant and for the
#       shipped library; then the instruction-level probe tools/probes/pk_opsel_mfma.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
build)
  SRCS="field_half" $R/tools/build_variant.sh slpfloor -fslp-vectorize -DCED_AB_NO_FRACT
  SRCS="field_half field_mixed" $R/tools/build_variant.sh k32 -DCED_HALF_MFMA_K32
  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o $R/tools/probes/pk_opsel_mfma $R/tools/probes/pk_opsel_mfma.hip
  ;;
run)
  for v in ${VARIANTS:-base slpfloor k16}; do
    echo "##### variant $v"
    if [ "$v" = "base" ]; then unset CED_NERF_LIB; else export CED_NERF_LIB=$R/build/variants/libcednerf_hip.$v.so; fi
    PRECS=f16x2,f16 timeout -k 10 240 python3 $R/tools/debug_half.py 2>&1 | grep -E "^n=" | cut -c1-150 || true
  done
  echo "##### probe"
  timeout -k 10 200 $R/tools/probes/pk_opsel_mfma 20000
  ;;
esac
