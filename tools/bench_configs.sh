#!/bin/bash
# one-frame-at-a-time timings of every BASELINE config shape in every MLP arithmetic mode
for prec in ${MODES:-f32+h16x2 f32 f16x2 f16}; do
  export PRECISION=$prec
  timeout -k 10 120 python tools/bench_render_image.py dnerf 2>&1 | grep ms/frame || exit 1
  timeout -k 10 120 python tools/bench_render_image.py dnerf f16 2>&1 | grep ms/frame || exit 1
  timeout -k 10 120 python tools/bench_render_image.py hypernerf 2>&1 | grep ms/frame || exit 1
  timeout -k 10 120 python tools/bench_render_image.py dynerf 2>&1 | grep ms/frame || exit 1
done
