#!/bin/bash
# round 4: the MFMA probe (all operand families) + replay of the oracle's own blocks + the bit-exactness check of the
# half-precision modes of the field kernel
set -e
mkdir -p gpurun_out/r4_half
python tools/probes/mfma_f16_order.py gen /tmp/in.bin
tools/probes/mfma_f16_order /tmp/in.bin gpurun_out/r4_half/out.bin
python tools/probes/mfma_f16_check.py gpurun_out/r4_half/out.bin > gpurun_out/r4_half/model_check.txt 2>&1
grep -v " 0 of" gpurun_out/r4_half/model_check.txt || true
python tools/probes/mfma_replay.py build/r4_mismatch.npz > gpurun_out/r4_half/replay.txt 2>&1
tail -n 3 gpurun_out/r4_half/replay.txt
timeout -k 10 900 python tools/check_half_exact.py "$@" > gpurun_out/r4_half/half_exact.txt 2>&1 || true
grep -v "differing sigma 0 geo 0 rgb 0" gpurun_out/r4_half/half_exact.txt | tail -n 40
