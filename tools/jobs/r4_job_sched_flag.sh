#!/bin/bash
# round 4: the scheduler strategy flag on other sources: f32 / mixed kernels standalone, and the pipelined bench with frame.hip rebuilt
set -e
mkdir -p gpurun_out/r4_ab
{
for lib in ced_nerf_amd/libcednerf_hip.so build/ab/lib_mc_field.so; do
  echo "== $lib f32"; CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib PRECISION=f32 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
done
for lib in ced_nerf_amd/libcednerf_hip.so build/ab/lib_mc_mixed.so; do
  echo "== $lib f32+h16x2"; CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib PRECISION=f32+h16x2 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "Gsamples"
done
for lib in ced_nerf_amd/libcednerf_hip.so build/ab/lib_mc_frame.so ced_nerf_amd/libcednerf_hip.so build/ab/lib_mc_frame.so; do
  echo "== $lib bench"; CED_NERF_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --also= --oracle-mode-frames 0 --no-single-frame 2>/dev/null | python tools/print_bench_line.py
done
} | tee gpurun_out/r4_ab/sched_flag.txt
