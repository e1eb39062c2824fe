#!/bin/bash
# Like ab_build.sh, but the extra flags (typically -mllvm options of the AMDGPU backend) reach the DEVICE compilation only:
# device compilation with the flags (an offload bundle) -> host compilation that embeds it.
# usage: tools/ab_build_device_flags.sh <name> <source.hip> <device-only flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/ab
COMMON="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17"
# (--cuda-device-only -c already yields the offload bundle the host compilation embeds)
hipcc $COMMON --cuda-device-only "$@" -c ced_nerf_amd/csrc/$src -o build/ab/$name.hipfb 2>/dev/null
hipcc $COMMON --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang build/ab/$name.hipfb -c ced_nerf_amd/csrc/$src -o build/ab/$name.$src.o 2>/dev/null
objs=$(ls build/obj/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/lib_$name.so $objs build/ab/$name.$src.o
echo build/ab/lib_$name.so
