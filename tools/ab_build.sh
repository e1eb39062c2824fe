#!/bin/bash
# Diagnostic A/B libraries: recompiles ONE source with extra -D flags and links it with the objects of the normal build
# into build/ab/lib_<name>.so (run a tool against it with CED_NERF_LIB=build/ab/lib_<name>.so).
# usage (here, cross-compiling): tools/ab_build.sh <name> <source.hip> <-Dflags...>
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/ab
objs=$(ls build/obj/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 "$@" -c ced_nerf_amd/csrc/$src -o build/ab/$name.$src.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/lib_$name.so $objs build/ab/$name.$src.o
echo build/ab/lib_$name.so
