#!/bin/bash
# Builds a variant of the library with extra -D flags applied to the marching / frame sources only:
#   tools/build_variant.sh NAME -DCED_FRAME_LOOK=8 ...   ->  build/variants/libcednerf_hip.NAME.so
# SRCS="field field_half" chooses which sources get the flags (default: frame accel march).
# Select it at run time with CED_NERF_LIB=<path>.  (Experiments only; the shipped library is _lib.build().)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/build/obj
mkdir -p $OBJ/var_$NAME $R/build/variants
python3 -c "import sys; sys.path.insert(0,'$R'); from ced_nerf_amd import _lib; _lib.build()"
for f in ${SRCS:-frame accel march}; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 "$@" -c $R/ced_nerf_amd/csrc/$f.hip -o $OBJ/var_$NAME/$f.hip.o &
done
wait
OBJS=""
for o in $OBJ/*.hip.o; do
  b=$(basename $o)
  if [ -f $OBJ/var_$NAME/$b ]; then OBJS="$OBJS $OBJ/var_$NAME/$b"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libcednerf_hip.$NAME.so $OBJS
echo built $R/build/variants/libcednerf_hip.$NAME.so
