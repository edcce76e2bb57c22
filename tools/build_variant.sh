#!/bin/bash
# build_variants/lib_NAME.so = the library with motion.hip recompiled with extra flags:  tools/build_variant.sh NAME -DFOO=1 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
C=$R/linux-fg_amd/csrc
mkdir -p $R/build_variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I$R/include -Wall -Wno-unused-function"
[ -n "$ONLY_LEAN" ] || /opt/rocm/bin/hipcc $FLAGS "$@" -c ${MOTION_SRC:-$C/motion.hip} -o /tmp/motion_$name.o
INTERP=$C/interpolate.hip.o
if [ -n "$ALSO_INTERPOLATE" ]; then     # the same flags for interpolate.hip (diagnostic switches that live there)
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/interpolate.hip -o /tmp/interpolate_$name.o
  INTERP=/tmp/interpolate_$name.o
fi
LEAN=$C/motion_lean.hip.o
if [ -n "$ALSO_LEAN" ]; then            # the same flags for motion_lean.hip
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/motion_lean.hip -o /tmp/motion_lean_$name.o
  LEAN=/tmp/motion_lean_$name.o
fi
if [ -n "$ONLY_LEAN" ]; then            # motion.hip as built for the product (saves its compile time)
  cp $C/motion.hip.o /tmp/motion_$name.o
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lib_$name.so $C/lfg_capi.cpp.o $C/lfg_comm.cpp.o $C/scale.hip.o $INTERP /tmp/motion_$name.o $LEAN -ldl
python3 $C/check_store_hazard.py $R/build_variants/lib_$name.so     # the same machine-code check as the product build
echo built build_variants/lib_$name.so
