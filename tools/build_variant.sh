#!/bin/bash
# build_variants/lib_NAME.so = the library with the motion files recompiled differently:
#     tools/build_variant.sh NAME [LFG_X=value ...] [-DFOO ...]
#   LFG_X=value   rewrites that constant in a COPY of csrc/lfg_motion_tune.hpp (the tuning constants are plain constexpr values
#                 since round 5; a -DLFG_X no longer compiles)
#   -DFOO         passed to the compiler: the diagnostic switches -DLFG_MOTION_STAMPS [-DLFG_STAMP_PHASES], -DLFG_LEAN_STATS ...
#   ONLY=lean|prefilter|...   recompile just motion_<that>.hip (the others as built for the product)
#   ALSO_INTERPOLATE=1        the same flags for interpolate.hip
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
C=$R/linux-fg_amd/csrc
W=$(mktemp -d /tmp/lfg_variant_XXXX)
mkdir -p $R/build_variants
cp $C/*.hip $C/*.hpp $C/*.inc $W/
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I$R/include -Wall -Wno-unused-function"
DEFS=()
for a in "$@"; do
  case "$a" in
    LFG_*=*) k=${a%%=*}; v=${a#*=}
             grep -q "^constexpr [a-z]* $k = " $W/lfg_motion_tune.hpp || { echo "no tuning constant $k in lfg_motion_tune.hpp"; exit 1; }
             sed -i -E "s|^(constexpr [a-z]+ $k = )[^;]*;|\1$v;|" $W/lfg_motion_tune.hpp ;;
    *) DEFS+=("$a") ;;
  esac
done
MOTION="motion_literal motion_order motion_lean motion_strip motion_prefilter motion_resolve motion_plan"
OBJS="$C/lfg_capi.cpp.o $C/lfg_comm.cpp.o $C/comm_probe.hip.o $C/scale.hip.o"
for m in $MOTION; do
  if [ -n "$ONLY" ] && [ "motion_$ONLY" != "$m" ]; then OBJS="$OBJS $C/$m.hip.o"; continue; fi
  /opt/rocm/bin/hipcc $FLAGS "${DEFS[@]}" -c $W/$m.hip -o $W/$m.o &
  OBJS="$OBJS $W/$m.o"
done
if [ -n "$ALSO_INTERPOLATE" ]; then
  /opt/rocm/bin/hipcc $FLAGS "${DEFS[@]}" -c $W/interpolate.hip -o $W/interpolate.o &
  OBJS="$OBJS $W/interpolate.o"
else
  OBJS="$OBJS $C/interpolate.hip.o"
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lib_$name.so $OBJS -ldl
python3 $C/check_store_hazard.py $R/build_variants/lib_$name.so     # the same machine-code check as the product build
rm -rf $W
echo built build_variants/lib_$name.so
