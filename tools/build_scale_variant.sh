#!/bin/bash
# build_variants/lib_NAME.so = the library with scale.hip recompiled with extra flags:  tools/build_scale_variant.sh NAME -DLFG_SCALE_STEPS=6 ...
# Run with LFG_LIB=build_variants/lib_NAME.so python bench.py --workload scale
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
C=$R/linux-fg_amd/csrc
mkdir -p $R/build_variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I$R/include -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/scale.hip -o /tmp/scale_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lib_$name.so $C/lfg_capi.cpp.o $C/lfg_comm.cpp.o /tmp/scale_$name.o $C/interpolate.hip.o $C/motion_literal.hip.o $C/motion_order.hip.o $C/motion_lean.hip.o $C/motion_strip.hip.o $C/motion_prefilter.hip.o $C/motion_resolve.hip.o $C/motion_plan.hip.o -ldl
if [ -z "$LFG_SKIP_HAZARD_CHECK" ]; then python3 $C/check_store_hazard.py $R/build_variants/lib_$name.so; else echo "store-hazard check skipped (LFG_SKIP_HAZARD_CHECK: a diagnostic build)"; fi
echo built build_variants/lib_$name.so
