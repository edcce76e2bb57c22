#!/bin/bash
# round 4: the interpolate kernel on content where it samples (static) and where it rejects both samples (the pan), per-kernel durations
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "interpolate or three_stage or golden or pitch" > $R/gpurun_out/r4/interp_tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/gpurun_out/r4/interp_tests.log
for c in static translated; do STAGE=interpolate REPS=200 bash $R/tools/gpu_ab_kernels.sh $c "$@"; done
