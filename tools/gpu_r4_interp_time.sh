#!/bin/bash
# per-kernel durations of the interpolate stage for library variants: usage gpu_r4_interp_time.sh variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in ${CONTENTS:-static translated}; do STAGE=interpolate REPS=200 bash $R/tools/gpu_ab_kernels.sh $c "$@" | grep -E "interpolate"; done
