#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mv build_variants/lib_scstamps.so /tmp/ 2>/dev/null
tools/gpu_scale_variants.sh > /dev/null 2>&1; cat gpurun_out/scale_variants.txt
echo "---- stamps of the shipped kernel's code (a -DLFG_DIAG_STAMPS build)"
LFG_LIB=/tmp/lib_scstamps.so python3 tools/scale_stamps.py 2>&1 | tail -22
