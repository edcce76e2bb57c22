#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for c in translated objects; do
LFG_MOTION_STRIP=0 tools/gpu_ab_bench.sh $c 3 st1024 | sed 's/st1024/nostrip/'
tools/gpu_ab_bench.sh $c 3 st1024 st512 st256
done
