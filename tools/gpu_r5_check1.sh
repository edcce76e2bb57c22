#!/bin/bash
# round 5, first GPU call: the split library against round 4's (build_variants/lib_r04.so) -- tests, fuzz, A/B frame rates
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_check1; mkdir -p $O
cp linux-fg_amd/liblinuxfg_hip.so build_variants/lib_new.so
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
LFG_FUZZ_CASES=24 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz1.log 2>&1; tail -1 $O/fuzz1.log
LFG_FUZZ_CASES=24 LFG_FUZZ_LANES=3 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz3.log 2>&1; tail -1 $O/fuzz3.log
for c in translated noisy objects; do tools/gpu_ab_bench.sh $c 3 r04 new; done 2>&1 | tee $O/ab.txt
tools/gpu_ab_bench.sh translated 1 r04 new 2>&1 | tee -a $O/ab.txt
