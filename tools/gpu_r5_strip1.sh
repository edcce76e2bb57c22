#!/bin/bash
# round 5: the strip kernel -- its tests, the whole suite, fuzz, A/B against the library without it (LFG_MOTION_STRIP=0)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_strip1; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strip" > $O/strip_tests.log 2>&1; rc=$?; echo "strip tests rc $rc"; tail -15 $O/strip_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
LFG_FUZZ_CASES=24 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz1.log 2>&1; tail -1 $O/fuzz1.log
LFG_FUZZ_CASES=24 LFG_FUZZ_LANES=3 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz3.log 2>&1; tail -1 $O/fuzz3.log
for c in translated objects occluded noisy; do LANES=3 CONTENT=$c STEPS=400 tools/gpu_quick_bench.sh "LFG_MOTION_STRIP=0" "-"; done 2>&1 | tee $O/ab.txt
LANES=1 CONTENT=translated STEPS=400 tools/gpu_quick_bench.sh "LFG_MOTION_STRIP=0" "-" 2>&1 | tee -a $O/ab.txt
