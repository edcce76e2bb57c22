#!/bin/bash
# round 5: the final library's extended fuzz (as every round's): one frame at a time, three in flight, the lean kernel forced
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_fuzz; mkdir -p $O
LFG_FUZZ_CASES=300 timeout -k 10 900 python3 tools/fuzz_motion_4k.py > $O/final_1lane.log 2>&1; tail -1 $O/final_1lane.log
LFG_FUZZ_CASES=200 LFG_FUZZ_LANES=3 timeout -k 10 900 python3 tools/fuzz_motion_4k.py > $O/final_3lanes.log 2>&1; tail -1 $O/final_3lanes.log
LFG_FUZZ_CASES=160 LFG_FUZZ_LANES=3 LFG_LEAN_FORCE=1 timeout -k 10 900 python3 tools/fuzz_motion_4k.py > $O/final_lean.log 2>&1; tail -1 $O/final_lean.log
LFG_FUZZ_CASES=160 LFG_FUZZ_LANES=3 LFG_MOTION_STRIP=1 timeout -k 10 900 python3 tools/fuzz_motion_4k.py > $O/final_strip.log 2>&1; tail -1 $O/final_strip.log
