#!/bin/bash
# round 5: extended fuzz with the persistent kernel's variant forced and noise of up to 12 levels, one lane and three
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_fuzz; mkdir -p $O
LFG_TIER_FORCE=1 LFG_FUZZ_MAX_AMP=12 LFG_FUZZ_CASES=96 timeout -k 10 500 python3 tools/fuzz_motion_4k.py > $O/tier_1lane.log 2>&1; tail -1 $O/tier_1lane.log
LFG_TIER_FORCE=1 LFG_FUZZ_MAX_AMP=12 LFG_FUZZ_CASES=96 LFG_FUZZ_LANES=3 timeout -k 10 500 python3 tools/fuzz_motion_4k.py > $O/tier_3lanes.log 2>&1; tail -1 $O/tier_3lanes.log
LFG_FUZZ_MAX_AMP=12 LFG_FUZZ_CASES=96 LFG_FUZZ_LANES=3 timeout -k 10 500 python3 tools/fuzz_motion_4k.py > $O/verdict_3lanes.log 2>&1; tail -1 $O/verdict_3lanes.log
LFG_FUZZ_CASES=96 LFG_FUZZ_LANES=3 LFG_LEAN_FORCE=1 timeout -k 10 500 python3 tools/fuzz_motion_4k.py > $O/lean_3lanes.log 2>&1; tail -1 $O/lean_3lanes.log
