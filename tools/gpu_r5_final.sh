#!/bin/bash
# round 5: the whole GPU suite, the fuzz runs (one lane, three lanes, with a communicator's CU masks), smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_final; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
LFG_FUZZ_CASES=24 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz1.log 2>&1; tail -1 $O/fuzz1.log
LFG_FUZZ_CASES=24 LFG_FUZZ_LANES=3 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz3.log 2>&1; tail -1 $O/fuzz3.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
