#!/bin/bash
# On the GPU box: per-kernel durations of lfg_motion on a content (default translated).  usage: gpu_motion_stats.sh [content]
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-translated}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$C
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_$C -- python3 $R/tools/run_stage.py motion 20 $C > /tmp/pm_$C.out 2>&1
python3 - $(find /tmp/pm_$C -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lfg::" in r["Name"]:
        print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
