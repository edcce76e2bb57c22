#!/bin/bash
# On the GPU box: prefiltered-vs-exact diff at 4K, then rocprof kernel timing of the motion stage.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python3 $R/tools/dbg_motion.py
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m -- python3 $R/tools/run_stage.py motion 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/prof_m/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:45], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
