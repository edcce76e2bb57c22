#!/bin/bash
# On the GPU box: rocprof kernel timing of the motion stage for the input size in $1 (e.g. 1904x1056).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for sz in "$@"; do
  rm -rf $R/gpurun_out/prof_t
  LFG_STAGE_INPUT=$sz rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t -- python3 $R/tools/run_stage.py motion 3 > /dev/null 2>&1
  echo "== input $sz"
  python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/prof_t/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'motion' in r['Name']: print(r['Name'][:45], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
done
