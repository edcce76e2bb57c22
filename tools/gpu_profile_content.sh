#!/bin/bash
# On the GPU box: rocprof kernel stats of bench.py for one content model ($1 = translated|uncorrelated).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python3 $R/bench.py --content $1 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_c.json 2>/dev/null
python3 - <<PY
import csv, glob, json
print(json.loads(open('/tmp/prof_c.json').read())['value'], 'frames/s')
for f in glob.glob('/tmp/prof_c/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:45], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
