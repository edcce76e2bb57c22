#!/bin/bash
# On the GPU box: A/B of scale-kernel builds under identical conditions: for each library, rocprofv3 kernel stats over
# 3000 back-to-back launches, three rounds interleaved.  usage: gpu_scale_ab.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do for f in "$@"; do
  n=$(basename $f .so); rm -rf /tmp/ab_$n
  LFG_LIB=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$n -- python3 $R/tools/run_stage.py scale 3000 > /dev/null 2>&1
  python3 - "$n" $(find /tmp/ab_$n -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "scale_2x" in r["Name"]:
        print(f"{sys.argv[1]:24s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}")
PY
done; done
