#!/bin/bash
# A/B of library variants on one content: rocprofv3 kernel time of the prefilter.  usage: gpu_ab_motion.sh content variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  rm -rf /tmp/ab_$v
  LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/tools/run_stage.py motion ${REPS:-10} $c > /tmp/ab_$v.out 2>&1
  python3 - $v $c $(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[3])):
    if "prefilter" in r["Name"] or "resolve" in r["Name"]:
        print(f"{sys.argv[2]:12s} {sys.argv[1]:10s} {r['Name'].split('(')[0][5:30]:26s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}")
PY
done; done
