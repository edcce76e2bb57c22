#!/bin/bash
# per-kernel mean durations of the pipeline with frames in flight, library variants.  usage: gpu_ab_lanes_trace.sh content lanes variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; n=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/abt_$v
  LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_$v -- python3 $R/bench.py --content $c --in-flight $n --steps 150 --warmup 9 --no-extras --no-cpu-baseline > /tmp/abt_$v.out 2>/tmp/abt_$v.err
  python3 - $v $c $(find /tmp/abt_$v -name "*kernel_stats.csv" | head -1) /tmp/abt_$v.out <<'PY'
import csv, sys, json
d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} {sys.argv[1]}: {d['value']:.1f} frames/s under rocprof")
for r in csv.DictReader(open(sys.argv[3])):
    if "lfg::" in r["Name"]:
        print(f"   {r['Name'].split('(')[0][:44]:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
