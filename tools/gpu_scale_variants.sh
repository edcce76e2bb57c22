#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the 2x scale kernel for the shipped library and every build_variants/lib_*.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/scale_variants.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for f in $R/linux-fg_amd/liblinuxfg_hip.so $R/build_variants/lib_*.so; do
  n=$(basename $f .so)
  rm -rf /tmp/pv_$n
  LFG_LIB=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$n -- python3 $R/tools/run_stage.py scale 300 > /tmp/pv_$n.out 2> /tmp/pv_$n.err
  python3 - "$n" $(find /tmp/pv_$n -name "*kernel_stats.csv" | head -1) >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "scale_2x" in r["Name"]:
        print(f"{sys.argv[1]:24s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}  max {float(r['MaxNs'])/1e3:6.2f}")
PY
done; done
cat $OUT
