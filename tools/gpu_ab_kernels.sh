#!/bin/bash
# every lfg:: kernel's mean duration for library variants on one content.  usage: [STAGE=motion|interpolate|scale|pipeline] [REPS=10] gpu_ab_kernels.sh content variant...
# (variant `product` = the library in the tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/abk_$v
  lib=$R/build_variants/lib_$v.so; [ "$v" = product ] && lib=$R/linux-fg_amd/liblinuxfg_hip.so
  LFG_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -- python3 $R/tools/run_stage.py ${STAGE:-motion} ${REPS:-10} $c > /tmp/abk_$v.out 2>&1
  python3 - $v $c ${REPS:-10} ${STAGE:-motion} $(find /tmp/abk_$v -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
tot = 0.0
reps = int(sys.argv[3]); want = "lfg::motion" if sys.argv[4] == "motion" else "lfg::"
for r in csv.DictReader(open(sys.argv[5])):
    if want in r["Name"]:
        n = int(r["Calls"]); per = float(r["TotalDurationNs"]) / 1e3 / reps
        tot += per
        print(f"{sys.argv[2]:12s} {sys.argv[1]:8s} {r['Name'].split('(')[0][5:34]:30s} calls {n:4d} avg {float(r['AverageNs'])/1e3:9.2f} us")
print(f"{sys.argv[2]:12s} {sys.argv[1]:8s} kernels per call ({reps} calls): {tot:9.2f} us")
PY
done
