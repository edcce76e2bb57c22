#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the prefilter kernel for library variants.  usage: gpu_ab_write.sh content variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pw_$v
    LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pw_$v -- python3 $R/tools/run_stage.py motion 4 $c > /tmp/pw_$v.out 2>&1
    python3 - $v $c /tmp/pw_$v <<'PY'
import csv, glob, os, sys
from collections import defaultdict
v, c, d = sys.argv[1:4]
vals = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("lfg::", "")
        vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
print(c, v, " ".join(f"{k[0]}:{k[1]}={sum(x)/len(x)/1e3:.1f}MB" for k, x in sorted(vals.items()) if "motion" in k[0]))
PY
  done
done
