#!/bin/bash
# SQ counters of the prefilter kernel for library variants on one content.  usage: gpu_ab_pmc.sh content variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    rm -rf /tmp/pm_$v
    LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm_$v -- python3 $R/tools/run_stage.py motion 3 $c > /tmp/pm_$v.out 2>&1
    python3 - $v $c /tmp/pm_$v <<'PY'
import csv, glob, os, sys
from collections import defaultdict
v, c, d = sys.argv[1:4]
vals = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "motion_prefilter" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(c, v, " ".join(f"{k}={sum(x)/len(x):.4g}" for k, x in sorted(vals.items())))
PY
  done
done
