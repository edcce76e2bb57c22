#!/bin/bash
# SQ counters of tools/bench_batch_eval variants.  usage: gpu_bbe_pmc.sh binary...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pm_bbe
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm_bbe -- $R/tools/$v 8 512 > /tmp/pm_bbe.out 2>&1
    python3 - $v /tmp/pm_bbe <<'PY'
import csv, glob, os, sys
from collections import defaultdict
v, d = sys.argv[1:3]
vals = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "batch_eval" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    if len(ids) < 3: continue
    for r in rows:
        if int(r["Dispatch_Id"]) == ids[2]:          # dispatches: check, mode 0 warm-up, mode 0 timed, mode 1 ...
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, "mode 0:", " ".join(f"{k}={sum(x):.4g}" for k, x in sorted(vals.items())))
PY
  done
done
