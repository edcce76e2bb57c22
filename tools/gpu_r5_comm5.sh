#!/bin/bash
# round 5: per-kernel durations with the library's streams CU-masked (a communicator's reservation) and not
set -o pipefail
mkdir -p gpurun_out/r5_comm
cat > /tmp/one.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
r = bench.measure_with_communicator(torch, capi, sharding, dev, 0, int(sys.argv[1]), contents=("translated",), probe_us=0)
print("cus", os.environ.get("LFG_COMM_CUS"), "lanes", sys.argv[1], {k[:12]: v["frames_per_s"] for k, v in r["by_content"].items()}, flush=True)
PY
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
for cus in 8 0; do for lanes in 1 3; do
  LFG_COMM_CUS=$cus rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${cus}_${lanes} -o p -- python3 /tmp/one.py $lanes > /tmp/one.out 2> /tmp/one.err; grep "^cus" /tmp/one.out || { tail -n 20 /tmp/one.err; exit 1; }
  f=$(find /tmp/prof_${cus}_${lanes} -name "*kernel_stats.csv" | head -n 1)
  echo "== cus $cus lanes $lanes" >> $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels.txt
  python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %6s avg %9.2f us  total %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf /tmp/prof_${cus}_${lanes}
done; done
cat $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels.txt
