#!/bin/bash
# round 5: which kernels carry a motion call on uncorrelated content and under noise (one frame at a time)
set -o pipefail
mkdir -p gpurun_out/r5_comm
cat > /tmp/one.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
r = bench.measure_with_communicator(torch, capi, sharding, dev, 0, 1, contents=(sys.argv[1],), probe_us=0, use_comm=False, steps=64)
print("content", sys.argv[1], os.environ.get("LFG_BENCH_NOISE_AMP"), {k[:12]: v["frames_per_s"] for k, v in r["by_content"].items()}, flush=True)
PY
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
for c in uncorrelated noisy:8 occluded; do
  name=${c%%:*}; amp=${c##*:}; [ "$amp" = "$c" ] && amp=2
  LFG_BENCH_NOISE_AMP=$amp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 /tmp/one.py $name > /tmp/one.out 2> /tmp/one.err; grep "^content" /tmp/one.out || { tail -n 20 /tmp/one.err; exit 1; }
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -n 1)
  echo "== $c" >> $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels2.txt
  python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels2.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("void lfg::motion") or r["Name"].startswith("lfg::motion"):
        print("%-60s calls %6s avg %9.2f us  max %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf /tmp/prof_$name
done
cat $GRAFT_REPO_ROOT/gpurun_out/r5_comm/kernels2.txt
