#!/bin/bash
# round 5: where does the N > 1 step with a communicator lose its frames?  (masks / probe / events, one at a time)
set -o pipefail
mkdir -p gpurun_out/r5_comm
timeout -k 10 600 python - > gpurun_out/r5_comm/ab.txt 2> gpurun_out/r5_comm/ab.err <<'PY'
import json, sys, os, torch
sys.path.insert(0, ".")
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
for cus in ("0",):
    for probe in (0, -1):
        os.environ["LFG_COMM_CUS"] = cus
        for rep in range(2):
            r = bench.measure_with_communicator(torch, capi, sharding, dev, 0, 3, contents=("translated",), probe_us=max(probe, 0), use_comm=probe >= 0)
            print("cus", cus, "probe_us", probe, {k[:40]: v["frames_per_s"] for k, v in r["by_content"].items()}, flush=True)
PY
echo "rc $?"; cat gpurun_out/r5_comm/ab.txt; tail -n 3 gpurun_out/r5_comm/ab.err
