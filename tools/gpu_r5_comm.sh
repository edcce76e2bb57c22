#!/bin/bash
# round 5: the CU reservation of a communicator -- its tests, and the N > 1 step with a communicator on one GPU
set -o pipefail
mkdir -p gpurun_out/r5_comm
python -m pytest tests/test_gpu_comm.py -x -q -s -m gpu > gpurun_out/r5_comm/tests.log 2>&1
echo "comm tests rc $?"; tail -n 12 gpurun_out/r5_comm/tests.log
timeout -k 10 300 python - > gpurun_out/r5_comm/with_comm.json 2> gpurun_out/r5_comm/with_comm.err <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
import importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
print(json.dumps(bench.measure_with_communicator(torch, capi, sharding, dev, 0, 3), indent=1))
PY
echo "with_communicator rc $?"; cat gpurun_out/r5_comm/with_comm.json; tail -n 5 gpurun_out/r5_comm/with_comm.err
