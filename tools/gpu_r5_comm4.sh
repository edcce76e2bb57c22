#!/bin/bash
# round 5: the same loop on a first, second, third ... context of one process (hardware queues of the streams?)
set -o pipefail
mkdir -p gpurun_out/r5_comm
cat > /tmp/ctxs.py <<'PY'
import sys, os, torch
sys.path.insert(0, ".")
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
mode = sys.argv[1]
for i in range(5):
    use = (mode == "comm") or (mode == "mixed" and i % 2 == 0)
    r = bench.measure_with_communicator(torch, capi, sharding, dev, 0, 3, contents=("translated",), probe_us=0, use_comm=use)
    print(mode, "queues", os.environ.get("GPU_MAX_HW_QUEUES"), "context", i, "comm" if use else "no comm", {k[:12]: v["frames_per_s"] for k, v in r["by_content"].items()}, flush=True)
PY
export LFG_COMM_CUS=0
timeout -k 10 200 python /tmp/ctxs.py nocomm > gpurun_out/r5_comm/ctxs.txt 2>/dev/null && \
timeout -k 10 200 python /tmp/ctxs.py comm >> gpurun_out/r5_comm/ctxs.txt 2>/dev/null && \
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python /tmp/ctxs.py nocomm >> gpurun_out/r5_comm/ctxs.txt 2>/dev/null && \
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python /tmp/ctxs.py comm >> gpurun_out/r5_comm/ctxs.txt 2>/dev/null
echo "rc $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/r5_comm/ctxs.txt
