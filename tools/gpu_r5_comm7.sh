#!/bin/bash
# round 5: with_communicator standalone and inside bench.py, by GPU_MAX_HW_QUEUES
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_comm; mkdir -p $O
cat > /tmp/one.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi"); sharding = importlib.import_module("linux_fg_amd.sharding")
dev = torch.device("cuda", 0)
r = bench.measure_with_communicator(torch, capi, sharding, dev, 0, 3, contents=("translated",))
print("standalone, queues", os.environ.get("GPU_MAX_HW_QUEUES"), {k[:12]: v["frames_per_s"] for k, v in r["by_content"].items()}, flush=True)
PY
for q in 8 16; do GPU_MAX_HW_QUEUES=$q python3 /tmp/one.py 2>/dev/null | grep standalone; done
for q in 8 16 24; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['with_communicator']
print('in bench.py, queues', d['config']['gpu_max_hw_queues'], 'value', d['value'], {k[:12]: v['frames_per_s'] for k, v in w['by_content'].items()}, 'config5', d['config5']['interpolated_frames_per_s'], 'stream', d['stream']['paced']['frames_per_s'])"
done
