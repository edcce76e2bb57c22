#!/bin/bash
# round 5: host time per call of the communicator path (is the N > 1 step bound by the host's enqueueing?)
set -o pipefail
mkdir -p gpurun_out/r5_comm
timeout -k 10 600 python - > gpurun_out/r5_comm/host.txt 2> gpurun_out/r5_comm/host.err <<'PY'
import sys, os, time, torch, numpy as np
sys.path.insert(0, ".")
import bench, importlib
capi = importlib.import_module("linux_fg_amd.capi")
dev = torch.device("cuda", 0)
w_in, h_in = 1920, 1080
for cus in ("8", "0", None):
    if cus is None: os.environ.pop("LFG_COMM_CUS", None)
    else: os.environ["LFG_COMM_CUS"] = cus
    ctx = capi.Context(0)
    ctx.lanes(3)
    if cus is not None:
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    t = torch.zeros((h_in, w_in, 4), dtype=torch.uint8, device=dev)
    f = capi.Context.wrap(t.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
    t4 = torch.zeros((2 * h_in, 2 * w_in, 4), dtype=torch.uint8, device=dev)
    f4 = capi.Context.wrap(t4.data_ptr(), 2 * w_in, 2 * h_in, capi.FORMAT_RGBA8)
    torch.cuda.synchronize()
    def host(fn, n=300):
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter()
        ctx.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
    print("comm cus", cus)
    print("  scale                 host %.1f us  total %.1f us" % host(lambda: ctx.scale(f, f4)))
    print("  lane_select x2        host %.1f us  total %.1f us" % host(lambda: (ctx.lane_select(1), ctx.lane_select(0))))
    print("  lane_mark             host %.1f us  total %.1f us" % host(lambda: ctx.lane_mark()))
    if cus is not None:
        print("  broadcast_frame       host %.1f us  total %.1f us" % host(lambda: ctx.broadcast_frame(f, 0)))
        print("  broadcast_frame_lane  host %.1f us  total %.1f us" % host(lambda: ctx.broadcast_frame_lane(f, 0)))
        print("  comm_wait             host %.1f us  total %.1f us" % host(lambda: ctx.comm_wait()))
        print("  comm_probe(8, 0)      host %.1f us  total %.1f us" % host(lambda: ctx.comm_probe(8, 0, False)))
        print("  comm_probe(8, 170)    host %.1f us  total %.1f us" % host(lambda: ctx.comm_probe(8, 170, False), 100))
    ctx.close()
PY
echo "rc $?"; cat gpurun_out/r5_comm/host.txt; tail -n 3 gpurun_out/r5_comm/host.err
