#!/bin/bash
# Round-3 probe: config 5 (4K->8K, three factors) with one and three frames in flight, its kernel trace, the motion workspace at
# 8K, and per-wave stamps of the prefilter on the contents the judge's sweep names.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p1
mkdir -p $O
cd $R
set -o pipefail
echo "== config 5, one frame at a time"; timeout -k 10 240 python3 bench.py --input 4k --factors 0.25,0.5,0.75 --in-flight 1 --steps 30 --warmup 4 --no-extras --no-cpu-baseline > $O/c5_lane1.json 2> $O/c5_lane1.err || { echo FAILED lane1; tail -5 $O/c5_lane1.err; }
echo "== config 5, three frames in flight"; timeout -k 10 240 python3 bench.py --input 4k --factors 0.25,0.5,0.75 --in-flight 3 --steps 30 --warmup 6 --no-extras --no-cpu-baseline > $O/c5_lane3.json 2> $O/c5_lane3.err || { echo FAILED lane3; tail -5 $O/c5_lane3.err; }
python3 - <<'PY' > $O/ws.txt 2>&1
import sys; sys.path.insert(0, '.')
from linux_fg_amd import capi
c = capi.Context(0)
for w, h in ((1920,1080),(3840,2160),(7680,4320)):
    print(w, h, c.motion_workspace_size(w, h))
c.lanes(3)
for w, h in ((3840,2160),(7680,4320)):
    print("lanes3", w, h, c.motion_workspace_size(w, h))
PY
cat $O/ws.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 $R/bench.py --input 4k --factors 0.25,0.5,0.75 --in-flight 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/c5_lane1_under_rocprof.json 2> $O/c5_prof.err
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $O/c5_lane1_kernel_stats.csv
cd $R
for c in objects occluded noisy; do
  echo "== stamps $c"
  LFG_LIB=$R/build_variants/lib_stamps.so timeout -k 10 200 python3 tools/run_stage.py motion 4 $c > $O/stamps_$c.txt 2>&1 || echo "stamps $c failed"
done
echo "== stamps 8K pan"
LFG_STAGE_INPUT=3840x2160 LFG_LIB=$R/build_variants/lib_stamps.so timeout -k 10 200 python3 tools/run_stage.py motion 4 translated > $O/stamps_8k.txt 2>&1 || echo "stamps 8k failed"
for c in translated objects occluded noisy; do bash tools/gpu_motion_stats.sh $c > $O/kstats_$c.txt 2>&1; done
LFG_STAGE_INPUT=3840x2160 bash tools/gpu_motion_stats.sh translated > $O/kstats_8k.txt 2>&1
echo done
