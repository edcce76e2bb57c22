#!/bin/bash
# On the GPU box: ablation timings of the 2x scale kernel + the streaming reference + instruction-fetch counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
bash $R/tools/gpu_scale_variants.sh 2>&1 | head -20
echo "== bench_stream under rocprof (kernel-side durations)"
rm -rf /tmp/ps; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- $R/tools/bench_stream > /tmp/ps.out 2>&1
cat /tmp/ps.out | tail -5
python3 - $(find /tmp/ps -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}")
PY
echo "== counters with ifetch/icache in their name"
rocprofv3 -L 2>/dev/null | grep -i -E "ifetch|icache|inst_level|SQC_" | head -40
