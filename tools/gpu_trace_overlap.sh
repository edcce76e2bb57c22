#!/bin/bash
# kernel overlap of the pipeline with frames in flight.  usage: [LANES=3] [CONTENT=translated] gpu_trace_overlap.sh "ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  rm -rf /tmp/tov
  for kv in $e; do export "$kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tov -- python3 $R/bench.py --content ${CONTENT:-translated} --in-flight ${LANES:-3} --steps 300 --warmup 9 --no-extras --no-cpu-baseline > /tmp/tov.out 2>/tmp/tov.err
  for kv in $e; do unset "${kv%%=*}"; done
  echo "== ${e:-(default)}"
  python3 $R/tools/trace_overlap.py $(find /tmp/tov -name "*kernel_trace.csv" | head -1)
done
