#!/bin/bash
# headline numbers for library variants / env settings, alternating: usage gpu_quick_bench.sh "ENV=.. ENV2=.." ...   (each argument: env assignments, "-" for none)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for e in "$@"; do
  [ "$e" = "-" ] && e=""
  for lanes in ${LANES:-3 1}; do
    env $e timeout -k 10 200 python3 bench.py --content ${CONTENT:-translated} --in-flight $lanes --steps ${STEPS:-600} --warmup 12 --no-extras --no-cpu-baseline > /tmp/qb.json 2> /tmp/qb.err || { echo "bench failed"; tail -3 /tmp/qb.err; }
    python3 - "$e" $lanes <<'PY'
import json, sys
d = json.loads(open('/tmp/qb.json').read().strip().splitlines()[-1])
v = d.get('verified') or {}
print(f"{sys.argv[1] or '(default)':40s} lanes {sys.argv[2]} {d['value']:8.1f} frames/s  {d['ms_per_step']:.4f} ms/step  motion {d['stages'].get('motion', {}).get('avg_ms', 0):.4f} ms  verified {v.get('ok')}")
PY
  done
done; done
