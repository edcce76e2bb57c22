#!/bin/bash
# quick check of a motion change: the motion parity tests, then frames/s on the contents given (default: all of the sweep's)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG:-quick}
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "motion or fuzz or rim or hand or lanes or noise" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for c in ${CONTENTS:-translated objects occluded noisy static uncorrelated}; do
  for n in ${LANES:-3 1}; do
    timeout -k 10 200 python3 bench.py --content $c --in-flight $n --steps ${STEPS:-300} --warmup 10 --no-extras --no-cpu-baseline > $O/b_${c}_$n.json 2> $O/b_${c}_$n.err || echo "bench $c $n failed"
    python3 - $O/b_${c}_$n.json $c $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:13s} lanes {sys.argv[3]}: {d['value']:8.1f} frames/s  motion {d['stages']['motion']['avg_ms']:.4f} ms  fallback {d['roofline']['dominant_stage'].get('fallback_tiles')}")
PY
  done
done
