#!/bin/bash
# the pipeline with lfg_interpolate_frames in the north-star order (fused) against one call per stage, alternating.  usage: [SEMANTICS=intended] gpu_fused_mi.sh [content ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in ${@:-translated}; do for n in 3 1; do for rep in 1 2; do for f in "" "--fused-motion-interpolate"; do
  timeout -k 10 200 python3 bench.py --content $c --in-flight $n --steps ${STEPS:-400} --warmup 12 --no-extras --no-cpu-baseline ${SEMANTICS:+--semantics $SEMANTICS} $f > /tmp/fm.json 2> /tmp/fm.err || { echo "bench failed"; tail -3 /tmp/fm.err; }
  python3 - "$c" $n "${f:-staged}" <<'PY'
import json, sys
d = json.loads(open('/tmp/fm.json').read().strip().splitlines()[-1])
import os
print(f"{os.environ.get('SEMANTICS', 'reference'):10s} {sys.argv[1]:12s} lanes {sys.argv[2]} {sys.argv[3]:28s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.4f} ms/step")
PY
done; done; done; done
