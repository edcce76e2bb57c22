#!/bin/bash
# frames/s of library variants, one content, frames in flight as given.  usage: gpu_ab_bench.sh content lanes variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; n=$2; shift 2
cd $R
for rep in 1 2; do for v in "$@"; do
  LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 200 python3 bench.py --content $c --in-flight $n --steps ${STEPS:-400} --warmup 12 --no-extras --no-cpu-baseline > /tmp/abb.json 2> /tmp/abb.err || { echo "bench $v failed"; tail -3 /tmp/abb.err; }
  python3 - $v $c $n <<'PY'
import json, sys
d = json.loads(open('/tmp/abb.json').read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} lanes {sys.argv[3]} {sys.argv[1]:10s} {d['value']:8.1f} frames/s  motion {d['stages']['motion']['avg_ms']:.4f} ms")
PY
done; done
