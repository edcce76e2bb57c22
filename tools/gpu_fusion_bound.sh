#!/bin/bash
# SURVEY.md 8(f)-1, north-star order: an UPPER BOUND on what keeping the motion vectors on-chip could save.  The diagnostic
# build `mvfree` (tools/build_variant.sh, ALSO_INTERPOLATE=1 -DLFG_DIAG_MV_UNIFORM -DLFG_DIAG_NO_MV_STORE) removes the vectors'
# whole round trip through memory -- the prefilter's settled segments do not store theirs, the interpolate kernel takes one
# vector for the frame instead of loading 16.6 MB -- and keeps every other byte and instruction of the step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in 3 1; do for rep in 1 2 3; do for v in ${VARIANTS:-new mvuniform mvnostore}; do
  LFG_LIB=$R/build_variants/lib_$v.so timeout -k 10 200 python3 bench.py --in-flight $n --steps 1500 --warmup 20 --no-extras --no-cpu-baseline > /tmp/fb.json 2> /tmp/fb.err || { echo "bench $v failed"; tail -3 /tmp/fb.err; }
  python3 - $v $n <<'PY'
import json, sys
d = json.loads(open('/tmp/fb.json').read().strip().splitlines()[-1])
st = d["stages"]
print(f"frames in flight {sys.argv[2]}  {sys.argv[1]:7s} {d['value']:8.1f} frames/s  step {d['ms_per_step']*1e3:7.2f} us   motion {st['motion']['avg_ms']*1e3:7.2f}  interpolate {st['interpolate']['avg_ms']*1e3:6.2f} us (one call at a time)")
PY
done; done; done
