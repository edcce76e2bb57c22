#!/bin/bash
# usage: tools/try_variants.sh <bench args...> ; runs bench.py once per build_variants/lib_*.so
for f in build_variants/lib_*.so; do
  cp "$f" linux-fg_amd/liblinuxfg_hip.so
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$f', {k:v['avg_ms'] for k,v in d['stages'].items()}, d['value'])"
done
