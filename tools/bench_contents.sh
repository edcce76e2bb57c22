#!/bin/bash
# On the GPU box: bench.py for every synthetic content (one line each) and the input-resolution variant.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for c in translated occluded objects noisy static uncorrelated fade; do python3 bench.py --content $c 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), 'fps; motion ms', round(d['stages']['motion']['avg_ms'],3), 'fallback', d['roofline'].get('fallback_tiles'), 'rec/px', d['roofline'].get('candidates_recorded_per_pixel'))"; done
python3 bench.py --workload pipeline_input_res 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('input_res', round(d['value'],1), 'fps')"
