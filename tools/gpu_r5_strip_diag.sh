#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5_strip_diag; mkdir -p $O
for l in 1 3; do
python3 bench.py --in-flight $l --steps 200 --warmup 6 --no-extras --no-cpu-baseline > $O/b$l.json 2>$O/b$l.err
python3 -c "
import json; d=json.loads(open('$O/b$l.json').read().strip().splitlines()[-1]); print('lanes $l value', d['value'], 'fallback', d['roofline']['dominant_stage'].get('fallback_tiles'), 'stages', {k:v['avg_ms'] for k,v in d['stages'].items()})"
done
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4k_modes_agree or strip" > $O/t.log 2>&1; grep -n "Error\|assert \|^E " $O/t.log | head; tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/run_stage.py pipeline 40 translated > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-150 | head -14
true
