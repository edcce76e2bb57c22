#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5_strip_diag; mkdir -p $O
for l in 1 3; do
python3 bench.py --in-flight $l --steps 200 --warmup 6 --no-extras --no-cpu-baseline > $O/b$l.json 2>$O/b$l.err
python3 -c "
import json; d=json.loads(open('$O/b$l.json').read().strip().splitlines()[-1]); print('lanes $l value', d['value'], 'fallback', d['roofline']['dominant_stage'].get('fallback_tiles'), 'stages', {k:v['avg_ms'] for k,v in d['stages'].items()})"
done
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4k_modes_agree or strip" > $O/t.log 2>&1; grep -n "^E " $O/t.log | head -5; tail -1 $O/t.log
tools/gpu_motion_stats.sh translated
LFG_LIB=$R/build_variants/lib_stamps.so python3 tools/run_stage.py motion 4 translated 2> $O/stamps.txt > /dev/null; grep -v "^unit \|late:\|  tile" $O/stamps.txt | cut -c1-400 | head -40
true
