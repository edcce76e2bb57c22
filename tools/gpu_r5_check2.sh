#!/bin/bash
# round 5, second GPU call: the new tests, the bench line with its new extras
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_check2; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean or wrong_guesses or small_fallback" > $O/new_tests.log 2>&1; echo "new tests rc $?"; tail -5 $O/new_tests.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_host.py -m gpu -x -q > $O/host_tests.log 2>&1; echo "host tests rc $?"; tail -2 $O/host_tests.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_check2/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'verified', (d.get('verified') or {}).get('ok'))
for k in ('stream', 'intended_semantics', 'pcie_inclusive'):
    print(k, json.dumps(d.get(k))[:1500])
PY
