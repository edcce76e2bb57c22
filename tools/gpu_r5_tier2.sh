#!/bin/bash
# round 5: the variant's lower threshold for the walks by SADs (300 / 220 / 150), forced on, by noise amplitude
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_tier; mkdir -p $O
rate() { python3 bench.py --content $1 --steps 300 --warmup 12 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for amp in 1 2 3 4; do
  export LFG_BENCH_NOISE_AMP=$amp
  line="noise +-$amp  off $(LFG_TIER_FORCE=0 rate noisy)"
  for v in esm300 esm220 esm150; do line="$line  $v $(LFG_LIB=$R/build_variants/lib_$v.so LFG_TIER_FORCE=1 rate noisy)"; done
  echo "$line" | tee -a $O/rates2.txt
done
