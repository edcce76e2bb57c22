#!/bin/bash
# round 5: the persistent kernel's variant for moderate noise (kTier 1) -- exactness with it forced on, frame rates by noise amplitude
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_tier; mkdir -p $O
LFG_TIER_FORCE=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/parity_tier1.log 2>&1; echo "parity suite with the variant forced: rc $?"; tail -2 $O/parity_tier1.log
LFG_TIER_FORCE=1 LFG_FUZZ_CASES=24 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz1.log 2>&1; tail -1 $O/fuzz1.log
LFG_TIER_FORCE=1 LFG_FUZZ_CASES=24 LFG_FUZZ_LANES=3 timeout -k 10 200 python3 tools/fuzz_motion_4k.py > $O/fuzz3.log 2>&1; tail -1 $O/fuzz3.log
rate() { python3 bench.py --content $1 --steps ${2:-300} --warmup 12 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for amp in 1 2 3 4 6 8 12 16 24; do
  export LFG_BENCH_NOISE_AMP=$amp
  a=$(LFG_TIER_FORCE=0 rate noisy); b=$(LFG_TIER_FORCE=1 rate noisy); c=$(rate noisy)
  echo "noise +-$amp  variant off $a  forced $b  by the verdict $c" | tee -a $O/rates.txt
done
unset LFG_BENCH_NOISE_AMP
for c in translated occluded objects; do
  a=$(LFG_TIER_FORCE=0 rate $c); b=$(LFG_TIER_FORCE=1 rate $c); d=$(rate $c)
  echo "$c  variant off $a  forced $b  by the verdict $d" | tee -a $O/rates.txt
done
