#!/bin/bash
# round 5: the fixed visiting order dealt out by LDS bank (order1) against the plain shuffle (order0): frame rates by content
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r5_order; mkdir -p $O
rate() { LFG_LIB=$R/build_variants/lib_$1.so python3 bench.py --content $2 --in-flight ${3:-3} --steps ${4:-300} --warmup 12 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for c in noisy objects occluded translated static; do
  echo "$c  shuffle $(rate order0 $c)  by bank $(rate order1 $c)  shuffle $(rate order0 $c)  by bank $(rate order1 $c)" | tee -a $O/rates.txt
done
echo "uncorrelated  shuffle $(rate order0 uncorrelated 3 30)  by bank $(rate order1 uncorrelated 3 30)" | tee -a $O/rates.txt
echo "fade  shuffle $(rate order0 fade 3 40)  by bank $(rate order1 fade 3 40)" | tee -a $O/rates.txt
for amp in 1 4 8; do export LFG_BENCH_NOISE_AMP=$amp; echo "noise +-$amp  shuffle $(rate order0 noisy)  by bank $(rate order1 noisy)" | tee -a $O/rates.txt; done; unset LFG_BENCH_NOISE_AMP
echo "translated, one lane  shuffle $(rate order0 translated 1)  by bank $(rate order1 translated 1)" | tee -a $O/rates.txt
echo "noisy, one lane  shuffle $(rate order0 noisy 1)  by bank $(rate order1 noisy 1)" | tee -a $O/rates.txt
