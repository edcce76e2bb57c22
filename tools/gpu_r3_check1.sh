#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c1
mkdir -p $O
cd $R
echo "== gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?"; tail -3 $O/gpu_tests.log
echo "== default bench (driver shape)"; ( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real; tail -c 600 $O/bench_driver.err
echo "== default bench"; ( time timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -c 600 $O/bench_default.err
echo "== --gpus 2 on one GPU"; python3 bench.py --gpus 2 --steps 5 > $O/gpus2.out 2>&1; echo "rc=$?"; cat $O/gpus2.out
echo "== rehearsal: two ranks share the card (no RCCL)"; LFG_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/share2.json 2> $O/share2.err; echo "rc=$?"; tail -c 400 $O/share2.err; cat $O/share2.json | cut -c1-400
