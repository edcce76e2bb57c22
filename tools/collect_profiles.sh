#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates every file profiles/README.md lists into gpurun_out/profiles_new/.
# usage: tools/collect_profiles.sh rNN
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # $1 = name, rest = bench args [env prefix via LFG_MOTION_MODE]
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extras \
      > $OUT/${TAG}_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
}
stats pipeline --steps 300 --warmup 6                      # the default command: three frames in flight (kernels of neighbouring steps overlap)
stats pipeline_one_lane --steps 200 --warmup 5 --in-flight 1   # one frame at a time: the durations bench.py's stages / dominant_stage quote
stats scale --workload scale --steps 2000 --warmup 50
LFG_MOTION_MODE=1 stats pipeline_exact_only --steps 5 --warmup 1
echo "kernel stats done"
python3 $R/bench.py > $OUT/${TAG}_pipeline_bench.json 2> /tmp/bench.err
python3 $R/bench.py --workload scale > $OUT/${TAG}_scale_bench.json 2>> /tmp/bench.err
python3 $R/bench.py --in-flight 1 --no-extras --no-cpu-baseline > $OUT/${TAG}_pipeline_one_lane_bench.json 2>> /tmp/bench.err   # one frame at a time
python3 $R/bench.py --workload pipeline_input_res --no-extras --no-cpu-baseline > $OUT/${TAG}_pipeline_input_res_bench.json 2>> /tmp/bench.err   # labelled variant: motion + interpolate at input resolution, then scale
echo "bench done"
N=10
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/run_stage.py pipeline $N > /dev/null 2>&1
done
python3 $R/tools/pmc_per_step.py $N /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $OUT/${TAG}_hbm_traffic_pmc.txt
rm -rf /tmp/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d /tmp/pmc_sq -- python3 $R/tools/run_stage.py pipeline $N > /dev/null 2>&1
python3 $R/tools/pmc_per_step.py $N /tmp/pmc_sq > $OUT/${TAG}_sq_counters.txt
echo "pmc done"; ls -la $OUT
