#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates every file profiles/README.md lists into gpurun_out/profiles_new/.
# usage: tools/collect_profiles.sh rNN
set -e
TAG=${1:-r04}
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
N=30
export LFG_STAGE_LANES=3      # the counter passes run the step as the headline does: three frames in flight (the lean kernel runs only there)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/run_stage.py pipeline $N > /dev/null 2>&1
done
python3 $R/tools/pmc_per_step.py $N /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $OUT/${TAG}_hbm_traffic_pmc.txt
rm -rf /tmp/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d /tmp/pmc_sq -- python3 $R/tools/run_stage.py pipeline $N > /dev/null 2>&1
python3 $R/tools/pmc_per_step.py $N /tmp/pmc_sq > $OUT/${TAG}_sq_counters.txt
unset LFG_STAGE_LANES
cp $OUT/${TAG}_hbm_traffic_pmc.txt $OUT/${TAG}_sq_counters.txt $R/profiles/    # (this box's copy of the tree: the bench lines below quote the tables of the library they ran)
stats pipeline --steps 300 --warmup 6                      # the default command: three frames in flight (kernels of neighbouring steps overlap)
stats pipeline_one_lane --steps 200 --warmup 5 --in-flight 1   # one frame at a time: the durations bench.py's stages / dominant_stage quote
stats scale --workload scale --steps 2000 --warmup 50
# the interpolate stage alone on content where it samples (static: zero vectors, 116 MB moved) and on the pan (both samples rejected: 49.8 MB)
for c in static translated; do
  rm -rf /tmp/prof_interp_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_interp_$c -- python3 $R/tools/run_stage.py interpolate 400 $c > /dev/null 2>&1
  cp $(find /tmp/prof_interp_$c -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_interpolate_${c}_kernel_stats.csv
done
LFG_MOTION_MODE=1 stats pipeline_exact_only --steps 5 --warmup 1
stats config5_one_lane --input 4k --factors 0.25,0.5,0.75 --in-flight 1 --steps 40 --warmup 4     # BASELINE config 5 on one GPU: 4K -> 8K, three generated frames per pair
stats config5 --input 4k --factors 0.25,0.5,0.75 --steps 60 --warmup 6
echo "kernel stats done"
python3 $R/bench.py > $OUT/${TAG}_pipeline_bench.json 2> /tmp/bench.err
python3 $R/bench.py --workload scale > $OUT/${TAG}_scale_bench.json 2>> /tmp/bench.err
python3 $R/bench.py --in-flight 1 --no-extras --no-cpu-baseline > $OUT/${TAG}_pipeline_one_lane_bench.json 2>> /tmp/bench.err   # one frame at a time
python3 $R/bench.py --workload pipeline_input_res --no-extras --no-cpu-baseline > $OUT/${TAG}_pipeline_input_res_bench.json 2>> /tmp/bench.err   # labelled variant: motion + interpolate at input resolution, then scale
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_pipeline_bench_driver_shape.json 2>> /tmp/bench.err   # the driver's command: a 7 ms region, repeated (the `repeats` field)
python3 $R/bench.py --input 4k --factors 0.25,0.5,0.75 --in-flight 1 --no-extras --no-cpu-baseline --steps 200 --warmup 10 > $OUT/${TAG}_config5_one_lane_bench.json 2>> /tmp/bench.err
python3 $R/bench.py --input 4k --factors 0.25,0.5,0.75 --no-extras --no-cpu-baseline --steps 300 --warmup 12 > $OUT/${TAG}_config5_bench.json 2>> /tmp/bench.err
rc=0; python3 $R/bench.py --gpus 2 --steps 5 > $OUT/${TAG}_gpus2_on_one_gpu.txt 2>&1 || rc=$?; echo "exit code $rc" >> $OUT/${TAG}_gpus2_on_one_gpu.txt   # the launcher's refusal on a box with one GPU
echo "bench done"
echo "pmc done"; ls -la $OUT
