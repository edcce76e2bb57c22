#!/bin/bash
# round 5: what the persistent kernel executes under noise of +-4 levels, default kernel and the variant (SQ counters, one frame at a time)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_tier; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LFG_BENCH_NOISE_AMP=4
for t in 0 1; do
  rm -rf /tmp/pmc_t$t
  LFG_TIER_FORCE=$t rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \
      --output-format csv -d /tmp/pmc_t$t -- python3 $R/tools/run_stage.py motion 12 noisy > /dev/null 2>&1
  echo "== LFG_TIER_FORCE=$t, noise +-4 levels, lfg_motion x 12, one frame at a time" >> $O/pmc.txt
  python3 $R/tools/pmc_per_step.py 12 /tmp/pmc_t$t | grep "prefilter\|lib_sha" | cut -c1-400 >> $O/pmc.txt
done
cat $O/pmc.txt
