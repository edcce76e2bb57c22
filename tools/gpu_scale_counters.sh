#!/bin/bash
# On the GPU box: cache / fetch counters of the 2x scale kernel (LFG_LIB selects the build).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/scale_counters.txt
: > $OUT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_IFETCH" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pc_$i -- python3 $R/tools/run_stage.py scale 20 > /tmp/pc_$i.out 2>&1 || echo "set $i failed: $set" >> $OUT
  python3 $R/tools/pmc_summary.py /tmp/pc_$i | grep -v copyBuffer | sed "s#/tmp/##" >> $OUT
done
cat $OUT
