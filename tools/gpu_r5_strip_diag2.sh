#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in nodpp norec nohard; do
  LFG_LIB=$R/build_variants/lib_sd_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python3 $R/tools/run_stage.py motion 12 translated > /dev/null 2>&1
  f=$(find /tmp/p_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep strip $f | cut -d, -f2-4)  prefilter: $(grep prefilter_kernel $f | cut -d'"' -f4- | cut -d, -f2-4)"
done
true
