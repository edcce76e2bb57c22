#!/bin/bash
# round 5: where a motion call spends its waves' time on the contents in the middle of the table (stamps build, one frame at a time)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5_stamps; mkdir -p $O
for c in objects occluded noisy; do
  LFG_LIB=$R/build_variants/lib_stamps.so timeout -k 10 120 python3 tools/run_stage.py motion 4 $c 2> $O/stamps_$c.txt > /dev/null
  echo "== $c"; grep -v "^unit \|late:\|  tile" $O/stamps_$c.txt | cut -c1-420 | head -40
done
