#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
LFG_DEBUG=1 LFG_LIB=$R/build_variants/lib_sstats.so python3 - <<'PY' 2>&1 | grep -v "^lfg: motion prefilter\|^lfg: lean\|fallback tile"
import sys; sys.path.insert(0,'.')
import numpy as np
from linux_fg_amd import capi, synth
import bench
w,h=1920,1080
prev_in,curr_in=bench.make_content('translated',w,h,0,0)
c=capi.Context(0)
p,q=c.frame_from(prev_in),c.frame_from(curr_in)
P,C=c.create_frame(2*w,2*h),c.create_frame(2*w,2*h); M=c.create_frame(2*w,2*h,capi.FORMAT_MV_S8X2)
c.scale(p,P); c.scale(q,C); c.motion(P,C,M); c.sync()
print(c.motion_strip_stats()); print("fallback", c.motion_last_stats()[1])
PY
