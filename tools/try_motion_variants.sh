#!/bin/bash
# On the GPU box: motion-stage time (HIP events via bench stage profile) for each build_variants/lib_*.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/linux-fg_amd/liblinuxfg_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so $R/build_variants/lib_*.so; do
  cp "$f" $R/linux-fg_amd/liblinuxfg_hip.so
  python3 - <<PY
import sys, time
sys.path.insert(0, '$R')
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
W, H = 3840, 2160
# the benchmark's frames: a translated 1080p pair, both upscaled on the device (CONTENT=uncorrelated: two noise frames)
import os
if os.environ.get('CONTENT') == 'uncorrelated':
    pin, cin = synth.make_uncorrelated_pair(W // 2, H // 2)
elif os.environ.get('CONTENT') == 'noisy':
    import numpy as np
    pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED); cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
    n = synth.noise_bytes(W // 2, H // 2, (synth.BASE_SEED + 15485863) & 0xFFFFFFFF) % 5
    cin = np.clip(cin.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
else:
    pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED); cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
Pin, Cin = ctx.frame_from(pin), ctx.frame_from(cin)
P, C = ctx.create_frame(W, H), ctx.create_frame(W, H); M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.scale(Pin, P); ctx.scale(Cin, C)
ctx.motion(P, C, M); ctx.sync()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(20): ctx.motion(P, C, M)
ctx.sync()
ms, n = ctx.profile_get(capi.STAGE_MOTION)
print('$f'.split('/')[-1], round(ms / n, 3), 'ms per motion call')
PY
done
cp /tmp/lib_base.so $R/linux-fg_amd/liblinuxfg_hip.so
