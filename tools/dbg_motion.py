import sys; sys.path.insert(0,'/root/repo')
import numpy as np, ctypes
from linux_fg_amd import capi, synth
ctx=capi.Context(0)
W,H=3840,2160
p,c=synth.make_pair(W,H,0)
P,C=ctx.frame_from(p),ctx.frame_from(c); M=ctx.create_frame(W,H,capi.FORMAT_MV_S8X2)
ctx.motion(P,C,M); ctx.sync()
print(ctx.motion_last_stats())
import torch
