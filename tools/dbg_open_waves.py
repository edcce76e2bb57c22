"""Debug aid: where on the benchmark's upscaled frames does the true motion vector NOT give a small block cost?
(waves that contain such pixels cannot use the partial-distortion test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
W, H = 3840, 2160
pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED); cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
Pin, Cin = ctx.frame_from(pin), ctx.frame_from(cin)
P, C = ctx.create_frame(W, H), ctx.create_frame(W, H); M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.scale(Pin, P); ctx.scale(Cin, C); ctx.sync()
p, c = ctx.download(P).astype(np.int32), ctx.download(C).astype(np.int32)
ctx.motion(P, C, M); ctx.sync(); mv = ctx.download(M)
d = np.zeros((H, W), np.float64)
# curr(Q) = prev(Q - (6,-4)): prev sampled at Q + (-6, +4)
ys, xs = slice(0, H - 4), slice(6, W)
diff = c[ys, xs] - p[4:H, 0:W - 6]
d[ys, xs] = np.sqrt((diff * diff).sum(-1))
d[H - 4:, :] = 510; d[:, :6] = 510
# block cost: positions px-4 .. px+3
cs = np.cumsum(np.cumsum(np.pad(d, ((1, 0), (1, 0))), 0), 1)
S = np.full((H, W), np.nan)
yy, xx = np.mgrid[4:H - 3, 4:W - 3]
S[4:H - 3, 4:W - 3] = cs[yy + 4, xx + 4] - cs[yy - 4, xx + 4] - cs[yy + 4, xx - 4] + cs[yy - 4, xx - 4]
bad = np.argwhere(S > 100)
inner = bad[(bad[:, 0] > 80) & (bad[:, 0] < H - 80) & (bad[:, 1] > 80) & (bad[:, 1] < W - 80)]
print("pixels with cost > 100 at the true vector:", len(bad), "of which away from the rim:", len(inner))
print("nonzero-distance positions away from rim:", int((d[80:H-80, 80:W-80] > 0).sum()))
if len(inner):
    print("rows", np.unique(inner[:, 0])[:40], "cols", np.unique(inner[:, 1])[:40])
    y, x = inner[0]; print("example", y, x, S[y, x], "mv there", mv[y, x])
vals, counts = np.unique(mv.reshape(-1, 2), axis=0, return_counts=True)
o = np.argsort(-counts)[:8]
print("most common vectors", [(tuple(vals[i]), int(counts[i])) for i in o])
di = d[80:H-80, 80:W-80]
print("distance histogram away from rim:", np.histogram(di[di > 0], bins=[0, 1.5, 2.5, 5, 20, 100, 300, 511])[0])
Si = S[80:H-80, 80:W-80]
print("block cost histogram away from rim:", np.histogram(Si[Si > 0], bins=[0, 2, 10, 100, 300, 510, 1000, 1e9])[0])
big = np.argwhere(di > 100)
print("big-distance positions (first 10, +80):", big[:10] + 80)
for (y, x) in (big[:3] + 80):
    print("curr", c[y, x], "prev@true", p[y + 4, x - 6], "input curr", cin[y // 2, x // 2], "input prev", pin[y // 2 + 2, x // 2 - 3])
