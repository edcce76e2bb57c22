#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a bench.py run: per steady-state step, how long each kernel ran (sum of its launches' durations), how
much of the wall clock had 0 / 1 / 2 / 3+ kernels in flight, and the wall clock per step.  usage: trace_overlap.py kernel_trace.csv [skip_fraction]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "lfg::" not in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "").replace("lfg::", "")[:32]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
t_hi = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - skip)
sel = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
steps = sum(1 for r in sel if r[2].startswith("interpolate_kernel"))
span = t_hi - t_lo
ev = []
for a, b, _ in sel:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
depth, last, hist = 0, t_lo, collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last
    last = t; depth += d
hist[min(depth, 4)] += t_hi - last
per = collections.defaultdict(lambda: [0, 0.0])
for a, b, n in sel:
    per[n][0] += 1; per[n][1] += (b - a)
print(f"window {span/1e3:.0f} us, {steps} steps: {span/1e3/max(steps,1):.1f} us per step")
print("kernels in flight: " + ", ".join(f"{k}{'+' if k == 4 else ''}: {100*v/span:.1f} %" for k, v in sorted(hist.items())))
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"   {n:34s} {c/max(steps,1):5.2f} launches/step  avg {t/c/1e3:8.2f} us  sum/step {t/1e3/max(steps,1):8.2f} us")
