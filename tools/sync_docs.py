#!/usr/bin/env python3
"""Rewrite the headline numbers of DESIGN.md / README.md from the committed bench lines
(profiles/rNN_pipeline_bench.json, profiles/rNN_pipeline_one_lane_bench.json), so that the prose never drifts from the
files it cites.  usage: tools/sync_docs.py [rNN]   (after tools/collect_profiles.sh rNN and the copy into profiles/)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def last_line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


d = last_line(f"{tag}_pipeline_bench.json")
d1 = last_line(f"{tag}_pipeline_one_lane_bench.json")
sw = {k: (v["frames_per_s"], v.get("motion_ms")) for k, v in d["content_sweep"]["frames_per_s_by_content"].items()}

path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()


def sub(pattern, replacement):
    global s
    s, n = re.subn(pattern, replacement, s, count=1)
    if n != 1:
        raise SystemExit(f"DESIGN.md: pattern not found: {pattern}")


sub(r"\| \*\*\d+\*\* \(one frame at a time: \d+\) \| [\d.]+ ms \|",
    f"| **{sw['translated'][0]:.0f}** (one frame at a time: {d1['value']:.0f}) | {sw['translated'][1]:.2f} ms |")
sub(r"hand-over\) \| \d+ \| [\d.]+ ms \|", f"hand-over) | {sw['occluded'][0]:.0f} | {sw['occluded'][1]:.2f} ms |")
sub(r"hand-over at their boundaries\) \| \d+ \| [\d.]+ ms \|",
    f"hand-over at their boundaries) | {sw['objects'][0]:.0f} | {sw['objects'][1]:.2f} ms |")
sub(r"four- and sixteen-point sums\) \| \d+ \| [\d.]+ ms \|",
    f"four- and sixteen-point sums) | {sw['noisy'][0]:.0f} | {sw['noisy'][1]:.2f} ms |")
sub(r"\| static \(curr = prev\) \| \d+ \| [\d.]+ ms \|",
    f"| static (curr = prev) | {sw['static'][0]:.0f} | {sw['static'][1]:.2f} ms |")
sub(r"every segment searches in full\) \| \d+ \| [\d.]+ ms \|",
    f"every segment searches in full) | {sw['uncorrelated'][0]:.0f} | {sw['uncorrelated'][1]:.1f} ms |")
sub(r"\*\*[\d,]+ interpolated frames/s\*\* \([\d.]+ ms/step; one\nframe at a time [\d,]+ = [\d.]+ ms/step, of which motion [\d.]+;",
    f"**{d['value']:,.0f} interpolated frames/s** ({d['ms_per_step']:.3f} ms/step; one\nframe at a time {d1['value']:,.0f} = "
    f"{d1['ms_per_step']:.3f} ms/step, of which motion {d1['stages']['motion']['avg_ms']:.2f};")
sub(r"motion kernel alone; [\d,]+ / [\d,]+ / [\d,]+ / [\d,]+ on occluded",
    f"motion kernel alone; {sw['occluded'][0]:,.0f} / {sw['objects'][0]:,.0f} / {sw['noisy'][0]:,.0f} / {sw['uncorrelated'][0]:,.0f} on occluded")
sub(r"with the round's last library [\d,]+ → [\d,]+ \(three\); occluded 711 → [\d,]+, noisy 737 → [\d,]+\.",
    f"with the round's last library {d1['value']:,.0f} → {d['value']:,.0f} (three); occluded 711 → {sw['occluded'][0]:,.0f}, "
    f"noisy 737 → {sw['noisy'][0]:,.0f}.")
open(path, "w").write(s)

path = os.path.join(ROOT, "README.md")
r = open(path).read()
r = re.sub(r"\*\*≈[\d,]+ interpolated frames/s\*\*", f"**≈{round(d['value'], -1):,.0f} interpolated frames/s**", r)
r = re.sub(r"[\d,]+ strictly one frame at a time", f"{d1['value']:,.0f} strictly one frame at a time", r)
lo = min(sw[k][0] for k in ("occluded", "objects", "noisy"))
hi = max(sw[k][0] for k in ("occluded", "objects", "noisy"))
r = re.sub(r"[\d,]+–[\d,]+ with occlusions, moving objects or sensor noise added", f"{round(lo, -1):,.0f}–{round(hi, -1):,.0f} with occlusions, moving objects or sensor noise added", r)
open(path, "w").write(r)
print(f"DESIGN.md / README.md: {d['value']:.0f} frames/s, one frame at a time {d1['value']:.0f}, library {d.get('library_sha16')}")
