#!/usr/bin/env python3
"""Per-kernel counter table of a run of `tools/run_stage.py <stage> N` under rocprofv3 --kernel-trace --pmc, one or
more passes (each pass = one directory).  Output (stdout), the format bench.py reads back:

    # lib_sha16 <sha>                         the library the run loaded
    # steps <N>
    kernel <name> launches_per_step <n> mean_us <t> <COUNTER> <mean value per launch> ...
    ...
    per_step <COUNTER> <sum over kernels of mean x launches_per_step> ...

usage: pmc_per_step.py N DIR [DIR ...]
"""
import csv
import glob
import hashlib
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    steps = int(sys.argv[1])
    vals = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
    dur = defaultdict(dict)                            # kernel -> dispatch -> us (of the first pass that saw it)
    launches = {}
    for d in sys.argv[2:]:
        seen = defaultdict(set)
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
                    if not k.startswith("lfg::"):
                        continue
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    seen[k].add(r["Dispatch_Id"])
                    dur[k].setdefault((d, r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, s in seen.items():
            launches[k] = len(s)
    lib = os.environ.get("LFG_LIB") or os.path.join(ROOT, "linux-fg_amd", "liblinuxfg_hip.so")
    print("# lib_sha16", hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16])
    print("# steps", steps)
    totals = defaultdict(float)
    for k in sorted(vals):
        per_step = max(1, round(launches[k] / steps))
        ds = list(dur[k].values())
        line = f"kernel {k} launches_per_step {per_step} mean_us {sum(ds) / len(ds):.2f}"
        for c, v in sorted(vals[k].items()):
            m = sum(v) / len(v)
            line += f" {c} {m:.6g}"
            totals[c] += m * per_step
        print(line)
    print("per_step " + " ".join(f"{c} {v:.6g}" for c, v in sorted(totals.items())))


if __name__ == "__main__":
    main()
