#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc/--kernel-trace CSV output: per kernel, mean duration and mean counter values.
usage: pmc_summary.py DIR [DIR ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"== {path}")
        for k in acc:
            ds = list(dur[k].values())
            print(f"{k}: launches {len(ds)}, mean {sum(ds) / len(ds):.2f} us (min {min(ds):.2f})")
            for c, v in sorted(acc[k].items()):
                print(f"    {c:28s} {sum(v) / len(v):.4g}")
