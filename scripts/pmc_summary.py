#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/<pass>/**/_counter_collection.csv) per kernel: every upx_*
kernel, keyed by the name bench.py reports (template arguments kept, the argument list dropped)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"pmc_{tag}")
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
# one CSV per pass: the newest (gpurun merges every run's files into the same directory)
paths = []
for pass_dir in sorted(glob.glob(os.path.join(root, "*", ""))):
    found = glob.glob(os.path.join(pass_dir, "**", "*_counter_collection.csv"), recursive=True)
    if found:
        paths.append(max(found, key=os.path.getmtime))
for path in paths:
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        m = re.search(r"(upx_\w+(<.*>)?)\(", name)
        if not m:
            continue
        key = m.group(1)
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[key][row["Counter_Name"]].add(row["Dispatch_Id"])
out = {}
for k in sorted(acc):
    out[k] = {c: acc[k][c] / max(len(calls[k][c]), 1) for c in sorted(acc[k])}
    out[k]["_launches_seen"] = max(len(v) for v in calls[k].values())
print(json.dumps(out, indent=1))
