#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per kernel: python tools/summarize_pmc.py <dir> [<dir> ...] [--kernel substr]
(one directory per --pmc pass, as MI355X_MICROARCH.md prescribes). FETCH_SIZE / WRITE_SIZE are KiB per dispatch."""
import csv
import glob
import json
import os
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
want = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kernel=")), "rollout")
out = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if want not in row["Kernel_Name"]:
                continue
            key = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])
            acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for (k, c), per in acc.items():
            v = sorted(per.values())
            out.setdefault(k, {})[c] = sum(v) / len(v)
            out[k]["dispatches"] = len(v)
print(json.dumps(out, indent=1))
