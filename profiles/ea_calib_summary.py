#!/usr/bin/env python3
"""Per kernel and dispatch of profiles/micro/ea_calib: the counters of the passes of profiles/ea_calib.sh beside the known byte counts."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
exp = json.load(open(os.path.join(out, "expected.json")))
res = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ("rd", "dram", "fetch", "tcc"):
    for f in glob.glob(os.path.join(out, grp, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            name = r["Kernel_Name"].split("(")[0]
            res[name][grp + ":" + r["Counter_Name"]].append(float(r["Counter_Value"]))
times = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        times[r["Name"].split("(")[0]] = float(r["AverageNs"])
summary = {"expected": exp, "kernels": {}}
for k, cs in res.items():
    d = {c: v for c, v in cs.items()}
    d["avg_ns"] = times.get(k)
    summary["kernels"][k] = d
json.dump(summary, sys.stdout, indent=1, sort_keys=True)
