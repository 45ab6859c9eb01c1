#!/usr/bin/env python3
"""kernel_stats.csv of a `rocprofv3 --kernel-trace --stats --output-format csv -d DIR` run -> one line per kernel (short name, calls, average ms).
usage: python profiles/kstat.py DIR [min_total_ms]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if float(r["TotalDurationNs"]) < floor * 1e6:
        continue
    m = re.search(r"wrapped_(\w+?)_config", n)
    short = "rocprim:" + m.group(1) if m else re.sub(r"^void ", "", n).split("(")[0][-46:]
    print("%-48s %5s avg %.3f ms" % (short, r["Calls"], float(r["AverageNs"]) / 1e6))
