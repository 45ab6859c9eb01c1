#!/usr/bin/env python3
"""Per-kernel medians of the rocprofv3 PMC passes made by profiles/prof_step.sh -> JSON on stdout.
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KB and on gfx950 FETCH_SIZE counts half of the
streamed read bytes (/opt/skills/guides/MI355X_MICROARCH.md, "HBM")."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def main():
    out = sys.argv[1]
    res = collections.defaultdict(dict)
    for grp in ("fetch", "write", "sq", "tcc", "inst"):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out, grp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in per.items():
            for c, v in cs.items():
                res[k][c] = med(v)
                res[k].setdefault("launches_seen", len(v))
    stats = {}
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
    for k, d in res.items():
        if "FETCH_SIZE" in d:
            d["hbm_bytes"] = int((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024)
        if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
        if d.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if d.get(c) is not None:
                    d[c + "_frac"] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        d.update(stats.get(k, {}))
        if d.get("SQ_INSTS_VALU") and d.get("avg_ns"):
            # a wave's vector instruction occupies its SIMD for 4 cycles (64 lanes over 16): the share of the launch during which the
            # 1024 SIMDs of the chip would be issuing them at 2.4 GHz -- the ALU-side roofline of integer kernels like these
            d["valu_issue_frac_at_2.4GHz"] = round(d["SQ_INSTS_VALU"] * 4 / (1024 * 2.4 * d["avg_ns"]), 3)
    json.dump({"kernels": res}, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
