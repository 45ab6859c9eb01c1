#!/usr/bin/env python3
"""Turn a rocprofv3 run triple (kernel stats, FETCH_SIZE pass, WRITE_SIZE pass) into the committed summaries.

    python profiles/make_summary.py <tag> <stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json>

writes profiles/<tag>_kernel_stats.csv (copy), profiles/<tag>_traffic.json, profiles/traffic_latest.json and
prints the markdown table of profiles/README.md.
FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of the streamed read bytes
(/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): hbm_bytes = (2*FETCH + WRITE) * 1024."""
import collections
import csv
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n.split("<")[0]


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def main():
    tag, stats, fetch, write, bjson = sys.argv[1:6]
    shutil.copy(stats, os.path.join(HERE, f"{tag}_kernel_stats.csv"))
    shutil.copy(bjson, os.path.join(HERE, f"{tag}_bench.json"))
    f, w = agg(fetch), agg(write)
    b = json.load(open(bjson))
    pairs = b["config"]["pairs_per_gpu"]
    traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --no-cpu --parity-sample 0`; KB per launch "
                       "(median over the workload's launches); hbm_bytes = (2*FETCH + WRITE)*1024 (gfx950: FETCH_SIZE counts half the "
                       "streamed read bytes; calibrated on k_pool_pack: 283 MB read -> FETCH 138,130 KB, 269 MB written -> WRITE 262,507 KB)",
               "pairs_per_gpu": pairs, "kernels": {}}
    for k_ in f:
        fv, wv = sorted(f[k_]), sorted(w.get(k_, [0]))
        fk, wk = fv[len(fv) // 2], wv[len(wv) // 2]
        if k_ == "k_pool_pack":        # two launches per step (primary, secondary): report their sum
            fk, wk = sum(fv) / (len(fv) / 2), sum(wv) / (len(wv) / 2)
        traffic["kernels"][k_] = {"fetch_kb": fk, "write_kb": wk, "hbm_bytes": int((2 * fk + wk) * 1024)}
    json.dump(traffic, open(os.path.join(HERE, f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(os.path.join(HERE, "traffic_latest.json"), "w"), indent=1, sort_keys=True)
    st = {}
    for r in csv.DictReader(open(stats)):
        st[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    ab = bench.algorithmic_bytes_per_pair(35)
    sb = bench.scorer_bytes(b["scorer_stats"], b["counts"]["windows"], b["counts"]["n_contigs_rank"], 35)
    kms = b["kernels_ms_per_step"]
    print("| kernel | ms/step (HIP events) | rocprofv3 avg ms | alg. bytes/launch | achieved GB/s | frac of 8 TB/s | HBM traffic/launch (PMC) | traffic / alg. |")
    print("|---|---|---|---|---|---|---|---|")
    for k_, ms in sorted(kms.items(), key=lambda kv: -kv[1]):
        algb = ab[k_] * pairs if k_ in ab else sb.get(k_)
        rp = st.get(k_)
        tr = traffic["kernels"].get(k_, {}).get("hbm_bytes")
        ach = algb / (ms * 1e-3) / 1e9 if algb else None
        print(f"| {k_} | {ms:.3f} | {rp[1] / 1e6:.3f} | {algb / 1e6:.0f} MB | {ach:.0f} | {ach / 8000:.4f} | {tr / 1e6:.0f} MB | {tr / algb:.2f} |"
              if algb and rp and tr else f"| {k_} | {ms:.3f} | {(rp[1] / 1e6 if rp else 0):.3f} | - | - | - | {(tr or 0) / 1e6:.0f} MB | - |")


if __name__ == "__main__":
    main()
