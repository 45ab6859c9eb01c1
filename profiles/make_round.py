#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.json (rocprofv3 PMC passes of `bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --parity-sample 0`, reduced by
summarize_pmc.py) + the bench line of the same run -> profiles/<round>_traffic.json (what bench.py reports as roofline.traffic,
stamped with the commit the passes were taken at) and the per-kernel table of profiles/README.md on stdout.

    python profiles/make_round.py r03a r03 $(git rev-parse --short HEAD)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402


def main():
    tag, rnd, commit = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
    pm = json.load(open(os.path.join(HERE, f"{tag}_pmc_summary.json")))["kernels"]
    b = json.load(open(os.path.join(HERE, f"{tag}_bench.json")))
    pairs, k = b["config"]["pairs_per_gpu"], 35
    alias = {"k_part_records": "k_part_records_g", "k_part_tuples": "k_part_tuples_g", "k_seg_hist": "k_seg_hist_g"}
    back = {v: k_ for k_, v in alias.items()}
    tr = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (profiles/prof_step.sh), median per launch; "
                  "both counters are in KB; on gfx950 FETCH_SIZE is 64 B per read request and a coalesced 16-byte-per-lane stream is one request "
                  "per 128 B (profiles/r05_ea_calib): hbm_bytes = (FETCH + min(FETCH, stream_in/2) + WRITE)*1024 where the kernel's streamed "
                  "input is stated (summarize_pmc.py STREAM_IN), else the bound (2*FETCH + WRITE)*1024; traffic_rule says which",
          "commit": commit, "passes": tag, "pairs_per_gpu": pairs, "k": k, "windows": "generator", "kernels": {}}
    for name, v in pm.items():
        if "hbm_bytes" in v and name.startswith("k_"):
            tr["kernels"][back.get(name, name)] = {"hbm_bytes": v["hbm_bytes"], "fetch_kb": v.get("FETCH_SIZE"), "write_kb": v.get("WRITE_SIZE"),
                                                   "traffic_rule": v.get("traffic_rule")}
    json.dump(tr, open(os.path.join(HERE, f"{rnd}_traffic.json"), "w"), indent=1, sort_keys=True)
    st = dict(b["scorer_stats"])
    gi = st.get("gated_instances")
    ab = bench.algorithmic_bytes_per_pair(k, 50, gi / pairs if gi else None)
    sb = bench.scorer_bytes(st, b["counts"]["windows"], b["counts"]["n_contigs_rank"], k)
    launches = {"k_pool_pack": 2, "k_map_classify": 5, "k_plan": 2}      # (classify: the windows in four pieces + the contigs, until the strings were read in place)
    for k_, v_ in b.get("roofline_by_kernel", {}).items():               # what the run itself counted
        if isinstance(v_.get("launches_per_step"), (int, float)) and k_ == "k_map_classify":      # (the pack's two launches are one timed scope there)
            launches[k_] = v_["launches_per_step"]
    print("| kernel | ms/step (HIP events) | rocprofv3 avg ms/launch | alg. bytes/step | achieved GB/s | frac of 8 TB/s | fabric traffic/step (PMC) | traffic / alg. | L2 hit | wave-cycles waiting | LDS conflict | VALU issue share |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, ms in sorted(b["kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
        v = pm.get(alias.get(name, name), {})
        algb = ab[name] * pairs if name in ab and name not in ("P", "input", "total", "gated_per_pair") else sb.get(name)
        t = v.get("hbm_bytes")
        t = t * launches.get(name, 1) if t else t
        if algb and t:
            ach = algb / (ms * 1e-3) / 1e9
            print(f"| {name} | {ms:.3f} | {v.get('avg_ns', 0) / 1e6:.3f} | {algb / 1e6:.0f} MB | {ach:.0f} | {ach / 8000:.3f} | {t / 1e6:.0f} MB | {t / algb:.2f} | "
                  f"{v.get('l2_hit')} | {v.get('SQ_WAIT_ANY_frac')} | {v.get('lds_conflict_frac')} | {v.get('valu_issue_frac_at_2.4GHz')} |")
        else:
            print(f"| {name} | {ms:.3f} | {v.get('avg_ns', 0) / 1e6:.3f} | - | - | - | {(t or 0) / 1e6:.0f} MB | - | {v.get('l2_hit')} | {v.get('SQ_WAIT_ANY_frac')} | {v.get('lds_conflict_frac')} | {v.get('valu_issue_frac_at_2.4GHz')} |")


if __name__ == "__main__":
    main()
