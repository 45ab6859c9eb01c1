#!/usr/bin/env python3
"""Per-kernel medians of the rocprofv3 PMC passes made by profiles/prof_step.sh -> JSON on stdout.

What the traffic figure is (round 5; calibrated on this stack with kernels of known byte counts, profiles/micro/ea_calib.hip ->
profiles/r05_ea_calib/summary.json):
  * FETCH_SIZE = 64 B x TCC_EA0_RDREQ (TCC_BUBBLE, "128-byte requests", and TCC_EA0_RDREQ_32B both read 0 on gfx950).
  * A wide coalesced stream (16 B per lane) makes ONE request per 128 bytes: FETCH_SIZE shows HALF its bytes (4 GiB read, 2 GiB
    counted) -- the guide's x2 (/opt/skills/guides/MI355X_MICROARCH.md, "HBM").
  * A random 4-, 8- or 16-byte read of a table larger than the caches makes ONE request per access, 67.1-67.8 M requests for 64 Mi
    accesses at 1.39-1.52 ms per launch whatever the width: at 64 B each that is 2.9 TB/s of random fills, at 128 B it would be
    5.7 TB/s -- more than the same chip streams (5.2 TB/s, same run): they are 64-byte requests and FETCH_SIZE is right as it stands.
  * So the x2 belongs to the coalesced share only.  The counters cannot tell the two kinds apart (one request either way), the
    kernel's source can: fabric_bytes = FETCH + min(FETCH, known coalesced input bytes / 2) + WRITE, with the coalesced input of every
    kernel stated below (STREAM_IN); fabric_bytes_max = 2 x FETCH + WRITE (round 4's figure) is the bound if everything streamed.
  * Infinity-Cache hits cannot be separated from HBM: TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ for every kernel, and a 64 MiB buffer
    streamed three times in a row shows the same 524,360 requests "to DRAM" each time.  The figure is traffic between the L2s and
    the memory side (fabric), an upper bound on HBM bytes."""
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


def stream_in(bench_line):
    """coalesced (16 B per lane, consecutive lanes on consecutive addresses) input bytes per launch of the kernels that stream, from the
    counts of the run itself"""
    if not bench_line:
        return {}
    b = bench_line
    pairs = b["config"]["pairs_per_gpu"]
    st = b.get("scorer_stats", {})
    sym, symw = st.get("kmer_build_sym", 0), st.get("kmer_build_sym_walk", 0)
    tuples = st.get("gated_instances", 0) / (2 if sym else 1)
    items = st.get("recount_items", 0)
    recs = 4 * pairs
    # bases (16) + gate mask (8) of every record: the couples' forms (sym) USE every other record only, but records are 16 / 8 bytes in
    # arrays of all records, so every 64-byte sector is still touched -- the arrays are swept whole (k_gated_hist: 482 MB counted for
    # 480 MB used = 964 MB moved)
    gate = 24 * recs
    return {"k_pool_pack": 101 * recs,                           # (everything it reads is a coalesced stream; two launches per step, the cap below is FETCH itself)
            "k_gated_hist": gate, "k_part_records_g": gate + 16 * tuples, "k_part_tuples_g": 16 * tuples, "k_seg_hist_g": 16 * tuples,
            "k_gated_reduce": 16 * tuples, "k_gated_local": 16 * tuples, "k_walk_items": 24 * recs,
            "k_part_items": 8 * items, "k_recount": 8 * items, "k_compact_partials": 0, "k_bucket_merge": 0}


# the composite scopes of bench.py's kernels_ms_per_step (vdjx_prof_scope names that bracket several launches) and their member kernels,
# as rocprofv3 names them: priced per STEP (sum over the members of median bytes x calls per step)
COMPOSITES = {
    "k_surv_table": ("k_surv_table2", "k_succ_links2"),
    "k_chain_order": ("k_chain_init", "k_chain_jump", "k_chain_len", "k_scan_sums", "k_scan_apply", "k_chain_place", "k_chain_permute", "k_table_remap",
                      "k_partner", "k_chain_words", "k_partner_check"),
    "k_node_order": ("k_rank_keys", "k_rank_scatter", "k_node_emit2"),      # (+ rocPRIM's radix sort kernels: counted in the step total, not named here)
    "k_shard_resolve": ("k_resolve_first", "k_resolve_add", "k_resolve_keep"),
}


def step_span(rows, steps):
    """rows of one counter pass in dispatch order -> those of the last `steps` steps of the process (prof_step.sh runs bench.py so that the
    timed steps are the last thing it launches): from the first k_pool_pack launch of the steps-th step from the end on.  A step has two
    pack launches (primary, secondary pool) unless the secondary is empty; the count per step is taken from the run itself."""
    packs = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_pool_pack") or short(r["Kernel_Name"]).startswith("k_pool_unpack")]
    walks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_walk_items"]
    if not packs or len(walks) < steps:
        return []
    chains = int(os.environ.get("PROF_CHAINS", "1"))          # (--config4: a step is three chains, each with its own packing and walk)
    per = max(1, round(len(packs) / max(1, len(walks)))) * chains
    first = packs[-per * steps] if len(packs) >= per * steps else packs[0]
    return rows[first:]


def main():
    out = sys.argv[1]
    steps_timed = int(os.environ.get("PROF_STEPS", "2"))
    bench_line = None
    try:
        bench_line = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        pass
    STREAM_IN = stream_in(bench_line)
    res = collections.defaultdict(dict)
    step_tot = {}
    for grp in ("fetch", "write", "sq", "tcc", "inst", "ea"):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out, grp, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            for r in rows:
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if grp in ("fetch", "write") and rows:
                # the whole step: every dispatch between the first launch of a timed step and the end of the process, all kernels (the
                # library's, rocPRIM's, the runtime's fills), per step and per kernel name
                key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else None
                if key:
                    rows.sort(key=lambda r: int(r[key]))
                span = step_span(rows, steps_timed)
                agg = collections.defaultdict(lambda: [0.0, 0])
                for r in span:
                    a = agg[short(r["Kernel_Name"])]
                    a[0] += float(r["Counter_Value"]) * 1024
                    a[1] += 1
                step_tot[grp] = {k_: (v_[0] / steps_timed, v_[1] / steps_timed) for k_, v_ in agg.items()}
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
            fetch, write = d["FETCH_SIZE"] * 1024, d.get("WRITE_SIZE", 0.0) * 1024
            stream = STREAM_IN.get(k)
            d["fabric_bytes_max"] = int(2 * fetch + write)              # everything taken as coalesced (round 4's figure)
            d["stream_in_bytes"] = None if stream is None else int(stream)
            # the x2 on the coalesced share only (see the module text); a kernel without a stated stream keeps the bound
            d["hbm_bytes"] = int(fetch + min(fetch, stream / 2) + write) if stream is not None else int(2 * fetch + write)
            d["traffic_rule"] = "FETCH + min(FETCH, stream_in/2) + WRITE" if stream is not None else "2*FETCH + WRITE (no stated stream: bound)"
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
    # ---- the step as a whole, and the composite scopes per step
    step = None
    if "fetch" in step_tot:
        fk, wk = step_tot["fetch"], step_tot.get("write", {})
        tot_f = tot_w = tot_cal = 0.0
        per_kernel = {}
        for k_ in set(fk) | set(wk):
            f_, n_ = fk.get(k_, (0.0, 0))
            w_ = wk.get(k_, (0.0, 0))[0]
            stream = STREAM_IN.get(k_)
            cal = f_ + (min(f_, stream * n_ / 2) if stream is not None else f_) + w_       # (the x2 on the stated coalesced share; no stated stream: the bound)
            per_kernel[k_] = {"fetch_bytes_per_step": int(f_), "write_bytes_per_step": int(w_), "launches_per_step": n_, "fabric_bytes_per_step": int(cal)}
            tot_f += f_; tot_w += w_; tot_cal += cal
        step = {"bytes_per_step": int(tot_cal), "bytes_per_step_max": int(2 * tot_f + tot_w), "fetch_counted": int(tot_f), "write_counted": int(tot_w),
                "kernels_counted": len(per_kernel), "steps": steps_timed,
                "rule": "every dispatch of the last `steps` steps of the process (all kernels: the library's, rocPRIM's, fills), FETCH_SIZE / WRITE_SIZE passes; "
                        "per kernel FETCH + min(FETCH, stated coalesced input / 2) + WRITE, kernels without a stated stream at the bound 2*FETCH + WRITE",
                "per_kernel": per_kernel}
        for comp, members in COMPOSITES.items():
            got = [per_kernel[m] for m in members if m in per_kernel]
            if got:
                res[comp]["composite_of"] = [m for m in members if m in per_kernel]
                res[comp]["hbm_bytes_per_step"] = int(sum(g["fabric_bytes_per_step"] for g in got))
                res[comp]["launches_per_step"] = sum(g["launches_per_step"] for g in got)
                ns_ = sum(res[m].get("avg_ns", 0.0) * per_kernel[m]["launches_per_step"] for m in members if m in per_kernel and m in res)
                res[comp]["ns_per_step"] = ns_
                for c in ("l2_hit", "SQ_WAIT_ANY_frac", "lds_conflict_frac"):      # the members' figures weighted by their time
                    ws = [(res[m].get(c), res[m].get("avg_ns", 0.0) * per_kernel[m]["launches_per_step"]) for m in members if m in res and m in per_kernel and res[m].get(c) is not None]
                    if ws and sum(w for _, w in ws) > 0:
                        res[comp][c] = round(sum(v * w for v, w in ws) / sum(w for _, w in ws), 4)
    json.dump({"kernels": res, "step": step}, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
