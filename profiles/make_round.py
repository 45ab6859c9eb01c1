#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.json (rocprofv3 PMC passes of `bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-index-leg --parity-sample 0 [flags]`,
reduced by summarize_pmc.py) + the bench line of the same run -> profiles/<round>_traffic[_<variant>].json (what bench.py reports as
roofline.traffic / step_traffic, stamped with the commit the passes were taken at) and the per-kernel table on stdout.
The variant (none = the default configs[2] step; k25 = configs[3]; shard = --force-shard; config4 = --config4 --gpus 1) is read from the
bench line's config.

    python profiles/make_round.py r06a r06 $(git rev-parse --short HEAD) > profiles/r06_table.md
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402


def main():
    tag, rnd, commit = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
    summ = json.load(open(os.path.join(HERE, f"{tag}_pmc_summary.json")))
    pm, step = summ["kernels"], summ.get("step")
    b = json.load(open(os.path.join(HERE, f"{tag}_bench.json")))
    cfg = b["config"]
    pairs, k, variant = cfg["pairs_per_gpu"], cfg.get("k", 35), cfg.get("variant", "")
    chains = len(cfg.get("chains", ["IGH"]))
    alias = {"k_part_records": "k_part_records_g", "k_part_tuples": "k_part_tuples_g", "k_seg_hist": "k_seg_hist_g"}
    back = {v: k_ for k_, v in alias.items()}
    tr = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (profiles/prof_step.sh), median per launch; "
                  "both counters are in KB; on gfx950 FETCH_SIZE is 64 B per read request and a coalesced 16-byte-per-lane stream is one request "
                  "per 128 B (profiles/r05_ea_calib): hbm_bytes = (FETCH + min(FETCH, stream_in/2) + WRITE)*1024 where the kernel's streamed "
                  "input is stated (summarize_pmc.py STREAM_IN), else the bound (2*FETCH + WRITE)*1024; traffic_rule says which.  Composite scopes "
                  "(several launches under one HIP-event name): hbm_bytes is PER STEP, summed over the member kernels (composite_of).  `step`: every "
                  "dispatch of a timed step, all kernels",
          "commit": commit, "passes": tag, "pairs_per_gpu": pairs, "k": k, "windows": cfg.get("windows", "generator"), "variant": variant, "kernels": {}}
    rbk = b.get("roofline_by_kernel", {})
    per_step = (step or {}).get("per_kernel", {})

    def scopes_per_step(name):          # how often the HIP-event scope of that name is opened per step (k_chain_order: twice; config4: per chain)
        v = rbk.get(name, {}).get("launches_per_step")
        return v if isinstance(v, (int, float)) and v > 0 else 1

    def step_bytes(name):               # counted fabric bytes per STEP of a scope: its kernel's (or, a composite, its member kernels') dispatches of a timed step
        v = pm.get(alias.get(name, name), {})
        members = v.get("composite_of") or [alias.get(name, name)]
        got = [per_step[m_]["fabric_bytes_per_step"] for m_ in members if m_ in per_step]
        return float(sum(got)) if got else None
    for name in b["kernels_ms_per_step"]:
        v = pm.get(alias.get(name, name), {})
        sbytes = step_bytes(name)
        if sbytes is None:
            continue
        tr["kernels"][name] = {"hbm_bytes": int(sbytes / scopes_per_step(name)), "hbm_bytes_per_step": int(sbytes), "scopes_per_step": scopes_per_step(name),
                               "composite_of": v.get("composite_of"), "fetch_kb_median_launch": v.get("FETCH_SIZE"), "write_kb_median_launch": v.get("WRITE_SIZE"),
                               "traffic_rule": v.get("traffic_rule") or "sum over the member kernels' dispatches of a timed step"}
    for name, v in pm.items():          # (kernels outside the step's scopes -- the read index's -- keep the median launch)
        if "hbm_bytes" in v and name.startswith("k_") and back.get(name, name) not in tr["kernels"]:
            tr["kernels"][back.get(name, name)] = {"hbm_bytes": v["hbm_bytes"], "traffic_rule": v.get("traffic_rule")}
    if step:
        tr["step"] = {k_: v_ for k_, v_ in step.items() if k_ != "per_kernel"}
        tr["step"]["note"] = "hbm_bytes of a scope = its dispatches' bytes of a timed step / the scope's openings per step (what bench.py divides by that scope's average duration)"
    json.dump(tr, open(os.path.join(HERE, f"{rnd}_traffic{'_' + variant if variant else ''}.json"), "w"), indent=1, sort_keys=True)
    st = dict(b["scorer_stats"])
    gi = st.get("gated_instances")
    ab = bench.algorithmic_bytes_per_pair(k, 50, gi / pairs if gi else None, sym=bool(st.get("kmer_build_sym")), sym_walk=bool(st.get("kmer_build_sym_walk")))
    sb = bench.scorer_bytes(st, b["counts"]["windows"], b["counts"]["n_contigs_rank"], k)
    svb = bench.survivor_bytes(int(b["counts"].get("nodes", 0)) + int(st.get("kmer_build_shadows", 0)), k)
    skip = ("P", "input", "total", "gated_per_pair", "model_all_records", "units_factor")
    print(f"<!-- {tag}: {cfg['workload']} -- commit {commit} -->")
    print("| kernel | ms/step (HIP events) | rocprofv3 avg ms/launch x launches/step | model bytes/step (units launched) | achieved GB/s | frac of 8 TB/s | fabric traffic/step (PMC) | frac on traffic | traffic / model | L2 hit | wave-cycles waiting | LDS conflict | VALU issue share |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, ms in sorted(b["kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
        v = pm.get(alias.get(name, name), {})
        t = step_bytes(name)
        if "hbm_bytes_per_step" in v:            # a composite scope
            rp = f"{v.get('ns_per_step', 0) / 1e6:.3f} ({v.get('launches_per_step', 0):.0f} launches of {len(v.get('composite_of', []))} kernels)"
        else:
            kl = per_step.get(alias.get(name, name), {}).get("launches_per_step", 1)
            rp = f"{v.get('avg_ns', 0) / 1e6:.3f} x {kl:g}"
        if name in sb and variant in ("shard", "config4") and name in ("k_group_pairs", "k_window_pairs", "k_window_cover"):
            algb = None          # (the sharded window scorer takes the two-call form -- pair lists, then the coverage test of the owned windows: the grouped model's counts do not describe it)
        elif name in sb:
            algb = sb[name] * chains
        elif name in svb:
            algb = svb[name] * chains
        elif name in ab and name not in skip:
            algb = ab[name] * pairs * chains
        else:
            algb = None
        cols = f"{v.get('l2_hit')} | {v.get('SQ_WAIT_ANY_frac')} | {v.get('lds_conflict_frac')} | {v.get('valu_issue_frac_at_2.4GHz')}"
        if algb and t:
            ach = algb / (ms * 1e-3) / 1e9
            print(f"| {name} | {ms:.3f} | {rp} | {algb / 1e6:.0f} MB | {ach:.0f} | {ach / 8000:.3f} | {t / 1e6:.0f} MB | {t / (ms * 1e-3) / 1e9 / 8000:.3f} | {t / algb:.2f} | {cols} |")
        else:
            tt = f"{t / 1e6:.0f} MB | {t / (ms * 1e-3) / 1e9 / 8000:.3f}" if t else "- | -"
            print(f"| {name} | {ms:.3f} | {rp} | - | - | - | {tt} | - | {cols} |")
    # kernels of the step that run outside every HIP-event scope (the begun read index on its own stream, rocPRIM's sorts, the runtime's
    # fills and copies): rocprofv3's own durations, whatever takes 0.1 ms per step or more
    covered = set()
    for name in b["kernels_ms_per_step"]:
        v = pm.get(alias.get(name, name), {})
        covered.update(v.get("composite_of") or [alias.get(name, name)])
    rest = []
    for kname, pk in per_step.items():
        if kname in covered:
            continue
        v = pm.get(kname, {})
        ns = v.get("avg_ns", 0.0) * pk.get("launches_per_step", 0)
        if ns >= 1e5:
            rest.append((ns, kname, pk, v))
    if rest:
        print("\nkernels of the step outside the HIP-event scopes (rocprofv3 durations; the read index begun on its own stream, rocPRIM, fills):\n")
        print("| kernel | rocprofv3 ms/step | launches/step | fabric traffic/step (PMC) | frac on traffic | L2 hit | wave-cycles waiting | LDS conflict | VALU issue share |")
        print("|---|---|---|---|---|---|---|---|---|")
        for ns, kname, pk, v in sorted(rest, reverse=True):
            t = pk["fabric_bytes_per_step"]
            print(f"| {kname[:60]} | {ns / 1e6:.3f} | {pk['launches_per_step']:g} | {t / 1e6:.0f} MB | {t / (ns * 1e-9) / 1e9 / 8000:.3f} | {v.get('l2_hit')} | {v.get('SQ_WAIT_ANY_frac')} | "
                  f"{v.get('lds_conflict_frac')} | {v.get('valu_issue_frac_at_2.4GHz')} |")
    if step:
        ms_step = b["ms_per_step"]
        print(f"\nstep as a whole: {step['bytes_per_step'] / 1e9:.2f} GB of counted fabric traffic per step ({step['kernels_counted']} kernels; bound {step['bytes_per_step_max'] / 1e9:.2f} GB) "
              f"in {ms_step:.3f} ms UNDER rocprofv3 (the bench line divides the same bytes by its own step time: roofline.step_traffic_frac) = {step['bytes_per_step'] / (ms_step * 1e-3) / 1e12:.2f} TB/s = {step['bytes_per_step'] / (ms_step * 1e-3) / 1e9 / 8000:.3f} of the 8 TB/s peak "
              f"(device busy {b.get('device_busy_frac')})")


if __name__ == "__main__":
    main()
