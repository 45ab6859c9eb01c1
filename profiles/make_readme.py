#!/usr/bin/env python3
"""Regenerate profiles/README.md from the committed JSON lines and the table printed by make_summary.py.

    python profiles/make_summary.py <tag> <stats.csv> <fetch.csv> <write.csv> <bench.json> > /tmp/prof_table.md
    python profiles/make_readme.py <tag> /tmp/prof_table.md
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tag, table = sys.argv[1], open(sys.argv[2]).read().strip()
    ld = lambda n: json.load(open(os.path.join(HERE, n)))
    d, d10, dk, fs = ld("r01_bench_final.json"), ld("r01_bench_10M_pairs.json"), ld("r01_bench_10M_k25_sensitive.json"), ld("r01_bench_force_shard_1gpu.json")
    cpu = d["cpu_baseline"]["value"]
    k10 = d10["kernels_ms_per_step"]
    ksum = sum(d["kernels_ms_per_step"].values())
    r = d["roofline"]
    s = f"""# profiles/ — round 1 evidence

All runs: one MI355X (gfx950), ROCm 7.2, `bench.py` on BASELINE.json configs[1] (1 M synthetic 50 bp pairs, 2,000 clones,
Zipf 1.1, 30 % noise, k=35 mf=3 mq=90 ins=175), ASCII pools resident in HBM, scorer inputs = the windows/contigs the host
traversal derives from this pool's graph (2,013 candidate windows, 132 final contigs).

| file | what |
|---|---|
| `r01_bench_final.json` | `python bench.py` (default flags: 10 steps, 3 warm-up): **{d['value']:.1f} M pairs/s**, {d['ms_per_step']:.2f} ms/step; CPU oracle {cpu:.3f} M pairs/s on 1 core (x{d['value'] / cpu:.0f}); parity gate inside the run: {str(d['parity_gate']).lower()}; kernel with the largest device time: `{r['kernel']}` ({r['avg_launch_ms']:.2f} ms, {r['frac'] * 100:.0f} % of HBM peak on its algorithmic bytes, PMC traffic {r['traffic'] / 1e6:.0f} MB per launch for {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB) — it is LDS/latency-bound (hash-table work per bucket), see the table for the bandwidth-bound kernels |
| `{tag}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu --parity-sample 0 --steps 5 --warmup 2` |
| `{tag}_bench.json` | the JSON line of that same (profiled) run |
| `{tag}_traffic.json`, `traffic_latest.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, same command with `--steps 2 --warmup 1`), reduced by `make_summary.py` |
| `r01_bench_10M_pairs.json` | configs[2] size: `bench.py --pairs 10000000 --clones 20000 --windows generator`: **{d10['value']:.1f} M pairs/s**, {d10['ms_per_step']:.1f} ms/step (one window per clone: 20,000 deep windows, 4.65 G read instances matched; k-mer build {d10['wall_ms_per_step']['kmer_build']:.0f} ms, window scoring {d10['wall_ms_per_step']['window_score']:.0f} ms, pair emission {d10['wall_ms_per_step']['map_emit']:.0f} ms of which 7 ms are the 208 MB of mapped pairs crossing PCIe) |
| `r01_bench_10M_k25_sensitive.json` | configs[3]: the same 10 M pairs with `--k 25 --mf 2 --mq 60 --mrs 20` (104 instances per pair, 1.04 G in all; 89.5 M distinct gated k-mers, 2.33 M nodes): {dk['value']:.1f} M pairs/s, {dk['ms_per_step']:.0f} ms/step |
| `r01_bench_force_shard_1gpu.json` | `bench.py --force-shard`: the k-mer build through the multi-GPU phases on ONE rank (real one-rank RCCL group): {fs['value']:.0f} M pairs/s, k-mer build {fs['wall_ms_per_step']['kmer_build']:.1f} ms vs {d['wall_ms_per_step']['kmer_build']:.1f} ms direct; {fs['exchange_bytes_per_step_rank0'] / 1e6:.0f} MB exchanged per rank and step; phases (ms): {fs['shard_wall_ms_per_step']} |
| `r01_bench_n2_one_device_dryrun.json` | plumbing check of the N>1 path on a 1-GPU box (`VDJX_BENCH_ONE_DEVICE=1`, two ranks sharing the GPU, gloo through host copies: NOT a performance number) |
| `r01d_*` | mid-round profile (100 M pairs/s): before the device-resident graph/roots, the weighted window scoring, the reduce/histogram/partition rework |
| `r01b_*`, `r01_bench_first.json` | earlier in the round: first bench line (16.8 M pairs/s) and the profile that exposed the 5.7x write amplification of the direct scatter (`k_kmer_scatter`: 5.86 GB written for 1.02 GB of tuples) which the LDS-staged partition removed |

## Per-kernel roofline, 1 M pairs (HBM-bound integer work; peak 8 TB/s)

`alg. bytes` = the bytes the kernel's job requires (DESIGN.md §4/§5; scorers from the run's actual counts, SURVEY §8d);
`HBM traffic` = (2·FETCH_SIZE + WRITE_SIZE)·1024 per launch (gfx950: FETCH_SIZE counts half of streamed read bytes; calibrated
on `k_pool_pack`).  rocprofv3's average for `k_pool_pack`/`k_window_hits` is per launch (two launches per step).

{table}

Reading: the tuple partition runs at 43 % of HBM peak and `k_part_records` at 30 %, both with traffic within 1.0-1.1x of their
algorithmic bytes; `k_graph_edges` follows successor links and moves exactly its algorithmic bytes.  `k_bucket_aggregate` is the
longest kernel: half of its time is the 1 GB tuple read (~3 TB/s while it loads), the other half LDS hash-table work (lookup of
every instance, recount) that the four resident workgroups per CU do not hide; `k_bucket_finalize` only walks the gated
instances of the candidates.  The window scorers evaluate identical read pairs once (weighted entries: 59.0 M matched read
instances -> 36.8 M evaluated entries), so their traffic is below the per-instance algorithmic figure.
Whole path: 3,324 B/pair x 1 M pairs / {d['ms_per_step']:.2f} ms = {3324e6 / (d['ms_per_step'] * 1e-3) / 8e12 * 100:.1f} % of HBM peak; the kernels sum to {ksum:.1f} ms, the rest is host-side (the mapped
pairs over PCIe into pinned buffers, ~45 launches, a dozen size round trips, Python); the graph's copy to the host runs beside the
root scorer.

Round-1 progression of the line of record (1 M pairs): 16.8 -> 27.8 -> 56 -> 74 -> 92 -> 103 -> 133 -> 148 -> 169 -> 183 -> 191 -> **{d['value']:.0f} M pairs/s**;
10 M pairs: 52 -> 66 -> 88 -> 106 -> 129 -> 140 -> 148 -> **{d10['value']:.0f} M pairs/s**.

10 M pairs, kernels per step (ms): part_records {k10['k_part_records']:.1f}, seg_hist {k10['k_seg_hist']:.1f}, part_tuples {k10['k_part_tuples']:.1f}, aggregate {k10['k_bucket_aggregate']:.1f},
finalize {k10['k_bucket_finalize']:.1f}, edges {k10['k_graph_edges']:.1f}, pack {k10['k_pool_pack']:.1f}; window_pairs {k10['k_window_pairs']:.1f}, cover {k10['k_window_cover']:.1f}, map_emit {k10['k_map_emit']:.1f} + gather {k10['k_gather_pairs']:.1f}.
"""
    open(os.path.join(HERE, "README.md"), "w").write(s)


if __name__ == "__main__":
    main()
