#!/usr/bin/env python3
"""bench.py -- M paired-reads/s through the hot path (k-mer build + contig scoring), IgH 50 bp PE.

One "step" = one pass of the hot path over one batch of synthetic reads already resident in HBM as the
reference's ASCII pool records:
    pool_pack -> k-mer table + prune + graph (a-1..a-3, a-5/a-6) -> root scorer over every root of that
    graph (a-7) -> window mapper + coverage test over the candidate windows the host traversal derives from
    this pool's graph (a-8, a-9) -> mapped-pair emission for the final contigs (a-10)
Default workload = BASELINE.json configs[2], the configuration the north-star target is quoted on: 10 M
synthetic pairs (20,000 clones, Zipf 1.1, 30 % noise, k=35 mf=3 mq=90, --ins 175) on one GPU.
`--pairs 1000000` is configs[1]; `--pairs 10000000 --k 25 --mf 2 --mq 60 --mrs 20` is configs[3].
With N GPUs every rank holds its own pool of that size and the k-mer partial aggregates are exchanged by
hash prefix through the C driver of `vdjer --gpus N` (vdjer_amd/libvdjmgpu.so, vdjer_amd/mgpu.py): weak scaling.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant HBM-streaming kernel of the step (the longest
kernel of the k-mer build; HIP events on the library's stream) against HBM peak, `roofline_by_kernel` every
kernel with a stated job size (the window scorer's inputs are cache resident: its figure is not an HBM one); `cpu_baseline` times the REFERENCE ITSELF (oracle/_ref/vdjer_ref, the
reference's own sources compiled -O0 as it ships, --t <host cores>) on a bounded sample of the same
generator, or -- where that binary is absent -- the C port (oracle/vdjx_oracle.c).  Neither is ever part
of the measured path.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vdjer_ref")


def algorithmic_bytes_per_pair(k: int, rl: int = 50, gated_per_pair: float | None = None, sym: bool = False, sym_walk: bool = False) -> dict:
    """SURVEY §8d.  P = 4*(rl-k+1) instances per pair; compulsory input of a pair = 2 reads x (packed bases + qualities);
    "every gated instance must reach its owner bucket once as a 16-B packed key, written once and read once"; the graph pass
    re-reads the packed input and probes the survivors with one 16-B key per instance.
    `total` is the SURVEY's per-pair figure for the whole k-mer build (it takes every instance as gated: 3,324 B at k=35) and
    prices the whole path (hot_path_frac).  Per kernel the job is priced with the MEASURED number of gated instances per pair
    (vdjx_stat "gated_instances": one instance in six of this generator's pools), because that is what the kernels move:
      pack: the four ASCII records of the pair in, their packed bases and masks out (32 B each; the quality characters stay in
      the resident records);  hist: the packed input once;  partition pass 1: packed input + one 16-B key per gated
      instance out;  pass 2: every such key in and out;  table + prune: every key in;
      walk (the graph pass): packed input once more + one 16-B survivor probe per instance (gated or not).
    **Units the launch processes (round 6).**  A pool made of couples (record, reverse complement: what add_to_buffer
    writes) is built over HALF the records: phase A (`sym`) moves one tuple per pair of mirrored gated instances and scans the couples'
    first records only; the walk (`sym_walk`) takes the first record of every couple and derives the second.  The per-kernel figures
    below are the §8d per-unit bytes x the units THAT launch processes -- so half the model's bytes for those kernels; the figure for
    all records is kept under `model_all_records` (what rounds 1-5 reported for the walk)."""
    P = 4 * (rl - k + 1)
    inp = 2 * ((rl + 3) // 4 + rl)
    g = P if gated_per_pair is None else gated_per_pair
    fa = 0.5 if sym else 1.0             # phase A: records scanned and tuples moved
    fw = 0.5 if sym_walk else 1.0        # the walk: records walked and instances probed
    allrec = {"k_gated_hist": inp, "k_part_records": inp + 16 * g, "k_part_tuples": 32 * g, "k_seg_hist": 16 * g, "k_gated_reduce": 16 * g,
              "k_gated_local": 16 * g, "k_walk_items": inp + 16 * P}
    out = {"P": P, "input": inp, "total": inp + 32 * P + inp + 16 * P, "gated_per_pair": g, "k_pool_pack": 4 * (2 * rl + 1) + 4 * 32,
           "model_all_records": allrec, "units_factor": {"phase_a": fa, "walk": fw}}
    for n_, v_ in allrec.items():
        out[n_] = v_ * (fw if n_ == "k_walk_items" else fa)
    return out


def survivor_bytes(ns: int, k: int, tmask_slots: int | None = None) -> dict:
    """The composite scopes of the k-mer build that work on the SURVIVORS (ns of them; 1.05 M at 10 M pairs), priced by their member
    kernels' array traffic per survivor (vdjx_kmer.hip stage_walk / stage_finish2) -- rounds 1-5 left them unpriced (1.05 ms of the step):
      k_surv_table = k_surv_table2 (key 16 in; table slot 16, key copy 16, filter bit out) + k_succ_links2 (four successor probes of 16,
                     counts 4 x 4; successor list 16, proposal 8, best successor 4 out)                                     = 156 B
      k_chain_order = k_chain_init (12 in, 8 out) + k_chain_jump x launches (8 + 8 in, 8 out: ~3 launches do work) + k_chain_len (8, 4)
                     + the scan (3 x 4) + k_chain_place (16) + k_chain_permute (key, counts, first, successor list: 60 in, 76 out)
                     + k_table_remap (the table: 2 slots of 16 per survivor in and out) + k_partner (one probe 16, 4 out)
                     + k_chain_words (successors 16, key 16 in; 12 out) + k_partner_check (4 + 8)                           = 420 B
      k_node_order = k_rank_keys (12 in, 12 out) + the radix sort (five 8-bit passes over 8-byte keys + 4-byte values, in and out: 120)
                     + k_rank_scatter (12, 4) + k_node_emit2 (key 16, counts 12, first 8, flags 2, rank 4, edge slots 4 x 16 in; the
                     node: 8 + 4 + 4 + 32 + 4 + k out)                                                                      = 326 + k B
      k_node_flags = key 16 in, two bitmap probes of 4, 2 out                                                               = 26 B"""
    return {"k_surv_table": 156 * ns, "k_chain_order": 420 * ns, "k_node_order": (326 + k) * ns, "k_node_flags": 26 * ns}


def scorer_bytes(stats: dict, n_windows: int, n_contigs: int, k: int, rl: int = 50, wlen: int = 486, clen: int = 360) -> dict:
    """The scorers' jobs (not proportional to pairs), priced on what each kernel must read and write in THIS design:
    classification: (len-rl) offsets per string x (one index slot in -- 32 B, or 64 B for an index over couples -- + one 16-B image entry out);
    window mapper: the image (16 B per offset) + one 8-B index entry per DISTINCT read-1 entry of the classes met (identical read
    pairs are one weighted entry; SURVEY 8d's "8 B per matched read instance" prices the reference's per-instance walk, 2.5 times
    as many) + 8 B per list entry out;  coverage: the list in twice;  pair emission: image + 8 B per hit + 24 B per mapped pair out,
    the gather 24 B in and 20 B out per pair;  root DP: k + 2k bytes per (root, seed hit)."""
    nw, nc = n_windows * (wlen - rl), n_contigs * (clen - rl)
    return {"k_part_items": 16 * stats.get("recount_items", 0),          # every run item (8 bytes) in and out
            "k_recount": 8 * stats.get("recount_items", 0),
            # (an index over couples keeps one 64-byte slot per pair {sequence, reverse complement}: k_ri_tab_canon; else 32-byte slots)
            "k_map_classify": (nw + nc) * ((64 if stats.get("read_index_sym") else 32) + 16),
            # the grouped mapper reads every image three times (numbering, counting, listing the occurrences), one 8-B entry per
            # DISTINCT entry of a GROUP's classes (a class shared by the group's windows is streamed once), and writes the lists;
            # k_window_pairs only looks at the flags of the windows the groups have done (none are left on this workload)
            "k_group_pairs": 3 * nw * 16 + 8 * stats.get("group_hits_distinct", 0) + 8 * stats.get("window_pairs_entries", 0),
            "k_window_pairs": (nw * 16 + 8 * stats.get("window_hits_distinct", 0) + 8 * stats.get("window_pairs_entries", 0)) if not stats.get("group_hits_distinct") else n_windows * 8,
            "k_window_cover": 8 * 2 * stats.get("window_pairs_entries", stats.get("window_pairs", 0)),
            "k_map_emit": nc * 16 + 8 * stats.get("map_hits", 0) + 24 * stats.get("mapped_pairs", 0),
            "k_gather_pairs": 44 * stats.get("mapped_pairs", 0),
            "k_root_dp": 3 * k * stats.get("root_dp_items", 0)}


def workload_label(pairs: int, k: int, mf: int, mq: int, mrs: int, ins: int, world: int, chains=("IGH",)) -> str:
    if len(chains) > 1:
        return (f"BASELINE.json configs[4]{'' if (pairs, world) == (12_500_000, 8) else ' geometry'}: {pairs} synthetic 50bp PE pairs per GPU x {world} GPU(s) = "
                f"{pairs * world} pairs per chain, hash-prefix sharded (pool dealt by pair, one bulk all-to-all of partial aggregates), "
                f"{' -> '.join(chains)} back to back in one process (set_chain_info params.c:13-30: own repertoire, ref-dir, anchors and pool per chain; "
                f"anchor sets, V region, pool packing and read index inside every step), k={k} mf={mf} mq={mq} mrs={mrs} ins={ins}; SURVEY §8d C5")
    if world > 1:
        tag = f"weak scaling of BASELINE.json configs[2]-sized pools over {world} GPUs (configs[4] is {world}x12.5 M)" if pairs >= 10_000_000 else "custom"
    elif pairs == 10_000_000 and (k, mf, mq) == (35, 3, 90):
        tag = "BASELINE.json configs[2]; SURVEY §8d C3"
    elif pairs == 10_000_000 and (k, mf, mq) == (25, 2, 60):
        tag = "BASELINE.json configs[3]; SURVEY §8d C4" + (" with --mrs 20" if mrs == 20 else "")
    elif pairs == 1_000_000 and (k, mf, mq) == (35, 3, 90):
        tag = "BASELINE.json configs[1]; SURVEY §8d C2"
    else:
        tag = "custom size"
    if REP_KW:
        tag += ("; --repertoire private: every clone over a germline V and J tail of its own, Zipf 0.25, one clone per 4,000 pairs -- BASELINE size on a repertoire the "
                "reference's serial traversal finishes (tests/golden/midscale.json cfg2_pv)")
    return f"synthetic {pairs} 50bp PE pairs per GPU, IGH, k={k} mf={mf} mq={mq} mrs={mrs} ins={ins} ({tag})"


def cli_at_size(args, rep, pool) -> dict | None:
    """--cli-at-size: this build's whole command line (vdjer_amd/vdjer) on the FULL-SIZE pool of --repertoire private -- the reads file written from
    the very pool the bench times -- against the committed digests of the reference's complete --t 1 run on it (tests/golden/midscale.json,
    made once in the build container by tests/golden/make_golden_midscale.py: ~25 minutes of the reference)."""
    from tests import midscale_util as M
    from vdjer_amd import synth
    flags = ["--k", str(args.k), "--mf", str(args.mf), "--mq", str(args.mq), "--mrs", str(args.mrs)]
    gold = next((c for c in M.cases().values() if c.get("private_v") and (c["pairs"], c["clones"], c["seed"], c["ins"]) == (args.pairs, args.clones, args.seed, args.ins)
                 and c["flags"] == flags), None)
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    if gold is None or not os.path.exists(exe):
        return {"skipped": "no committed reference digest for this workload (tests/golden/midscale.json) or vdjer not built"}
    with tempfile.TemporaryDirectory() as td:
        t_w = time.perf_counter()
        (pool.to_host() if hasattr(pool, "to_host") else pool).write_reads_file(os.path.join(td, "reads.txt"))
        synth.write_ref_dir(rep, os.path.join(td, "ref"))
        t_w = time.perf_counter() - t_w
        cmd = [exe, "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", str(args.ins), "--t", str(min(os.cpu_count() or 1, 100))] + flags
        t0 = time.perf_counter()
        ms = {}
        with open(os.path.join(td, "out.sam"), "wb") as so:
            pr = subprocess.Popen(cmd, cwd=td, stdout=so, stderr=subprocess.PIPE, text=True, errors="replace", env=dict(os.environ, VDJH_ROOT_LOG="roots.log", VDJX_TIMES="1"))
            tail = []
            for line in pr.stderr:
                if line.startswith("VDJX_TIMES\t"):
                    f_ = line.rstrip("\n").split("\t")
                    ms.setdefault(f_[1], float(f_[2]))
                tail.append(line)
                tail = tail[-12:]
            pr.wait()
        wall = time.perf_counter() - t0
        mine = {key: (M.digest_file(os.path.join(td, fn)) if os.path.exists(os.path.join(td, fn)) else None)
                for key, fn in (("fasta", "vdj_contigs.fa"), ("sam", "out.sam"), ("dot", "vdjer.dot"), ("root_log", "roots.log"))}

    def seg(a, b):
        return round(ms[b] - ms[a], 1) if a in ms and b in ms else None
    return {"pairs": args.pairs, "clones": args.clones, "exit_code": pr.returncode, "wall_s_process": round(wall, 2), "reads_file_written_s": round(t_w, 1),
            "contigs": (mine["fasta"] or {}).get("lines", 0) // 2, "sam_lines": (mine["sam"] or {}).get("lines"),
            "outputs_identical_to_reference": all(mine[f_] == gold[f_] for f_ in ("fasta", "sam", "dot")), "root_verdicts_identical": mine["root_log"] == gold["root_log"],
            "differs": [f_ for f_ in ("fasta", "sam", "dot", "root_log") if mine[f_] != gold[f_]],
            "reference": {"contigs": gold["contigs"], "roots": gold["roots"], "roots_accepted": gold["roots_accepted"], "wall_s": gold.get("reference_wall_s"),
                          "complete_runs": gold.get("complete_runs"), "source": "tests/golden/midscale.json (complete --t 1 runs of oracle/_ref/vdjer_ref, all identical)"},
            "process_breakdown_ms": {"input_read_and_parsed": seg("START", "(inputs read)"), "pool_load": seg("POST_VJF_INIT", "(pool loaded)"),
                                     "read_index_plus_kmer_build": seg("(pool loaded)", "(read index ended)"),
                                     "traversal_and_scorers": seg("POST_BUILD_GRAPH2", "THREADS_DONE"), "overlap_removal_fasta_sam": seg("THREADS_DONE", "PRE_CLEANUP")},
            "stderr_tail": None if pr.returncode == 0 else "".join(tail)[-1500:]}


REP_KW = {}          # generator parameters of --repertoire private (make_repertoire keywords), set by main()


def make_workload(n_pairs: int, n_clones: int, seed: int, rank: int, world: int, device: str, chain: str = "IGH", ci: int = 0):
    """Rank r holds its own library: n_pairs read pairs from n_clones clones over its own germline (rank 0: exactly the
    one-GPU workload).  N GPUs = N independent libraries processed as ONE job (one k-mer table, one graph, one traversal):
    N x pairs, N x clones, and a ref-dir that is the union of the N germlines, so the per-GPU work stays what it is on one
    GPU -- the definition of weak scaling (SURVEY §8d scales clones with pairs the same way: 1 M / 2,000 ... 100 M / 100,000).
    The pool is generated in HBM by the counter-based generator (bit-identical to its CPU evaluation)."""
    from vdjer_amd import synth
    libs = [synth.make_repertoire(n_clones, seed=seed + 15485863 * r + 7 * ci, chain=chain, **REP_KW) for r in range(world)]
    rep = libs[rank]
    # (--repertoire private on one GPU: the very pool of tests/golden/midscale.json's cfg2_pv -- seed + 13 -- so that its reference digests apply)
    pool = synth.make_reads_cb(rep, n_pairs, noise_frac=0.3, seed=seed + (13 if REP_KW and world == 1 and ci == 0 else 104729 * rank + 1000 * ci), device=device)
    vc = np.array(sorted({synth.seq_to_int(a) for lb in libs for a in lb.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for lb in libs for a in lb.j_anchors}), dtype=np.uint32)
    return rep, pool, vc, jc, [lb.v_region for lb in libs]


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N=1 only; bounded samples of the same generator)
# ---------------------------------------------------------------------------------------------------------------
def cpu_reference(n_sample: int, seed: int, k: int, mf: int, mq: int, mrs: int, ins: int, threads: int) -> dict | None:
    """The reference itself: its own sources compiled where they lie (oracle/Makefile, -O0 exactly like the reference's
    Makefile:8), whole `assemble` (A2:1350-1507) with --t <threads>, on a pool of n_sample pairs of the same generator
    (clones scaled 1 per 500 pairs like configs[1..4]).  Timed from the PRE_PRE_GRAPH1 marker (the pools are extracted,
    A2:1388) to FINIS (A2:1473): k-mer table, prune, graph, traversal with the root / window scorers inside, SAM mapping."""
    if not os.path.exists(REF_BIN):
        return None
    from vdjer_amd import synth
    rep = synth.make_repertoire(max(4, n_sample // 500), seed=seed)
    pool = synth.make_reads_cb(rep, n_sample, noise_frac=0.3, seed=seed + 13)
    with tempfile.TemporaryDirectory() as td:
        pool.write_reads_file(os.path.join(td, "reads.txt"))
        synth.write_ref_dir(rep, os.path.join(td, "ref"))
        cmd = [REF_BIN, "run", "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", str(ins), "--t", str(threads),
               "--k", str(k), "--mf", str(mf), "--mq", str(mq), "--mrs", str(mrs)]
        def timed_run(argv, cwd, sam_name, env=None):
            mk = {}
            t_start = time.perf_counter()
            with open(os.path.join(cwd, sam_name), "wb") as so:
                pr = subprocess.Popen(argv, cwd=cwd, stdout=so, stderr=subprocess.PIPE, text=True, errors="replace", env=env)
                for line in pr.stderr:
                    if line.startswith("ELAPSED_SECS\t"):
                        mk.setdefault(line.split("\t")[1], time.perf_counter())
                    elif line.startswith("VDJX_TIMES\t"):          # (this build's own milliseconds beside the reference's whole seconds)
                        f_ = line.rstrip("\n").split("\t")
                        mk.setdefault("_ms", {}).setdefault(f_[1], float(f_[2]))
                    elif line.startswith("num root nodes:"):
                        mk["_roots"] = int(line.split(":")[1])
                    elif line.startswith("HARNESS_ROOTS_SCORED"):
                        mk["_scored"] = int(line.split("\t")[1])
                pr.wait()             # (the reference's exit status is meaningless: its main falls off its end)
            return mk, t_start, time.perf_counter(), pr.returncode

        def digests(d, sam_name):
            from tests import midscale_util as M
            out = {}
            for key, fn in (("fasta", "vdj_contigs.fa"), ("sam", sam_name), ("dot", "vdjer.dot"), ("root_log", "roots.log")):
                out[key] = M.digest_file(os.path.join(d, fn)) if os.path.exists(os.path.join(d, fn)) else None
            return out

        def root_log(path):
            try:
                return dict(l.rstrip("\n").split("\t") for l in open(path))
            except OSError:
                return {}
        # the timed run: --t <threads>, the harness logging what score_seq said of every root it consumed (one fwrite per root)
        marks, _, _, _ = timed_run(cmd, td, "sam.out", dict(os.environ, VDJX_REF_ROOT_LOG="roots.log"))
        ref_complete = marks.get("_roots") is not None and marks.get("_roots") == marks.get("_scored")
        n_contigs = sum(1 for l in open(os.path.join(td, "vdj_contigs.fa")) if l.startswith(">")) if os.path.exists(os.path.join(td, "vdj_contigs.fa")) else -1
        # ---- the whole command line of THIS build on the same file, same flags (one GPU): the first number that covers the read
        # index, the serial host traversal and the text output as well
        exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
        cli = None
        if os.path.exists(exe):
            d2 = os.path.join(td, "hip")
            os.makedirs(d2)
            os.symlink(os.path.join(td, "reads.txt"), os.path.join(d2, "reads.txt"))
            os.symlink(os.path.join(td, "ref"), os.path.join(d2, "ref"))
            mk2, t0c, t1c, rc2 = timed_run([exe] + cmd[2:], d2, "sam.out", dict(os.environ, VDJH_ROOT_LOG="roots.log", VDJX_TIMES="1"))
            mine = digests(d2, "sam.out")
            # ---- (1) byte for byte against the reference at --t 1 (its race-free order): the committed digests of complete
            # --t 1 runs when this sample IS a committed mid-scale golden (tests/golden/midscale.json), a live --t 1 run otherwise
            t1 = {"source": None, "identical": None}
            try:
                from tests import midscale_util as M
                gold = next((c for c in M.cases().values() if (c["pairs"], c["clones"], c["seed"], c["noise"], c["chain"], c["ins"]) ==
                             (n_sample, len(rep.clones), seed, 0.3, "IGH", ins) and c["flags"] == cmd[cmd.index("--k"):]), None)
            except Exception:  # noqa: BLE001
                gold = None
            if gold is not None:
                t1 = {"source": "tests/golden/midscale.json: digests of complete --t 1 runs of the compiled reference on this very sample "
                                f"({gold['complete_runs']} runs, all identical; tests/golden/make_golden_midscale.py)",
                      "identical": all(mine[f_] == gold[f_] for f_ in ("fasta", "sam", "dot")), "root_verdicts_identical": mine["root_log"] == gold["root_log"],
                      "differs": [f_ for f_ in ("fasta", "sam", "dot", "root_log") if mine[f_] != gold[f_]]}
            elif os.environ.get("VDJX_BENCH_REF_T1", "1") != "0":
                for attempt in range(2):          # (the reference loses its last roots to a wake-up race even at --t 1: a run counts only when it scored them all)
                    d1 = os.path.join(td, f"t1_{attempt}")
                    os.makedirs(d1)
                    os.symlink(os.path.join(td, "reads.txt"), os.path.join(d1, "reads.txt"))
                    os.symlink(os.path.join(td, "ref"), os.path.join(d1, "ref"))
                    c1 = list(cmd)
                    c1[c1.index("--t") + 1] = "1"
                    mk1, _, _, _ = timed_run(c1, d1, "sam.out", dict(os.environ, VDJX_REF_ROOT_LOG="roots.log"))
                    if mk1.get("_roots") is not None and mk1.get("_roots") == mk1.get("_scored"):
                        theirs = digests(d1, "sam.out")
                        t1 = {"source": f"a complete --t 1 run of oracle/_ref/vdjer_ref beside this one (attempt {attempt + 1})",
                              "identical": all(mine[f_] == theirs[f_] for f_ in ("fasta", "sam", "dot")), "root_verdicts_identical": mine["root_log"] == theirs["root_log"],
                              "differs": [f_ for f_ in ("fasta", "sam", "dot", "root_log") if mine[f_] != theirs[f_]]}
                        break
                    t1["source"] = "two --t 1 runs of the reference both lost roots to its wake-up race (SURVEY §0-3): nothing to compare with"
            # ---- (2) the timed --t N run of the reference against that: at --t > 1 its workers share ONE scoring matrix (seq_score.c:14,
            # written by every worker, seq_score.c:93-98), so root verdicts flip with thread timing, and its windows enter the
            # order-defining tables in thread-timing order: its contig set, order and numbering are not a function of the input
            same_set = None
            try:
                fa_a, fa_b = open(os.path.join(d2, "vdj_contigs.fa"), "rb").read(), open(os.path.join(td, "vdj_contigs.fa"), "rb").read()
                same_set = sorted(fa_a.split(b"\n")[1::2]) == sorted(fa_b.split(b"\n")[1::2])       # the contigs, whatever their order and numbering
                only_ref = sorted(set(fa_b.split(b"\n")[1::2]) - set(fa_a.split(b"\n")[1::2]))
                only_here = sorted(set(fa_a.split(b"\n")[1::2]) - set(fa_b.split(b"\n")[1::2]))
            except OSError:
                only_ref = only_here = []
            la, lb = root_log(os.path.join(d2, "roots.log")), root_log(os.path.join(td, "roots.log"))
            flips = sorted(k_ for k_ in lb if k_ in la and la[k_] != lb[k_])
            tn = {"threads": threads, "scored_all_roots": ref_complete, "roots": marks.get("_roots"), "roots_scored": marks.get("_scored"),
                  "same_contig_set": same_set, "contigs_only_in_reference_run": len(only_ref), "contigs_only_here": len(only_here),
                  "root_verdicts_differing_from_t1": len(flips), "roots_accepted_here_and_at_t1": sum(1 for v_ in la.values() if v_ == "1"),
                  "roots_accepted_by_this_reference_run": sum(1 for v_ in lb.values() if v_ == "1"),
                  "flipped_roots_sample": [f"{k_} t1={la[k_]} t{threads}={lb[k_]}" for k_ in flips[:6]],
                  "why": "seq_score_matrix is one global scratch written by every worker (seq_score.c:14,93-98): at --t > 1 a root's DP is "
                         "overwritten mid-way by its neighbours' and score_seq accepts roots the serial run rejects (or the reverse); every such "
                         "root is listed by k-mer.  The extra accepted roots can add contigs, and the accepted windows' insertion order -- thread "
                         "timing -- decides which of two overlapping windows A2:875-898 erases.  This build emits the --t 1 bytes whatever --t."}

            def sp(m, a, b):
                return round(m[b] - m[a], 3) if a in m and b in m else None
            stages = (("kmer_table", "PRE_PRE_GRAPH1", "POST_PRE_GRAPH1"), ("prune", "POST_PRE_GRAPH1", "POST_PRUNE_PRE_GRAPH1"),
                      ("graph", "POST_PRUNE_PRE_GRAPH1", "POST_BUILD_GRAPH2"), ("traversal_and_scorers", "POST_BUILD_GRAPH2", "THREADS_DONE"),
                      ("output_and_sam", "THREADS_DONE", "PRE_CLEANUP"), ("assemble_total", "PRE_PRE_GRAPH1", "FINIS"))
            ms = mk2.get("_ms", {})

            def seg(a, b):
                return round(ms[b] - ms[a], 1) if a in ms and b in ms else None
            breakdown = {"process_start_to_first_marker": round((mk2["START"] - t0c) * 1e3, 1) if "START" in mk2 else None,
                         "input_read_and_parsed (the HIP runtime starting beside it on a thread)": seg("START", "(inputs read)"),
                         "wait_for_hip_init": seg("(inputs read)", "(vdjx_init done)"),
                         "anchor_sets_and_v_region": seg("(vdjx_init done)", "POST_VJF_INIT"),
                         "pool_load": seg("POST_VJF_INIT", "(pool loaded)"), "read_index_begin": seg("(pool loaded)", "(read index begun)"),
                         "read_names_to_device": seg("(read index begun)", "POST_READ_EXTRACT"),
                         "kmer_build (the read index beside it)": seg("PRE_PRE_GRAPH1", "(k-mer build done)"),
                         "read_index_waited_for_after_the_build": seg("(k-mer build done)", "(read index ended)"),
                         "read_index_plus_kmer_build": seg("(pool loaded)", "(read index ended)"),
                         "traversal_and_scorers": seg("POST_BUILD_GRAPH2", "THREADS_DONE"),
                         "overlap_removal_fasta_sam": seg("THREADS_DONE", "PRE_CLEANUP"), "teardown": seg("PRE_CLEANUP", "FINIS"),
                         "finis_to_process_end": round((t1c - mk2["FINIS"]) * 1e3, 1) if "FINIS" in mk2 else None}
            cli = {"pairs": n_sample, "exit_code": rc2, "wall_s_process": round(t1c - t0c, 3), "cli_process_breakdown_ms": breakdown,
                   "stages_s": {n_: sp(mk2, a, b) for n_, a, b in stages}, "reference_stages_s": {n_: sp(marks, a, b) for n_, a, b in stages},
                   "outputs_identical_to_reference": t1["identical"], "reference_t1": t1, "reference_tN": tn,
                   "note": "vdjer_amd/vdjer (C host over libvdjx, one GPU) and oracle/_ref/vdjer_ref on the same extracted-reads file and ref-dir, "
                           "timed by their ELAPSED_SECS stage markers as they arrive; process wall time includes reading the file, loading the "
                           "pool, building the read index and the GPU start-up.  outputs_identical_to_reference: vdj_contigs.fa, SAM and "
                           "vdjer.dot byte for byte against the reference at --t 1"}
    if "PRE_PRE_GRAPH1" not in marks or "FINIS" not in marks:
        return None

    def span(a, b):
        return round(marks[b] - marks[a], 2) if a in marks and b in marks else None
    secs = marks["FINIS"] - marks["PRE_PRE_GRAPH1"]
    return {"value": n_sample / secs / 1e6, "unit": "M paired-reads/s", "cores": threads, "kind": "reference", "cli_end_to_end": cli,
            "sample": f"{n_sample} pairs / {len(rep.clones)} clones of the same generator through oracle/_ref/vdjer_ref "
                      f"(the reference's own sources, g++ -O0 as it ships; -O1+ crashes on its missing returns, SURVEY §0-2), --t {threads}: "
                      f"k-mer table {span('PRE_PRE_GRAPH1', 'POST_PRE_GRAPH1')}s (1 thread by construction, A2:1388-1390), prune "
                      f"{span('POST_PRE_GRAPH1', 'POST_PRUNE_PRE_GRAPH1')}s, graph {span('POST_PRUNE_PRE_GRAPH1', 'POST_BUILD_GRAPH2')}s, "
                      f"traversal + root/window scorers on {threads} threads {span('POST_BUILD_GRAPH2', 'THREADS_DONE')}s, SAM mapping "
                      f"{span('THREADS_DONE', 'PRE_CLEANUP')}s; {n_contigs} contigs; includes the host traversal, which `value` does not "
                      f"(see host_side)",
            "seconds": round(secs, 2)}


def cpu_port(n_sample: int, seed: int, k: int, mf: int, mq: int, mrs: int, ins: int, opt0: bool) -> dict:
    """The C port of the reference's algorithm (oracle/vdjx_oracle.c; test infrastructure, used here only as a timed baseline)
    on the same sample: k-mer table + prune + graph, every root scored, one window per clone mapped + validated, the valid
    ones mapped again for the SAM records.  1 thread."""
    from oracle import oracle
    from vdjer_amd import synth
    L = oracle.lib(opt0=True) if opt0 else oracle.lib()
    rep = synth.make_repertoire(max(4, n_sample // 500), seed=seed)
    pool = synth.make_reads_cb(rep, n_sample, noise_frac=0.3, seed=seed + 13)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    wins = [w for w in rep.windows() if w]
    t0 = time.perf_counter()
    tb = oracle.KmerTable(pool, k, L=L)
    tb.prune(mf, mq)
    g = oracle.Graph(tb, vc, jc)
    t1 = time.perf_counter()
    sc = oracle.RootScorer([rep.v_region], 15, L=L)
    roots = [oracle.inst_kmer(pool, int(g.first[i]), k) for i in np.flatnonzero(g.from_deg == 0)]
    t1b = time.perf_counter()
    n_ok = sum(sc.score(r, mrs) for r in roots)
    t2 = time.perf_counter()
    ix = oracle.ReadIndex(pool, L=L)
    t3 = time.perf_counter()           # the index belongs to extraction in the reference (bam_read.c:228,243): not timed
    nvalid = 0
    contigs = []
    for w in wins:
        pairs, starts = ix.quick_map(w)
        if ix.coverage_is_valid(starts, len(w), ins):
            nvalid += 1
            contigs.append(w[51:411])
    for c in contigs:
        ix.quick_map(c)
    t4 = time.perf_counter()
    secs = (t1 - t0) + (t2 - t1b) + (t4 - t3)
    return {"value": n_sample / secs / 1e6, "unit": "M paired-reads/s", "cores": 1, "kind": "port",
            "sample": f"{n_sample} pairs / {len(rep.clones)} clones: kmer+prune+graph {t1 - t0:.2f}s, {len(roots)} roots ({n_ok} accepted) "
                      f"{t2 - t1b:.2f}s, {len(wins)} windows ({nvalid} valid) + SAM mapping {t4 - t3:.2f}s; oracle/vdjx_oracle.c "
                      f"{'-O0' if opt0 else '-O2'}, 1 thread, no traversal",
            "seconds": round(secs, 2)}


def _traffic_file(args, world: int):
    """the newest committed PMC pass of THIS workload: profiles/rNN_traffic[_<variant>].json (variant: k25, shard, config4; none = the
    default configs[2] step).  The file names the commit and the flags it was taken with."""
    import glob
    variant = "config4" if getattr(args, "config4", False) else ("shard" if args.force_shard else ("k25" if args.k == 25 else ""))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic{'_' + variant if variant else ''}.json")))
    if not files or world != 1:
        return None, None
    tj = json.load(open(files[-1]))
    same = (tj.get("pairs_per_gpu") == args.pairs and tj.get("k") == args.k and tj.get("windows", "traversal") == args.windows
            and tj.get("variant", "") == variant)
    return (tj, files[-1]) if same else (None, None)


def load_traffic(args, world: int, kernel: str):
    """PMC-measured fabric bytes per launch of `kernel` (a composite scope: per step, summed over its member kernels) from the newest
    committed rocprofv3 pass of the same command (separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections, profiles/README.md)"""
    tj, _ = _traffic_file(args, world)
    if not tj:
        return None
    for name, v in tj.get("kernels", {}).items():
        if name.split("<")[0] == kernel:
            return v["hbm_bytes"]
    return None


def step_traffic(args, world: int):
    tj, _ = _traffic_file(args, world)
    if not tj or not tj.get("step"):
        return None
    return tj["step"]


def traffic_source(args=None, world: int = 1) -> dict | None:
    if args is None:
        return None
    tj, f = _traffic_file(args, world)
    if not tj:
        return None
    return {"file": os.path.relpath(f, ROOT), "commit": tj.get("commit"), "note": "PMC passes are separate runs of the same command (profiles/prof_step.sh)"}


# the HBM-streaming kernels of the k-mer build: the roofline kernel is the longest of them
BUILD_KERNELS = ("k_pool_pack", "k_gated_hist", "k_part_records", "k_part_tuples", "k_seg_hist", "k_gated_reduce", "k_gated_local",
                 "k_walk_items", "k_part_items", "k_recount")
# every scope with a stated job size (algorithmic_bytes_per_pair, scorer_bytes, survivor_bytes): the roofline kernel is the longest of THESE
PRICED_KERNELS = BUILD_KERNELS + ("k_surv_table", "k_chain_order", "k_node_order", "k_node_flags", "k_map_classify", "k_group_pairs", "k_window_pairs",
                                  "k_window_cover", "k_map_emit", "k_gather_pairs", "k_root_dp")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (default 10 M; 12.5 M with --config4)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE.json configs[4]: 12.5 M pairs per GPU, IGH -> IGK -> IGL back to back in one process (a step = the three chains: anchor sets, "
                         "V region, pool packing, read index, sharded k-mer build over the pool dealt by pair, scorers -- per chain).  The default with --gpus 8")
    ap.add_argument("--no-config4", action="store_true", help="--gpus 8 as weak scaling of configs[2]-sized IGH pools, like --gpus 2 / 4")
    ap.add_argument("--chains", default=None, help="comma-separated chain presets of a --config4 step (default IGH,IGK,IGL)")
    ap.add_argument("--clones", type=int, default=0, help="clones per GPU (default: pairs / 500; --repertoire private: pairs / 4000)")
    ap.add_argument("--repertoire", choices=["survey", "private"], default="survey",
                    help="survey: SURVEY 8d's repertoire (60 germline V segments shared by all clones, Zipf 1.1) -- the reference's contig enumeration does not "
                         "end on it above ~1.5 M pairs, so the windows come from the generator there.  private: every clone over a germline V and a J + constant tail of its "
                         "own, abundance Zipf 0.25, one clone per 4,000 pairs (tests/golden/make_golden_midscale.py cfg2_pv): the serial traversal TERMINATES at "
                         "10 M pairs (reference: ~25 min at --t 1, 2 k contigs), so the scorers get the windows the reference's DFS really asks for and the "
                         "whole command line is compared with the reference's bytes at BASELINE size (--cli-at-size)")
    ap.add_argument("--cli-at-size", action="store_true",
                    help="--repertoire private only: run this build's `vdjer` on the full-size pool (written as a reads file) and compare vdj_contigs.fa / SAM / "
                         "vdjer.dot / root verdicts with the committed digests of the reference's complete --t 1 run (tests/golden/midscale.json cfg2_pv)")
    ap.add_argument("--k", type=int, default=35)
    ap.add_argument("--mf", type=int, default=3)
    ap.add_argument("--mq", type=int, default=90)
    ap.add_argument("--mrs", type=int, default=30)
    ap.add_argument("--ins", type=int, default=175)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="pairs the reference is timed on")
    ap.add_argument("--cpu-port-sample", type=int, default=200_000, help="pairs the C port is timed on (-O2 and -O0 legs)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-fed (PCIe-inclusive) variant of the step")
    ap.add_argument("--force-shard", action="store_true",
                    help="N=1 only: run the k-mer build and the window scorer through the multi-GPU phases (kernels and host logic of the sharded path; a lone rank "
                         "skips the collectives that would hand it its own data back -- VDJX_SHARD_SELF_COLLECTIVES=1 keeps them, over a one-rank RCCL group); "
                         "not the line of record")
    ap.add_argument("--windows", choices=["auto", "traversal", "generator"], default="auto",
                    help="scorer inputs: the candidate windows/contigs the host traversal derives from the graph, or one window per "
                         "clone straight from the generator.  auto = traversal up to 1.5 M pairs in all, generator above: the "
                         "reference's contig enumeration (A2:939-1061, no path cap below 5e8 per root) explodes on this repertoire "
                         "(60 V germlines shared by thousands of clones): 21 k contig candidates at 1 M pairs, 699 k at 2 M, and at "
                         "10 M pairs it does not finish in an hour on 8 cores (DESIGN.md §5)")
    ap.add_argument("--parity-sample", type=int, default=20000, help="pairs of the parity gate (SURVEY §8d)")
    ap.add_argument("--no-two-in-flight", action="store_true",
                    help="N=1: skip the leg that runs, after the timed region, the same steps again with TWO batches in flight -- a second context (own stream, workspaces, anchor sets) and a "
                         "second host thread, the steps dealt alternately: what the host waits and kernel tails of one step leave idle is filled by the other "
                         "(reported as value_two_in_flight; never `value`)")
    ap.add_argument("--no-index-leg", action="store_true", help="skip value_with_read_index (profiles/prof_step.sh: the process then ends with the timed steps)")
    args = ap.parse_args()
    if args.gpus == 8 and not args.no_config4:
        args.config4 = True
    chains = tuple((args.chains or "IGH,IGK,IGL").split(",")) if args.config4 else ("IGH",)
    if args.pairs is None:
        args.pairs = 12_500_000 if args.config4 else 10_000_000
    if args.config4:
        args.force_shard = True          # configs[4] runs only through the sharded path (one rank: the same phases, nothing to exchange)
        args.no_e2e = True
        args.windows = "generator"
    if args.repertoire == "private":
        REP_KW.update(private_v=True, private_j=True, zipf_s=0.25)
        if args.clones <= 0:
            args.clones = max(4, args.pairs // 4000)
        if args.windows == "auto":
            args.windows = "traversal"
    if args.clones <= 0:
        # clones per GPU: 1 per 500 pairs like configs[1..3] (1 M / 2,000, 10 M / 20,000); configs[4] is 100 M pairs of 100,000 clones
        # (SURVEY §8d C5): 1 per 1,000
        args.clones = max(4, args.pairs // (1000 if args.config4 else 500))
    if args.windows == "auto":
        args.windows = "traversal" if args.pairs * args.gpus <= 1_500_000 else "generator"

    # ONE JSON line on stdout: libraries that print banners to the process's stdout (RCCL does at its first collective) are sent to
    # stderr; the line itself goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    # the step hands tens of MB of results to the host (graph arrays, mapped pairs); with glibc's default mmap threshold
    # every such buffer is a fresh mapping (page faults + munmap per step).  Keep them on the heap instead.
    try:
        import ctypes
        _libc = ctypes.CDLL(None)
        _libc.mallopt(-3, 1 << 30)      # M_MMAP_THRESHOLD
        _libc.mallopt(-1, 1 << 30)      # M_TRIM_THRESHOLD
    except Exception:  # noqa: BLE001
        pass

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # dry run of the N>1 path on a box with one GPU: VDJX_BENCH_ONE_DEVICE=1 puts every rank on device 0 and moves
    # the bytes with gloo (RCCL refuses two ranks on one device).  Never used for reported numbers.
    one_device = os.environ.get("VDJX_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from vdjer_amd import api

    t_gen = time.perf_counter()
    # W: the chains of one step.  One entry (IGH) but for --config4, where a step is IGH -> IGK -> IGL back to back: a repertoire,
    # ref-dir (anchor sets, V region), pool and candidate windows per chain, all resident before the timed region
    W = []
    for ci_, chain_ in enumerate(chains):
        rep_, pool_, vc_, jc_, vl_ = make_workload(args.pairs, args.clones, args.seed, rank, world, f"cuda:{local_rank}", chain_, ci_)
        W.append({"chain": chain_, "ci": ci_, "rep": rep_, "pool": pool_, "vc": vc_, "jc": jc_, "v_lines": vl_, "d_pri": pool_.primary, "d_sec": pool_.secondary})
    rep, pool, vc, jc, v_lines = (W[0][n_] for n_ in ("rep", "pool", "vc", "jc", "v_lines"))
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    rl = pool.rl
    ctx = api.Context(local_rank, pinned_results=True)
    ctx.anchor_sets_load(vc, jc)
    ctx.vregion_load(v_lines, 15)
    d_pri, d_sec = pool.primary, pool.secondary

    if world > 1 or args.force_shard:
        # ONE multi-GPU driver, the product's (round 6): csrc/host/vdjx_mgpu.c + vdjx_comm.c -- what `vdjer --gpus N` runs -- as libvdjmgpu.so
        # through ctypes.  The ranks are the launcher's processes; they meet over sockets (vdjx_comm_rendezvous), the bulk bytes move over
        # RCCL (or host sockets when the ranks share one device: dry runs only).  torch.distributed is left with the bench's own
        # bookkeeping (barrier, the slowest rank's time); vdjer_amd/shard.py remains the test-side model of the protocol
        from vdjer_amd import mgpu
        engine = mgpu.Driver(ctx, rank, world, local_rank, "host" if one_device else "rccl")
    else:
        engine = None

    def all_gather_obj(x):
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    state = {}
    host_side = {"pool_generate_s": round(t_gen, 2)}
    # the read index belongs to extraction in the reference (add_read_info is called from extract, bam_read.c:228,243):
    # built once, outside the timed region, over a pool handle that stays alive.  With several GPUs every rank
    # holds the index of the WHOLE pool (replicate the index, shard the windows: no communication).
    t_ix = time.perf_counter()
    # every rank indexes ITS pairs only (the pool is split by pair: both mates of a pair on one rank); windows are mapped by
    # every rank against its own reads and the pair lists meet on the window's owner (vdjx_mgpu.c:do_window_score)
    p_index = ctx.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
    # the per-record read info {pair, read_num, is_rc, registration rank} resident in HBM like the pools; the index itself is
    # built on the device (vdjx_rindex.hip): timed here, reported beside `value` (row a-8's index, quick_map3.c:126-149)
    for w_ in W:
        w_["ri_dev"] = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (w_["pool"].pair_id, w_["pool"].read_num, w_["pool"].is_rc, w_["pool"].reg_rank)]
    ri_dev = W[0]["ri_dev"]
    if args.config4 and world > 1:
        # the pool dealt BY PAIR over the ranks (what `vdjer --gpus N` keeps): the four records of local pair j of rank r sit at scan
        # positions ((j * world) + r) * 4 ... of the primary part (then the secondary part) of the whole pool -- interleaved with every
        # other rank's.  The build works on local record numbers and translates first instances on their way out (vdjx_shard_begin_share)
        for w_ in W:
            npri_, nsec_ = int(w_["d_pri"].shape[0]), int(w_["d_sec"].shape[0])
            mx = torch.tensor([npri_, nsec_], dtype=torch.int64, device=dev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            mp_, ms_ = int(mx[0].item()), int(mx[1].item())
            i_ = torch.arange(npri_, dtype=torch.int64, device=dev)
            j_ = torch.arange(nsec_, dtype=torch.int64, device=dev)
            pos = torch.cat([((i_ // 4) * world + rank) * 4 + i_ % 4, world * mp_ + ((j_ // 4) * world + rank) * 4 + j_ % 4])
            w_["total"] = world * (mp_ + ms_)
            if w_["total"] >= 1 << 32:
                raise SystemExit("scan positions beyond 2^32")
            w_["scan"] = (pos & 0xFFFFFFFF).to(torch.int64).numpy(force=True).astype(np.uint32).view(np.int32)
            w_["scan"] = torch.from_numpy(w_["scan"]).to(dev)
    torch.cuda.synchronize()
    ix_times = []

    def build_index():
        ctx.read_index_build_device(p_index, ri_dev[0].data_ptr(), ri_dev[1].data_ptr(), ri_dev[2].data_ptr(), ri_dev[3].data_ptr(), pool.n_pairs)
    build_index()                       # (first call: allocations)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(3):
        t_ix = time.perf_counter()
        build_index()
        ix_times.append(time.perf_counter() - t_ix)
    ri_prof = ctx.profile_get()
    ctx.profile(False)
    ctx.profile_reset()
    host_side["read_index_build_s"] = round(min(ix_times), 4)
    host_side["read_index"] = {n_: ctx.stat("read_index_" + n_) for n_ in ("classes", "r1_members", "r1_distinct", "rank_order")}
    host_side["read_index_kernels_ms"] = {k_: round(v[0] / 3, 4) for k_, v in ri_prof.items()}
    scorer_src = ("one 486-nt window per clone from the generator (the window the reference derives for that clone's transcript) and, as contigs, the "
                  "[51,411) slices of the windows the coverage test accepts; the host traversal is not run at this size: its contig "
                  "enumeration explodes combinatorially on this repertoire (see --windows)")
    if args.windows == "traversal":
        # the windows the reference would hand to quick_map/coverage for THIS pool: run the serial host stage once, outside the timed
        # region, and keep what it asked the scorers.  Several ranks: rank 0 runs it (as in `vdjer --gpus N`) and calls on the others'
        # scorers, which serve until it yields; the windows and contigs it ends up with are told to everybody
        from vdjer_amd import host
        p0 = ctx.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
        g0 = engine.kmer_build(p0, args.k, args.mf, args.mq) if engine else ctx.kmer_build(p0, args.k, args.mf, args.mq)
        p0.free()
        wins, contigs_fixed, t_tr, st_tr = None, None, 0.0, None
        if rank == 0:
            prm = host.make_params("IGH", ins=args.ins, k=args.k, mf=args.mf, mq=args.mq, mrs=args.mrs, rl=rl)
            rs_fn, ws_fn, _ = host.gpu_hooks(ctx, None, None, prm)
            if engine is not None:
                def ws_fn(w, _ins=args.ins):          # noqa: F811
                    return engine.window_score(w, _ins)[0]
            asked = []

            def ws_capture(w):
                asked.extend(w)
                return ws_fn(w)
            with tempfile.TemporaryDirectory() as td:
                t_tr = time.perf_counter()
                st_tr = host.assemble(prm, g0, rs_fn, ws_capture, None, vc, jc, os.path.join(td, "c.fa"), None, None)
                t_tr = time.perf_counter() - t_tr
                fa = open(os.path.join(td, "c.fa")).read().split("\n")
            wins = asked
            contigs_fixed = [fa[i] for i in range(1, len(fa), 2)]
            if engine is not None:
                engine.yield_step()
        elif engine is not None:
            engine.serve_step(1)
        if world > 1:
            wins, contigs_fixed, t_tr, st_tr = all_gather_obj((wins, contigs_fixed, t_tr, st_tr))[0]
        host_side["traversal_s"] = round(t_tr, 2)
        scorer_src = (f"host traversal of this pool's graph ({t_tr:.1f}s serial incl. its scorer calls, untimed): {st_tr['n_contig_candidates']} contig "
                      f"candidates -> {len(wins)} distinct windows, {len(contigs_fixed)} final contigs")
        del g0
    else:
        for w_ in W:
            libs_w = [w for w in w_["rep"].windows() if w]
            if world > 1:
                from vdjer_amd import synth
                libs_w = [w for r in range(world) for w in synth.make_repertoire(args.clones, seed=args.seed + 15485863 * r + 7 * w_["ci"], chain=w_["chain"], **REP_KW).windows() if w]
            w_["wins"] = libs_w
        wins = W[0]["wins"]
        contigs_fixed = None
    # one GPU: all windows / contigs.  Several GPUs: every rank maps ALL of them against its own reads (the k-mer instances and
    # the reads are sharded, not the windows)
    my_wins = wins
    my_contigs = contigs_fixed
    W[0].setdefault("wins", wins)
    for w_ in W:
        tg = "" if w_["ci"] == 0 else str(w_["ci"])
        w_["wins_packed"] = ctx.pin_strings(w_["wins"], "windows" + tg)
        w_["wins_rows"] = np.frombuffer("".join(w_["wins"]).encode(), np.uint8).reshape(len(w_["wins"]), -1) if w_["wins"] else np.zeros((0, 486), np.uint8)
    my_wins_packed, my_wins_rows = W[0]["wins_packed"], W[0]["wins_rows"]
    my_contigs_packed = ctx.pin_strings(my_contigs, "contigs") if my_contigs else None

    def gather_bytes(a):
        """all_gather of a small uint8 vector whose length differs per rank (bookkeeping of the bench, not the data path)"""
        return [np.asarray(x) for x in all_gather_obj(np.ascontiguousarray(a))]

    wall = {}

    def lap(name, t):
        wall[name] = wall.get(name, 0.0) + (time.perf_counter() - t)
        return time.perf_counter()

    host_fwd = {}

    def step(from_host: bool = False):
        for w_ in W:
            step_chain(w_, from_host)

    def step_chain(w_, from_host: bool = False):
        t = time.perf_counter()
        tg = "" if w_["ci"] == 0 else str(w_["ci"])
        d_pri, d_sec = w_["d_pri"], w_["d_sec"]
        if args.config4:           # a new chain: its ref-dir's anchor sets (two 2^32-bit bitmaps) and V region go up with it
            ctx.anchor_sets_load(w_["vc"], w_["jc"])
            ctx.vregion_load(w_["v_lines"], 15)
            t = lap("chain_ref_dir", t)
        if from_host:      # end-to-end variant: the reads start in page-locked HOST memory (forward records only) and cross PCIe;
            # the NEXT step's pool is uploaded and packed on the copy stream while this step computes (two pools in flight)
            def upload():
                if host_fwd.get("packed"):
                    return ctx.pool_load_packed(host_fwd["pri"].data_ptr(), host_fwd["sec"].data_ptr(), rl, host_fwd["pri"].shape[0],
                                                host_fwd["sec"].shape[0], wait=False)
                return ctx.pool_load_forward(host_fwd["pri"].data_ptr(), host_fwd["sec"].data_ptr(), rl, host_fwd["pri"].shape[0],
                                             host_fwd["sec"].shape[0], wait=False)
            p = host_fwd.pop("next", None) or upload()
            p.wait()
            host_fwd["next"] = upload()
        else:
            p = ctx.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
        t = lap("pool_pack", t)
        if args.config4:           # every chain's pool is seen once: its read index (quick_map3.c:126-149) is part of the step
            rd_ = w_["ri_dev"]
            # begun on the library's index stream: built beside the k-mer build of the same pool (both only read the packed records), ended by
            # the window scorer's first call (VDJX_BENCH_INDEX_SERIAL=1: waited for here, as until round 5)
            ctx.read_index_build_device(p, rd_[0].data_ptr(), rd_[1].data_ptr(), rd_[2].data_ptr(), rd_[3].data_ptr(), w_["pool"].n_pairs,
                                        wait=os.environ.get("VDJX_BENCH_INDEX_SERIAL") == "1")
            t = lap("read_index_begin", t)
        if engine is None:
            g = ctx.kmer_build(p, args.k, args.mf, args.mq, async_export=True)
        elif w_.get("scan") is not None:
            g = engine.kmer_build(p, args.k, args.mf, args.mq, async_export=True, scan_index=w_["scan"], total_records=w_["total"])
        else:
            g = engine.kmer_build(p, args.k, args.mf, args.mq, async_export=True)
        t = lap("kmer_build", t)
        # every root of the graph (nodes without predecessor), scored where the graph lives while its copy for the host traversal
        # is still crossing PCIe; rank r takes roots r, r+world, ...
        # (queued, not waited for: the verdicts -- a few kilobytes -- are looked at after the window scorer has had the device; the windows
        # of this bench do not depend on them)
        root_ids, ok = ctx.root_score_graph(g, args.mrs, rank, world, wait=False)
        t = lap("root_score", t)
        if engine is None:
            valid, npairs = ctx.window_score(w_["wins_packed"], args.ins)
        elif rank == 0:           # rank 0 calls, the others serve (vdjx_mgpu_window_score2 / vdjx_mgpu_serve_step): every rank ends up with all verdicts
            valid, npairs = engine.window_score(w_["wins_packed"], args.ins)
            engine.yield_step()
        else:
            valid, npairs, _ = engine.serve_step(len(w_["wins"]))
        t = lap("window_score", t)
        # generator windows: the contigs are the [51,411) slices of the windows the coverage test accepts (page-locked, like the windows)
        if my_contigs is not None:
            cpk = my_contigs_packed if my_contigs_packed is not None else (b"", 0, 0)
        else:
            cpk = ctx.pin_rows_take(w_["wins_rows"], np.flatnonzero(valid), "contigs" + tg, 51, 360)
        n_contigs = cpk[1]
        # the mapped pairs (20 B each) cross PCIe on the copy stream while the next step's kernels run; they are waited for before
        # the next map_emit reuses the stream (and after the last step, inside the timed region)
        offs, pairs = ctx.map_emit(cpk, async_copy=True)
        t = lap("map_emit", t)
        ctx.root_score_wait()        # the root verdicts are on the host
        g.wait()                     # the graph arrays are on the host
        g.free()
        t = lap("graph_copy_wait", t)
        if world > 1:      # every rank learns every root verdict (a few KB); the window verdicts already are global
            ok = np.concatenate(gather_bytes(ok))
            t = lap("gather_results", t)
        per_chain = dict(nodes=g.n, pre=g.pre_nodes, roots=int(g.n_roots), roots_ok=int(ok.sum()), windows=len(w_["wins"]),
                         valid=int(valid.sum()), contigs=n_contigs if world == 1 else None, mapped_this_rank=int(pairs.shape[0]),
                         window_pairs_this_rank=int(npairs.sum()), n_contigs_rank=n_contigs)
        state.setdefault("per_chain", {})[w_["chain"]] = per_chain
        if w_["ci"] == 0:
            state.update(per_chain)
            state.update(graph=g, last=dict(root_ids=root_ids, ok=ok, valid=valid, npairs=npairs, offs=offs, pairs=pairs))
        p.free()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_step_ms, first_step_phases = None, None
    if args.warmup:
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        step()
        ctx.map_emit_wait()
        torch.cuda.synchronize()
        first_step_ms = (time.perf_counter() - t_first) * 1e3
        first_step_phases = {k_: round(v * 1e3, 3) for k_, v in wall.items()}     # where the first call of a process spends it: workspaces, page-locked result buffers
    for _ in range(max(0, args.warmup - 1)):
        step()
    # Every kernel bracketed by HIP events is 80 event records per step: 0.16 ms of a 10 M-pair step, 0.29 ms of a 1 M-pair one -- the
    # instrument, not the path.  So: (1) n_prof UNTIMED steps with every kernel bracketed -> kernels_ms_per_step, roofline_by_kernel, the
    # busy fraction, and the name of the roofline kernel (the longest HBM-streaming kernel of the build); (2) the timed region with that
    # ONE kernel bracketed (vdjx_profile_only): roofline.achieved is measured live inside the timed region, on the stream the kernel is
    # launched on.  VDJX_BENCH_ALL_EVENTS=1: every kernel inside the timed region, as in rounds 1-4; VDJX_BENCH_NO_EVENTS=1: none anywhere.
    no_events = os.environ.get("VDJX_BENCH_NO_EVENTS") == "1"
    all_events = os.environ.get("VDJX_BENCH_ALL_EVENTS") == "1"
    prof_full, prof_ms_step, n_prof, dom_name = None, None, 0, None
    if not no_events and not all_events:
        n_prof = max(1, min(args.steps, 10))
        ctx.profile(True)
        step()                          # (the events are made on their first use: not a step to measure)
        ctx.map_emit_wait()
        ctx.profile_reset()
        barrier()
        t_p = time.perf_counter()
        for _ in range(n_prof):
            step()
        ctx.map_emit_wait()
        barrier()
        prof_ms_step = (time.perf_counter() - t_p) / n_prof * 1e3
        prof_full = ctx.profile_get()
        # the roofline kernel: the LONGEST scope of the whole step (per step, all its launches), whatever it is -- k-mer build, scorer or one
        # of the composite survivor scopes (round 6; rounds 1-5 looked among the k-mer build's streaming kernels only)
        cand = {n_: v_ for n_, v_ in prof_full.items() if n_ in PRICED_KERNELS}
        dom_name = max(cand.items(), key=lambda kv: kv[1][0])[0] if cand else None
        ctx.profile_only(dom_name)
    ctx.profile(not no_events)
    ctx.profile_reset()
    wall.clear()
    bytes_before = engine.bytes_exchanged if engine else 0
    import gc
    gc.collect()
    gc.disable()                       # no collector pauses inside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.map_emit_wait()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    prof_timed = ctx.profile_get()
    ctx.profile(False)
    ctx.profile_only(None)
    if prof_full is None:               # (every kernel was bracketed inside the timed region, or none at all)
        prof, prof_steps, prof_ms_step = prof_timed, args.steps, dt / args.steps * 1e3
    else:
        prof, prof_steps = dict(prof_full), n_prof
    # ---- parity gate inside the benchmark.  (1) What the LAST TIMED STEP produced -- graph, root verdicts, every
    # window's verdict and pair count, the mapped-pair stream -- against the oracle's digests of this very workload
    # (tests/golden/fullsize_digests.json, made by tests/golden/make_fullsize_digests.py on the same counter-based pool), when the
    # flags name that workload; (2) a small sample of the same generator re-checked against the oracle itself.
    timed_check = None
    dg_path = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
    if rank == 0 and world == 1 and not args.config4 and os.path.exists(dg_path):          # (--force-shard too: the C driver's one-rank build and scorers against the same digests)
        import hashlib
        dg = json.load(open(dg_path))
        case = {(35, 3, 90, 30): "k35", (25, 2, 60, 20): "k25", (35, 3, 60, 30): "k35_mq60"}.get((args.k, args.mf, args.mq, args.mrs))
        if (dg["n_pairs"], dg["n_clones"], dg["seed"]) == (args.pairs, args.clones, args.seed) and case in dg["cases"]:
            def sha(a):
                return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
            d, g, last = dg["cases"][case], state["graph"], state["last"]
            bad = []
            if (g.pre_nodes, g.n) != (d["pre_nodes"], d["nodes"]):
                bad.append("node counts")
            for f in ("first_inst", "freq", "has_v", "has_j", "to_ids", "from_ids"):
                if sha(getattr(g, f)) != d[f]:
                    bad.append("graph." + f)
            if sha(last["root_ids"].astype(np.uint32)) != d["root_ids"] or sha(last["ok"].astype(np.uint8)) != d["root_verdicts"]:
                bad.append("root verdicts")
            checked = ["graph", "roots"]
            bs = dg.get("bench_scorers")
            if bs and args.windows == "generator" and args.ins == bs["ins"] and len(wins) == bs["n_windows"]:
                if sha(last["valid"].astype(np.uint8)) != bs["valid"] or sha(last["npairs"].astype(np.uint32)) != bs["npairs"]:
                    bad.append("window verdicts / pair counts")
                per_contig = np.diff(last["offs"].astype(np.uint64))
                if int(last["pairs"].shape[0]) != bs["pairs_total"] or sha(per_contig.astype(np.uint64)) != bs["pairs_per_contig"]:
                    bad.append("mapped pairs per contig")
                if sha(last["pairs"]) != bs["pairs"]:
                    bad.append("mapped-pair stream")
                checked += ["windows(all)", "mapped pairs(all)"]
            timed_check = {"case": case, "checked": checked, "ok": not bad, "differs": bad}
            if bad and not os.environ.get("VDJX_BENCH_ABLATION"):     # (kernel ablations under profiles/: wrong results on purpose; the line says ok: false)
                raise SystemExit(f"parity gate failed: the timed step's {bad} differ from the oracle digests")
    # ---- the same step fed from the HOST (never `value`): the reads as extracted (one 101-byte record per read; the reverse
    # complement records are derived on the chip) in page-locked memory, uploaded in chunks beside the packing
    e2e = None
    if world == 1 and not args.no_e2e:
        def host_fed(packed: bool):
            """the step of `value` fed from page-locked HOST memory: the reads as extracted (forward records only; the reverse complement
            records are derived on the chip), as 101-byte ASCII records or in the packed host format (vdjx_pool_load_packed: 64 bytes per read)"""
            asc = [torch.empty((d_.shape[0] // 2, d_.shape[1]), dtype=torch.uint8, pin_memory=True) for d_ in (d_pri, d_sec)]
            asc[0].copy_(d_pri[0::2])
            asc[1].copy_(d_sec[0::2])
            torch.cuda.synchronize()
            if packed:
                S = ctx.packed_read_bytes(rl)
                pk = [torch.empty((a_.shape[0], S), dtype=torch.uint8, pin_memory=True) for a_ in asc]
                t_pk = time.perf_counter()
                for a_, p_ in zip(asc, pk):
                    api.check(ctx.L.vdjx_pack_reads(ctypes.c_void_p(a_.data_ptr()), a_.shape[0], rl, ctypes.c_void_p(p_.data_ptr())), "vdjx_pack_reads")
                t_pk = time.perf_counter() - t_pk
                host_fwd["pri"], host_fwd["sec"] = pk
                del asc
            else:
                t_pk = None
                host_fwd["pri"], host_fwd["sec"] = asc
            host_fwd["packed"] = packed
            n_e2e = max(2, min(args.steps, 8))
            step(True)
            wall.clear()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n_e2e):
                step(True)
            ctx.map_emit_wait()
            barrier()
            dte = time.perf_counter() - t1
            up_bytes = (host_fwd["pri"].numel() + host_fwd["sec"].numel())
            res = {"value": round(args.pairs * n_e2e / dte / 1e6, 4), "unit": "M paired-reads/s", "ms_per_step": round(dte / n_e2e * 1e3, 3), "steps": n_e2e,
                   "upload_bytes_per_step": int(up_bytes), "upload_bytes_per_pair": round(up_bytes / args.pairs, 1), "upload_GBps": round(up_bytes / (dte / n_e2e) / 1e9, 1),
                   "pool_load_ms_per_step": round(wall.get("pool_pack", 0.0) / n_e2e * 1e3, 3),
                   "host_pack_s_one_core": round(t_pk, 2) if t_pk is not None else None}
            wall.clear()
            nxt = host_fwd.pop("next", None)
            if nxt is not None:
                nxt.wait()
                nxt.free()
            del host_fwd["pri"], host_fwd["sec"]
            return res
        wall_keep = dict(wall)
        e2e = host_fed(True)
        e2e["note"] = ("host (page-locked) forward reads in the PACKED host format (vdjx_pool_load_packed_begin: 2-bit bases + quality bytes, 64 B per "
                       "50 bp read) -> results: the NEXT step's pool is uploaded in 256 K-read chunks and unpacked on the copy stream while this step "
                       "computes; everything else as in `value`.  ascii_records: the same with add_to_buffer's 101-byte records (vdjx_pool_load_forward_begin)")
        e2e["ascii_records"] = host_fed(False)
        wall.update(wall_keep)
    # ---- the same step with the read index of its pool built inside it (row a-8's index, quick_map3.c:126-149: the reference
    # builds it during extraction, bam_read.c:228,243, once per pool -- like a command line that sees every pool once)
    with_index = None
    if world == 1 and not args.force_shard and not args.no_index_leg:
        n_wi = max(2, min(args.steps, 10))
        wall_keep = dict(wall)

        def index_leg(overlap: bool):
            barrier()
            t1 = time.perf_counter()
            for _ in range(n_wi):
                if overlap:       # begun on the index stream; the step's k-mer build runs beside it; the step's window scorer ends it
                    ctx.read_index_build_device(p_index, ri_dev[0].data_ptr(), ri_dev[1].data_ptr(), ri_dev[2].data_ptr(), ri_dev[3].data_ptr(), pool.n_pairs, wait=False)
                else:
                    build_index()
                step()
            ctx.map_emit_wait()
            barrier()
            return time.perf_counter() - t1
        dts = index_leg(False)
        dtw = index_leg(True)
        with_index = {"value": round(args.pairs * n_wi / dtw / 1e6, 4), "unit": "M paired-reads/s", "ms_per_step": round(dtw / n_wi * 1e3, 3), "steps": n_wi,
                      "serial": {"value": round(args.pairs * n_wi / dts / 1e6, 4), "ms_per_step": round(dts / n_wi * 1e3, 3)},
                      "note": "the step of `value` with the read index of its pool built in it, every step (per-record arrays resident in HBM like the pools): "
                              "vdjx_read_index_build_device_begin -- the index on a stream, a workspace and a thread of its own BESIDE the k-mer build (both only read "
                              "the packed records; quick_map3.c:126-149 vs A2:1388), ended by the window scorer's first call.  `serial`: vdjx_read_index_build_device, "
                              "waited for, then the step (rounds 1-5)"}
        wall.clear()
        wall.update(wall_keep)
    # ---- two batches in flight (never `value`): how much of the step is the device waiting for the host and for kernel tails
    two = None
    if world == 1 and engine is None and not args.no_two_in_flight and not args.no_index_leg:
        import threading
        cx2 = api.Context(local_rank, pinned_results=True)
        cx2.anchor_sets_load(vc, jc)
        cx2.vregion_load(v_lines, 15)
        wp = {id(ctx): W[0]["wins_packed"], id(cx2): cx2.pin_strings(W[0]["wins"], "windows")}
        cfix = {id(ctx): my_contigs_packed, id(cx2): (cx2.pin_strings(my_contigs, "contigs") if my_contigs else None)}
        p_index2 = cx2.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
        cx2.read_index_build_device(p_index2, ri_dev[0].data_ptr(), ri_dev[1].data_ptr(), ri_dev[2].data_ptr(), ri_dev[3].data_ptr(), pool.n_pairs)
        res2 = {}

        def mini_step(cx):
            p_ = cx.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
            g_ = cx.kmer_build(p_, args.k, args.mf, args.mq, async_export=True)
            cx.root_score_graph(g_, args.mrs, 0, 1, wait=False)
            valid_, np_ = cx.window_score(wp[id(cx)], args.ins)
            cpk_ = cfix[id(cx)] if my_contigs is not None else cx.pin_rows_take(W[0]["wins_rows"], np.flatnonzero(valid_), "contigs", 51, 360)
            offs_, pairs_ = cx.map_emit(cpk_ if cpk_ is not None else (b"", 0, 0), async_copy=True)
            cx.root_score_wait()
            g_.wait()
            res2[id(cx)] = (g_.n, int(valid_.sum()), int(pairs_.shape[0]))
            g_.free()
            p_.free()

        pix = {id(ctx): p_index, id(cx2): p_index2}

        def worker(cx, n_, with_ix=False):
            for _ in range(n_):
                if with_ix:       # the pool's read index begun in every step, as in value_with_read_index
                    cx.read_index_build_device(pix[id(cx)], ri_dev[0].data_ptr(), ri_dev[1].data_ptr(), ri_dev[2].data_ptr(), ri_dev[3].data_ptr(), pool.n_pairs, wait=False)
                mini_step(cx)
            cx.map_emit_wait()
        for cx in (ctx, cx2):
            mini_step(cx)           # (first calls of the second context: allocations)
            cx.map_emit_wait()
        n_two = max(2, args.steps // 2 * 2)
        barrier()
        t1 = time.perf_counter()
        worker(ctx, n_two)
        barrier()
        dt_one = time.perf_counter() - t1
        ths = [threading.Thread(target=worker, args=(cx, n_two // 2)) for cx in (ctx, cx2)]
        barrier()
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        barrier()
        dt_two = time.perf_counter() - t1
        ths = [threading.Thread(target=worker, args=(cx, n_two // 2, True)) for cx in (ctx, cx2)]
        barrier()
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        barrier()
        dt_two_ix = time.perf_counter() - t1
        same = res2[id(ctx)] == res2[id(cx2)] == (state["nodes"], state["valid"], state["mapped_this_rank"])
        two = {"value": round(args.pairs * n_two / dt_two / 1e6, 4), "unit": "M paired-reads/s", "ms_per_step": round(dt_two / n_two * 1e3, 3), "steps": n_two,
               "one_in_flight_same_loop": {"value": round(args.pairs * n_two / dt_one / 1e6, 4), "ms_per_step": round(dt_one / n_two * 1e3, 3)},
               "with_read_index": {"value": round(args.pairs * n_two / dt_two_ix / 1e6, 4), "ms_per_step": round(dt_two_ix / n_two * 1e3, 3),
                                   "note": "the same with every pool's read index begun in its step (value_with_read_index, two in flight)"},
               "results_equal_the_timed_steps": bool(same),
               "note": "two contexts on the one device (own streams, workspaces, anchor sets, read index), two host threads, the steps dealt alternately: the "
                       "device's idle time inside one step -- host waits for four sizes, kernel tails -- is filled by the other batch.  Not `value`: a step "
                       "of `value` is one batch alone on the device"}
        p_index2.free()
        cx2.close()
    if world == 1 and engine is not None and not args.no_two_in_flight and not args.no_index_leg:
        # the same through the C driver (--force-shard / --config4 on one GPU): a second context WITH a second one-rank driver; the chains of
        # the steps (config4: IGH, IGK, IGL, IGH, ...) dealt alternately to two host threads -- chain i + 1's packing, index and build under
        # chain i's scorers, which is what two `vdjer --gpus N` runs side by side on the same devices would do
        import threading
        from vdjer_amd import mgpu as _mgpu
        cx2 = api.Context(local_rank, pinned_results=True)
        drv2 = _mgpu.Driver(cx2, 0, 1, local_rank, "rccl")
        pairs_ctx = ((ctx, engine), (cx2, drv2))
        winp = {}
        for cx, _ in pairs_ctx:
            cx.anchor_sets_load(vc, jc)
            cx.vregion_load(v_lines, 15)
            for w_ in W:
                winp[(id(cx), w_["ci"])] = w_["wins_packed"] if cx is ctx else cx.pin_strings(w_["wins"], "windows" + ("" if w_["ci"] == 0 else str(w_["ci"])))
        p_index2 = None
        if not args.config4:          # (the index is outside the step: the second context needs one of its own over the same pool)
            p_index2 = cx2.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
            cx2.read_index_build_device(p_index2, W[0]["ri_dev"][0].data_ptr(), W[0]["ri_dev"][1].data_ptr(), W[0]["ri_dev"][2].data_ptr(), W[0]["ri_dev"][3].data_ptr(), pool.n_pairs)
        res2 = {}

        def chain_step(cx, drv, w_):
            tg = "" if w_["ci"] == 0 else str(w_["ci"])
            if args.config4:
                cx.anchor_sets_load(w_["vc"], w_["jc"])
                cx.vregion_load(w_["v_lines"], 15)
            dp_, ds_ = w_["d_pri"], w_["d_sec"]
            p_ = cx.pool_load_device(dp_.data_ptr(), dp_.shape[0], ds_.data_ptr(), ds_.shape[0], rl)
            if args.config4:
                rd_ = w_["ri_dev"]
                cx.read_index_build_device(p_, rd_[0].data_ptr(), rd_[1].data_ptr(), rd_[2].data_ptr(), rd_[3].data_ptr(), w_["pool"].n_pairs, wait=False)
            g_ = drv.kmer_build(p_, args.k, args.mf, args.mq, async_export=True)
            cx.root_score_graph(g_, args.mrs, 0, 1, wait=False)
            valid_, np_ = drv.window_score(winp[(id(cx), w_["ci"])], args.ins)
            drv.yield_step()
            cpk_ = cx.pin_rows_take(w_["wins_rows"], np.flatnonzero(valid_), "contigs" + tg, 51, 360) if my_contigs is None else (my_contigs_packed if cx is ctx else cfix2)
            offs_, pairs_ = cx.map_emit(cpk_ if cpk_ is not None else (b"", 0, 0), async_copy=True)
            cx.root_score_wait()
            g_.wait()
            res2[(id(cx), w_["ci"])] = (g_.n, int(valid_.sum()), int(pairs_.shape[0]))
            g_.free()
            p_.free()
        cfix2 = cx2.pin_strings(my_contigs, "contigs") if my_contigs else None

        def worker(cx, drv, items):
            for w_ in items:
                chain_step(cx, drv, w_)
            cx.map_emit_wait()
        for cx, drv in pairs_ctx:          # (first calls of the second context: allocations)
            worker(cx, drv, list(W))
        n_two = max(2, min(args.steps, 10) // 2 * 2)
        seq = [w_ for _ in range(n_two) for w_ in W]
        barrier()
        t1 = time.perf_counter()
        worker(ctx, engine, seq)
        barrier()
        dt_one = time.perf_counter() - t1
        ths = [threading.Thread(target=worker, args=(cx, drv, seq[i::2])) for i, (cx, drv) in enumerate(pairs_ctx)]
        barrier()
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        barrier()
        dt_two = time.perf_counter() - t1
        pc = state.get("per_chain", {})
        same = all(res2.get((id(cx), w_["ci"])) in (None, (pc[w_["chain"]]["nodes"], pc[w_["chain"]]["valid"], pc[w_["chain"]]["mapped_this_rank"])) for cx, _ in pairs_ctx for w_ in W)
        tp = args.pairs * len(W) * n_two
        two = {"value": round(tp / dt_two / 1e6, 4), "unit": "M paired-reads/s", "ms_per_step": round(dt_two / n_two * 1e3, 3), "steps": n_two,
               "one_in_flight_same_loop": {"value": round(tp / dt_one / 1e6, 4), "ms_per_step": round(dt_one / n_two * 1e3, 3)},
               "results_equal_the_timed_steps": bool(same),
               "note": "two contexts on the one device, each with a one-rank driver of its own (libvdjmgpu.so), two host threads, the chains of the steps dealt "
                       "alternately: chain i + 1's packing, read index and build run under chain i's scorers.  Not `value`"}
        if p_index2 is not None:
            p_index2.free()
        drv2.close()
        cx2.close()
    del ri_dev
    laps = None
    if os.environ.get("VDJX_LAPS"):          # (diagnostic: host-side microseconds per step inside the scorer calls, vdjx_common.h vdjx_laps)
        names = ("plan_issue", "plan_wait", "ws_cover_wait", "me_key", "me_plan", "me_kernel_wait", "me_second_call", "me_prev_copy_wait", "me_copy_issue")
        laps = {n_: round(ctx.stat("us_" + n_) / (args.warmup + args.steps), 1) for n_ in names}
    stats = {n_: ctx.stat(n_) for n_ in ("window_hits", "window_hits_max", "window_hits_distinct", "window_pairs", "window_work_items", "map_hits", "root_dp_items", "recount_items", "recount_instances", "gated_instances",
                                             "pool_symmetric", "kmer_build_sym", "kmer_build_sym_walk", "kmer_build_sym_walk_retries", "kmer_build_shadows", "read_index_sym")}
    stats["mapped_pairs"] = int(state["last"]["pairs"].shape[0]) if state.get("last") else 0
    stats["window_pairs_entries"] = ctx.stat("window_pairs_entries")
    for n_ in ("group_hits_distinct", "group_overflows", "group_classes", "group_queued", "group_clocks_sum", "group_clocks_max"):      # k_group_pairs: what the groups of windows shared
        stats[n_] = ctx.stat(n_)
    if world > 1:
        dt = max(all_gather_obj(dt))            # the slowest rank's time

    # ---- parity gate, part 2 (part 1, the timed step's own outputs, ran right after the timed loop): a small sample of the same
    # generator re-checked against the oracle itself
    parity = None
    if rank == 0 and args.parity_sample > 0 and world == 1:
        from oracle import oracle
        from vdjer_amd import synth
        n_s = min(args.parity_sample, args.pairs)
        sp = synth.make_reads_cb(rep, n_s, noise_frac=0.3, seed=args.seed + 99)
        ctx.anchor_sets_load(vc, jc)          # (a --config4 step leaves its last chain's sets behind)
        pp = ctx.pool_load(sp.primary, sp.secondary, rl)
        hg = ctx.kmer_build(pp, args.k, args.mf, args.mq)
        pp.free()
        tb = oracle.KmerTable(sp, args.k)
        tb.prune(args.mf, args.mq)
        og = oracle.Graph(tb, vc, jc)
        parity = bool(hg.n == og.n and np.array_equal(hg.first_inst, og.first) and np.array_equal(hg.freq, og.freq)
                      and np.array_equal(hg.to_ids, og.to_ids) and np.array_equal(hg.from_ids, og.from_ids))
        if not parity:
            raise SystemExit("parity gate failed: HIP k-mer build differs from the oracle")

    if engine is not None:          # the job ends the way `vdjer --gpus N` ends: rank 0 releases the others (VDJX_TIMES=1: its per-phase laps on stderr)
        if rank == 0:
            engine.finish()
        else:
            engine.serve_step(1)
    if rank != 0:
        return
    ms_step = dt / args.steps * 1e3
    total_pairs = args.pairs * world * len(W)          # (a --config4 step takes its chains' pools one after the other)
    value = total_pairs * args.steps / dt / 1e6
    ab = algorithmic_bytes_per_pair(args.k, rl, stats.get("gated_instances", 0) / args.pairs if stats.get("gated_instances") else None,      # (the last build's count: one chain's pool)
                                    sym=bool(stats.get("kmer_build_sym")), sym_walk=bool(stats.get("kmer_build_sym_walk")))
    svb = survivor_bytes(int(state.get("nodes", 0)) + int(stats.get("kmer_build_shadows", 0)), args.k)
    # `roofline`: the dominant kernel among the HBM-streaming ones, i.e. the longest kernel of the k-mer build.  The scorer kernels
    # are priced in `roofline_by_kernel` too, but k_window_pairs / k_window_cover work on cache-resident inputs (every read class is
    # looked at by many windows: PMC fabric traffic is a third of SURVEY 8d's per-instance bytes), so an HBM fraction says nothing
    # about them.
    sb = scorer_bytes(stats, len(my_wins), state.get("n_contigs_rank", 0), args.k, rl)
    build_kernels = BUILD_KERNELS

    def price(name, tot_ms, launches):
        avg_ms = tot_ms / max(1, launches)
        per_step = launches / prof_steps
        per_pair = ab.get(name) if name in ab and name not in ("P", "input", "total", "gated_per_pair", "model_all_records", "units_factor") else None
        if name in sb:
            bpl = sb[name] * len(W) / max(1.0, per_step)
        elif name in svb:
            bpl = svb[name] * len(W) / max(1.0, per_step)
        elif per_pair is not None:
            bpl = per_pair * args.pairs * len(W) / max(1.0, per_step)
        else:
            return None                      # a kernel without a stated job size is not priced (never the whole path's bytes)
        ach = bpl / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None
        res = {"avg_launch_ms": round(avg_ms, 4), "launches_per_step": round(per_step, 2), "algorithmic_bytes_per_launch": int(bpl),
               "algorithmic_bytes_per_pair": per_pair, "achieved": round(ach, 2) if ach else None,
               "frac": round(ach / HBM_PEAK_GBS, 5) if ach else None, "traffic": load_traffic(args, world, name)}
        allrec = ab["model_all_records"].get(name)
        if allrec is not None and per_pair and allrec != per_pair and ach:       # (a couples' form: what the same time would be on the model's bytes for ALL records)
            res["frac_model_all_records"] = round(ach * allrec / per_pair / HBM_PEAK_GBS, 5)
        return res

    by_kernel = {}
    for name, (tot_ms, launches) in prof.items():
        pr = price(name, tot_ms, launches)
        if pr:
            pr["hbm_streaming"] = name in build_kernels
            pr["ms_per_step"] = round(tot_ms / prof_steps, 4)
            by_kernel[name] = pr
    # the read index's kernels (one build per pool, outside `value`; inside `value_with_read_index`): bytes this design must move
    ri = host_side.get("read_index", {})
    R_ix, n1_ix, ncls_ix, nd_ix = 4 * args.pairs, ri.get("r1_members", 0), ri.get("classes", 0), ri.get("r1_distinct", 0)
    ri_bytes = {"k_ri_insert": R_ix * 28 + 2 * R_ix * 4,            # packed read + mask in, a 4-byte slot claimed, record -> slot out
                "k_ri_number": 2 * 2 * R_ix * 4 * 2,                  # the build's table (2 R slots) read twice, written once
                "k_ri_records": R_ix * (4 + 4 + 4 + 1 + 1 + 4 + 4) + 2 * args.pairs * 12,
                "k_ri_members": R_ix * (4 + 1 + 4) + n1_ix * (4 + 8 + 8 + 8 + 4 + 4 + 4 + 8),
                "ri_sort_members": n1_ix * (2 * 12 + 3 * 2 * 8 + 8 + 24),       # merge of the two runs, three 8-bit passes over 8-byte keys, the gather
                "k_ri_fold": n1_ix * 8 + nd_ix * 8 + ncls_ix * 12,
                "k_ri_tab": ncls_ix * (4 + 16 + 12 + 64)}
    for name, (tot_ms, launches) in ri_prof.items():
        if name in ri_bytes and tot_ms > 0:
            avg = tot_ms / 3
            by_kernel[name] = {"avg_launch_ms": round(avg, 4), "launches_per_step": "one per index build (3 timed builds)", "algorithmic_bytes_per_launch": int(ri_bytes[name]),
                               "algorithmic_bytes_per_pair": round(ri_bytes[name] / args.pairs, 1), "achieved": round(ri_bytes[name] / (avg * 1e-3) / 1e9, 2),
                               "frac": round(ri_bytes[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": load_traffic(args, world, name), "hbm_streaming": False}
    for v_ in by_kernel.values():       # what the counters say the kernel moved, against the same peak (null without a PMC pass of this workload)
        v_["frac_on_traffic"] = round(v_["traffic"] / (v_["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if v_.get("traffic") and v_["avg_launch_ms"] else None
    dom = max(((n_, v_) for n_, v_ in prof.items() if n_ in PRICED_KERNELS and n_ in by_kernel), key=lambda kv: kv[1][0], default=(None, (0.0, 0)))
    if dom_name and dom_name in prof_timed and dom_name in by_kernel:
        # the roofline kernel's own figures come from INSIDE the timed region (it alone was bracketed there); the untimed pass's are kept beside them
        tt, tl = prof_timed[dom_name]
        live = dict(by_kernel[dom_name])
        scale = (by_kernel[dom_name]["avg_launch_ms"] / (tt / max(1, tl))) if tt > 0 else 1.0
        live.update(avg_launch_ms=round(tt / max(1, tl), 4), avg_launch_ms_untimed_pass=by_kernel[dom_name]["avg_launch_ms"],
                    achieved=round(by_kernel[dom_name]["achieved"] * scale, 2) if by_kernel[dom_name].get("achieved") else None,
                    frac=round(by_kernel[dom_name]["frac"] * scale, 5) if by_kernel[dom_name].get("frac") else None)
        if live.get("frac_model_all_records"):
            live["frac_model_all_records"] = round(live["frac_model_all_records"] * scale, 5)
        if live.get("traffic") and live["avg_launch_ms"]:
            live["frac_on_traffic"] = round(live["traffic"] / (live["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        by_kernel[dom_name] = live
        dom = (dom_name, prof[dom_name])
    roof = None
    if dom[0]:
        pr = by_kernel[dom[0]]
        longest = max(prof.items(), key=lambda kv: kv[1][0] / max(1, kv[1][1]))
        lk = by_kernel.get(longest[0], {})
        step_tr = step_traffic(args, world)
        roof = {"bound": "hbm", "kernel": dom[0], "achieved": pr["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pr["frac"],
                "traffic": pr["traffic"], "traffic_source": traffic_source(args, world), "avg_launch_ms": pr["avg_launch_ms"],
                "algorithmic_bytes_per_launch": pr["algorithmic_bytes_per_launch"],
                "algorithmic_bytes_per_pair": pr["algorithmic_bytes_per_pair"],
                "rule": "the longest scope of the whole step among all priced kernels (k-mer build, survivor scopes, scorers); `achieved` = SURVEY 8d's "
                        "per-unit bytes x the units THIS launch processes / its average duration measured inside the timed region; all priced kernels "
                        "in roofline_by_kernel",
                "units": ({"records_walked_per_pair": 4 * ab["units_factor"]["walk"], "instances_probed_per_pair": ab["P"] * ab["units_factor"]["walk"],
                           "note": "a pool of couples (read, reverse complement): the walk takes the first record of every couple and derives the second (DESIGN 4.1)"}
                          if dom[0] == "k_walk_items" else None),
                "longest_kernel_overall": {"kernel": longest[0], "avg_launch_ms": round(longest[1][0] / max(1, longest[1][1]), 4),
                                           "frac": lk.get("frac"), "achieved": lk.get("achieved"), "traffic": lk.get("traffic"),
                                           "hbm_streaming": lk.get("hbm_streaming")},
                "frac_on_traffic": pr.get("frac_on_traffic"),
                "frac_model_all_records": pr.get("frac_model_all_records"),
                "frac_note": "`frac` = model bytes of the units the launch processes / time / peak (round 6: for a pool of couples the walk's 575 B/pair, "
                             "not the 1,150 of all four records -- `frac_model_all_records` keeps that figure, which charged the kernel for bytes it never "
                             "touched).  `frac_on_traffic` = fabric bytes the counters saw / time / peak (calibrated: profiles/r05_ea_calib).  Neither says "
                             "the walk is bandwidth-bound: it waits on dependent table / chain-word probes (L2 hit 0.80, 72 % of wave-cycles waiting)",
                # the whole step against the peak: every byte the counters saw between the L2s and the memory side, all kernels of a step
                "step_traffic_frac": (round(step_tr["bytes_per_step"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if step_tr else None),
                "step_traffic": step_tr,
                "hot_path_frac": round(ab["total"] * args.pairs * len(W) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # SURVEY 8d's per-pair figure takes EVERY instance as gated (3,324 B at k=35); with the gated instances this run measured
                "hot_path_frac_gated": round((2 * ab["input"] + 32 * ab["gated_per_pair"] + 16 * ab["P"]) * args.pairs * len(W) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    cpu = None
    cpu_port_legs = None
    if not args.no_cpu and world == 1:
        ncores = min(os.cpu_count() or 1, 100)          # `thread_info threads[100]`, A2:1285
        cpu = cpu_reference(min(args.cpu_sample, args.pairs), args.seed, args.k, args.mf, args.mq, args.mrs, args.ins, ncores)
        cpu_port_legs = [cpu_port(min(args.cpu_port_sample, args.pairs), args.seed, args.k, args.mf, args.mq, args.mrs, args.ins, o0)
                         for o0 in (False, True)]
        if cpu is None:           # no compiled reference on this box: the -O2 port is the stated baseline
            cpu = cpu_port_legs[0]
    cli_e2e = cpu.pop("cli_end_to_end", None) if cpu else None
    cli_size = cli_at_size(args, rep, pool) if (args.cli_at_size and REP_KW and world == 1) else None
    kern_ms = {k_: round(v[0] / prof_steps, 4) for k_, v in prof.items()}
    out = {
        "metric": "M paired-reads/sec (k-mer build + contig score), IgH 50bp PE", "value": round(value, 4),
        "unit": "M paired-reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/u64 (2-bit packed bases, integer counts)", "data": "synthetic",
        "config": {"workload": workload_label(args.pairs, args.k, args.mf, args.mq, args.mrs, args.ins, world, chains), "chains": list(chains),
                   "pairs_per_gpu": args.pairs, "clones_per_gpu": args.clones, "noise": 0.3, "parallelism": f"hash-prefix x{world}",
                   "k": args.k, "mf": args.mf, "mq": args.mq, "mrs": args.mrs, "ins": args.ins, "windows": args.windows,
                   "variant": "config4" if args.config4 else ("shard" if args.force_shard else ("k25" if args.k == 25 else "")),
                   "multi_gpu_input": "one independent library (own germline, clones, reads) per GPU; ONE k-mer table / graph / traversal over all of them",
                   "scorer_inputs": scorer_src},
        "roofline": roof, "roofline_by_kernel": by_kernel, "cpu_baseline": cpu, "cpu_baseline_port_legs": cpu_port_legs,
        "speedup_vs_cpu_baseline": round(value / cpu["value"], 1) if cpu else None,
        "value_end_to_end": e2e, "value_with_read_index": with_index, "value_two_in_flight": two, "first_step_ms": round(first_step_ms, 3) if first_step_ms else None, "first_step_phases_ms": first_step_phases,
        "cli_end_to_end": cli_e2e, "cli_at_size": cli_size,
        "kernels_ms_per_step": kern_ms, "kernels_sum_ms_per_step": round(sum(kern_ms.values()), 3),
        "device_busy_frac": round(sum(kern_ms.values()) / prof_ms_step, 4) if prof_ms_step else None,
        "instrumentation": ({"timed_region_events": f"{dom_name} only (the roofline kernel: roofline.avg_launch_ms / achieved / frac are measured inside the timed region)",
                             "per_kernel_pass": f"{n_prof} untimed steps before the timed region with every kernel bracketed by HIP events: kernels_ms_per_step, "
                                                "roofline_by_kernel, device_busy_frac (against that pass's own wall time)",
                             "per_kernel_pass_ms_per_step": round(prof_ms_step, 3),
                             "why": "80 event records per step are 0.16 ms of a 10 M-pair step and 0.29 ms of a 1 M-pair one; VDJX_BENCH_ALL_EVENTS=1 puts them back "
                                    "into the timed region (rounds 1-4)"} if prof_full is not None else
                            {"timed_region_events": "none" if no_events else "every kernel (VDJX_BENCH_ALL_EVENTS=1)"}),
        "exchange_bytes_per_step_rank0": ((engine.bytes_exchanged - bytes_before) // args.steps) if engine else 0,
        "multi_gpu_driver": ("vdjer_amd/libvdjmgpu.so (csrc/host/vdjx_mgpu.c + vdjx_comm.c: the driver of `vdjer --gpus N`), transport "
                             + ("host sockets (ranks share one device: dry run)" if one_device else "rccl")) if engine else None,
        "wall_ms_per_step": {k_: round(v / args.steps * 1e3, 3) for k_, v in wall.items()},
        "host_side": host_side,
        "counts": {k_: v for k_, v in state.items() if k_ not in ("graph", "last")}, "scorer_stats": stats, **({"host_laps_us_per_step": laps} if laps else {}),
        "shard_stats_rank0": ({n_: ctx.stat("shard_" + n_) for n_ in ("partials_received", "open_kmers", "questions", "decided_at_merge",
                                                                    "kept_after_answers")} if engine else None),
        "parity_gate": parity, "parity_gate_timed_step": timed_check,
    }
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
