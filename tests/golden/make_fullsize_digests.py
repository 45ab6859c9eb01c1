#!/usr/bin/env python3
"""Oracle digests of the BASELINE.json-sized configurations (configs[2], configs[3]) -> tests/golden/fullsize_digests.json.

The 10 M-pair pools are far too large to commit, but they come out of a counter-based generator
(vdjer_amd/synth.py:make_reads_cb) that yields the SAME bytes on any device, so the fixture only has to hold what
the CPU oracle (oracle/vdjx_oracle.c, pinned on the reference's dumps by tests/test_oracle_vs_golden.py) computes
from them: SHA-256 digests of the graph arrays, of the root-scorer verdicts and of the window scorer's
(verdict, mapped pairs) lists.  tests/test_gpu_fullsize.py regenerates the pool in HBM and compares the HIP
result's digests: seconds on the GPU box, minutes of oracle time here, once.

    python tests/golden/make_fullsize_digests.py            # ~25 min, ~25 GB on 8 cores
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SEED = 20261002
N_PAIRS, N_CLONES = 10_000_000, 20_000
WINDOW_STEP = 10          # every 10th clone's generator window: 2,000 deep windows
CASES = [("k35", 35, 3, 90, 30), ("k25", 25, 2, 60, 20),     # configs[2]; configs[3] (--mrs 20: SURVEY §0-6)
         ("k35_mq60", 35, 3, 60, 30)]                        # reference point of the 40 M-pair (4x duplicated pool) property test


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def graph_digests(g_first, g_freq, g_hv, g_hj, g_to, g_from) -> dict:
    return {"first_inst": sha(g_first.astype(np.uint64)), "freq": sha(g_freq.astype(np.uint32)),
            "has_v": sha(g_hv.astype(np.uint8)), "has_j": sha(g_hj.astype(np.uint8)),
            "to_ids": sha(g_to.astype(np.uint32)), "from_ids": sha(g_from.astype(np.uint32))}


_IX = None


def _score_window(w):
    pairs, starts = _IX.quick_map(w)
    return int(_IX.coverage_is_valid(starts, len(w), 175)), len(pairs)


def _score_and_map(w):
    """bench.py's scorer workload for one generator window: verdict, mapped pairs, and -- for an accepted window -- the mapped
    pairs of its [51,411) slice in the reference's output order (quick_map3.c:152-181), as the bytes of the vdjx_pair array"""
    pairs, starts = _IX.quick_map(w)
    v = int(_IX.coverage_is_valid(starts, len(w), 175))
    cp = _IX.quick_map(w[51:411])[0].tobytes() if v else None
    return v, len(pairs), cp


def scorers_all(args):
    """--scorers-all: every generator window of the bench workload (one per clone) and the mapped-pair stream of the accepted
    ones -> "bench_scorers" of the existing digest file.  ~15 min on 8 cores."""
    global _IX
    from oracle import oracle
    from vdjer_amd import synth
    out = json.load(open(args.out))
    assert (out["n_pairs"], out["n_clones"], out["seed"]) == (N_PAIRS, N_CLONES, SEED)
    t0 = time.time()
    rep = synth.make_repertoire(N_CLONES, seed=SEED)
    pool = synth.make_reads_cb(rep, N_PAIRS, noise_frac=0.3, seed=SEED)
    assert sha(pool.primary[:100000]) == out["pool"]["primary_head_sha"]
    wins = [w for w in rep.windows() if w]
    _IX = oracle.ReadIndex(pool)
    print(f"pool + read index ({time.time() - t0:.0f}s), {len(wins)} windows", flush=True)
    h = hashlib.sha256()
    valid, npairs, per_contig = [], [], []
    with mp.get_context("fork").Pool(8) as pl:
        for i, (v, n, cp) in enumerate(pl.imap(_score_and_map, wins, chunksize=8)):
            valid.append(v)
            npairs.append(n)
            if v:
                h.update(cp)
                per_contig.append(len(cp) // oracle.PAIR_DTYPE.itemsize)
            if i % 2000 == 0:
                print(f"  window {i} ({time.time() - t0:.0f}s)", flush=True)
    valid, npairs, per_contig = np.array(valid, np.uint8), np.array(npairs, np.uint32), np.array(per_contig, np.uint64)
    out["bench_scorers"] = {"n_windows": len(wins), "ins": 175, "valid": sha(valid), "npairs": sha(npairs), "n_valid": int(valid.sum()),
                            "npairs_sum": int(npairs.astype(np.int64).sum()), "contigs": int(valid.sum()),
                            "contig_rule": "the [51,411) slice of every accepted window, in window order",
                            "pairs_total": int(per_contig.sum()), "pairs_per_contig": sha(per_contig), "pairs": h.hexdigest()}
    print(out["bench_scorers"], f"({time.time() - t0:.0f}s)", flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


def main():
    global _IX, N_PAIRS, N_CLONES
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=N_PAIRS)
    ap.add_argument("--clones", type=int, default=N_CLONES)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "fullsize_digests.json"))
    ap.add_argument("--only", default="", help="comma-separated case names to (re)compute and merge into the existing file; no window pass")
    ap.add_argument("--scorers-all", action="store_true", help="only (re)compute bench_scorers: all windows + the mapped-pair stream")
    args = ap.parse_args()
    N_PAIRS, N_CLONES = args.pairs, args.clones
    if args.scorers_all:
        return scorers_all(args)
    from oracle import oracle
    from vdjer_amd import synth
    out = {"seed": SEED, "n_pairs": N_PAIRS, "n_clones": N_CLONES, "noise": 0.3, "generator": "synth.make_reads_cb",
           "window_step": WINDOW_STEP, "cases": {}}
    only = [x for x in args.only.split(",") if x]
    if only:
        out = json.load(open(args.out))
        assert (out["n_pairs"], out["n_clones"], out["seed"]) == (N_PAIRS, N_CLONES, SEED)
    t0 = time.time()
    rep = synth.make_repertoire(N_CLONES, seed=SEED)
    pool = synth.make_reads_cb(rep, N_PAIRS, noise_frac=0.3, seed=SEED)
    out["pool"] = {"n_primary": int(pool.primary.shape[0]), "n_secondary": int(pool.secondary.shape[0]),
                   "primary_head_sha": sha(pool.primary[:100000]), "secondary_tail_sha": sha(pool.secondary[-100000:])}
    print(f"pool: {pool.primary.shape[0]} + {pool.secondary.shape[0]} records ({time.time() - t0:.0f}s)", flush=True)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    for name, k, mf, mq, mrs in CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        tb = oracle.KmerTable(pool, k)
        pre = tb.size()
        tb.prune(mf, mq)
        g = oracle.Graph(tb, vc, jc)
        d = graph_digests(g.first, g.freq, g.has_v, g.has_j, g.to_ids, g.from_ids)
        d.update(k=k, mf=mf, mq=mq, mrs=mrs, pre_nodes=int(pre), nodes=int(g.n), freq_sum=int(g.freq.astype(np.int64).sum()))
        roots = np.flatnonzero(g.from_deg == 0)
        sc = oracle.RootScorer([rep.v_region], 15)
        ok = np.array([sc.score(oracle.inst_kmer(pool, int(g.first[i]), k), mrs) for i in roots], dtype=np.uint8)
        d.update(n_roots=int(roots.shape[0]), roots_ok=int(ok.sum()), root_ids=sha((roots + 1).astype(np.uint32)), root_verdicts=sha(ok))
        out["cases"][name] = d
        print(f"{name}: pre={pre} nodes={g.n} roots={roots.shape[0]} ok={int(ok.sum())} ({time.time() - t0:.0f}s)", flush=True)
        del tb, g
    if only:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")
        return
    # window scorer (a-8, a-9) on deep windows of the same pool
    t0 = time.time()
    wins = [w for w in rep.windows()[::WINDOW_STEP] if w]
    _IX = oracle.ReadIndex(pool)
    print(f"read index ({time.time() - t0:.0f}s)", flush=True)
    with mp.get_context("fork").Pool(8) as pl:
        res = pl.map(_score_window, wins, chunksize=8)
    valid = np.array([r[0] for r in res], dtype=np.uint8)
    npairs = np.array([r[1] for r in res], dtype=np.uint32)
    out["windows"] = {"n": len(wins), "ins": 175, "valid": sha(valid), "npairs": sha(npairs), "n_valid": int(valid.sum()),
                      "npairs_sum": int(npairs.astype(np.int64).sum())}
    print(f"windows: {len(wins)} scored, {int(valid.sum())} valid, {int(npairs.sum())} pairs ({time.time() - t0:.0f}s)", flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
