"""SAM text on the device at size (no oracle: does it run, how fast, are the lines well-formed): python profiles/samtext.py [pairs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vdjer_amd import api, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rep = synth.make_repertoire(max(4, pairs // 500), seed=20261002)
pool = synth.make_reads_cb(rep, pairs, noise_frac=0.3, seed=20261002 + 7, device="cuda:0")
ctx = api.Context(0)
p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
wins = [w for w in rep.windows() if w]
valid, npairs = ctx.window_score(wins, 175)
contigs = [w[51:411] for w, v in zip(wins, valid) if v]
ids = [f"vdj_{i}" for i in range(len(contigs))]
t0 = time.perf_counter()
ctx.sam_names_load([f"r{i}" for i in range(pool.n_pairs)])
t1 = time.perf_counter()
txt = ctx.sam_text_device(contigs, ids)
t2 = time.perf_counter()
offs, mapped = ctx.map_emit(contigs)
nl = txt.count(b"\n")
print(f"{pairs} pairs: {len(contigs)} contigs, {len(mapped)} mapped pairs, SAM text {len(txt) / 1e6:.1f} MB, {nl} lines (2 per pair: {nl == 2 * len(mapped)}), "
      f"names {t1 - t0:.2f} s, text {t2 - t1:.3f} s")
first, last = txt[:400].split(b"\n")[0], txt[-400:].split(b"\n")[-2]
print(first.decode()); print(last.decode())
assert all(len(l.split(b"\t")) >= 11 for l in (first, last))
