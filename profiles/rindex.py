#!/usr/bin/env python3
"""The read index's build on its own (vdjx_rindex.hip): kernel times by HIP events, the index's sizes, and what the folding of
identical read-1 entries leaves the window mapper to stream (window_hits_distinct).  usage: python profiles/rindex.py [pairs]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vdjer_amd import api, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rep = synth.make_repertoire(max(4, pairs // 500), seed=20261002)
pool = synth.make_reads_cb(rep, pairs, noise_frac=0.3, seed=20261002, device="cuda:0")
ctx = api.Context(0)
p = ctx.pool_load_device(pool.primary, d_secondary=pool.secondary)
ri = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank)]
torch.cuda.synchronize()


def build():
    ctx.read_index_build_device(p, ri[0].data_ptr(), ri[1].data_ptr(), ri[2].data_ptr(), ri[3].data_ptr(), pool.n_pairs)


build()
ctx.profile(True)
ctx.profile_reset()
ts = []
for _ in range(3):
    t = time.perf_counter()
    build()
    ts.append(time.perf_counter() - t)
prof = ctx.profile_get()
ctx.profile(False)
out = {"pairs": pairs, "build_ms": [round(x * 1e3, 3) for x in ts], "kernels_ms": {k: round(v[0] / 3, 4) for k, v in prof.items()},
       "index": {n: ctx.stat("read_index_" + n) for n in ("classes", "r1_members", "r1_distinct", "rank_order")}}
out["kernels_sum_ms"] = round(sum(out["kernels_ms"].values()), 3)
wins = [w for w in rep.windows() if w]
valid, npairs = ctx.window_score(wins, 175)
out["windows"] = {"n": len(wins), "valid": int(valid.sum()), "npairs": int(npairs.astype(np.int64).sum()),
                  **{n: ctx.stat(n) for n in ("window_hits", "window_hits_distinct", "window_work_items", "group_hits_distinct", "group_overflows")}}
print(json.dumps(out))
