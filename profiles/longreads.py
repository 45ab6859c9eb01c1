"""The k-mer build on long reads (the long-read record format): python profiles/longreads.py <read length> [pairs] [k]
-> per-kernel milliseconds (HIP events) of a pool load + build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vdjer_amd import api, synth  # noqa: E402

rl = int(sys.argv[1])
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 35
rep = synth.make_repertoire(max(4, pairs // 500), seed=20261002)
pool = synth.make_reads_cb(rep, pairs, noise_frac=0.3, rl=rl, seed=20261002 + 7, device="cuda:0")
vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
ctx = api.Context(0, pinned_results=True)
ctx.anchor_sets_load(vc, jc)
ctx.profile(True)
import time  # noqa: E402
for it in range(4):
    if it == 1:
        ctx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    g = ctx.kmer_build(p, k, 3, 90)
    n = g.n
    del g, p
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
# the scorers on the same pool (no oracle here: does it run at size, and how fast)
p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
wins = [w for w in rep.windows() if w]
ins = max(175, rl + 40)
torch.cuda.synchronize()
t1 = time.perf_counter()
valid, npairs = ctx.window_score(wins, ins)
contigs = [w[51:411] for w, v in zip(wins, valid) if v]
offs, mapped = ctx.map_emit(contigs) if contigs else (np.zeros(1, np.uint64), np.zeros(0))
torch.cuda.synchronize()
t_sc = time.perf_counter() - t1
print(f"scorers: {len(wins)} windows, {int(valid.sum())} valid, {int(npairs.astype(np.int64).sum())} window pairs, {len(mapped)} mapped pairs, {t_sc * 1e3:.1f} ms")
pr = ctx.profile_get()
print(f"rl {rl}, {pairs} pairs, k {k}: {dt * 1e3:.2f} ms per load + build = {pairs / dt / 1e6:.1f} M pairs/s, {n} nodes")
print({k_: round(v[0] / max(v[1], 1), 3) for k_, v in pr.items()})
