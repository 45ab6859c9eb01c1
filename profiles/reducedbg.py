"""Ablation of k_gated_reduce (VDJX_RD_DBG: the kernel returns after 1 sweep 1, 4 sweep 2 (first round), 5 the quality sums; 0 = whole).
One process per level: python profiles/reducedbg.py <level> [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VDJX_RD_DBG"] = sys.argv[1]
# the phase switches live in the ablation build only (make -C vdjer_amd/csrc ablate)
os.environ.setdefault("VDJX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vdjer_amd", "libvdjx_ablate.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from vdjer_amd import api  # noqa: E402

pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
rep, pool, vc, jc, v_lines = bench.make_workload(pairs, pairs // 500, 20240607, 0, 1, "cuda:0")
ctx = api.Context(0)
ctx.anchor_sets_load(vc, jc)
ctx.vregion_load(v_lines, 15)
ctx.profile(True)
for it in range(4):
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    if it == 1:
        ctx.profile_reset()
    try:
        g = ctx.kmer_build(p, 35, 3, 90)
    except Exception as e:  # noqa: BLE001  (a cut kernel leaves no survivors)
        g = None
        if it == 0:
            print('kmer_build:', e)
    torch.cuda.synchronize()
    del g, p
pr = ctx.profile_get()
print("dbg", sys.argv[1], {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items() if "reduce" in k or "part" in k or "walk" in k or "recount" in k})
if os.environ.get("VDJX_WALK_DBG"):
    print("walk:", {n: ctx.stat("walk_dbg_" + n) for n in ("rows", "rounds", "filter", "lookup", "chain", "item_trips")}, {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items() if "walk" in k})
