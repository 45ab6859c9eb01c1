"""Ablation of k_gated_hist / k_part_records (ablation build: make -C vdjer_amd/csrc ablate).  VDJX_HIST_DBG: 1 = loads and gate masks
only, 2 = + the dense listing, 3 = + k-mers cut out and hashed, 4 = + LDS histogram (no flush), 0 = whole.  One process per setting:
python profiles/histdbg.py <level> [pairs] (other knobs through the environment)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VDJX_HIST_DBG"] = sys.argv[1]
os.environ.setdefault("VDJX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vdjer_amd", "libvdjx_ablate.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from vdjer_amd import api  # noqa: E402

pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
rep, pool, vc, jc, v_lines = bench.make_workload(pairs, pairs // 500, 20240607, 0, 1, "cuda:0")
ctx = api.Context(0)
ctx.anchor_sets_load(vc, jc)
ctx.profile(True)
for it in range(4):
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    if it == 1:
        ctx.profile_reset()
    try:
        g = ctx.kmer_build(p, 35, 3, 90)
    except Exception as e:  # noqa: BLE001  (a cut kernel leaves nothing to build)
        g = None
        if it == 0:
            print("kmer_build:", str(e)[:80])
    torch.cuda.synchronize()
    del g, p
pr = ctx.profile_get()
print("dbg", sys.argv[1], {k_: os.environ[k_] for k_ in os.environ if k_.startswith("VDJX_") and k_ != "VDJX_LIB_PATH"},
      {k_: round(v[0] / max(v[1], 1), 3) for k_, v in pr.items() if "hist" in k_ or "part_rec" in k_ or "pack" in k_})
