"""The FIRST k-mer build of a process, stage by stage (VDJX_SYNC_DEBUG names every stage on stderr after waiting for it; pipe through
`python profiles/first_call.py --stamp` to put a time in front of every line):
    VDJX_SYNC_DEBUG=1 python profiles/first_call.py <k> <mf> <mq> [pairs] 2>&1 | python profiles/first_call.py --stamp"""
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--stamp":
    t0 = last = time.perf_counter()
    for line in sys.stdin:
        now = time.perf_counter()
        sys.stdout.write("%9.3f (+%8.3f ms)  %s" % (now - t0, (now - last) * 1e3, line))
        last = now
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vdjer_amd import api  # noqa: E402

k, mf, mq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
rep, pool, vc, jc, v_lines = bench.make_workload(pairs, pairs // 500, 20240607, 0, 1, "cuda:0")
torch.cuda.synchronize()
ctx = api.Context(0)
ctx.anchor_sets_load(vc, jc)
if os.environ.get("FIRST_CALL_SMALL"):         # a small pool through the same context first (bench.py's parity gate does that)
    n_s = int(os.environ["FIRST_CALL_SMALL"])
    sp, ss = pool.primary[:n_s * 2].contiguous(), pool.secondary[:n_s * 2].contiguous()
    p = ctx.pool_load_device(sp.data_ptr(), sp.shape[0], ss.data_ptr(), ss.shape[0], pool.rl)
    g = ctx.kmer_build(p, k, mf, mq)
    print(f"--- small build: {g.n} nodes", file=sys.stderr, flush=True)
    del g, p
for it in range(2):
    print(f"--- build {it}", file=sys.stderr, flush=True)
    t = time.perf_counter()
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    g = ctx.kmer_build(p, k, mf, mq)
    torch.cuda.synchronize()
    print(f"--- build {it}: {(time.perf_counter() - t) * 1e3:.1f} ms, {g.n} nodes", file=sys.stderr, flush=True)
    del g, p
