#!/usr/bin/env python3
"""Where the begun read-index build (round 6) wins and loses beside the k-mer build: three phases of N repetitions each, 60 ms of idle
device between them (so that a rocprofv3 --kernel-trace of this process splits into the phases by its gaps):
  A  the index alone (vdjx_read_index_build_device)        B  the k-mer build alone        C  index begun, k-mer build, index ended
    python profiles/overlap.py [pairs]                      -> wall ms per repetition
    rocprofv3 --kernel-trace --output-format csv -d D -- python3 profiles/overlap.py ; python profiles/overlap.py --trace D
                                                              -> per kernel: average duration alone / beside the other stream"""
import csv
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]


def trace(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
    rows.sort()
    phases, cur = [], []
    for r in rows:
        if cur and r[0] - max(x[1] for x in cur[-50:]) > 40e6:
            phases.append(cur)
            cur = []
        cur.append(r)
    phases.append(cur)
    phases = [p for p in phases if any(x[2].startswith("k_ri_") or x[2] == "k_walk_items" for x in p)][-3:]
    out = {}
    for name, p in zip("ABC", phases):
        agg = {}
        for s, e, k, q in p:
            a = agg.setdefault(k, [0, 0])
            a[0] += e - s
            a[1] += 1
        span = (max(x[1] for x in p) - min(x[0] for x in p)) / 1e6
        busy = sum(v[0] for v in agg.values()) / 1e6
        out[name] = {"span_ms": round(span, 3), "kernel_ms_sum": round(busy, 3), "queues": sorted({x[3] for x in p}),
                     "kernels": {k: [round(v[0] / v[1] / 1e3, 1), v[1]] for k, v in agg.items()}}
    names = sorted(set(out["C"]["kernels"]) if "C" in out else [])
    print(json.dumps({n: {k: v for k, v in o.items() if k != "kernels"} for n, o in out.items()}))
    print(f"{'kernel':28s} {'alone us':>10s} {'beside us':>10s} {'x':>6s} {'launches':>8s}")
    for k in sorted(names, key=lambda k: -out["C"]["kernels"][k][0] * out["C"]["kernels"][k][1]):
        alone = out["A"]["kernels"].get(k) or out["B"]["kernels"].get(k)
        c = out["C"]["kernels"][k]
        if alone and c[0] * c[1] > 200:
            print(f"{k:28s} {alone[0]:10.1f} {c[0]:10.1f} {c[0] / alone[0]:6.2f} {c[1]:8d}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        return trace(sys.argv[2])
    import numpy as np
    import torch
    import bench
    from vdjer_amd import api
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    N = 6
    rep, pool, vc, jc, vl = bench.make_workload(pairs, max(4, pairs // 500), 20261002, 0, 1, "cuda:0")
    dev = torch.device("cuda", 0)
    ctx = api.Context(0, pinned_results=True)
    ctx.anchor_sets_load(vc, jc)
    ri = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank)]
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    torch.cuda.synchronize()

    def index(wait=True):
        ctx.read_index_build_device(p, ri[0].data_ptr(), ri[1].data_ptr(), ri[2].data_ptr(), ri[3].data_ptr(), pool.n_pairs, wait=wait)

    def build():
        g = ctx.kmer_build(p, 35, 3, 90, keep_device=True, export=True, async_export=False)
        g.free()
    index(); build(); index(False); build(); ctx.read_index_wait()      # (first calls: allocations)
    res = {}
    for name, fn in (("A_index_alone", lambda: index()), ("B_build_alone", build), ("C_index_beside_build", lambda: (index(False), build(), ctx.read_index_wait()))):
        ctx.sync(); torch.cuda.synchronize()
        time.sleep(0.06)
        t = time.perf_counter()
        for _ in range(N):
            fn()
        ctx.sync()
        res[name] = round((time.perf_counter() - t) / N * 1e3, 3)
    time.sleep(0.06)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
