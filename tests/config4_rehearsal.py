#!/usr/bin/env python3
"""BASELINE.json configs[4] rehearsed on ONE GPU (test infrastructure; run by tests/test_gpu_fullsize.py in a fresh process).

configs[4] = 100 M 50 bp pairs, hash-prefix sharded over 8 GPUs, IGH then IGK then IGL back to back.  No 8-GPU node has been
available to any round, so its GEOMETRY runs here: 8 ranks as threads of one process (tests/fake_dist.py: the real driver
vdjer_amd/shard.py and the real HIP phases, collectives as tensor copies), each with its own context and 12.5 M pairs
(50 M records) of a 100 M-pair library, global instance ids record << 6 | offset running up to 2.56e10 (past 2^32 from the
second rank on), 8 x 1.3 GB of partial aggregates exchanged -- and the three chain presets (set_chain_info, params.c:13-30) one
after the other in the SAME process and contexts (arenas and exchange buffers reused, a new repertoire / ref-dir / pool each).
Checked: every rank ends with the same graph, and it is the graph the one-GPU build makes of the union pool (400 M records in
one context: itself pinned to the oracle at 10 M and 40 M pairs by the tests around it).  Prints one JSON line per chain with
the phases' wall times and the bytes a rank exchanged.

usage: config4_rehearsal.py [pairs_per_rank=12500000] [ranks=8]
"""
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests.fake_dist import ThreadDist  # noqa: E402
from vdjer_amd import api, shard, synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    k, mf, mq = 35, 3, 90
    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info()
    need = world * per * 2200 + per * world * 1500                       # (measured: ~2.2 KB per pair and rank in flight, ~1.5 KB per pair for the union build)
    if free < need:
        print(json.dumps({"skipped": f"needs about {need >> 30} GiB of device memory, {free >> 30} GiB free"}))
        return 0
    ctxs = [api.Context(0) for _ in range(world)]
    dist = ThreadDist(world)
    t_all = time.perf_counter()
    for ci, chain in enumerate(("IGH", "IGK", "IGL")):
        t0 = time.perf_counter()
        rep = synth.make_repertoire(max(4, per * world // 1000), seed=20261002 + 7 * ci, chain=chain)
        vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
        jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
        pools = [synth.make_reads_cb(rep, per, noise_frac=0.3, seed=20261002 + 1000 * ci, device="cuda:0", pair0=r * per) for r in range(world)]
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        stride = pools[0].primary.shape[0] + pools[0].secondary.shape[0]
        assert all(p.primary.shape[0] + p.secondary.shape[0] == stride for p in pools)
        assert (stride * (world - 1)) << 6 >= 1 << 32 or per < 12_500_000
        out, errs, laps, moved = [None] * world, [], [None] * world, [0] * world

        def work(r):
            try:
                dist.set_rank(r)
                c = ctxs[r]
                c.anchor_sets_load(vc, jc)
                p = c.pool_load_device(pools[r].primary, d_secondary=pools[r].secondary)
                drv = shard.ShardedHotPath(c, dist, dev, stride=stride)
                g = drv.kmer_build(p, k, mf, mq)
                out[r] = {f: sha(getattr(g, f)) for f in ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids")}
                out[r]["n"], out[r]["pre"] = g.n, g.pre_nodes
                if r == 0:
                    out[r]["graph"] = g
                laps[r] = dict(drv.laps)
                moved[r] = drv.bytes_exchanged
                p.free()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
                dist.barrier.abort()

        t1 = time.perf_counter()
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        t_shard = time.perf_counter() - t1
        if errs:
            print(json.dumps({"chain": chain, "error": errs[:2]}))
            return 1
        for r in range(1, world):
            assert {k_: v for k_, v in out[r].items() if k_ != "graph"} == {k_: v for k_, v in out[0].items() if k_ != "graph"}, f"rank {r} differs from rank 0"
        gs = out[0]["graph"]
        # the union pool in ONE context: rank-major concatenation = the sharded build's scan order
        t2 = time.perf_counter()
        cat = torch.cat([x for p in pools for x in (p.primary, p.secondary)])
        del pools
        c0 = ctxs[0]
        pu = c0.pool_load_device(cat)
        gu = c0.kmer_build(pu, k, mf, mq)
        torch.cuda.synchronize()
        t_union = time.perf_counter() - t2
        same = gu.n == gs.n and gu.pre_nodes == gs.pre_nodes and all(np.array_equal(getattr(gu, f), getattr(gs, f)) for f in
                                                                      ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids"))
        line = {"chain": chain, "ranks": world, "pairs_per_rank": per, "records": stride * world, "largest_instance_id": int(gs.first_inst.max()) if gs.n else 0,
                "nodes": int(gs.n), "pre_nodes": int(gs.pre_nodes), "sharded_equals_union": bool(same),
                "seconds": {"generate": round(t_gen, 2), "sharded_build_all_ranks_on_one_device": round(t_shard, 2), "union_build_one_context": round(t_union, 2)},
                "phase_wall_ms_rank0": {k_: round(v * 1e3, 2) for k_, v in laps[0].items()},
                "bytes_exchanged_per_rank": moved, "shard_stats_rank0": {n_: ctxs[0].stat("shard_" + n_) for n_ in ("partials_received", "open_kmers", "questions", "decided_at_merge", "kept_after_answers")}}
        print(json.dumps(line), flush=True)
        pu.free()
        del cat, gu, gs, out
        if not same:
            return 1
    print(json.dumps({"done": True, "seconds": round(time.perf_counter() - t_all, 1), "peak_device_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
