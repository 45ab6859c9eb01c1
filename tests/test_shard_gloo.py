"""world_size-2 `gloo` run (CPU) of the multi-rank driver vdjer_amd/shard.py: the all-to-all / all_gather /
all_reduce choreography (partial aggregates, questions, answers), split arithmetic and record numbering, with the pure-Python phase engine of
tests/shard_ref_engine.py standing in for the GPU engine.  The sharded result must equal the oracle's
single-process result on the rank-major concatenation of the shards."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(rank, n_pairs=90, seed=5):
    from vdjer_amd import synth
    rep = synth.make_repertoire(2, seed=seed)
    pool = synth.make_reads(rep, n_pairs, noise_frac=0.2, seed=100 + rank, err=0.01, n_rate=0.004)
    # reads that exist ONCE per rank, identical on all ranks (plus one variant with weak qualities on rank 1 only): no rank sees
    # two different reads for their k-mers, so the distinct-read flag and the quality sums are settled by the question round
    shared = synth.make_reads(synth.make_repertoire(1, seed=77), 4, noise_frac=0.0, seed=9, err=0.0, n_rate=0.0).primary.copy()
    if rank == 1:
        weak = shared[:2].copy()
        weak[:, 1 + 50:] = ord("6")                      # Phred 21: passes the gate, adds little to the sums
        weak[0, 10] = ord("A") if weak[0, 10] != ord("A") else ord("C")
        shared = np.concatenate([shared, weak])
    else:                                                # same record count on every rank (the oracle's union has no stride gaps)
        blank = np.frombuffer(("0" + "N" * 50 + "I" * 50).encode(), np.uint8)
        shared = np.concatenate([shared, np.stack([blank, blank])])
    import dataclasses
    n_extra = shared.shape[0]
    pool = dataclasses.replace(pool, primary=np.concatenate([pool.primary, shared]),
                               pair_id=np.concatenate([pool.pair_id[:pool.primary.shape[0]], np.zeros(n_extra, np.uint32), pool.pair_id[pool.primary.shape[0]:]]),
                               read_num=np.concatenate([pool.read_num[:pool.primary.shape[0]], np.ones(n_extra, np.uint8), pool.read_num[pool.primary.shape[0]:]]),
                               is_rc=np.concatenate([pool.is_rc[:pool.primary.shape[0]], np.zeros(n_extra, np.uint8), pool.is_rc[pool.primary.shape[0]:]]),
                               reg_rank=np.arange(pool.n_records + n_extra, dtype=np.uint32))
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    return rep, pool, vc, jc


def _worker(rank, world, port, k, mf, mq, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from tests.shard_ref_engine import RefShardEngine
    from vdjer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rep, pool, vc, jc = _make(rank)
        drv = shard.ShardedHotPath(None, dist, torch.device("cpu"), engine=RefShardEngine(vc, jc))
        g = drv.kmer_build(pool, k, mf, mq)
        q.put((rank, g.n, g.pre_nodes, g.first_inst.tolist(), g.freq.tolist(), g.gated_count.tolist(), g.has_v.tolist(),
               g.has_j.tolist(), g.to_ids.tolist(), g.from_ids.tolist(), drv.bytes_exchanged, drv.engine.stats))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,mf,mq", [(2, 35, 2, 60), (2, 25, 2, 40), (3, 35, 2, 60)])
def test_sharded_driver_gloo(world, k, mf, mq):
    import torch.multiprocessing as mp
    from oracle import oracle
    from vdjer_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, mf, mq, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the union pool in rank-major scan order
    pools = [_make(r)[1] for r in range(world)]
    rep, _, vc, jc = _make(0)
    cat = np.concatenate([np.concatenate([p.primary, p.secondary]) for p in pools])
    union = synth.ReadPool(50, cat, np.zeros((0, 101), np.uint8), np.zeros(cat.shape[0], np.uint32), np.zeros(cat.shape[0], np.uint8),
                           np.zeros(cat.shape[0], np.uint8), np.arange(cat.shape[0], dtype=np.uint32), 0)
    t = oracle.KmerTable(union, k)
    pre = t.size()
    t.prune(mf, mq)
    first, count, _, _ = t.export()
    og = oracle.Graph(t, vc, jc)
    assert og.n > 20
    for r in res:                          # identical on every rank
        assert r[1] == og.n and r[2] == pre
        assert r[3] == og.first.tolist()
        assert r[4] == og.freq.tolist()
        assert r[6] == og.has_v.tolist() and r[7] == og.has_j.tolist()
        assert r[8] == og.to_ids.tolist() and r[9] == og.from_ids.tolist()
        assert r[10] > 0
    # the question round did real work: k-mers whose flag was open after the merge (some settled either way), and low counts
    st = [r[11] for r in res]
    assert sum(x["open_flag"] for x in st) > 0 and sum(x["low_count"] for x in st) > 0
    assert 0 < sum(x["flag_set_by_answers"] for x in st) < sum(x["open_flag"] for x in st)
    gc = {int(f): int(c) for f, c in zip(first, count)}
    assert sorted(res[0][5]) == sorted(gc.values())


def _score_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from tests.shard_ref_engine import RefScorerEngine
    from vdjer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool, wins = _scorer_case()
        mine = _pair_share(pool, rank, world)
        drv = shard.ShardedHotPath(None, dist, torch.device("cpu"), engine=object())
        valid, npairs = drv.window_score(RefScorerEngine(mine), wins, 175)
        q.put((rank, valid.tolist(), npairs.tolist()))
    finally:
        dist.destroy_process_group()


def _scorer_case():
    from vdjer_amd import synth
    rep = synth.make_repertoire(3, seed=11)
    pool = synth.tile_reads(rep, [0, 1], copies=2)
    noisy = synth.make_reads(rep, 400, noise_frac=0.1, seed=3)
    import dataclasses
    # dense tiling of two clones (valid windows) + a thin random sample of all three (invalid ones)
    pool = dataclasses.replace(pool, primary=np.concatenate([pool.primary, noisy.primary, noisy.secondary]),
                               pair_id=np.concatenate([pool.pair_id, noisy.pair_id + pool.n_pairs]),
                               read_num=np.concatenate([pool.read_num, noisy.read_num]), is_rc=np.concatenate([pool.is_rc, noisy.is_rc]),
                               reg_rank=np.arange(pool.n_records + noisy.n_records, dtype=np.uint32), n_pairs=pool.n_pairs + noisy.n_pairs)
    wins = [w for w in rep.windows() if w]
    return pool, wins + [w[::-1] for w in wins[:1]]


def _pair_share(pool, rank, world):
    """the records of the pairs p with p % world == rank, in pool order (both mates of a pair stay together)"""
    from vdjer_amd import synth
    sel = np.flatnonzero(pool.pair_id % world == rank)
    allrec = np.concatenate([pool.primary, pool.secondary])
    return synth.ReadPool(pool.rl, allrec[sel], np.zeros((0, allrec.shape[1]), np.uint8), pool.pair_id[sel], pool.read_num[sel], pool.is_rc[sel],
                          np.arange(sel.shape[0], dtype=np.uint32), pool.n_pairs)


def test_sharded_window_scorer_world2_gloo():
    """every rank maps all windows against ITS pairs, the lists meet on the window's owner: same verdicts and pair counts as the
    oracle over the whole pool"""
    import torch.multiprocessing as mp
    from oracle import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_score_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pool, wins = _scorer_case()
    ix = oracle.ReadIndex(pool)
    exp_valid, exp_np = [], []
    for w in wins:
        pairs, starts = ix.quick_map(w)
        exp_np.append(len(pairs))
        exp_valid.append(ix.coverage_is_valid(starts, len(w), 175))
    assert sum(exp_valid) >= 1 and sum(exp_valid) < len(wins)
    for r in res:
        assert r[1] == exp_valid and r[2] == exp_np


def _chunk_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from vdjer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cm = shard.Comm(dist, torch.device("cpu"))
        cm.native = True                      # the code path of the RCCL runs (P2P pieces, chunked reductions), on CPU tensors over gloo
        cm.CHUNK_BYTES = 96                   # 3 rows of 32 bytes per transfer: many rounds, ragged tails, peers with different numbers of rounds
        rng = np.random.default_rng(1234)
        counts = rng.integers(0, 23, (world, world))          # counts[a][b]: rows a sends to b (the same matrix on every rank)
        counts[0, world - 1] = 0                              # an empty pair
        send = torch.cat([torch.full((int(counts[rank, b]), 4), 1000 * rank + b, dtype=torch.int64) +
                          torch.arange(int(counts[rank, b]), dtype=torch.int64).view(-1, 1) * 7 for b in range(world)])
        recv = torch.empty((int(counts[:, rank].sum()), 4), dtype=torch.int64)
        cm.all_to_all_v(send, [int(v) for v in counts[rank]], recv, [int(v) for v in counts[:, rank]])
        exp = torch.cat([torch.full((int(counts[a, rank]), 4), 1000 * a + rank, dtype=torch.int64) +
                         torch.arange(int(counts[a, rank]), dtype=torch.int64).view(-1, 1) * 7 for a in range(world)])
        ok = bool(torch.equal(recv, exp))
        # chunked all_reduce (MIN on the sign-flipped ids, SUM) and all_gather over more rows than one transfer holds
        x = torch.arange(100, dtype=torch.int64) * (rank + 1)
        cm.all_reduce(x, dist.ReduceOp.SUM)
        ok = ok and bool(torch.equal(x, torch.arange(100, dtype=torch.int64) * sum(range(1, world + 1))))
        y = torch.arange(40, dtype=torch.int64).view(10, 4) + 100 * rank
        g = cm.all_gather_cat(y)
        ok = ok and bool(torch.equal(g, torch.cat([torch.arange(40, dtype=torch.int64).view(10, 4) + 100 * a for a in range(world)])))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_in_pieces_over_gloo(world):
    """shard.Comm cuts every transfer into pieces of CHUNK_BYTES (RCCL delivered the second half of a 1.09 GB transfer wrongly):
    the piecewise send/receive rounds, with uneven and empty pairs, and the piecewise reductions and gathers, on CPU tensors."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_chunk_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
