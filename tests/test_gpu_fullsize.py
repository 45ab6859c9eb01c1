"""BASELINE.json-sized parity (-m gpu).  1 M pairs (configs[1]) is compared with the CPU oracle in full; the
10 M-pair configs (configs[2], configs[3]) are compared with committed oracle digests (tests/golden/fullsize_digests.json:
the pool is regenerated in HBM, bit-identical, by the counter-based generator); on top of that come size-independent
properties: determinism, pool duplication (counts double, the graph keeps its shape), and sharded == single-GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _codes(rep):
    from vdjer_amd import synth
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    return vc, jc


def _full_compare(n_pairs, n_clones, k, mf, mq):
    from oracle import oracle
    from vdjer_amd import api, synth
    rep = synth.make_repertoire(n_clones, seed=20261002)
    pool = synth.make_reads(rep, n_pairs, noise_frac=0.3, seed=20261002 + 7919)
    vc, jc = _codes(rep)
    ctx = api.Context(0)
    ctx.anchor_sets_load(vc, jc)
    p = ctx.pool_load(pool.primary, pool.secondary, pool.rl)
    g = ctx.kmer_build(p, k, mf, mq)
    g2 = ctx.kmer_build(p, k, mf, mq)                      # determinism
    for f in ("first_inst", "freq", "gated_count", "to_ids", "from_ids", "kmers"):
        np.testing.assert_array_equal(getattr(g, f), getattr(g2, f))
    t = oracle.KmerTable(pool, k)
    assert g.pre_nodes == t.size()
    t.prune(mf, mq)
    og = oracle.Graph(t, vc, jc)
    assert g.n == og.n
    np.testing.assert_array_equal(g.first_inst, og.first)
    np.testing.assert_array_equal(g.freq, og.freq)
    np.testing.assert_array_equal(g.has_v, og.has_v)
    np.testing.assert_array_equal(g.has_j, og.has_j)
    np.testing.assert_array_equal(g.to_ids, og.to_ids)
    np.testing.assert_array_equal(g.from_ids, og.from_ids)
    p.free()
    ctx.close()
    return g


def test_config1_1M_pairs_k35_full_oracle_compare():
    g = _full_compare(1_000_000, 2000, 35, 3, 90)
    assert g.n > 50_000


def test_sensitive_mode_300k_pairs_k25_full_oracle_compare():
    # configs[3] parameters (--k 25 --mf 2 --mq 60) at a size the oracle finishes in seconds
    _full_compare(300_000, 2000, 25, 2, 60)


# ---- configs[2] / configs[3] at full size: the pool is regenerated in HBM by the counter-based generator (identical bytes on any
# device) and the HIP result is compared with the committed oracle digests (tests/golden/make_fullsize_digests.py)
_FS = {}


def _fullsize_state():
    if _FS:
        return _FS
    import hashlib
    import json
    import torch
    from vdjer_amd import api, synth
    dg = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")))
    rep = synth.make_repertoire(dg["n_clones"], seed=dg["seed"])
    pool = synth.make_reads_cb(rep, dg["n_pairs"], noise_frac=dg["noise"], seed=dg["seed"], device="cuda:0")
    torch.cuda.synchronize()

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert pool.primary.shape[0] == dg["pool"]["n_primary"] and pool.secondary.shape[0] == dg["pool"]["n_secondary"]
    assert sha(pool.primary[:100000].cpu().numpy()) == dg["pool"]["primary_head_sha"], "the generator is not device-independent"
    assert sha(pool.secondary[-100000:].cpu().numpy()) == dg["pool"]["secondary_tail_sha"]
    vc, jc = _codes(rep)
    ctx = api.Context(0)
    ctx.anchor_sets_load(vc, jc)
    ctx.vregion_load([rep.v_region], 15)
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    _FS.update(dg=dg, rep=rep, pool=pool, ctx=ctx, p=p, sha=sha)
    return _FS


@pytest.mark.parametrize("case", ["k35", "k25"])
def test_config2_and_3_10M_pairs_vs_oracle_digests(case):
    st = _fullsize_state()
    d, ctx, sha = st["dg"]["cases"][case], st["ctx"], st["sha"]
    g = ctx.kmer_build(st["p"], d["k"], d["mf"], d["mq"], keep_device=True)
    assert (g.pre_nodes, g.n) == (d["pre_nodes"], d["nodes"])
    assert int(g.freq.astype(np.int64).sum()) == d["freq_sum"]
    for f, key in (("first_inst", "first_inst"), ("freq", "freq"), ("has_v", "has_v"), ("has_j", "has_j"), ("to_ids", "to_ids"),
                   ("from_ids", "from_ids")):
        assert sha(getattr(g, f)) == d[key], f"{case}: {f} differs from the oracle"
    # every k-mer text is the pool slice its first instance names
    rec, off = (g.first_inst >> np.uint64(6)).astype(np.int64), (g.first_inst & np.uint64(63)).astype(np.int64)
    npri = st["pool"].primary.shape[0]
    sel = np.arange(0, g.n, 997)
    import torch
    for i in sel[:200]:
        r = int(rec[i])
        row = st["pool"].primary[r] if r < npri else st["pool"].secondary[r - npri]
        assert bytes(row[1 + int(off[i]):1 + int(off[i]) + d["k"]].cpu().numpy()) == g.kmers[i].tobytes()
    # a-7 over every root of the graph
    ids, ok = ctx.root_score_graph(g, d["mrs"])
    assert ids.shape[0] == d["n_roots"] and sha(ids.astype(np.uint32)) == d["root_ids"]
    assert int(ok.sum()) == d["roots_ok"] and sha(ok.astype(np.uint8)) == d["root_verdicts"]
    # the pool is made of couples (record, reverse complement) and k is odd: both phases took the form that exploits it, and the mirrored
    # walk never had to be done again over every record
    assert ctx.stat("kmer_build_sym") == 1 and ctx.stat("kmer_build_sym_walk") == 1 and ctx.stat("kmer_build_sym_walk_retries") == 0
    assert ctx.stat("kmer_build_shadows") > 0          # k-mers whose own quality test failed and whose reverse complement's passed: no nodes
    g.free()


def test_config2_10M_pairs_sharded_world1_rccl_vs_oracle_digests():
    """The sharded driver over real RCCL with one rank at full size: 34 M partials (1.09 GB) leave and come back, 600 k questions and
    answers, the survivors' gather and the MIN / SUM reductions -- against the same digests.  (A single 1.09 GB all_to_all_single
    came back with its second half wrong on this stack: shard.Comm cuts every transfer into 128 MB pieces.)"""
    import torch
    import torch.distributed as dist
    from vdjer_amd import shard
    st = _fullsize_state()
    d, ctx, sha = st["dg"]["cases"]["k35"], st["ctx"], st["sha"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", 0)
    mine = not dist.is_initialized()
    if mine:
        dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
    try:
        # the exchange itself, on a buffer as large as the partials': what arrives is what left
        x = torch.arange(0, (1088 << 20) // 8, dtype=torch.int64, device=dev).view(-1, 4)
        y = torch.empty_like(x)
        shard.Comm(dist, dev).all_to_all_v(x, [x.shape[0]], y, [x.shape[0]])
        assert torch.equal(x, y)
        del x, y
        g = shard.ShardedHotPath(ctx, dist, dev).kmer_build(st["p"], d["k"], d["mf"], d["mq"])
    finally:
        if mine:
            dist.destroy_process_group()
    assert (g.pre_nodes, g.n) == (d["pre_nodes"], d["nodes"])
    for f in ("first_inst", "freq", "has_v", "has_j", "to_ids", "from_ids"):
        assert sha(getattr(g, f)) == d[f], f"sharded: {f} differs from the oracle"
    g.free()


def test_config2_10M_pairs_window_scorer_vs_oracle_digests():
    st = _fullsize_state()
    dg, ctx, sha, pool = st["dg"], st["ctx"], st["sha"], st["pool"]
    ctx.read_index_build(st["p"], pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    wins = [w for w in st["rep"].windows()[::dg["window_step"]] if w]
    assert len(wins) == dg["windows"]["n"]
    valid, npairs = ctx.window_score(wins, dg["windows"]["ins"])
    assert int(npairs.astype(np.int64).sum()) == dg["windows"]["npairs_sum"] and int(valid.sum()) == dg["windows"]["n_valid"]
    assert sha(npairs.astype(np.uint32)) == dg["windows"]["npairs"]
    assert sha(valid.astype(np.uint8)) == dg["windows"]["valid"]


_PROD_SCORER_CASE = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from vdjer_amd import api, synth
assert "VDJX_HIT_CHUNK" not in os.environ and "VDJX_MAP_SLICE" not in os.environ and "VDJX_GROUP_MIN" not in os.environ
dg = json.load(open(os.path.join(%r, "tests", "golden", "fullsize_digests.json")))
bs = dg["bench_scorers"]
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
rep = synth.make_repertoire(dg["n_clones"], seed=dg["seed"])
pool = synth.make_reads_cb(rep, dg["n_pairs"], noise_frac=dg["noise"], seed=dg["seed"], device="cuda:0")
ctx = api.Context(0)
p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
wins = [w for w in rep.windows() if w]
assert len(wins) == bs["n_windows"]
valid, npairs = ctx.window_score(wins, bs["ins"])
if os.environ.get("VDJX_WINDOW_GROUP") == "0":
    assert ctx.stat("window_work_items") == len(wins)    # (at this depth no window reaches 512 K distinct hits: one workgroup each)
else:
    assert ctx.stat("window_work_items") > len(wins)     # (grouped: what the groups leave over goes in slices of 32,768 hits, as in bench.py)
assert (int(valid.sum()), int(npairs.astype(np.int64).sum())) == (bs["n_valid"], bs["npairs_sum"])
assert sha(valid.astype(np.uint8)) == bs["valid"] and sha(npairs.astype(np.uint32)) == bs["npairs"]
contigs = [w[51:411] for w, v in zip(wins, valid) if v]
offs, pairs = ctx.map_emit(contigs)
assert int(pairs.shape[0]) == bs["pairs_total"] and sha(np.diff(offs.astype(np.uint64)).astype(np.uint64)) == bs["pairs_per_contig"]
assert sha(pairs) == bs["pairs"], "mapped-pair stream differs from the oracle"
print("PROD_SCORER_OK", len(wins), int(valid.sum()), int(pairs.shape[0]))
'''


@pytest.mark.parametrize("grouped", ["1", "0"])
def test_config2_10M_pairs_scorers_production_slicing_vs_oracle_digests(grouped):
    """The scorers exactly as bench.py times them -- shipped VDJX_HIT_CHUNK / slice sizes (the suite otherwise runs with 128-hit
    pieces, tests/conftest.py), all 20,000 generator windows, the mapped pairs of the 3,846 accepted ones -- against the oracle's
    digests (`bench_scorers`, tests/golden/make_fullsize_digests.py --scorers-all).  Fresh process: the knobs are read once.
    Once with the windows mapped in groups of eight (k_group_pairs, the shipped way), once one by one (k_window_pairs alone)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k_: v for k_, v in os.environ.items() if k_ not in ("VDJX_HIT_CHUNK", "VDJX_MAP_SLICE", "VDJX_GROUP_MIN")}
    e["VDJX_WINDOW_GROUP"] = grouped
    r = subprocess.run([sys.executable, "-c", _PROD_SCORER_CASE % (root, root)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0 and "PROD_SCORER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_40M_pairs_one_gpu_beyond_2_31_instances():
    """One GPU, 40 M pairs (160 M records, 2.56 G k-mer instances: past the 2^31 the round-1 instance ids could hold; BASELINE
    configs[4] gives every GPU 12.5 M pairs of a 100 M-pair pool): the 10 M-pair pool four times over.  With mf raised four-fold
    and mq = 60 (a k-mer of >= 3 gated instances of this generator always reaches it) the copies change nothing but the counts:
    same k-mers, same first sights, same edges as the oracle's graph of the single pool (digest case k35_mq60), every count x 4."""
    import torch
    from vdjer_amd import api
    st = _fullsize_state()
    d, sha, pool = st["dg"]["cases"]["k35_mq60"], st["sha"], st["pool"]
    ctx = st["ctx"]
    allrec = torch.cat([pool.primary, pool.secondary])
    big = torch.cat([allrec] * 4)
    del allrec
    torch.cuda.synchronize()
    empty = torch.zeros((0, big.shape[1]), dtype=torch.uint8, device=big.device)
    p4 = ctx.pool_load_device(big.data_ptr(), big.shape[0], 0, 0, pool.rl)
    assert p4.n_records * 16 > 2 ** 31
    ctx.profile(True)
    ctx.profile_reset()
    g = ctx.kmer_build(p4, d["k"], 4 * d["mf"], d["mq"])
    pr = ctx.profile_get()
    ctx.profile(False)
    # 400 M gated instances = 200 M tuples (the pool is made of couples: one tuple per pair of mirrored instances): more than 4,096 per
    # bucket of the histogram's 2^15, so the buckets were cut a second time before the reduce kernel's table could fill
    assert ctx.stat("kmer_build_sym") == 1 and ctx.stat("kmer_build_sym_walk") == 1 and "k_seg_hist" in pr, sorted(pr)
    p4.free()
    del big, empty
    torch.cuda.empty_cache()
    assert (g.n, g.pre_nodes) == (d["nodes"], d["pre_nodes"])
    # primary records come before secondary ones in the single pool and in every copy: same scan order inside the first copy
    for f in ("first_inst", "has_v", "has_j", "to_ids", "from_ids"):
        assert sha(getattr(g, f)) == d[f], f
    g1 = ctx.kmer_build(st["p"], d["k"], d["mf"], d["mq"])
    assert sha(g1.freq) == d["freq"]
    np.testing.assert_array_equal(g.freq, np.minimum(4 * g1.freq.astype(np.int64), 32765).astype(np.uint32))
    np.testing.assert_array_equal(g.gated_count, np.minimum(4 * g1.gated_count.astype(np.int64), 32765).astype(np.uint32))


def test_fullsize_release():
    """frees the 10 M-pair pool (4 GB of ASCII + the packed pool) before the remaining tests"""
    if _FS:
        _FS["p"].free()
        _FS["ctx"].close()
        _FS.clear()
    import torch
    torch.cuda.empty_cache()


def test_pool_duplication_property_2M_pairs():
    """Appending a copy of the pool doubles every count (below saturation) and cannot remove a node; first
    instances and edge order are unchanged because the copy comes later in scan order."""
    from vdjer_amd import api, synth
    rep = synth.make_repertoire(2000, seed=7)
    pool = synth.make_reads(rep, 1_000_000, noise_frac=0.3, seed=8)
    vc, jc = _codes(rep)
    ctx = api.Context(0)
    ctx.anchor_sets_load(vc, jc)
    allrec = np.concatenate([pool.primary, pool.secondary])
    p1 = ctx.pool_load(allrec, np.zeros((0, 101), np.uint8), 50)
    g1 = ctx.kmer_build(p1, 35, 3, 254)                     # mq 254: only quality-saturated k-mers survive in both
    p1.free()
    p2 = ctx.pool_load(np.concatenate([allrec, allrec]), np.zeros((0, 101), np.uint8), 50)
    g2 = ctx.kmer_build(p2, 35, 3, 254)
    p2.free()
    k1 = {g1.kmer(i): i for i in range(0, g1.n, 97)}
    idx2 = {g2.kmers[i].tobytes(): i for i in range(g2.n)}
    assert g2.n >= g1.n
    for km, i in k1.items():
        j = idx2[km.encode()]
        assert g2.first_inst[j] == g1.first_inst[i]
        assert g2.freq[j] == min(2 * int(g1.freq[i]), 32765)
        assert g2.gated_count[j] == min(2 * int(g1.gated_count[i]), 32765)
    ctx.close()


_KNOB_CASE = r'''
import sys
sys.path.insert(0, %r)
from tests.test_gpu_fullsize import _full_compare
from vdjer_amd import api, shard, synth
import numpy as np
g = _full_compare(120000, 300, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
# the same pool through the sharded phases (their local aggregation uses the same partition code), two ranks as threads
import threading, torch
from tests.fake_dist import ThreadDist
rep = synth.make_repertoire(300, seed=20261002)
pool = synth.make_reads(rep, 120000, noise_frac=0.3, seed=20261002 + 7919)
vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
half = pool.primary.shape[0] // 2
dist, out, errs = ThreadDist(2), [None, None], []
def work(r):
    try:
        dist.set_rank(r)
        c = api.Context(0)
        c.anchor_sets_load(vc, jc)
        pri = pool.primary[:half] if r == 0 else np.concatenate([pool.primary[half:], pool.secondary])
        if r == 0:
            pad = np.tile(np.frombuffer(("0" + "N" * 50 + "I" * 50).encode(), np.uint8), (pool.n_records - 2 * half, 1))
            pri = np.concatenate([pri, pad])            # equal strides: the union scan order stays the single-pool order
        p = c.pool_load(pri, np.zeros((0, 101), np.uint8), 50)
        out[r] = shard.ShardedHotPath(c, dist, torch.device("cuda", 0)).kmer_build(p, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
        p.free(); c.close()
    except Exception as e:
        errs.append(e); dist.barrier.abort()
th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
assert not errs, errs
for s in out:
    assert s.n == g.n and s.pre_nodes == g.pre_nodes
    np.testing.assert_array_equal(s.freq, g.freq); np.testing.assert_array_equal(s.gated_count, g.gated_count)
    np.testing.assert_array_equal(s.to_ids, g.to_ids); np.testing.assert_array_equal(s.from_ids, g.from_ids)
    np.testing.assert_array_equal(s.kmers, g.kmers)
print("KNOB_CASE_OK", g.n, g.pre_nodes)
'''


@pytest.mark.parametrize("k,mf,mq,env", [
    (35, 3, 90, {"VDJX_GATED_BUCKET": "4", "VDJX_REFINE_TUPLES": "4"}),                             # > 2^15 buckets: counting pass + 1024-way pass 2
    (25, 2, 60, {"VDJX_GATED_BUCKET": "16", "VDJX_REFINE_TUPLES": "50", "VDJX_SUB_TUPLES": "64"}),     # long buckets split up front (verified dry run)
    (35, 3, 90, {"VDJX_GATED_BUCKET": "100000", "VDJX_SUB_TUPLES": "1000000000"}),                   # few huge buckets: table overflow -> sub-passes
    (35, 3, 90, {"VDJX_RD_DBG": "9"}),                                                               # table + prune: the list of unsettled tuples "ran over": second sweep by rescan
    (25, 2, 60, {"VDJX_RD_DBG": "9", "VDJX_GATED_BUCKET": "100000", "VDJX_SUB_TUPLES": "1000000000"}),   # ... inside hash-selected sub-passes
    (35, 3, 90, {"VDJX_RC_MAX_RANGES": "4", "VDJX_RC_MAX_SHIFT": "9", "VDJX_RC_WIDE": "1"}),         # recount: two partition levels, 64-bit ids
    (25, 2, 60, {"VDJX_RC_MAX_RANGES": "8", "VDJX_RC_MAX_SHIFT": "12"}),                             # recount: the largest ranges
    (35, 3, 90, {"VDJX_RC_LEN_BITS": "0"}),                                                          # > 2^25 survivors: one item per instance
    (25, 2, 60, {"VDJX_RC_LEN_BITS": "2", "VDJX_RC_WIDE": "1", "VDJX_RC_MAX_SHIFT": "8"}),           # > 2^23 survivors: runs of at most 4, 64-bit ids, 256-position ranges
])
def test_large_pool_code_paths_on_a_small_pool(k, mf, mq, env):
    """The paths 10 M-pair pools take (more than 2^15 buckets, long-bucket handling) forced on 120 k pairs through the tuning
    knobs, in a fresh process (the knobs are read once): full comparison with the oracle, single GPU and two sharded ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _KNOB_CASE % root, str(k), str(mf), str(mq)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1200)
    assert r.returncode == 0 and "KNOB_CASE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_20M_pairs_buckets_are_cut_before_the_table_fills():
    """20 M pairs on one GPU: 200 M gated tuples want 2^16 buckets, the histogram has 2^15.  Round 3 found the second cut of the
    buckets starting only at 8,192 tuples per bucket: at 6,100 every bucket's 2,080 distinct k-mers probed the reduce kernel's
    2,048-slot table to the brim, overflowed and started over in sub-passes -- same graph, 52 ms instead of 3.  No oracle at this
    size: the test pins the path (the second-cut kernel runs) and the order of magnitude of the kernel's time."""
    import torch
    from vdjer_amd import api, synth
    torch.cuda.empty_cache()
    rep = synth.make_repertoire(40_000, seed=20261002)
    pool = synth.make_reads_cb(rep, 20_000_000, noise_frac=0.3, seed=20261002 + 7, device="cuda:0")
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ctx = api.Context(0)
    ctx.anchor_sets_load(vc, jc)
    ctx.profile(True)
    p = ctx.pool_load_device(pool.primary.data_ptr(), pool.primary.shape[0], pool.secondary.data_ptr(), pool.secondary.shape[0], pool.rl)
    g = ctx.kmer_build(p, 35, 3, 90)
    ctx.profile_reset()
    g2 = ctx.kmer_build(p, 35, 3, 90)
    pr = ctx.profile_get()
    assert g.n == g2.n and g.n > 1_500_000
    # the buckets are cut a second time exactly when the 2^15 the histogram has would hold more than 4,096 tuples each.  (Round 5: a pool
    # made of couples moves half the tuples -- 100 M at 20 M pairs --, so THIS pool no longer needs the second cut; the 40 M-pair test
    # above pins it)
    tuples = ctx.stat("gated_instances") // (2 if ctx.stat("kmer_build_sym") else 1)
    assert ("k_seg_hist" in pr) == (tuples // 32768 > 4096), (sorted(pr), tuples)
    ms = pr["k_gated_reduce"][0] / max(pr["k_gated_reduce"][1], 1)
    assert ms < 15.0, f"k_gated_reduce took {ms:.1f} ms at 20 M pairs"
    p.free()
    ctx.close()
    del pool
    torch.cuda.empty_cache()


@pytest.mark.parametrize("per,chains,check", [(8_000_000, "IGH", "sharded_equals_one_gpu_build_of_the_union_pool"),
                                              (12_500_000, "IGH,IGK,IGL", "equals_the_build_by_4_ranks_of_twice_the_size")])
def test_config4_geometry_rehearsed_on_one_gpu(per, chains, check):
    """BASELINE.json configs[4] (100 M pairs hash-prefix sharded over 8 GPUs, IGH + IGK + IGL back to back) has never met an 8-GPU
    node; tests/config4_rehearsal.py runs its GEOMETRY here: 8 ranks as threads with their own contexts, 12.5 M pairs each, one
    process for the three chain presets, instance ids past 2^32, 1.8 GB of exchange per rank.  At 8 x 8 M pairs the sharded graph is
    compared with the ONE-GPU build of the union pool (the most one context takes: records x offsets < 2^32), at the full size with the
    build of the same union by 4 ranks of 25 M pairs.  Fresh process (the 8 contexts hold ~160 GB of workspace between them)."""
    import gc
    import json
    import subprocess
    import sys
    # what this process still holds on the device (the 10 M / 40 M-pair tests' context keeps its workspace: tens of GB) goes back first
    if _FS:
        try:
            _FS["p"].free()
            _FS["ctx"].close()
        except Exception:  # noqa: BLE001
            pass
        _FS.clear()
    gc.collect()
    try:
        import torch
        torch.cuda.empty_cache()
    except Exception:  # noqa: BLE001
        pass
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the full geometry also runs the rest of the path on the shares (round 5): read index per rank, the sharded window scorer over 20,000
    # windows, the SAM records of 400 contigs formatted per rank and merged on rank 0 -- 8 ranks == 4 ranks of twice the size, every chain
    mode = "score" if per == 12_500_000 else "build"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "config4_rehearsal.py"), str(per), "8", chains, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-3000:], r.stderr[-3000:])
    if "skipped" in lines[0]:
        pytest.skip(lines[0]["skipped"])
    per_chain = [l for l in lines if "chain" in l]
    assert [l["chain"] for l in per_chain] == chains.split(",") and lines[-1].get("done")
    for l in per_chain:
        assert l["all_ranks_agree"] and l["instance_ids_pass_2_32"] and l["nodes"] > 1_000_000 and l["records"] == per * 32
        moved = [m["build"] if isinstance(m, dict) else m for m in l["bytes_exchanged_per_rank"]]
        assert min(moved) > 400_000_000            # (one 32-byte aggregate per pair of mirrored k-mers since round 5: 0.76 GB at 8 x 8 M pairs, 1.2 GB at 12.5 M)
        if mode == "score":
            assert l["scorers_equal_those_of_half_as_many_ranks"] is True and l["equals_the_build_by_4_ranks_of_twice_the_size"] is True
            assert l["scorers"]["windows"] >= 20_000 and l["scorers"]["windows_valid"] > 0 and l["scorers"]["sam_lines"] > 10_000
        print(json.dumps({k_: l[k_] for k_ in ("chain", "records", "nodes", "pre_nodes", "seconds", "phase_wall_ms_rank0", "bytes_exchanged_per_rank", "scorers") if k_ in l}))
    assert per_chain[-1][check] is True
