"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (libvdjx.so via ctypes), against
  (1) the golden dumps of the compiled reference in tests/golden/ and
  (2) the CPU oracle (oracle/vdjx_oracle.c) on seeded inputs.
Everything is integer / byte / index work: comparisons are bit-exact.
"""
import os

import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from vdjer_amd import api
    c = api.Context(0)
    yield c
    c.close()


def assert_graph_equals_oracle(hg, og, pool, k):
    from oracle import oracle
    assert hg.n == og.n
    np.testing.assert_array_equal(hg.first_inst, og.first)
    np.testing.assert_array_equal(hg.freq, og.freq)
    np.testing.assert_array_equal(hg.has_v, og.has_v)
    np.testing.assert_array_equal(hg.has_j, og.has_j)
    np.testing.assert_array_equal(hg.to_deg, og.to_deg)
    np.testing.assert_array_equal(hg.from_deg, og.from_deg)
    np.testing.assert_array_equal(hg.to_ids, og.to_ids)
    np.testing.assert_array_equal(hg.from_ids, og.from_ids)
    for i in range(0, hg.n, max(1, hg.n // 200)):
        assert hg.kmer(i) == oracle.inst_kmer(pool, int(og.first[i]), k)


def run_both(ctx, case_pool, v_codes, j_codes, k, mf, mq):
    from oracle import oracle
    t = oracle.KmerTable(case_pool, k)
    pre = t.size()
    t.prune(mf, mq)
    first, count, _, _ = t.export()
    og = oracle.Graph(t, v_codes, j_codes)
    ctx.anchor_sets_load(v_codes, j_codes)
    p = ctx.pool_load(case_pool.primary, case_pool.secondary, case_pool.rl)
    hg = ctx.kmer_build(p, k, mf, mq)
    p.free()
    assert hg.pre_nodes == pre
    assert_graph_equals_oracle(hg, og, case_pool, k)
    # gated count per surviving k-mer (pre_node.frequency)
    o_cnt = {oracle.inst_kmer(case_pool, int(f), k): int(n) for f, n in zip(first, count)}
    h_cnt = {hg.kmer(i): int(hg.gated_count[i]) for i in range(hg.n)}
    assert o_cnt == h_cnt
    return hg


@pytest.mark.parametrize("case,tag", [("noisy", "noisy_k35"), ("noisy", "noisy_k25"), ("noisy", "noisy_mq230"),
                                      ("noisy", "noisy_mq20"), ("pre", "pre_k35")])
def test_kmer_build_vs_reference_dump(ctx, case, tag):
    c = G.Case(case)
    info = G.manifest()[case][tag]
    p = G.flags_to_params(info["flags"])
    hg = run_both(ctx, c.pool, c.v_codes, c.j_codes, p["k"], p["mf"], p["mq"])
    # the goldens' pools are couples (record, reverse complement) as add_to_buffer writes them (bam_read.c:206-244) and k is odd: the build
    # took the form that moves one tuple per pair of mirrored instances (vdjx_pool::sym)
    assert ctx.stat("pool_symmetric") == 1 and ctx.stat("kmer_build_sym") == (p["k"] & 1)
    # ... and phase B walked the couples' first records only, the survivor set closed under reverse complement and the chains mirror
    # images of each other (k_walk_items SYM); the fallback -- every record -- was not needed
    assert ctx.stat("kmer_build_sym_walk") == (p["k"] & 1) and ctx.stat("kmer_build_sym_walk_retries") == 0
    assert hg.pre_nodes == info["pre"]
    nodes = G.rows(f"{tag}.nodes.tsv.gz")
    assert hg.n == len(nodes) == info["nodes"]
    for i, r in enumerate(nodes):
        assert hg.kmer(i) == r[1]
        assert int(hg.freq[i]) == int(r[2])
        assert (int(hg.has_v[i]), int(hg.has_j[i])) == (int(r[3]), int(r[4]))
        assert list(hg.to_ids[i, :hg.to_deg[i]]) == [int(x) for x in r[5].split(",") if x]
        assert list(hg.from_ids[i, :hg.from_deg[i]]) == [int(x) for x in r[6].split(",") if x]
    surv = {r[0]: int(r[1]) for r in G.rows(f"{tag}.survivors.tsv.gz")}
    assert {hg.kmer(i): int(hg.gated_count[i]) for i in range(hg.n)} == surv


@pytest.mark.parametrize("k,mf,mq,n_pairs,seed", [(35, 3, 90, 40000, 1), (25, 2, 60, 20000, 2), (35, 2, 254, 8000, 3),
                                                   (16, 3, 90, 3000, 4), (12, 2, 40, 1500, 5), (48, 2, 60, 6000, 6),
                                                   (50, 2, 60, 6000, 7), (31, 1, 100, 5000, 8)])
def test_kmer_build_vs_oracle_seeded(ctx, k, mf, mq, n_pairs, seed):
    from vdjer_amd import synth
    rep = synth.make_repertoire(12, seed=100 + seed)
    pool = synth.make_reads(rep, n_pairs, noise_frac=0.3, seed=seed, err=0.004, n_rate=0.002)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    run_both(ctx, pool, vc, jc, k, mf, mq)


def test_kmer_build_edge_cases(ctx):
    from vdjer_amd import synth, api
    rep = synth.make_repertoire(2, seed=77)
    vc = np.array([synth.seq_to_int(a) for a in rep.v_anchors], dtype=np.uint32)
    jc = np.array([synth.seq_to_int(a) for a in rep.j_anchors], dtype=np.uint32)
    # empty pools
    empty = synth.ReadPool(50, np.zeros((0, 101), np.uint8), np.zeros((0, 101), np.uint8), np.zeros(0, np.uint32),
                           np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint32), 0)
    ctx.anchor_sets_load(vc, jc)
    p = ctx.pool_load(empty.primary, empty.secondary, 50)
    g = ctx.kmer_build(p, 35, 3, 90)
    assert g.n == 0 and g.pre_nodes == 0
    p.free()
    # secondary-only / primary-only, k == rl (one k-mer per record), all-N and all-low-quality reads
    pool = synth.make_reads(rep, 1200, noise_frac=0.5, seed=9)
    pool.primary[::7, 1:51] = ord("N")
    pool.secondary[::5, 51:101] = ord("#")
    for k, mf, mq in ((50, 2, 60), (35, 3, 90)):
        run_both(ctx, pool, vc, jc, k, mf, mq)
    only_sec = synth.ReadPool(50, np.zeros((0, 101), np.uint8), np.concatenate([pool.primary, pool.secondary]), pool.pair_id,
                              pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    run_both(ctx, only_sec, vc, jc, 35, 3, 90)
    # a malformed record must fail loudly, not be skipped
    bad = pool.primary.copy()
    bad[3, 0] = ord("x")
    with pytest.raises(api.VdjxError):
        ctx.pool_load(bad, pool.secondary, 50)
    with pytest.raises(api.VdjxError):
        ctx.pool_load(np.full((2, 323), ord("0"), np.uint8), np.zeros((0, 323), np.uint8), 161)   # rl > 160


def test_tandem_repeats_and_homopolymers(ctx):
    """k-mers that are their own successor (poly-A) or lie on a short cycle ((AC)n, (ACG)n, (ACGTT)n): the chain order of the
    survivors must leave cycles alone (no head to rank from), and reads that go round a cycle follow it through the successor lists."""
    from vdjer_amd import synth
    rep = synth.make_repertoire(3, seed=15)
    pool = synth.make_reads(rep, 3000, noise_frac=0.2, seed=16)
    rng = np.random.default_rng(3)
    units = ["A", "AC", "ACG", "ACGTT", "T", "GGC"]
    rows = np.arange(0, pool.primary.shape[0], 9)
    for i, r in enumerate(rows):
        u = units[i % len(units)]
        ph = int(rng.integers(0, len(u)))
        tail = 25 if i % 11 == 0 else 10                  # every read leaves the repeat (distinct reads, A2:150-170), a few half way
        seq = (u * 60)[ph:ph + 50 - tail] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, tail))
        pool.primary[r, 1:51] = np.frombuffer(seq.encode(), np.uint8)
        pool.primary[r, 51:101] = ord("I")
    vc = np.array([synth.seq_to_int(a) for a in rep.v_anchors], dtype=np.uint32)
    jc = np.array([synth.seq_to_int(a) for a in rep.j_anchors], dtype=np.uint32)
    for k, mf, mq in ((35, 2, 60), (25, 3, 90), (50, 2, 60)):
        hg = run_both(ctx, pool, vc, jc, k, mf, mq)
        kmers = {hg.kmer(i) for i in range(hg.n)}
        assert k > 40 or ("A" * k in kmers and ("AC" * k)[:k] in kmers and ("ACG" * k)[:k] in kmers)


def test_hot_kmer_skew(ctx):
    """One clone at extreme depth: a handful of k-mers with tens of thousands of instances in one bucket."""
    from vdjer_amd import synth
    rep = synth.make_repertoire(1, seed=5)
    pool = synth.make_reads(rep, 60000, noise_frac=0.02, seed=11)
    vc = np.array([synth.seq_to_int(a) for a in rep.v_anchors], dtype=np.uint32)
    jc = np.array([synth.seq_to_int(a) for a in rep.j_anchors], dtype=np.uint32)
    hg = run_both(ctx, pool, vc, jc, 35, 3, 90)
    assert int(hg.freq.max()) > 1000


@pytest.mark.parametrize("tag", ["k35_t30", "k35_t25", "k35_t34", "k25_t20"])
def test_root_score_vs_reference_dump(ctx, tag):
    c = G.Case("noisy")
    info = G.manifest()["score"][tag]
    ctx.vregion_load([c.v_region], 15)
    rows = G.rows(f"score_{tag}.tsv.gz")
    got = ctx.root_score([r[0] for r in rows], info["k"], info["thr"])
    assert got.tolist() == [int(r[1]) for r in rows]


def test_root_score_vs_oracle_multiline(ctx):
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(5, seed=300)
    lines = [rep.v_region[:6000], rep.v_region[6000:6100], rep.v_region[9000:]]
    s = oracle.RootScorer(lines, 15)
    ctx.vregion_load(lines, 15)
    rng = np.random.default_rng(4)
    qs = []
    for _ in range(300):
        src = rep.v_region if rng.random() < 0.7 else "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
        st = int(rng.integers(0, len(src) - 35))
        q = list(src[st:st + 35])
        for _m in range(int(rng.integers(0, 9))):
            q[int(rng.integers(0, 35))] = "ACGT"[int(rng.integers(0, 4))]
        qs.append("".join(q))
    for thr in (30, 22, 35, 0):
        got = ctx.root_score(qs, 35, thr)
        assert got.tolist() == [s.score(q, thr) for q in qs]


@pytest.mark.parametrize("k,thr", [(35, 30), (25, 20), (35, 0)])
def test_root_score_graph_on_device(ctx, k, thr):
    """roots taken from the device-resident graph == roots of the exported graph pushed through vdjx_root_score == oracle"""
    from oracle import oracle
    c = G.Case("noisy")
    ctx.anchor_sets_load(c.v_codes, c.j_codes)
    ctx.vregion_load([c.v_region], 15)
    p = ctx.pool_load(c.pool.primary, c.pool.secondary, c.pool.rl)
    g = ctx.kmer_build(p, k, 3 if k == 35 else 2, 90 if k == 35 else 60, keep_device=True)
    p.free()
    want_ids = np.flatnonzero(g.from_deg == 0) + 1
    assert g.n_roots == want_ids.shape[0] > 0
    ids, ok = ctx.root_score_graph(g, thr)
    assert ids.tolist() == want_ids.tolist()
    assert ok.tolist() == ctx.root_score(g.kmers[want_ids - 1], k, thr).tolist()
    sc = oracle.RootScorer([c.v_region], 15)
    assert ok.tolist() == [sc.score(g.kmer(int(i) - 1), thr) for i in ids]
    # strided parts (what rank r of `world` ranks scores) tile the whole list
    parts = [ctx.root_score_graph(g, thr, r, 3) for r in range(3)]
    merged = np.zeros(g.n_roots, np.uint8)
    for r, (pi, po) in enumerate(parts):
        assert pi.tolist() == want_ids[r::3].tolist()
        merged[r::3] = po
    assert merged.tolist() == ok.tolist()
    g.free()
    with pytest.raises(Exception):
        ctx.root_score_graph(g, thr)


def test_pinned_result_buffers():
    """results delivered into page-locked buffers (vdjx_host_alloc) are the same bytes"""
    from vdjer_amd import api
    c = G.Case("noisy")
    a, b = api.Context(0), api.Context(0, pinned_results=True)
    try:
        out = []
        for cx in (a, b):
            cx.anchor_sets_load(c.v_codes, c.j_codes)
            p = cx.pool_load(c.pool.primary, c.pool.secondary, c.pool.rl)
            g1 = cx.kmer_build(p, 35, 3, 90)
            snap = [np.array(x) for x in (g1.first_inst, g1.gated_count, g1.freq, g1.has_v, g1.has_j, g1.to_deg, g1.to_ids,
                                          g1.from_deg, g1.from_ids, g1.kmers)]
            g2 = cx.kmer_build(p, 25, 2, 60)     # reuses (pinned) or reallocates the result buffers
            assert g2.n != g1.n
            p.free()
            out.append(snap)
        for x, y in zip(*out):
            assert np.array_equal(x, y)
    finally:
        a.close()
        b.close()


def test_index_generator_vs_reference_rows(ctx):
    """f-3: vdjx_index_generate against the rows the reference's process_kmers printed (tests/golden/index_rows.tsv.gz)"""
    anchors, ranges = G.index_case()
    for s, e, codes, dists in ranges:
        c, d = ctx.index_generate(anchors, s, e)
        assert c.tolist() == codes.tolist() and d.tolist() == dists.tolist()
    # other radii, a range that straddles chunk borders, an empty result, more anchors than one LDS batch
    from oracle import oracle
    rng = np.random.default_rng(8)
    many = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    for a, s, e, md in [(anchors, (1 << 24) - 5000, (1 << 24) + 70000, 3), (anchors, 12345678, 12345678 + 40000, 0),
                        (anchors[:1], int(anchors[0]) - 300, int(anchors[0]) + 300, 5), (many, 3_000_000_000, 3_000_000_000 + 30000, 4),
                        (anchors, 5, 4, 5), (anchors, 2 ** 32 - 100, 2 ** 32 + 50, 5)]:
        c, d = ctx.index_generate(a, s, e, md)
        oc, od = oracle.index_rows(a, s, e, md)
        assert c.tolist() == oc.tolist() and d.tolist() == od.tolist()


@pytest.mark.parametrize("am", [4, 5, 0, 9])
def test_anchor_sets_from_anchors_equal_the_loaded_index(am):
    """the Hamming-ball bitmaps == the sets the reference builds from the index files (load_kmers, vj_filter.c:56-68):
    every row of the full-range generated index with distance <= am, for V and J separately"""
    from vdjer_amd import api, synth
    anchors, _ = G.index_case()
    va, ja = anchors[:7], anchors[7:]
    a, b = api.Context(0), api.Context(0)
    try:
        a.anchor_sets_from_anchors(va, ja, am)
        sets = []
        for x in (va, ja):
            codes, dists = b.index_generate(x, 0, 2 ** 32 - 1, 5)
            assert codes.shape[0] > 1_000_000 and np.all(np.diff(codes.astype(np.int64)) > 0)
            sets.append(codes[dists <= am])
        b.anchor_sets_load(sets[0], sets[1])
        rng = np.random.default_rng(3)
        probe = np.concatenate([sets[0][::max(1, sets[0].shape[0] // 20000)], sets[1][::max(1, sets[1].shape[0] // 20000)],
                                rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32),
                                (anchors[:, None] ^ (np.uint32(3) << (2 * rng.integers(0, 16, (anchors.shape[0], 400)).astype(np.uint32)))).ravel(),
                                np.array([0, 1, 2 ** 32 - 1], np.uint32)])
        contig = "".join(synth.int_to_seq(int(c)) for c in probe) + "A" * 16
        (av, aj), (bv, bj) = a.anchor_probe(contig), b.anchor_probe(contig)
        assert np.array_equal(av, bv) and np.array_equal(aj, bj)
        assert av[::16].sum() > 0 or am == 0
        if am >= 0:
            assert bool(av[::16][:1][0]) == bool(sets[0].shape[0])     # a member of the V set probes as one
    finally:
        a.close()
        b.close()


def test_async_graph_export():
    """vdjx_graph_export_begin/_end: same bytes as the blocking export, with work on the context in between"""
    from vdjer_amd import api
    c = G.Case("noisy")
    cx = api.Context(0, pinned_results=True)
    try:
        cx.anchor_sets_load(c.v_codes, c.j_codes)
        cx.vregion_load([c.v_region], 15)
        p = cx.pool_load(c.pool.primary, c.pool.secondary, c.pool.rl)
        ref = cx.kmer_build(p, 35, 3, 90)
        want = [np.array(x) for x in (ref.first_inst, ref.gated_count, ref.freq, ref.has_v, ref.has_j, ref.to_deg, ref.to_ids, ref.from_deg,
                                      ref.from_ids, ref.kmers)]
        for _ in range(3):
            g = cx.kmer_build(p, 35, 3, 90, async_export=True)
            ids, ok = cx.root_score_graph(g, 30)             # device work while the copy is in flight
            g.wait()
            got = (g.first_inst, g.gated_count, g.freq, g.has_v, g.has_j, g.to_deg, g.to_ids, g.from_deg, g.from_ids, g.kmers)
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
            assert ids.shape[0] == g.n_roots == int((ref.from_deg == 0).sum())
            g.free()
        p.free()
    finally:
        cx.close()


def test_anchor_probe(ctx):
    """seq_to_int + matches_vmer/jmer per contig offset (vj_filter.c:221-238) against the oracle's seq_to_int (pinned on the
    reference's known answers, tests/test_oracle_vs_golden.py) and the golden code sets"""
    from oracle import oracle
    c = G.Case("noisy")
    ctx.anchor_sets_load(c.v_codes, c.j_codes)
    vs, js = set(c.v_codes.tolist()), set(c.j_codes.tolist())
    for t in c.clones:
        ov, oj = ctx.anchor_probe(t)
        exp_v = [int(oracle.seq_to_int(t[i:i + 16]) in vs) for i in range(len(t) - 16)]
        exp_j = [int(oracle.seq_to_int(t[i:i + 16]) in js) for i in range(len(t) - 16)]
        assert ov.tolist() == exp_v and oj.tolist() == exp_j
        assert sum(exp_v) >= 1 and sum(exp_j) >= 1


def _load_index(ctx, pool):
    p = ctx.pool_load(pool.primary, pool.secondary, pool.rl)
    ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    return p


@pytest.mark.parametrize("tag", ["ins175", "ins150_rf2", "ins200_ms20"])
def test_window_score_and_map_vs_reference_dump(ctx, tag):
    from tests.test_oracle_vs_golden import parse_map
    c = G.Case("map")
    info = G.manifest()["map"][tag]
    prm = dict(rs=35, ms=48, rf=1)
    it = iter(info["flags"])
    for f in it:
        prm[f.lstrip("-")] = int(next(it))
    p = _load_index(ctx, c.pool)
    wins = G.text("map_windows.txt.gz").split()
    ref = parse_map(f"map_{tag}.txt.gz")
    by_len = {}
    for i, w in enumerate(wins):
        by_len.setdefault(len(w), []).append(i)
    for ln, idxs in by_len.items():
        ws = [wins[i] for i in idxs]
        valid, npairs = ctx.window_score(ws, info["ins"], rs=prm["rs"], ms=prm["ms"], floor=prm["rf"])
        assert npairs.tolist() == [ref[i]["n"] for i in idxs]
        assert valid.tolist() == [ref[i]["valid"] for i in idxs]
        offs, pairs = ctx.map_emit(ws)
        for j, i in enumerate(idxs):
            mine = [(f"r{q['pair_id']}", int(q["pos1"]), int(q["pos2"]), int(q["insert"]), int(q["rc1"]), int(q["rc2"]))
                    for q in pairs[int(offs[j]):int(offs[j + 1])]]
            assert mine == ref[i]["pairs"]
    p.free()


@pytest.mark.parametrize("tag", ["e2e_tiled", "e2e_mixed", "e2e_k25", "e2e_rl100"])
def test_sam_of_reference_contigs(ctx, tag):
    """a-10: map the reference's own final contigs and reproduce its SAM body byte for byte -- from the pairs (host formatting) and
    with the text formatted on the device (vdjx_sam_text; names with a leading '@' lose it, quick_map3.c:160-163)."""
    from vdjer_amd import api
    c = G.Case(tag)
    p = _load_index(ctx, c.pool)
    fa = G.text(f"{tag}.contigs.fa.gz").splitlines()
    ids = [fa[i][1:] for i in range(0, len(fa), 2)]
    seqs = [fa[i + 1] for i in range(0, len(fa), 2)]
    offs, pairs = ctx.map_emit(seqs)
    body = api.sam_text((c.pool.primary, c.pool.secondary), c.pool.names(), ids, offs, pairs, c.pool.rl)
    head = "@HD\tVN:1.4\tSO:unsorted\n" + "".join(f"@SQ\tSN:{i}\tLN:{len(s)}\n" for i, s in zip(ids, seqs))
    assert head + body == G.text(f"{tag}.sam.gz")
    ctx.sam_names_load(["@" + nm if i % 3 == 0 else nm for i, nm in enumerate(c.pool.names())])
    assert head + ctx.sam_text_device(seqs, ids).decode() == G.text(f"{tag}.sam.gz")
    assert ctx.sam_text_device(seqs[:1], ids[:1]).decode() == body[:len(ctx.sam_text_device(seqs[:1], ids[:1]))]
    p.free()


def test_window_score_vs_oracle_seeded(ctx):
    """Deeper pool, every clone window plus shifted/mutated ones, three parameter sets."""
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(8, seed=900)
    pool = synth.make_reads(rep, 30000, noise_frac=0.1, seed=12)
    ix = oracle.ReadIndex(pool)
    p = _load_index(ctx, pool)
    wins = []
    for w in rep.windows():
        if w is None:
            continue
        wins += [w, w[::-1], synth.revcomp(w)]
    for t in rep.clones:
        for s in (0, 7, 40, 100):
            wins.append(t[s:s + 486])
    for ins, rs, ms, fl in ((175, 35, 48, 1), (160, 30, 60, 2), (175, 35, 48, 0), (230, 35, 48, 1)):
        valid, npairs = ctx.window_score(wins, ins, rs=rs, ms=ms, floor=fl)
        if os.environ.get("VDJX_WP_BUDGET_MB") == "1":           # (the fresh-process run below: the windows go in slices)
            assert ctx.stat("window_slices") > 1
        else:
            assert ctx.stat("window_slices") == 1
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs)
            exp = 1 if fl == 0 else ix.coverage_is_valid(starts, len(w), ins, rs=rs, ms=ms, floor=fl)
            assert int(valid[i]) == exp, (i, ins, rs, ms, fl)
    p.free()


def test_deep_windows_many_slices_and_multiplicities(ctx):
    """One clone at very high depth with clean reads: tens of thousands of hits per window/contig (several MAP_SLICE slices of
    k_map_emit, HIT_CHUNK slices of k_window_pairs) and large multiplicities in the weighted window entries.
    Pairs, their order, counts and verdicts against the oracle."""
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(2, seed=321)
    pool = synth.tile_reads(rep, [0, 1], ins=175, copies=500, step=2)          # every second start, 500 identical pairs each
    ix = oracle.ReadIndex(pool)
    p = _load_index(ctx, pool)
    wins = [w for w in rep.windows() if w] + [t[30:516] for t in rep.clones]
    valid, npairs = ctx.window_score(wins, 175)
    assert ctx.stat("window_hits") > 10 * ctx.stat("window_hits_distinct") > 0          # multiplicities at work
    assert ctx.stat("window_hits_max") > 65536                                          # more than one slice of hits ...
    assert ctx.stat("window_work_items") > len(wins)                                    # ... and the split happened (conftest: VDJX_HIT_CHUNK)
    for i, w in enumerate(wins):
        pairs, starts = ix.quick_map(w)
        assert int(npairs[i]) == len(pairs) > 10000, (i, int(npairs[i]), len(pairs))
        assert int(valid[i]) == ix.coverage_is_valid(starts, len(w), 175)
    contigs = [w[51:411] for w in wins]
    offs, got = ctx.map_emit(contigs)
    assert ctx.stat("map_hits") > 8 * 4096                                              # several slices per contig
    for j, c in enumerate(contigs):
        pairs, _ = ix.quick_map(c)
        mine = got[int(offs[j]):int(offs[j + 1])]
        assert mine.shape[0] == len(pairs) > 4096
        for fld in ("pair_id", "rec1", "rec2", "pos1", "pos2", "insert", "rc1", "rc2"):
            assert np.array_equal(mine[fld].astype(np.int64), pairs[fld].astype(np.int64)), (j, fld)
    p.free()


@pytest.mark.parametrize("world,k,mf,mq,stride,rl", [(2, 35, 3, 90, None, 50), (4, 25, 2, 60, None, 50), (2, 48, 2, 60, None, 50), (8, 35, 2, 60, None, 50),
                                                     (2, 35, 3, 90, 1 << 30, 50), (4, 25, 2, 60, (1 << 30) - 12345, 50),
                                                     (3, 35, 3, 90, None, 50), (6, 25, 2, 60, None, 50), (5, 35, 2, 60, (1 << 29) + 77, 50),
                                                     (2, 35, 3, 90, None, 100), (3, 25, 2, 60, (1 << 28) + 5, 151), (4, 50, 2, 60, None, 75)])
def test_sharded_build_ranks_as_threads(ctx, world, k, mf, mq, stride, rl):
    """The real multi-rank driver + the HIP phase engine with `world` ranks on this one GPU (ranks are threads,
    collectives are tensor copies: tests/fake_dist.py).  Result == single-GPU build of the union pool == oracle.
    stride: the ranks' records numbered 2^30 apart, so that global instance ids (record << 6 | offset) run past 2^32 as they
    do for BASELINE configs[4] (400 M records): same graph, first instances shifted by the rank's base"""
    import threading
    import torch
    from tests.fake_dist import ThreadDist
    from vdjer_amd import api, shard, synth
    rep = synth.make_repertoire(6, seed=41)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    import dataclasses
    ob = 6 if rl <= 64 else 8                                # instance ids: record << 6 | offset, or << 8 with reads of more than 64 bases (the long-read record format)
    pools = [synth.make_reads(rep, 6000 if world < 8 else 1500, noise_frac=0.3, seed=500 + r, err=0.004, n_rate=0.002, rl=rl) for r in range(world)]
    # reads present once on EVERY rank (no rank sees two different reads of their k-mers: the owner has to ask), a variant with
    # weak qualities on rank 1 only (settles some of those flags; low quality sums), blanks elsewhere to keep the strides equal
    shared = synth.make_reads(synth.make_repertoire(2, seed=78), 12, noise_frac=0.0, seed=10, err=0.0, n_rate=0.0, rl=rl).primary.copy()
    weak = shared[:6].copy()
    weak[:, 1 + rl:] = ord("6")
    weak[::2, 12] = np.where(weak[::2, 12] == ord("A"), ord("C"), ord("A"))
    blank = np.frombuffer(("0" + "N" * rl + "I" * rl).encode(), np.uint8)
    for r in range(world):
        tail = weak if r == 1 else np.stack([blank] * weak.shape[0])
        pools[r] = dataclasses.replace(pools[r], primary=np.concatenate([pools[r].primary, shared, tail]))
    cat = np.concatenate([np.concatenate([p.primary, p.secondary]) for p in pools])
    R = cat.shape[0]
    union = synth.ReadPool(rl, cat, np.zeros((0, 2 * rl + 1), np.uint8), np.zeros(R, np.uint32), np.zeros(R, np.uint8),
                           np.zeros(R, np.uint8), np.arange(R, dtype=np.uint32), 0)
    ref = run_both(ctx, union, vc, jc, k, mf, mq)           # single-GPU == oracle, and the reference result
    dist = ThreadDist(world)
    out, errs, stats = [None] * world, [], [None] * world

    def work(r):
        try:
            dist.set_rank(r)
            c = api.Context(0)
            c.anchor_sets_load(vc, jc)
            p = c.pool_load(pools[r].primary, pools[r].secondary, rl)
            drv = shard.ShardedHotPath(c, dist, torch.device("cuda", 0), stride=stride)
            out[r] = drv.kmer_build(p, k, mf, mq)
            stats[r] = {n_: c.stat("shard_" + n_) for n_ in ("partials_received", "open_kmers", "questions", "decided_at_merge", "kept_after_answers")}
            p.free()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            dist.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    # the question round did real work, and settled open k-mers both ways
    tot = {n_: sum(st[n_] for st in stats) for n_ in stats[0]}
    assert tot["questions"] >= tot["open_kmers"] > 0
    assert 0 < tot["kept_after_answers"] < tot["open_kmers"]
    assert tot["decided_at_merge"] + tot["kept_after_answers"] == ref.n
    exp_first = ref.first_inst
    if stride is not None:          # union record rank*R_r + i is global record rank*stride + i
        per = pools[0].n_records
        assert all(p_.n_records == per for p_ in pools)
        rec = ref.first_inst >> np.uint64(ob)
        exp_first = (((rec // np.uint64(per)) * np.uint64(stride) + rec % np.uint64(per)) << np.uint64(ob)) | (ref.first_inst & np.uint64((1 << ob) - 1))
        assert (stride * (world - 1)) << ob >= 1 << 32         # the later ranks' instance ids do not fit 32 bits
    for g in out:
        assert g.n == ref.n and g.pre_nodes == ref.pre_nodes
        np.testing.assert_array_equal(g.first_inst, exp_first)
        np.testing.assert_array_equal(g.freq, ref.freq)
        np.testing.assert_array_equal(g.gated_count, ref.gated_count)
        np.testing.assert_array_equal(g.has_v, ref.has_v)
        np.testing.assert_array_equal(g.to_ids, ref.to_ids)
        np.testing.assert_array_equal(g.from_ids, ref.from_ids)
        np.testing.assert_array_equal(g.kmers, ref.kmers)


@pytest.mark.parametrize("world,k,mf,mq,rl,base", [(2, 35, 3, 90, 50, 0), (3, 25, 2, 60, 50, 0), (4, 35, 2, 60, 50, (1 << 31) + 12345), (8, 35, 2, 60, 50, 0),
                                                   (3, 35, 3, 90, 100, 0), (2, 50, 2, 60, 151, (1 << 29) + 7), (5, 48, 2, 60, 50, 0)])
def test_sharded_build_over_shares_by_pair(ctx, world, k, mf, mq, rl, base):
    """vdjx_shard_begin_share: every rank holds a SHARE of the pool dealt by pair (both mates, all four records of a pair on one rank --
    what `vdjer --gpus N` keeps), its records at scattered places of the scan order, and builds over local record numbers; first
    instances go through the share's scan positions where they leave the rank.  No record moves.  Result == the one-GPU build of
    the whole pool == oracle.  base: the pool's records sit at scan positions base + i of a larger pool (positions and instance
    ids beyond 2^32 >> 6, as at BASELINE configs[4])."""
    import threading
    import torch
    from tests.fake_dist import ThreadDist
    from vdjer_amd import api, shard, synth
    rep = synth.make_repertoire(6, seed=43)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ob = 6 if rl <= 64 else 8
    pool = synth.make_reads(rep, 9000 if world < 8 else 12000, noise_frac=0.3, seed=611, err=0.004, n_rate=0.002, rl=rl)
    ref = run_both(ctx, pool, vc, jc, k, mf, mq)
    npri, R = pool.primary.shape[0], pool.n_records
    cat = np.concatenate([pool.primary, pool.secondary])
    owner = ((pool.pair_id.astype(np.uint64) * np.uint64(2654435761)) >> np.uint64(9)) % np.uint64(world)
    dist = ThreadDist(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            dist.set_rank(r)
            c = api.Context(0)
            c.anchor_sets_load(vc, jc)
            mine = np.flatnonzero(owner == r)                              # ascending scan positions
            p = c.pool_load(np.ascontiguousarray(cat[mine[mine < npri]]), np.ascontiguousarray(cat[mine[mine >= npri]]), rl)
            scan = torch.from_numpy((mine.astype(np.uint64) + np.uint64(base)).astype(np.uint32).view(np.int32)).to("cuda:0")
            drv = shard.ShardedHotPath(c, dist, torch.device("cuda", 0))
            out[r] = drv.kmer_build(p, k, mf, mq, scan_index=scan, total_records=base + R)
            p.free()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            dist.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    exp_first = ref.first_inst + (np.uint64(base) << np.uint64(ob))
    if base:
        assert int(exp_first.max()) >= 1 << 32
    for g in out:
        assert g.n == ref.n and g.pre_nodes == ref.pre_nodes
        np.testing.assert_array_equal(g.first_inst, exp_first)
        np.testing.assert_array_equal(g.freq, ref.freq)
        np.testing.assert_array_equal(g.gated_count, ref.gated_count)
        np.testing.assert_array_equal(g.has_v, ref.has_v)
        np.testing.assert_array_equal(g.to_ids, ref.to_ids)
        np.testing.assert_array_equal(g.from_ids, ref.from_ids)
        np.testing.assert_array_equal(g.kmers, ref.kmers)


@pytest.mark.parametrize("rl", [50, 36, 51, 52, 64])
def test_packed_host_format_equals_the_forward_load(ctx, rl):
    """vdjx_pool_load_packed (2-bit bases + quality bytes, 64 bytes per 50 bp read over PCIe instead of 101) makes the pool that
    vdjx_pool_load_forward makes of the same reads' ASCII records -- and that vdjx_pool_load makes of add_to_buffer's full buffers
    (bam_read.c:206-244): same graph (k-mers, counts, first sights, edges: bases, masks AND quality rows of both records of every
    couple take part), same SAM text (sequences and qualities of both orientations), incl. N, other IUPAC codes and low qualities"""
    from oracle import oracle
    from vdjer_amd import api, synth
    rep = synth.make_repertoire(5, seed=91)
    pool = synth.make_reads(rep, 5000, noise_frac=0.25, seed=92, rl=rl, err=0.004, n_rate=0.004)
    pri, sec = pool.primary.copy(), pool.secondary.copy()
    for a in (pri, sec):                                        # a few IUPAC codes other than N (packed as N; both records of the couple)
        for r in range(0, a.shape[0] - 1, 97 * 2):
            a[r, 1 + 7] = ord("R")
            a[r + 1, 1 + rl - 1 - 7] = ord("Y")
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ctx.anchor_sets_load(vc, jc)
    k = min(35, rl - 8) | 1
    graphs, texts = [], []
    names = [f"r{i}" for i in range(pool.n_pairs)]
    wins = [w for w in rep.windows() if w]
    for how in ("full", "forward", "packed"):
        if how == "full":
            p = ctx.pool_load(pri, sec, rl)
        elif how == "forward":
            p = ctx.pool_load_forward(pri[0::2], sec[0::2], rl)
        else:
            p = ctx.pool_load_packed(api.Context.pack_reads(pri[0::2], rl), api.Context.pack_reads(sec[0::2], rl), rl)
        assert ctx.stat("pool_symmetric") == 1
        g = ctx.kmer_build(p, k, 2, 60)
        graphs.append(g)
        ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        ctx.sam_names_load(names)
        contigs = [w[51:411] for w in wins]
        texts.append(ctx.sam_text_device(contigs, [f"c{i}" for i in range(len(contigs))]))
        p.free()
    assert len(texts[0]) > 10000
    for g in graphs[1:]:
        assert g.n == graphs[0].n > 100 and g.pre_nodes == graphs[0].pre_nodes
        for f in ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids", "kmers"):
            np.testing.assert_array_equal(getattr(g, f), getattr(graphs[0], f), err_msg=f)
    assert texts[1] == texts[0] and texts[2] == texts[0]


@pytest.mark.parametrize("rl", [50, 36, 64])
def test_read_index_over_couples_equals_the_general_build(ctx, monkeypatch, rl):
    """A pool made of couples (record 2i + 1 = reverse complement of record 2i) gets its read index built over the couples
    (k_ri_insert_sym, k_ri_tab_canon: one insertion and one table slot per pair of mirrored sequences; round 5); the general build of
    the same pool (VDJX_NO_SYM_INDEX=1) numbers the classes differently and must give the same index to its users: the same class
    counts, the same verdicts and pair counts of every window, the same mapped pairs of every contig, the same SAM text.  Even read
    lengths (a read can be its own reverse complement: planted) and 64 bases (the reverse complement's own code path) included."""
    from vdjer_amd import synth
    rep = synth.make_repertoire(6, seed=191)
    pool = synth.make_reads(rep, 6000, noise_frac=0.25, seed=192, rl=rl, err=0.004, n_rate=0.003)
    pri = pool.primary.copy()
    if rl % 2 == 0:                                            # reads that are their own reverse complement, as read 1 and as read 2
        pal = ("ACGT" * 16)[:rl // 2]
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        pal = pal + "".join(comp[c] for c in reversed(pal))
        for r in (0, 2, 4 * 37, 4 * 37 + 2):
            pri[r, 1:1 + rl] = np.frombuffer(pal.encode(), dtype=np.uint8)
            pri[r + 1, 1:1 + rl] = pri[r, 1:1 + rl]
    names = [f"r{i}" for i in range(pool.n_pairs)]
    wins = [w for w in rep.windows() if w]
    wl = len(wins[0])
    if rl % 2 == 0:
        wins.append((pal + "ACGT" * 200)[:wl])                 # a window that holds the planted read
    contigs = [w[51:411] for w in wins]
    p = ctx.pool_load(pri, pool.secondary, rl)
    assert ctx.stat("pool_symmetric") == 1
    got = []
    for no_sym in (False, True):
        if no_sym:
            monkeypatch.setenv("VDJX_NO_SYM_INDEX", "1")
        ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        assert ctx.stat("read_index_sym") == (0 if no_sym else 1)
        ctx.sam_names_load(names)
        valid, npairs = ctx.window_score(wins, 175)
        offs, pairs = ctx.map_emit(contigs)
        text = ctx.sam_text_device(contigs, [f"c{i}" for i in range(len(contigs))])
        st = {n: ctx.stat("read_index_" + n) for n in ("r1_members", "r1_distinct")}
        got.append((valid.copy(), npairs.copy(), offs.copy(), pairs.copy(), text, st, ctx.stat("read_index_classes")))
    monkeypatch.delenv("VDJX_NO_SYM_INDEX")
    a, b = got
    assert a[1].sum() > 1000 and len(a[4]) > 10000
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    for f in a[3].dtype.names:
        if f not in ("cls",):
            np.testing.assert_array_equal(a[3][f], b[3][f], err_msg=f)
    assert a[4] == b[4] and a[5] == b[5]
    # (the couples' build numbers both sides of every pair: one class more than the general one per read that is its own reverse complement)
    assert a[6] >= b[6] and a[6] - b[6] <= 8
    p.free()


@pytest.mark.parametrize("rl", [50, 100])
def test_read_index_begun_beside_the_kmer_build(rl):
    """vdjx_read_index_build[_device]_begin / _end (round 6): the index on a stream, a workspace and a thread of its own beside the k-mer
    build of the same pool (quick_map3.c:126-149 against A2:1388: nothing orders the two before the first quick_map_process_contig).
    Same graph as without it (vs the oracle), same window verdicts / pair counts / mapped pairs / SAM text as the waiting build;
    ended by _end, or by the first scorer call; several rounds in one context (the index arrays are recycled); a bad pair id
    comes back from _end as the waiting call's error, and the scorers then refuse."""
    import torch
    from vdjer_amd import api, synth
    from vdjer_amd._lib import VdjxError
    c = api.Context(0)
    rep = synth.make_repertoire(6, seed=411)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    c.anchor_sets_load(vc, jc)
    wins = [w for w in rep.windows() if w]
    contigs = [w[51:411] for w in wins]
    for it, n_pairs in enumerate((5000, 9000, 5000)):
        pool = synth.make_reads(rep, n_pairs, noise_frac=0.25, seed=412 + it, rl=rl, err=0.004, n_rate=0.003)
        names = [f"r{i}" for i in range(pool.n_pairs)]
        p = c.pool_load(pool.primary, pool.secondary, rl)
        # the waiting build: what every result is compared with
        c.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        c.sam_names_load(names)
        want = (c.window_score(wins, 175), c.map_emit(contigs), c.sam_text_device(contigs, [f"c{i}" for i in range(len(contigs))]))
        want = ([x.copy() for x in want[0]], [x.copy() for x in want[1]], want[2])
        g0 = c.kmer_build(p, 35, 3, 90)
        for mode in ("host_end", "device_first_scorer_call"):
            if mode == "host_end":
                c.read_index_build_begin(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
            else:
                dv = [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in (pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank)]
                torch.cuda.synchronize()
                c.read_index_build_device(p, dv[0].data_ptr(), dv[1].data_ptr(), dv[2].data_ptr(), dv[3].data_ptr(), pool.n_pairs, wait=False)
            g = c.kmer_build(p, 35, 3, 90)                       # <- runs beside the index
            if mode == "host_end":
                c.read_index_wait()
            assert g.n == g0.n and g.n > 0
            for f in ("first_inst", "freq", "to_ids", "from_ids", "has_v", "has_j"):
                np.testing.assert_array_equal(getattr(g, f), getattr(g0, f), err_msg=f)
            c.sam_names_load(names)
            valid, npairs = c.window_score(wins, 175)             # (device mode: this call ends the build)
            np.testing.assert_array_equal(valid, want[0][0])
            np.testing.assert_array_equal(npairs, want[0][1])
            offs, pairs = c.map_emit(contigs)
            np.testing.assert_array_equal(offs, want[1][0])
            for f in pairs.dtype.names:
                if f != "cls":
                    np.testing.assert_array_equal(pairs[f], want[1][1][f], err_msg=f)
            assert c.sam_text_device(contigs, [f"c{i}" for i in range(len(contigs))]) == want[2]
            assert c.stat("read_index_classes") > 0
        if it == 2:
            bad = pool.pair_id.copy()
            bad[7] = pool.n_pairs + 5
            c.read_index_build_begin(p, bad, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
            c.kmer_build(p, 35, 3, 90, export=False)
            with pytest.raises(VdjxError, match="pair id"):
                c.read_index_wait()
            with pytest.raises(VdjxError, match="read_index_build first"):
                c.window_score(wins, 175)
        p.free()
    c.close()


def test_read_index_begun_many_times_every_way_it_can_end():
    """300 begun index builds in one context over three pools: ended by _end after a k-mer build (the gate opened by the build), by _end
    at once (the join opens the gate), by the first scorer call after a build, and by the first scorer call with no build at all --
    no deadlock, and every time the window verdicts and pair counts of the first build of that pool"""
    from vdjer_amd import api, synth
    c = api.Context(0)
    rep = synth.make_repertoire(4, seed=3)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    c.anchor_sets_load(vc, jc)
    wins = [w for w in rep.windows() if w]
    pools = [synth.make_reads(rep, n, noise_frac=0.3, seed=10 + i) for i, n in enumerate((800, 3000, 12000))]
    ref = {}
    for it in range(300):
        pool = pools[it % 3]
        p = c.pool_load(pool.primary, pool.secondary, pool.rl)
        c.read_index_build_begin(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        mode = it % 4
        if mode == 0:
            c.kmer_build(p, 35, 3, 90, export=False)
            c.read_index_wait()
        elif mode == 1:
            c.read_index_wait()
        elif mode == 2:
            c.kmer_build(p, 35, 3, 90, export=False)
        valid, npairs = c.window_score(wins, 175)
        if it % 3 in ref:
            np.testing.assert_array_equal(ref[it % 3][0], valid, err_msg=str(it))
            np.testing.assert_array_equal(ref[it % 3][1], npairs, err_msg=str(it))
        else:
            ref[it % 3] = (valid.copy(), npairs.copy())
        p.free()
    c.close()
    assert sum(int(r[1].sum()) for r in ref.values()) > 1000


def test_read_index_table_survives_many_builds_and_changes_of_kind(monkeypatch):
    """The lookup table of the couples' index is not cleared per build: a slot counts as taken only if its claim word carries the build's
    number (7 bits), and the table is cleared when the numbers are used up (every 127 builds), when the buffer is new, and when the
    other kind of build (VDJX_NO_SYM_INDEX: 32-byte slots) has written it.  140 builds over two different pools in one context, the
    kinds mixed: every build's window verdicts and pair counts are those of a fresh context."""
    from vdjer_amd import api, synth
    pools, wins, want = [], [], []
    for seed in (311, 322):
        rep = synth.make_repertoire(4, seed=seed)
        pool = synth.make_reads(rep, 1500, noise_frac=0.25, seed=seed + 1, rl=50, err=0.004)
        pools.append(pool)
        wins.append([w for w in rep.windows() if w])
        fresh = api.Context(0)
        p = fresh.pool_load(pool.primary, pool.secondary, 50)
        fresh.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        want.append(fresh.window_score(wins[-1], 175))
        p.free()
        fresh.close()
    assert sum(int(w[1].sum()) for w in want) > 500
    ctx = api.Context(0)
    ps = [ctx.pool_load(pool.primary, pool.secondary, 50) for pool in pools]
    try:
        for it in range(140):
            which = it % 2 if it % 7 else 0
            plain = it in (5, 60, 61, 129)                      # the other kind of build in between
            if plain:
                monkeypatch.setenv("VDJX_NO_SYM_INDEX", "1")
            pool = pools[which]
            ctx.read_index_build(ps[which], pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
            if plain:
                monkeypatch.delenv("VDJX_NO_SYM_INDEX")
            assert ctx.stat("read_index_sym") == (0 if plain else 1)
            valid, npairs = ctx.window_score(wins[which], 175)
            np.testing.assert_array_equal(valid, want[which][0], err_msg=f"build {it}")
            np.testing.assert_array_equal(npairs, want[which][1], err_msg=f"build {it}")
    finally:
        for p in ps:
            p.free()
        ctx.close()


def test_root_scorer_begun_and_ended_equals_the_waiting_call():
    """vdjx_root_score_graph_begin / _end (queued on the stream, other scorer calls behind it, verdicts read at the end) against
    vdjx_root_score_graph: same ids, same verdicts -- also when the guessed item count falls short (a second, larger graph: the
    call is then repeated the ordinary way inside _end)"""
    from vdjer_amd import api, synth
    c = api.Context(0, pinned_results=True)
    rep = synth.make_repertoire(8, seed=101)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    c.anchor_sets_load(vc, jc)
    c.vregion_load([rep.v_region], 15)
    for n_pairs in (3000, 3000, 40000):                      # (the third graph has many more roots than the guess from the second)
        pool = synth.make_reads(rep, n_pairs, noise_frac=0.3, seed=102 + n_pairs)
        p = c.pool_load(pool.primary, pool.secondary, pool.rl)
        g = c.kmer_build(p, 35, 3, 90, keep_device=True)
        ids0, ok0 = (np.array(x) for x in c.root_score_graph(g, 30))
        ids1, ok1 = c.root_score_graph(g, 30, wait=False)
        wins = [w for w in rep.windows() if w]
        c.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        c.window_score(wins, 175)                           # (work queued behind the begun call)
        c.root_score_wait()
        assert ids0.shape[0] > 10 and np.array_equal(ids0, ids1) and np.array_equal(ok0, ok1) and 0 < int(ok0.sum()) < ok0.shape[0]
        g.free()
        p.free()
    c.close()


def test_kmer_build_between_root_begin_and_end_with_a_short_guess():
    """the header allows any call between vdjx_root_score_graph_begin and _end.  A k-mer build reads 8,208
    bytes of status into the context's page-locked scratch; the begun call's item count used to wait at byte 8,192 of it, was
    overwritten with 0, and _end then took a short guess for complete.  Second graph much larger than the first (the guess falls
    short), a build of a third pool in between: ids and verdicts must equal the waiting call's."""
    from vdjer_amd import api, synth
    c = api.Context(0, pinned_results=True)
    rep = synth.make_repertoire(8, seed=131)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    c.anchor_sets_load(vc, jc)
    c.vregion_load([rep.v_region], 15)
    other = synth.make_reads(rep, 5000, noise_frac=0.3, seed=77)
    po = c.pool_load(other.primary, other.secondary, other.rl)
    for n_pairs in (2000, 40000):
        pool = synth.make_reads(rep, n_pairs, noise_frac=0.3, seed=132 + n_pairs)
        p = c.pool_load(pool.primary, pool.secondary, pool.rl)
        g = c.kmer_build(p, 35, 3, 90, keep_device=True)
        ids0, ok0 = (np.array(x) for x in c.root_score_graph(g, 30))
        if n_pairs == 40000:
            c.root_score_graph(g, 30)          # (hint := this graph's item count) ... then a SMALL graph's call shrinks it again below
        pool_s = synth.make_reads(rep, 1500, noise_frac=0.3, seed=5)
        ps = c.pool_load(pool_s.primary, pool_s.secondary, pool_s.rl)
        gs = c.kmer_build(ps, 35, 3, 90, keep_device=True)
        c.root_score_graph(gs, 30)             # the guess for the next call is now this small graph's count
        gs.free()
        ps.free()
        ids1, ok1 = c.root_score_graph(g, 30, wait=False)
        n_mid, _ = c.kmer_build(po, 35, 3, 90, export=False)          # <- the call in between
        c.root_score_wait()
        assert n_mid > 0 and ids0.shape[0] > 5
        assert np.array_equal(ids0, ids1) and np.array_equal(ok0, ok1), n_pairs
        g.free()
        p.free()
    po.free()
    c.close()


def test_all_gated_all_distinct_overflows_the_lds_table(ctx):
    """High-quality reads with 80 % noise: nearly every instance is gated and distinct, so buckets hold more
    distinct k-mers than one LDS table pass takes and the sub-pass split / restart path runs."""
    from vdjer_amd import synth
    rep = synth.make_repertoire(20, seed=61)
    pool = synth.make_reads(rep, 150000, noise_frac=0.8, seed=62, clean=True)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    hg = run_both(ctx, pool, vc, jc, 35, 2, 60)
    assert hg.pre_nodes > 6_000_000 and hg.n > 1000


@pytest.mark.parametrize("rl,k,mf,mq", [(64, 35, 2, 60), (36, 25, 2, 40), (75 - 11, 50, 2, 60), (40, 40, 2, 30),
                                        (65, 35, 2, 60), (75, 35, 3, 90), (100, 35, 3, 90), (100, 25, 2, 60), (151, 35, 3, 90), (151, 50, 2, 60),
                                        (160, 21, 2, 40), (96, 50, 2, 60), (128, 35, 2, 60), (129, 35, 2, 230), (150, 150 - 100, 1, 20)])
def test_other_read_lengths(ctx, rl, k, mf, mq):
    """rl != 50 (records of 2*rl+1 bytes): the short-read kernels up to 64 bases, including k == rl (one k-mer per record), and
    the long-read record format (words, vdjx_pool) up to the 160-base limit -- 75 / 100 / 151 bp libraries; the oracle and the
    reference take them as they are (bam_read.c:208)"""
    from vdjer_amd import synth
    rep = synth.make_repertoire(6, seed=71)
    pool = synth.make_reads(rep, 6000, noise_frac=0.3, seed=72, rl=rl, err=0.004, n_rate=0.002)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    run_both(ctx, pool, vc, jc, k, mf, mq)


@pytest.mark.parametrize("rl", [75, 100, 151])
def test_long_reads_forward_load_and_scorers(ctx, rl):
    """reads of more than 64 bases through the rest of the path: the forward-only load (reverse-complement records derived on the
    device) equals the full load, and the read index / window mapper / coverage test / mapped pairs equal the oracle's"""
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(4, seed=91)
    pool = synth.make_reads(rep, 12000, noise_frac=0.2, seed=92, rl=rl, ins_mean=max(175.0, rl + 40.0), err=0.003, n_rate=0.001)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ctx.anchor_sets_load(vc, jc)
    p1 = ctx.pool_load(pool.primary, pool.secondary, rl)
    g1 = ctx.kmer_build(p1, 35, 3, 90)
    p2 = ctx.pool_load_forward(pool.primary[0::2], pool.secondary[0::2], rl)
    g2 = ctx.kmer_build(p2, 35, 3, 90)
    for f in ("first_inst", "freq", "gated_count", "to_ids", "from_ids", "kmers"):
        np.testing.assert_array_equal(getattr(g1, f), getattr(g2, f))
    p2.free()
    ix = oracle.ReadIndex(pool)
    ctx.read_index_build(p1, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    wins = [w for w in rep.windows() if w] + [t[s:s + 486] for t in rep.clones for s in (0, 40, 117)]
    ins = int(max(175, rl + 40))
    valid, npairs = ctx.window_score(wins, ins)
    for i, w in enumerate(wins):
        pairs, starts = ix.quick_map(w)
        assert int(npairs[i]) == len(pairs), (i, rl)
        assert int(valid[i]) == ix.coverage_is_valid(starts, len(w), ins), (i, rl)
    contigs = [w[51:411] for w in wins]
    offs, pairs = ctx.map_emit(contigs)
    for i, cg in enumerate(contigs):
        op, _ = ix.quick_map(cg)
        mine = pairs[int(offs[i]):int(offs[i + 1])]
        assert mine.tobytes() == op.tobytes(), (i, rl)
    p1.free()


def test_scorers_other_read_length_and_window_geometry(ctx):
    """rl=36 reads, 300-base windows, shifted eval range and spans: mapper + coverage vs the oracle"""
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(4, seed=81)
    pool = synth.make_reads(rep, 20000, noise_frac=0.1, seed=82, rl=36, ins_mean=150.0)
    ix = oracle.ReadIndex(pool)
    p = _load_index(ctx, pool)
    wins = []
    for t in rep.clones:
        for s in (0, 33, 100, 250):
            wins.append(t[s:s + 300])
    for ins, e0, e1, rs, ms, fl in ((150, 30, 250, 25, 30, 1), (150, 40, 200, 20, 48, 2), (120, 30, 250, 25, 30, 1)):
        valid, npairs = ctx.window_score(wins, ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=fl)
        nvalid = 0
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs)
            exp = ix.coverage_is_valid(starts, len(w), ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=fl)
            assert int(valid[i]) == exp, (i, ins, e0, e1, rs, ms, fl)
            nvalid += exp
    offs, pairs = ctx.map_emit([w[:200] for w in wins])
    for i, w in enumerate(wins):
        op, _ = ix.quick_map(w[:200])
        mine = pairs[int(offs[i]):int(offs[i + 1])]
        assert [(int(q["pair_id"]), int(q["pos1"]), int(q["pos2"]), int(q["rec1"]), int(q["rec2"])) for q in mine] == \
               [(int(q["pair_id"]), int(q["pos1"]), int(q["pos2"]), int(q["rec1"]), int(q["rec2"])) for q in op]
    p.free()


def test_scorer_argument_errors(ctx):
    from vdjer_amd import api, synth
    rep = synth.make_repertoire(2, seed=91)
    pool = synth.make_reads(rep, 500, seed=92)
    p = _load_index(ctx, pool)
    with pytest.raises(api.VdjxError):
        ctx.window_score(["ACGT" * 10], 175)                 # shorter than a read
    with pytest.raises(api.VdjxError):
        ctx.window_score(["A" * 2000], 175)                  # longer than the supported window
    v, n = ctx.window_score(["N" * 486, "A" * 486], 175)     # nothing maps; never valid with floor 1
    assert v.tolist() == [0, 0] and n.tolist() == [0, 0]
    with pytest.raises(api.VdjxError):
        ctx.root_score(["A" * 35], 35, 30) if False else ctx.kmer_build(p, 51, 3, 90)   # k > 50
    p.free()


def test_sharded_build_real_rccl_world1(ctx, monkeypatch):
    """The RCCL code path of the driver (async all_gather, all_to_all_single, all_reduce on device tensors) with a
    real one-rank `nccl` process group: the most of the multi-GPU path one GPU can execute for real."""
    monkeypatch.setenv("VDJX_SHARD_SELF_COLLECTIVES", "1")      # (a lone rank would skip the collectives: here they are the point)
    import socket
    import torch
    import torch.distributed as dist
    from vdjer_amd import shard, synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rep = synth.make_repertoire(6, seed=43)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    pool = synth.make_reads(rep, 30000, noise_frac=0.3, seed=44, err=0.004, n_rate=0.002)
    ref = run_both(ctx, pool, vc, jc, 35, 3, 90)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    try:
        p = ctx.pool_load(pool.primary, pool.secondary, 50)
        drv = shard.ShardedHotPath(ctx, dist, dev)
        assert drv.comm.async_ok
        for _ in range(2):
            g = drv.kmer_build(p, 35, 3, 90)
            assert g.n == ref.n and g.pre_nodes == ref.pre_nodes
            np.testing.assert_array_equal(g.first_inst, ref.first_inst)
            np.testing.assert_array_equal(g.freq, ref.freq)
            np.testing.assert_array_equal(g.to_ids, ref.to_ids)
            np.testing.assert_array_equal(g.from_ids, ref.from_ids)
        assert drv.bytes_exchanged > 0
        p.free()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_window_scorer_ranks_as_threads(ctx, world):
    """The window scorer over a pool sharded BY PAIR (vdjer_amd/shard.py:window_score + the HIP halves vdjx_window_pairs / _fetch /
    vdjx_window_cover), `world` ranks as threads on this GPU: verdicts and pair counts equal the single-GPU scorer over the whole
    pool, which equals the oracle."""
    import threading
    import torch
    from oracle import oracle
    from tests.fake_dist import ThreadDist
    from tests.test_shard_gloo import _pair_share, _scorer_case
    from vdjer_amd import api, shard
    pool, wins = _scorer_case()
    p = ctx.pool_load(pool.primary, pool.secondary, pool.rl)
    ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    valid1, np1 = ctx.window_score(wins, 175)
    ix = oracle.ReadIndex(pool)
    for i, w in enumerate(wins):
        pairs, starts = ix.quick_map(w)
        assert int(np1[i]) == len(pairs) and int(valid1[i]) == ix.coverage_is_valid(starts, len(w), 175)
    assert 0 < int(valid1.sum()) < len(wins)
    dist, out, errs = ThreadDist(world), [None] * world, []

    def work(r):
        try:
            dist.set_rank(r)
            c = api.Context(0)
            mine = _pair_share(pool, r, world)
            pr = c.pool_load(mine.primary, mine.secondary, mine.rl)
            c.read_index_build(pr, mine.pair_id, mine.read_num, mine.is_rc, mine.reg_rank, mine.n_pairs)
            drv = shard.ShardedHotPath(c, dist, torch.device("cuda", 0))
            out[r] = drv.window_score(shard.HipScorerEngine(c, torch.device("cuda", 0), mine.rl), wins, 175)
            pr.free()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            dist.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for valid, npairs in out:
        np.testing.assert_array_equal(valid, valid1)
        np.testing.assert_array_equal(npairs, np1)
    p.free()


def test_pool_load_forward_derives_the_reverse_complement_records(ctx):
    """vdjx_pool_load_forward: the reads as extracted only; the device derives every read's reverse-complement record
    (add_to_buffer, bam_read.c:231-243).  Same packed pool as the full buffers: same graph, same mapped pairs."""
    c = G.Case("e2e_mixed")
    ctx.anchor_sets_load(c.v_codes, c.j_codes)
    pool = c.pool
    full = ctx.pool_load(pool.primary, pool.secondary, pool.rl)
    g1 = ctx.kmer_build(full, 35, 3, 90)
    fwd = ctx.pool_load_forward(pool.primary[0::2], pool.secondary[0::2], pool.rl)
    assert fwd.n_records == full.n_records
    g2 = ctx.kmer_build(fwd, 35, 3, 90)
    for f in ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids", "kmers"):
        np.testing.assert_array_equal(getattr(g1, f), getattr(g2, f))
    assert g1.pre_nodes == g2.pre_nodes and g1.n > 100
    contigs = [l for l in G.text("e2e_mixed.contigs.fa.gz").split("\n") if l and not l.startswith(">")]
    res = []
    for p in (full, fwd):
        ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
        offs, pairs = ctx.map_emit(contigs)
        res.append((offs.copy(), pairs.copy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[0][1].tobytes() == res[1][1].tobytes() and res[0][1].shape[0] > 100
    full.free()
    fwd.free()


def test_iupac_bases_are_counted_and_read_as_N(ctx):
    """bases other than ACGTN (the BAM alphabet has them, bam_read.c:52) are packed as N: k-mers over them are dropped like the
    reference drops k-mers over N, and the count is visible (vdjx_stat "pool_other_bases")"""
    c = G.Case("noisy")
    ctx.anchor_sets_load(c.v_codes, c.j_codes)
    pri = c.pool.primary.copy()
    as_n = pri.copy()
    rows = np.arange(0, pri.shape[0], 7)
    pri[rows, 20] = ord("R")
    as_n[rows, 20] = ord("N")
    p1 = ctx.pool_load(pri, c.pool.secondary, c.pool.rl)
    assert ctx.stat("pool_other_bases") == rows.shape[0]
    g1 = ctx.kmer_build(p1, 35, 3, 90)
    p2 = ctx.pool_load(as_n, c.pool.secondary, c.pool.rl)
    assert ctx.stat("pool_other_bases") == 0
    g2 = ctx.kmer_build(p2, 35, 3, 90)
    for f in ("first_inst", "freq", "gated_count", "to_ids", "from_ids", "kmers"):
        np.testing.assert_array_equal(getattr(g1, f), getattr(g2, f))
    p1.free()
    p2.free()


def test_window_scorer_without_grouping_in_a_fresh_process():
    """k_group_pairs takes groups of windows that share read classes; windows it leaves (unrelated ones, long ones) and every
    window under VDJX_WINDOW_GROUP=0 go through k_window_pairs with its slices of deep windows.  The knob is read once per
    process: the window-scorer tests of this file and of the randomised suite run again with grouping off."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("VDJX_WINDOW_GROUP") == "0":
        pytest.skip("already the ungrouped run")
    env = dict(os.environ, VDJX_WINDOW_GROUP="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        "-m", "gpu", "-x", "-q", "-k", "(window or scorers or deep_windows) and not in_slices"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_window_scorer_in_slices_in_a_fresh_process():
    """The pair lists of a vdjx_window_score call are sized by its windows' hits; what does not fit the device is done in slices of
    windows (400 k windows over 10 M pairs would ask for 600 GB).  VDJX_WP_BUDGET_MB=1 (read once per process) puts the scorer tests'
    small calls through the slices: same verdicts, same pair counts."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("VDJX_WP_BUDGET_MB"):
        pytest.skip("already the sliced run")
    env = dict(os.environ, VDJX_WP_BUDGET_MB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        "-m", "gpu", "-x", "-q", "-k", "(window_score or scorers or deep_windows or grouped_window) and not fresh_process"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_grouped_window_mapper_edge_cases(ctx):
    """k_group_pairs takes eight windows at a time: identical windows in one group (every class shared), a single window, a
    group that is not full, windows nothing maps to, windows in which a read class repeats (tandem repeats: the slow walk of the
    occurrence lists), unrelated random windows (more distinct classes than a group's table: left to k_window_pairs) and
    windows longer than the grouped kernel takes -- pair counts and verdicts against the oracle, and the statistics say which
    kernel did the work."""
    from oracle import oracle
    from vdjer_amd import synth
    import os
    grouping = os.environ.get("VDJX_WINDOW_GROUP") != "0"        # (this test runs again with grouping off: test_window_scorer_without_grouping_...)
    rng = np.random.default_rng(77)
    rep = synth.make_repertoire(9, seed=4321)
    # a clone with a tandem repeat inside its window: reads of the repeat unit recur at several offsets of the same window
    unit = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 25))
    rep_clone = rep.clones[0][:150] + unit * 8 + rep.clones[0][350:]
    rep2 = synth.Repertoire(rep.v_germ, rep.j_germ, rep.clones + [rep_clone], rep.clone_v + [0], rep.clone_j + [0], np.full(10, 0.1), 99, j_codon=rep.j_codon)
    both = synth.tile_reads(rep2, range(10), ins=175, copies=2, step=2)
    ix = oracle.ReadIndex(both)
    p = _load_index(ctx, both)
    base = [w for w in rep.windows() if w]
    rand = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, 486)) for _ in range(16)]
    cases = {
        "one": base[:1],
        "same8": [base[0]] * 8,
        "nine": (base * 3)[:9],
        "repeat": [rep_clone[100:586], rep_clone[120:606], base[0], rep_clone[100:586]],
        "nothing": rand[:5],
        "mixed": base + rand + [rep_clone[100:586]] + [t[s:s + 486] for t in rep.clones for s in (3, 21)],
    }
    for name, wins in cases.items():
        valid, npairs = ctx.window_score(wins, 175)
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs), (name, i)
            assert int(valid[i]) == ix.coverage_is_valid(starts, len(w), 175), (name, i)
        if name == "same8" and grouping:
            assert ctx.stat("group_overflows") == 0 and ctx.stat("group_classes") <= 436         # eight windows, one set of classes
    # eight windows of eight different clones, every offset of each one a read class of its own: 3,000 distinct classes do not fit a
    # group's table (2,048), the group is left to k_window_pairs
    for name, wins in (("eight_clones", [t[20:506] for t in rep.clones[:8]]),
                       ("shifted", [t[s:s + 486] for s in (0, 5, 11, 17, 23, 29, 37, 41) for t in rep.clones])):
        valid, npairs = ctx.window_score(wins, 175)
        if name == "eight_clones" and grouping:
            assert ctx.stat("group_overflows") == 1
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs), (name, i)
            assert int(valid[i]) == ix.coverage_is_valid(starts, len(w), 175), (name, i)
    # longer than the grouped kernel takes (more than 512 offsets): k_window_pairs alone
    long_w = [t[:600] for t in rep.clones]
    valid, npairs = ctx.window_score(long_w, 175, e1=500)
    for i, w in enumerate(long_w):
        pairs, starts = ix.quick_map(w)
        assert int(npairs[i]) == len(pairs), ("long", i)
    p.free()


@pytest.mark.parametrize("rl,n_pairs", [(100, 200_000), (151, 120_000)])
def test_long_reads_at_the_histogram_cap(ctx, rl, n_pairs):
    """Enough long reads for the gated histogram to want its full resolution: the long-read kernels stage 5.6 KB of records per
    wave beside it, and 2^15 counters did not fit (every build beyond about a million pairs of 2 x 100 bp failed with "invalid
    argument"; the pools of test_other_read_lengths are too small to get there).  2^14 counters and a second cut of the buckets
    now: graph against the oracle."""
    from vdjer_amd import synth
    rep = synth.make_repertoire(n_pairs // 500, seed=171)
    pool = synth.make_reads_cb(rep, n_pairs, noise_frac=0.3, rl=rl, seed=172)
    assert pool.n_records * (rl - 35 + 1) // 3072 > (1 << 14)            # the histogram is capped
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    hg = run_both(ctx, pool, vc, jc, 35, 3, 90)
    assert hg.n > 10_000


def test_calls_that_grow_and_shrink(ctx):
    """The root DP is launched for the LAST call's item count and the mapped pairs are gathered into the LAST call's buffer before the
    host knows this call's numbers: a small call, a much larger one (the guess is short: a second launch / a new buffer), the small one
    again (the guess is long), each against the oracle; and the contigs picked out of window rows in page-locked memory."""
    from oracle import oracle
    from vdjer_amd import synth
    rep = synth.make_repertoire(6, seed=77)
    lines = [rep.v_region]
    s = oracle.RootScorer(lines, 15)
    ctx.vregion_load(lines, 15)
    rng = np.random.default_rng(8)
    qs = []
    for _ in range(2000):
        st = int(rng.integers(0, len(rep.v_region) - 35))
        q = list(rep.v_region[st:st + 35])
        for _m in range(int(rng.integers(0, 6))):
            q[int(rng.integers(0, 35))] = "ACGT"[int(rng.integers(0, 4))]
        qs.append("".join(q))
    exp = [s.score(q, 30) for q in qs]
    for sel in (slice(0, 3), slice(0, 2000), slice(5, 9), slice(0, 700), slice(0, 0), slice(0, 2000)):
        assert ctx.root_score(qs[sel], 35, 30).tolist() == exp[sel]
        if qs[sel]:
            assert ctx.stat("root_dp_items") > 0
    pool = synth.tile_reads(rep, list(range(6)), ins=175, copies=6, step=3)
    ix = oracle.ReadIndex(pool)
    p = _load_index(ctx, pool)
    wins = [w for w in rep.windows() if w]
    rows = np.frombuffer("".join(wins).encode(), np.uint8).reshape(len(wins), -1)
    contigs = [w[51:411] for w in wins]
    want = [ix.quick_map(c)[0] for c in contigs]
    for idx in ([0], list(range(len(contigs))), [2, 1], [], list(range(len(contigs))), [1]):
        packed = ctx.pin_rows_take(rows, np.array(idx, dtype=np.int64), "contigs_test", 51, 360)
        assert packed[1] == len(idx) and packed[2] == 360
        offs, got = ctx.map_emit(packed)
        assert int(offs[-1]) == sum(len(want[j]) for j in idx)
        for n_, j in enumerate(idx):
            mine = got[int(offs[n_]):int(offs[n_ + 1])]
            assert mine.shape[0] == len(want[j]) > 0
            for fld in ("pair_id", "rec1", "rec2", "pos1", "pos2", "insert", "rc1", "rc2"):
                assert np.array_equal(mine[fld].astype(np.int64), want[j][fld].astype(np.int64)), (idx, j, fld)
    p.free()


@pytest.mark.parametrize("rl", [65, 75, 100, 130, 151, 160])
def test_long_reads_resident_records(ctx, rl):
    """vdjx_pool_load_device on reads of more than 64 bases (records resident on the device, packed four characters at a time) gives
    the pool of vdjx_pool_load (character by character) on the same records -- N, other letters, low qualities and every tail
    length included: same graph, same count of other bases"""
    import torch
    from vdjer_amd import synth
    rep = synth.make_repertoire(3, seed=191 + rl)
    pool = synth.make_reads(rep, 9000, noise_frac=0.2, seed=192 + rl, rl=rl, ins_mean=max(175.0, rl + 40.0), err=0.003, n_rate=0.002)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ctx.anchor_sets_load(vc, jc)
    pri, sec = pool.primary.copy(), pool.secondary.copy()
    rng = np.random.default_rng(rl)
    for a in (pri, sec):                                     # letters the BAM alphabet has besides ACGTN, at every column of the read
        rows = rng.integers(0, a.shape[0], 300)
        a[rows, 1 + rng.integers(0, rl, 300)] = np.frombuffer(b"RYKMSWacgtn.", np.uint8)[rng.integers(0, 12, 300)]
        rows = rng.integers(0, a.shape[0], 600)
        a[rows, 1 + rl + rng.integers(0, rl, 600)] = rng.integers(33, 54, 600).astype(np.uint8)      # Phred 0..20 around the gate
    p1 = ctx.pool_load(pri, sec, rl)
    other = ctx.stat("pool_other_bases")
    assert other > 0
    g1 = ctx.kmer_build(p1, 35, 3, 90)
    # (device records must be 16-byte aligned and stay alive while the pool reads their quality characters)
    d_pri, d_sec = torch.from_numpy(pri).cuda(), torch.from_numpy(sec).cuda()
    p2 = ctx.pool_load_device(d_pri.data_ptr(), d_pri.shape[0], d_sec.data_ptr(), d_sec.shape[0], rl)
    assert ctx.stat("pool_other_bases") == other
    g2 = ctx.kmer_build(p2, 35, 3, 90)
    assert g1.n == g2.n > 100 and g1.pre_nodes == g2.pre_nodes
    for f in ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids", "kmers"):
        np.testing.assert_array_equal(getattr(g1, f), getattr(g2, f))
    p1.free()
    p2.free()


def test_pool_load_device_holds_the_buffers_it_points_into():
    """vdjx_pool_load_device does not copy the quality characters (include/vdjx.h): the Python wrapper keeps the tensors it is
    given alive until Pool.free(), so a caller that drops its own references -- and then lets torch's caching allocator hand the
    block to somebody else -- still gets the reference's quality sums (api.py pool_load_device).  mq 230 makes every
    low-count k-mer's verdict depend on those sums (A2:454-465)."""
    import gc

    import torch
    from oracle import oracle
    from vdjer_amd import api, synth
    rep = synth.make_repertoire(3, seed=5)
    pool = synth.make_reads(rep, 3000, noise_frac=0.3, seed=9)
    ctx = api.Context(0)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    ctx.anchor_sets_load(vc, jc)
    d_pri, d_sec = torch.from_numpy(pool.primary).cuda(), torch.from_numpy(pool.secondary).cuda()
    p = ctx.pool_load_device(d_pri, d_secondary=d_sec)
    assert p.rl == pool.rl and p.n_records == pool.primary.shape[0] + pool.secondary.shape[0]
    del d_pri, d_sec
    gc.collect()
    junk = [torch.full((pool.primary.shape[0], 2 * pool.rl + 1), 33, dtype=torch.uint8, device="cuda") for _ in range(4)]      # would land in the freed blocks
    torch.cuda.synchronize()
    g = ctx.kmer_build(p, 35, 2, 230)
    t = oracle.KmerTable(pool, 35)
    t.prune(2, 230)
    og = oracle.Graph(t, vc, jc)
    assert g.n == og.n > 0
    np.testing.assert_array_equal(g.first_inst, og.first)
    np.testing.assert_array_equal(g.freq, og.freq)
    assert p._src is not None
    p.free()
    assert p._src is None
    del junk
    ctx.close()
