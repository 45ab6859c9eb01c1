"""Randomised differential test of the k-mer build (a-1 ... a-3) against the oracle: read length, k, mf, mq, repertoire size, pool
size, noise, error and N rates drawn at random, some pools with a third of their records duplicated or with scrambled qualities.
The first version of this file (run by hand at the end of round 2) found what the fixed-case tests did not: for reads of more than
32 offsets (2*(rl-k-o) reaches 64) the 128-bit shift by a per-lane amount in the dense-listing kernels gave a few wrong k-mers in
ten thousand, depending on timing (vdjx_common.h: vdjx_kmer_at_lane)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_SCALE = int(os.environ.get("VDJX_FUZZ_SCALE", "1"))          # more configurations per run (and other seeds: VDJX_FUZZ_SEED)
_SEED = int(os.environ.get("VDJX_FUZZ_SEED", "0"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _draw(rng, mode):
    rl = int(rng.integers(65, 161)) if mode == "long" else int(rng.choice([36, 40, 50, 50, 50, 64]))      # long: the word-per-32-bases record format
    k = int(rng.integers(8, min(50, rl) + 1))
    if mode == "gt32" and rl - k + 1 <= 32:
        k = max(8, rl - 32 - int(rng.integers(0, 8)))
    if mode == "le32" and rl - k + 1 > 32:
        k = min(50, rl - 31 + int(rng.integers(0, 10)))
    return dict(rl=rl, k=k, mf=int(rng.integers(1, 5)), mq=int(rng.choice([20, 40, 60, 90, 120, 214, 254])),
                clones=int(rng.integers(1, 30)), pairs=int(rng.integers(200, 12000)), noise=float(rng.choice([0.0, 0.1, 0.3, 0.6])),
                err=float(rng.choice([0.0, 0.002, 0.01, 0.03])), n_rate=float(rng.choice([0.0, 0.002, 0.02])),
                seed=int(rng.integers(0, 1 << 30)), dup=bool(rng.random() < 0.3), scramble=bool(rng.random() < 0.3))


@pytest.fixture(scope="module")
def ctx():
    from vdjer_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("mode,seed,n", [("gt32", 101, 40), ("le32", 102, 40), ("any", 103, 40), ("long", 104, 30)])
def test_kmer_build_random_configurations_vs_oracle(ctx, mode, seed, n):
    from vdjer_amd import synth
    import test_gpu_parity as T
    rng = np.random.default_rng(seed + _SEED)
    for it in range(n * _SCALE):
        c = _draw(rng, mode)
        rep = synth.make_repertoire(c["clones"], seed=c["seed"])
        pool = synth.make_reads(rep, c["pairs"], noise_frac=c["noise"], seed=c["seed"] + 1, rl=c["rl"], err=c["err"], n_rate=c["n_rate"])
        if c["dup"]:                        # a third of the records twice (identical reads: the distinct-read rule, A2:349-352)
            m = pool.primary.shape[0] // 3
            pool.primary[m:2 * m] = pool.primary[:m]
        if c["scramble"]:                   # qualities around the Phred-20 gate
            rows = rng.integers(0, pool.primary.shape[0], 50)
            pool.primary[rows, 1 + c["rl"]:1 + 2 * c["rl"]] = rng.integers(33, 75, (50, c["rl"]), dtype=np.uint8)
        vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
        jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
        try:
            T.run_both(ctx, pool, vc, jc, c["k"], c["mf"], c["mq"])
        except Exception as e:
            raise AssertionError(f"configuration {it} of ({mode}, seed {seed}): {c}") from e


@pytest.mark.parametrize("seed,n", [(201, 12)])
def test_scorers_random_configurations_vs_oracle(ctx, seed, n):
    """a-8 / a-9 / a-10 with random read length, window length, insert size, evaluation range, spans and floor, window starts all
    over the transcripts (and reverse complements, and windows nothing maps to), contig slices of random length: mapped pairs in the
    reference's order, counts and verdicts against the oracle."""
    from oracle import oracle
    from vdjer_amd import synth
    import test_gpu_parity as T
    rng = np.random.default_rng(seed + _SEED)
    for it in range(n * _SCALE):
        rl = int(rng.choice([36, 50, 50, 64]))
        wlen = int(rng.integers(rl + 120, 640))
        ins = int(rng.integers(120, 260))
        # every other configuration: many clones over two V and one J gene (windows of different clones share most of their read
        # classes, the grouped window mapper's case); else a few unrelated clones
        if it % 2:
            rep = synth.make_repertoire(int(rng.integers(6, 20)), seed=int(rng.integers(0, 1 << 30)), n_v=2, n_j=1)
        else:
            rep = synth.make_repertoire(int(rng.integers(1, 8)), seed=int(rng.integers(0, 1 << 30)))
        pool = synth.make_reads(rep, int(rng.integers(500, 9000)), noise_frac=float(rng.choice([0.0, 0.2])), seed=int(rng.integers(0, 1 << 30)),
                                rl=rl, ins_mean=float(ins), err=float(rng.choice([0.0, 0.004])))
        ix = oracle.ReadIndex(pool)
        p = T._load_index(ctx, pool)
        wins = []
        for t in rep.clones:
            for _ in range(4):
                st = int(rng.integers(0, max(1, len(t) - wlen)))
                w = t[st:st + wlen]
                if len(w) == wlen:
                    wins.append(w if rng.random() < 0.8 else synth.revcomp(w))
        wins.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, wlen)))
        e0 = int(rng.integers(1, wlen // 3))
        e1 = int(rng.integers(e0 + 20, wlen - rl))
        rs, ms, fl = int(rng.integers(10, rl)), int(rng.integers(10, 80)), int(rng.integers(0, 4))
        cfg = dict(it=it, rl=rl, wlen=wlen, ins=ins, e0=e0, e1=e1, rs=rs, ms=ms, fl=fl, n_wins=len(wins))
        valid, npairs = ctx.window_score(wins, ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=fl)
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs), (cfg, i)
            exp = 1 if fl == 0 else ix.coverage_is_valid(starts, len(w), ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=fl)
            assert int(valid[i]) == exp, (cfg, i)
        clen = int(rng.integers(rl + 20, wlen))
        c0 = int(rng.integers(0, wlen - clen + 1))
        contigs = [w[c0:c0 + clen] for w in wins]
        offs, got = ctx.map_emit(contigs)
        for j, c in enumerate(contigs):
            pairs, _ = ix.quick_map(c)
            mine = got[int(offs[j]):int(offs[j + 1])]
            assert mine.shape[0] == len(pairs), (cfg, j)
            for fld in ("pair_id", "rec1", "rec2", "pos1", "pos2", "insert", "rc1", "rc2"):
                assert np.array_equal(mine[fld].astype(np.int64), pairs[fld].astype(np.int64)), (cfg, j, fld)
        p.free()


@pytest.mark.parametrize("seed,n", [(251, 20)])
def test_coverage_floor_one_random_configurations_vs_oracle(ctx, seed, n):
    """a-9 at the reference's default floor (params.c:64), the bit-grid form of k_window_cover (and, where the mate span passes 64
    deltas, its fall-back to the difference arrays): random read length incl. 75 / 100 / 151 bases, window length, insert size,
    evaluation range and spans over deep pools, so that valid AND invalid windows occur; verdicts and pair counts vs the oracle's
    coverage_is_valid (coverage.c:10-130)."""
    from oracle import oracle
    from vdjer_amd import synth
    import test_gpu_parity as T
    rng = np.random.default_rng(seed + _SEED)
    seen = {0: 0, 1: 0}
    for it in range(n * _SCALE):
        rl = int(rng.choice([36, 50, 50, 75, 100, 151]))
        wlen = int(rng.integers(rl + 160, rl + 520))
        ins = int(rng.integers(rl + 60, rl + 220))
        rep = synth.make_repertoire(int(rng.integers(1, 5)), seed=int(rng.integers(0, 1 << 30)))
        pool = synth.make_reads(rep, int(rng.choice([300, 600, 1200, 2500, 6000])), noise_frac=0.0, seed=int(rng.integers(0, 1 << 30)), rl=rl, ins_mean=float(ins),
                                err=float(rng.choice([0.0, 0.002])))
        ix = oracle.ReadIndex(pool)
        p = T._load_index(ctx, pool)
        wins = []
        for t in rep.clones:
            for _ in range(6):
                st = int(rng.integers(0, max(1, len(t) - wlen)))
                w = t[st:st + wlen]
                if len(w) == wlen:
                    wins.append(w if rng.random() < 0.8 else synth.revcomp(w))
        if not wins:
            p.free()
            continue
        # evaluation ranges from a few positions to most of the window; mate spans on both sides of the 64 deltas one word holds
        e0 = int(rng.integers(1, max(2, wlen // 2)))
        e1 = int(rng.integers(e0 + 5, max(e0 + 6, wlen - rl // 2)))
        rs = int(rng.integers(10, rl))
        ms = int(rng.choice([10, 30, 48, 48, 63, 64, 65, 90]))
        cfg = dict(it=it, rl=rl, wlen=wlen, ins=ins, e0=e0, e1=e1, rs=rs, ms=ms, n_wins=len(wins))
        valid, npairs = ctx.window_score(wins, ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=1)
        for i, w in enumerate(wins):
            pairs, starts = ix.quick_map(w)
            assert int(npairs[i]) == len(pairs), (cfg, i)
            exp = ix.coverage_is_valid(starts, len(w), ins, e0=e0, e1=e1, rs=rs, ms=ms, floor=1)
            assert int(valid[i]) == exp, (cfg, i)
            seen[int(exp)] += 1
        p.free()
    assert seen[0] and seen[1], seen          # (both verdicts occurred)


@pytest.mark.parametrize("seed,n", [(301, 10)])
def test_sharded_build_random_configurations_vs_single_gpu(ctx, seed, n):
    """The sharded driver with 2 or 4 ranks as threads on this GPU (tests/fake_dist.py) on random configurations (including reads of
    more than 32 offsets, duplicated records, k up to 50): the graph of every rank == the single-GPU build of the union pool (which
    run_both also checks against the oracle)."""
    import threading
    import torch
    from fake_dist import ThreadDist
    from vdjer_amd import api, shard, synth
    import test_gpu_parity as T
    rng = np.random.default_rng(seed + _SEED)
    for it in range(n * _SCALE):
        c = _draw(rng, "any" if it % 2 else "gt32")
        world = int(rng.choice([2, 4]))
        rl = c["rl"]
        rep = synth.make_repertoire(c["clones"], seed=c["seed"])
        per = max(200, c["pairs"] // world)
        pools = [synth.make_reads(rep, per, noise_frac=c["noise"], seed=c["seed"] + 1 + r, rl=rl, err=c["err"], n_rate=c["n_rate"]) for r in range(world)]
        if c["dup"]:                        # rank 1 holds rank 0's reads again: k-mers whose reads differ only across ranks
            m = min(pools[0].primary.shape[0], pools[1].primary.shape[0]) // 2
            pools[1].primary[:m] = pools[0].primary[:m]
        vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
        jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
        nrec = [p_.primary.shape[0] + p_.secondary.shape[0] for p_ in pools]
        stride = max(nrec)
        blank = np.frombuffer(("0" + "N" * rl + "I" * rl).encode(), np.uint8)
        parts = []
        for r, p_ in enumerate(pools):      # the union in global numbering: rank r's records start at r * stride
            rows = np.concatenate([p_.primary, p_.secondary])
            pad = np.stack([blank] * (stride - rows.shape[0])) if stride > rows.shape[0] else np.zeros((0, 2 * rl + 1), np.uint8)
            parts += [rows, pad]
        cat = np.concatenate(parts)
        R = cat.shape[0]
        union = synth.ReadPool(rl, cat, np.zeros((0, 2 * rl + 1), np.uint8), np.zeros(R, np.uint32), np.zeros(R, np.uint8),
                               np.zeros(R, np.uint8), np.arange(R, dtype=np.uint32), 0)
        cfg = dict(c, world=world, it=it)
        try:
            ref = T.run_both(ctx, union, vc, jc, c["k"], c["mf"], c["mq"])
        except Exception as e:
            raise AssertionError(f"single-GPU build of the union: {cfg}") from e
        dist = ThreadDist(world)
        out, errs = [None] * world, []

        def work(r):
            try:
                dist.set_rank(r)
                cx = api.Context(0)
                cx.anchor_sets_load(vc, jc)
                p = cx.pool_load(pools[r].primary, pools[r].secondary, rl)
                out[r] = shard.ShardedHotPath(cx, dist, torch.device("cuda", 0), stride=stride).kmer_build(p, c["k"], c["mf"], c["mq"])
                p.free()
                cx.close()
            except Exception as e:  # noqa: BLE001
                errs.append(e)
                dist.barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, (cfg, errs)
        for g in out:
            assert (g.n, g.pre_nodes) == (ref.n, ref.pre_nodes), cfg
            for f in ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids", "kmers"):
                assert np.array_equal(getattr(g, f), getattr(ref, f)), (cfg, f)


REF_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "vdjer_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/vdjer_ref (the reference's own sources, compiled by oracle/Makefile) is not built")
@pytest.mark.parametrize("seed,n", [(401, 5)])
def test_cli_end_to_end_random_pools_vs_the_compiled_reference(seed, n, tmp_path):
    """The whole command line against THE REFERENCE ITSELF (oracle/_ref/vdjer_ref: its own sources compiled where they lie, exactly
    how the goldens were made, tests/golden/make_golden.py) on pools drawn here: tiled clones (they pass the coverage test) or noisy
    reads, k / mf / mq / mrs at random.  vdj_contigs.fa, the SAM on stdout and vdjer.dot byte for byte.  A run of the reference
    counts when it scored every root (its root threads race otherwise, as in make_golden.py).  Every pool also goes through
    `vdjer --gpus N` (N = 2..4 process-ranks on the one device): the sharded path from C against the reference's bytes."""
    import re
    import subprocess
    from vdjer_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "vdjer_amd", "vdjer")
    rng = np.random.default_rng(seed + _SEED)
    with_contigs = 0
    for it in range(n * _SCALE):
        n_clones = int(rng.integers(2, 7))
        chain = str(rng.choice(["IGH", "IGH", "IGK", "IGL"]))
        rep = synth.make_repertoire(n_clones, seed=int(rng.integers(0, 1 << 30)), chain=chain)
        if rng.random() < 0.5:
            pool = synth.tile_reads(rep, list(range(int(rng.integers(1, n_clones + 1)))), copies=int(rng.integers(2, 5)))
        else:
            pool = synth.make_reads(rep, int(rng.integers(3000, 9000)), noise_frac=float(rng.choice([0.1, 0.3])), seed=int(rng.integers(0, 1 << 30)))
        k = int(rng.choice([25, 31, 35]))
        flags = ["--k", str(k), "--mf", str(int(rng.integers(2, 4))), "--mq", str(int(rng.choice([60, 90]))), "--mrs", str(int(rng.choice([20, 30])))]
        cfg = dict(it=it, chain=chain, clones=n_clones, pairs=pool.n_pairs, flags=flags)
        outs = {}
        gpus = int(rng.integers(2, 5))          # the same pool through `vdjer --gpus N` as well (N process-ranks sharing the one device: host transport)
        for who, binary in (("ref", [REF_BIN, "run"]), ("hip", [exe]), ("hipN", [exe])):
            d = tmp_path / f"{it}_{who}"
            d.mkdir()
            pool.write_reads_file(str(d / "reads.txt"))
            synth.write_ref_dir(rep, str(d / "ref"))
            cmd = binary + ["--in", "reads.txt", "--chain", chain, "--ref-dir", "ref", "--ins", "175", "--t", "1"] + flags + (["--gpus", str(gpus)] if who == "hipN" else [])
            for attempt in range(8):
                r = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                                   env=dict(os.environ, VDJX_MGPU_ONE_DEVICE="1", VDJX_MGPU_TIMEOUT_S="120") if who == "hipN" else None)
                err = r.stderr.decode(errors="replace")
                if who != "ref":
                    assert r.returncode == 0, (cfg, who, gpus, err[-2000:])
                    break
                m1, m2 = re.search(r"num root nodes: (\d+)", err), re.search(r"HARNESS_ROOTS_SCORED\t(\d+)", err)
                if m1 and m2 and m1.group(1) == m2.group(1):
                    break
            else:
                pytest.skip(f"the reference never scored all its roots: {cfg}")
            outs[who] = ((d / "vdj_contigs.fa").read_bytes() if (d / "vdj_contigs.fa").exists() else b"", r.stdout,
                         (d / "vdjer.dot").read_bytes() if (d / "vdjer.dot").exists() else b"")
        assert outs["hip"][0] == outs["ref"][0], (cfg, "vdj_contigs.fa differs")
        assert outs["hip"][1] == outs["ref"][1], (cfg, "SAM differs")
        assert outs["hip"][2] == outs["ref"][2], (cfg, "vdjer.dot differs")
        assert outs["hipN"] == outs["ref"], (cfg, f"--gpus {gpus} differs from the reference")
        with_contigs += 1 if outs["ref"][0].count(b">") else 0
    assert with_contigs >= 1          # at least one configuration assembled contigs
