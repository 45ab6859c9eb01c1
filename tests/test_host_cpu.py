"""CPU tests of the serial host stage (vdjer_amd/csrc/host: sparsehash-order emulation, condensation, contig
enumeration, window discovery, acceptance, overlap removal, FASTA/SAM/dot text) against dumps of the compiled
reference.  The scorers behind the stage are the CPU oracle here (the GPU tests run the same stage behind libvdjx)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle
from tests import golden_util as G
from vdjer_amd import api, host


def test_dense_table_iteration_order_vs_sparsehash():
    """insert / erase / resize(0) sequences on dense_hash_map<const char*, ..., contig_hash> (360-char keys)"""
    L = host.lib()
    t = host.SphTable()
    L.sph_init(C.byref(t), 360, 0)
    keep = []
    got = []
    for op in G.text("order_ops.txt.gz").splitlines():
        if op[0] == "I":
            kb = C.create_string_buffer(op[2:].encode())
            keep.append(kb)
            L.sph_map_put(C.byref(t), C.cast(kb, C.c_char_p), None, None)
        elif op[0] == "E":
            L.sph_erase(C.byref(t), op[2:].encode())
        elif op[0] == "R":
            L.sph_resize0(C.byref(t))
        elif op[0] == "D":
            got.append(f"D\t{L.sph_size(C.byref(t))}\t{t.nbuckets}")
            b = L.sph_next(C.byref(t), 0)
            while b < t.nbuckets:
                got.append(t.b[b].key.decode())
                b = L.sph_next(C.byref(t), b + 1)
    assert got == G.text("order_out.txt.gz").splitlines()
    assert L.sph_murmur64a(b"ACGTACGTACGTACGTACGTACGTACGTACGTACG", 35, 97) == 5958308921596863335


def _golden_graph(tag, k):
    rows = G.rows(f"{tag}.nodes.tsv.gz")
    n = len(rows)
    g = api.Graph(k, n, 0, np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8),
                  np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32), np.zeros(n, np.uint8),
                  np.zeros((n, 4), np.uint32), np.zeros((n, k), np.uint8))
    for i, r in enumerate(rows):
        g.kmers[i] = np.frombuffer(r[1].encode(), np.uint8)
        g.freq[i], g.has_v[i], g.has_j[i] = int(r[2]), int(r[3]), int(r[4])
        to = [int(x) for x in r[5].split(",") if x]
        fr = [int(x) for x in r[6].split(",") if x]
        g.to_deg[i], g.from_deg[i] = len(to), len(fr)
        g.to_ids[i, :len(to)] = to
        g.from_ids[i, :len(fr)] = fr
    return g


@pytest.mark.parametrize("tag,k", [("noisy_k35", 35), ("noisy_k25", 25), ("noisy_mq230", 35), ("pre_k35", 35)])
def test_nodes_table_order(tag, k):
    g = _golden_graph(tag, k)
    assert host.node_order(g).tolist() == [int(r[0]) for r in G.rows(f"{tag}.node_order.tsv.gz")]


def test_window_discovery_vs_reference():
    z = np.load(os.path.join(G.GOLD, "vjf_codes.npz"))
    p = host.make_params("IGH", ins=175)
    contigs = G.text("vjf_contigs.txt.gz").split()
    exp, cur = [], None
    for l in G.text("vjf_out.txt.gz").splitlines():
        f = l.split("\t")
        if f[0] == "C":
            cur = []
            exp.append(cur)
        else:
            cur.append((f[0], f[1]))
    assert len(exp) == len(contigs)
    for c, e in zip(contigs, exp):
        assert host.vjf_search(p, c, z["v_codes"], z["j_codes"]) == e
    assert sum(len(e) for e in exp) >= 5


def _oracle_hooks(pool, v_region, p):
    sc = oracle.RootScorer([v_region], p.vregion_kmer_size)
    ix = oracle.ReadIndex(pool)

    def root_score(km, k, thr):
        return [sc.score(km[i].tobytes().decode(), thr) for i in range(km.shape[0])]

    def window_score(wins):
        out = []
        for w in wins:
            pairs, starts = ix.quick_map(w)
            out.append(1 if p.read_filter_floor == 0 else ix.coverage_is_valid(
                starts, len(w), p.insert_len, e0=p.eval_start, e1=p.eval_stop, rs=p.filter_read_span, ms=p.filter_mate_span,
                floor=p.read_filter_floor))
        return out

    def sam_body(ids, contigs):
        txt = []
        for cid, c in zip(ids, contigs):
            pairs, _ = ix.quick_map(c)
            for q in pairs:
                txt.append(ix.sam_pair(cid, f"r{q['pair_id']}", q))
        return "".join(txt)

    return root_score, window_score, sam_body


def _e2e_info(tag):
    m = G.manifest()
    return m["e2e"][tag] if tag in m["e2e"] else m["e2e_chains"][tag]


@pytest.mark.parametrize("threads", [1, 5])
@pytest.mark.parametrize("tag", ["e2e_tiled", "e2e_mixed", "e2e_k25", "e2e_igk", "e2e_igl"])
def test_host_stage_end_to_end_vs_reference(tag, threads, tmp_path):
    """graph (from the oracle) -> host stage -> vdj_contigs.fa, SAM and vdjer.dot byte-identical to the reference's
    (e2e_igk / e2e_igl: the light-chain presets of set_chain_info, params.c:20-31)"""
    c = G.Case(tag)
    info = _e2e_info(tag)
    fl = G.flags_to_params(info["flags"])
    # --t: the roots are enumerated by `threads` workers and merged in dispatch order: same bytes whatever the thread count
    p = host.make_params(info.get("chain", "IGH"), ins=175, t=threads, k=fl["k"], mf=fl["mf"], mq=fl["mq"], mcs=fl["mcs"], mrs=fl["mrs"], rl=c.pool.rl)
    t = oracle.KmerTable(c.pool, fl["k"])
    t.prune(fl["mf"], fl["mq"])
    og = oracle.Graph(t, c.v_codes, c.j_codes)
    g = api.Graph(fl["k"], og.n, 0, og.first, np.zeros(og.n, np.uint32), og.freq, og.has_v, og.has_j, og.to_deg, og.to_ids,
                  og.from_deg, og.from_ids,
                  np.frombuffer("".join(oracle.inst_kmer(c.pool, int(f), fl["k"]) for f in og.first).encode(), np.uint8).reshape(og.n, fl["k"]))
    fa, dot, sam = tmp_path / "c.fa", tmp_path / "g.dot", tmp_path / "o.sam"
    marks = []
    st = host.assemble(p, g, *_oracle_hooks(c.pool, c.v_region, p), c.v_codes, c.j_codes, str(fa), str(dot), str(sam), status=marks.append)
    assert st["n_roots"] == info["roots"]
    # the stage markers assemble() prints between the graph build and FINIS, in the reference's order (A2:1417-1464)
    import json
    ref = json.load(open(os.path.join(G.GOLD, "stage_markers.json")))["markers"]
    assert marks == ref[ref.index("POST_GRAPH_BLOCK"):ref.index("FINIS")]
    assert fa.read_text() == G.text(f"{tag}.contigs.fa.gz")
    assert dot.read_text() == G.text(f"{tag}.dot.gz")
    assert sam.read_text() == G.text(f"{tag}.sam.gz")
    assert st["n_contigs_out"] == info["contigs"]


def test_host_stage_midscale_vs_reference_digests(tmp_path):
    """The serial host stage at MID scale -- 400,000 pairs / 800 clones: 4,381 roots, 989 contig candidates, 319 candidate windows,
    55 contigs, the order-defining tables grown several times (A2:775-914 overlap removal with erase-during-iteration, vjf_windows_temp
    A2:806, print_windows vj_filter.c:209-309) -- behind the oracle's graph and scorers: FASTA, SAM, vdjer.dot and the per-root verdict
    log byte-identical (by digest) to complete --t 1 runs of the compiled reference (tests/golden/midscale.json)."""
    from tests import midscale_util as M
    from vdjer_amd import synth
    case = M.cases()["mid_400k"]
    rep = synth.make_repertoire(case["clones"], seed=case["seed"])
    pool = synth.make_reads_cb(rep, case["pairs"], noise_frac=case["noise"], seed=case["seed"] + 13)
    vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
    jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
    fl = G.flags_to_params(case["flags"])
    t = oracle.KmerTable(pool, fl["k"])
    t.prune(fl["mf"], fl["mq"])
    og = oracle.Graph(t, vc, jc)
    g = api.Graph(fl["k"], og.n, 0, og.first, np.zeros(og.n, np.uint32), og.freq, og.has_v, og.has_j, og.to_deg, og.to_ids, og.from_deg, og.from_ids,
                  np.frombuffer("".join(oracle.inst_kmer(pool, int(f), fl["k"]) for f in og.first).encode(), np.uint8).reshape(og.n, fl["k"]))
    p = host.make_params(case["chain"], ins=case["ins"], t=4, k=fl["k"], mf=fl["mf"], mq=fl["mq"], mcs=fl["mcs"], mrs=fl["mrs"], rl=pool.rl)
    fa, dot, sam, rlog = tmp_path / "c.fa", tmp_path / "g.dot", tmp_path / "o.sam", tmp_path / "roots.log"
    os.environ["VDJH_ROOT_LOG"] = str(rlog)
    try:
        st = host.assemble(p, g, *_oracle_hooks(pool, rep.v_region, p), vc, jc, str(fa), str(dot), str(sam))
    finally:
        del os.environ["VDJH_ROOT_LOG"]
    assert (st["n_roots"], st["n_roots_accepted"], st["n_contigs_out"]) == (case["roots"], case["roots_accepted"], case["contigs"])
    assert M.digest_file(str(rlog)) == case["root_log"]
    assert M.digest_file(str(fa)) == case["fasta"]
    assert M.digest_file(str(dot)) == case["dot"]
    assert M.digest_file(str(sam)) == case["sam"]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- `vdjer --gpus N` without GPUs: the exchange arithmetic, the slices the ranks read, and what happens when ranks fail ----
def _host_lib():
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "vdjer_amd", "libvdjhost.so"))

    class Step(C.Structure):
        _fields_ = [("round", C.c_uint32), ("peer", C.c_int32), ("send_off", C.c_uint64), ("send_len", C.c_uint64),
                    ("recv_off", C.c_uint64), ("recv_len", C.c_uint64)]
    L.vdjx_a2a_plan.restype = C.c_size_t
    L.vdjx_a2a_plan.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]
    return L, Step


def test_a2a_plan_moves_every_byte_once_in_bounded_pieces():
    """vdjx_a2a_plan (vdjx_mgpu.c's all-to-all-v): simulated with G ranks in plain Python -- every rank's plan executed against every
    other's, sends matched with receives round by round -- the receive buffers must come out as the concatenation by source rank,
    no piece above the chunk, self copies right.  Sizes around the chunk boundary and empty rows included."""
    import ctypes as C
    L, Step = _host_lib()
    rng = np.random.default_rng(7)
    for trial in range(30):
        G = int(rng.choice([1, 2, 3, 4, 8]))
        row = int(rng.choice([1, 4, 8, 32, 200]))
        chunk = int(rng.choice([64, 100, 256, 4096]))
        rows = rng.integers(0, 40, size=(G, G)).astype(np.uint64)          # rows[s][d]: rows rank s holds for rank d
        if trial % 5 == 0:
            rows[rng.integers(0, G)] = 0
        send = [rng.integers(0, 256, size=int(rows[s].sum()) * row, dtype=np.uint8) for s in range(G)]
        recv = [np.full(int(rows[:, d].sum()) * row, 255, np.uint8) for d in range(G)]
        plans = []
        for me in range(G):
            sr, rr = np.ascontiguousarray(rows[me]), np.ascontiguousarray(rows[:, me])
            self3 = (C.c_uint64 * 3)()
            n = L.vdjx_a2a_plan(G, me, sr.ctypes.data, rr.ctypes.data, row, chunk, None, 0, self3)
            st = (Step * max(n, 1))()
            assert L.vdjx_a2a_plan(G, me, sr.ctypes.data, rr.ctypes.data, row, chunk, st, n, self3) == n
            plans.append([st[i] for i in range(n)])
            if self3[2]:
                recv[me][self3[1]:self3[1] + self3[2]] = send[me][self3[0]:self3[0] + self3[2]]
        for me in range(G):
            for s_ in plans[me]:
                assert s_.peer != me and s_.send_len <= chunk and s_.recv_len <= chunk
                if s_.send_len:        # the matching receive of the peer in the same round
                    m_ = [q for q in plans[s_.peer] if q.round == s_.round and q.peer == me]
                    assert len(m_) == 1 and m_[0].recv_len == s_.send_len
                    recv[s_.peer][m_[0].recv_off:m_[0].recv_off + s_.send_len] = send[me][s_.send_off:s_.send_off + s_.send_len]
        for d in range(G):
            exp, at = [], [0] * G
            for s in range(G):
                off = int(rows[s][:d].sum()) * row
                exp.append(send[s][off:off + int(rows[s][d]) * row])
            np.testing.assert_array_equal(recv[d], np.concatenate(exp) if exp else recv[d])


def _tiny_cli_inputs(d, n_pairs=300, rl=50):
    from vdjer_amd import synth
    rep = synth.make_repertoire(3, seed=5)
    pool = synth.make_reads(rep, n_pairs, noise_frac=0.3, seed=6, rl=rl)
    pool.write_reads_file(os.path.join(d, "reads.txt"))
    synth.write_ref_dir(rep, os.path.join(d, "ref"))
    return pool


def _dump_share(exe, cwd, rk, nr, inp="reads.txt"):
    import subprocess
    r = subprocess.run([exe, "--in", inp, "--chain", "IGH", "--ref-dir", "ref", "--ins", "175"], cwd=cwd,
                       env=dict(os.environ, VDJX_DUMP_SHARE=f"{rk},{nr}"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    lines = r.stderr.decode().splitlines()
    head = next(l for l in lines if l.startswith("share\t")).split("\t")
    recs = [l.split("\t")[1:] for l in lines if l.startswith("rec\t")]
    return {"rl": int(head[4]), "records": int(head[6]), "total": int(head[8]), "pairs": int(head[10]), "recs": recs, "bytes": r.stdout}


def test_ranks_keep_their_share_of_the_pool_by_pair(tmp_path):
    """rank r of `vdjer --gpus N` keeps the pairs whose read name hashes to it -- both mates, all four records -- and knows every
    record's place in the scan order of the WHOLE pool (primary pool, then secondary: A2:1388-1390) and its registration rank over the
    whole pool (the order of the add_read_info calls, bam_read.c:228,243).  The shares of all ranks, each record put back at its
    place, are the pool the one-GPU run loads; a share is in ascending scan order; no pair is split."""
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    if not os.path.exists(exe):
        pytest.skip("vdjer is not built")
    pool = _tiny_cli_inputs(str(tmp_path))
    whole = np.concatenate([pool.primary, pool.secondary])
    R, rec = whole.shape
    one = _dump_share(exe, tmp_path, 0, 1)
    assert one["records"] == one["total"] == R and one["bytes"] == whole.tobytes()
    reg_one = {int(x[0]): int(x[1]) for x in one["recs"]}
    for nr in (2, 3, 4):
        got = np.zeros_like(whole)
        seen = np.zeros(R, bool)
        owner = {}
        sizes = []
        for rk in range(nr):
            sh = _dump_share(exe, tmp_path, rk, nr)
            assert sh["total"] == R and sh["rl"] == pool.rl and len(sh["recs"]) == sh["records"]
            rows = np.frombuffer(sh["bytes"], np.uint8).reshape(-1, rec)
            scan = np.array([int(x[0]) for x in sh["recs"]])
            assert np.all(np.diff(scan) > 0)                         # ascending: what goes to a rank's slice is one run of the share
            assert not seen[scan].any()
            seen[scan] = True
            got[scan] = rows
            for x in sh["recs"]:
                assert reg_one[int(x[0])] == int(x[1])               # the registration rank is the whole pool's, not the share's
                assert owner.setdefault(x[5], rk) == rk              # a read name lives on one rank
            assert len({x[5] for x in sh["recs"]}) == sh["pairs"]
            sizes.append(sh["records"])
        assert seen.all() and np.array_equal(got, whole)
        assert max(sizes) < 1.5 * R / nr + 40                        # shares are about 1/N of the pool


def test_ranks_keep_their_share_of_a_bam(tmp_path):
    """the same through the BAM route (bamx_extract_filtered): every rank runs the extraction's passes and stores only its share"""
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    if not os.path.exists(exe):
        pytest.skip("vdjer is not built")
    from tests import golden_util as G
    from tests.test_cli_e2e import _bam_inputs, _write_inputs
    c = G.Case("e2e_mixed")
    _write_inputs(c, str(tmp_path))
    _bam_inputs(c, str(tmp_path))
    one = _dump_share(exe, tmp_path, 0, 1, "in.bam")
    R = one["total"]
    rec = 2 * one["rl"] + 1
    assert one["records"] == R > 0
    whole = np.frombuffer(one["bytes"], np.uint8).reshape(R, rec)
    reg_one = {int(x[0]): int(x[1]) for x in one["recs"]}
    got = np.zeros_like(whole)
    n = 0
    for rk in range(3):
        sh = _dump_share(exe, tmp_path, rk, 3, "in.bam")
        scan = np.array([int(x[0]) for x in sh["recs"]], dtype=np.int64)
        assert np.all(np.diff(scan) > 0) and all(reg_one[int(x[0])] == int(x[1]) for x in sh["recs"])
        got[scan] = np.frombuffer(sh["bytes"], np.uint8).reshape(-1, rec)
        n += sh["records"]
    assert n == R and np.array_equal(got, whole)


def test_multi_gpu_cli_fails_fast_and_leaves_no_rank_behind(tmp_path):
    """no GPU here: every rank of `vdjer --gpus 4` fails at vdjx_init.  The run must end non-zero within seconds and none of the
    forked ranks may survive it (round-2 advice: a failing rank left the others blocked in a collective, orphaned, holding their GPUs)"""
    import subprocess
    import time
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("needs a box without GPUs")
    except ImportError:
        pass
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    if not os.path.exists(exe):
        pytest.skip("vdjer is not built")
    _tiny_cli_inputs(str(tmp_path))
    t0 = time.time()
    pr = subprocess.Popen([exe, "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--gpus", "4"], cwd=tmp_path,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = pr.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        os.killpg(pr.pid, 9)
        raise AssertionError("vdjer --gpus 4 hung without GPUs")
    assert pr.returncode != 0 and time.time() - t0 < 100
    time.sleep(0.5)
    left = subprocess.run(["ps", "-o", "pid=", "-g", str(pr.pid)], stdout=subprocess.PIPE, text=True).stdout.split()
    assert not left, f"ranks left behind: {left}"
