"""The counter-based workload generator (vdjer_amd/synth.py:make_reads_cb): integer-only tensor arithmetic pinned against
Python integers, sub-range consistency (what lets the GPU box regenerate exactly the pool the oracle digests were taken
from), record layout, and a committed digest of a small pool so that any change of the stream is noticed on the CPU."""
import hashlib

import numpy as np
import torch

from vdjer_amd import synth


def test_splitmix_tensor_arithmetic_matches_python_integers():
    xs = [0, 1, 2 ** 63 - 1, -1, -2 ** 63, 123456789123456789, -987654321987654321]
    got = [int(v) & synth._M64 for v in synth._sm64(torch.tensor(xs, dtype=torch.int64))]
    assert got == [synth._sm64_py(x) for x in xs]
    assert synth._sm64_py(0) == 0 and synth._sm64_py(1) == 0x5692161D100B05E5       # splitmix64 finalizer known answer


def test_subrange_equals_the_same_pairs_of_a_larger_pool():
    rep = synth.make_repertoire(50, seed=3)
    big = synth.make_reads_cb(rep, 6000, seed=11)
    sub = synth.make_reads_cb(rep, 1000, seed=11, pair0=5000)
    npri = big.primary.shape[0]
    in_p = np.isin(big.pair_id[:npri], np.arange(5000, 6000))
    in_s = np.isin(big.pair_id[npri:], np.arange(5000, 6000))
    assert np.array_equal(big.primary[in_p], sub.primary) and np.array_equal(big.secondary[in_s], sub.secondary)
    small_chunks = synth.make_reads_cb(rep, 6000, seed=11, chunk=777)
    assert np.array_equal(big.primary, small_chunks.primary) and np.array_equal(big.secondary, small_chunks.secondary)


def test_record_layout_and_read_model():
    rep = synth.make_repertoire(40, seed=5)
    pool = synth.make_reads_cb(rep, 20000, seed=9)
    rl = pool.rl
    a = np.concatenate([pool.primary, pool.secondary])
    assert a.shape[1] == 2 * rl + 1 and (a[:, 0] == ord("0")).all() and a.shape[0] == 4 * pool.n_pairs
    seq, qual = a[:, 1:1 + rl], a[:, 1 + rl:]
    assert set(np.unique(seq)) <= set(b"ACGTN") and set(np.unique(qual)) <= {40 + 33, 30 + 33, 12 + 33}
    # record 2i+1 is the reverse complement of record 2i with reversed qualities (bam_read.c:206-244)
    comp = np.arange(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    assert np.array_equal(comp[seq[0::2][:, ::-1]], seq[1::2]) and np.array_equal(qual[0::2][:, ::-1], qual[1::2])
    assert 0.0005 < (seq == ord("N")).mean() < 0.002 and 0.03 < (qual == 45).mean() < 0.07
    assert 0.25 < pool.secondary.shape[0] / a.shape[0] < 0.35
    # clone reads come from the transcripts (most are error-free)
    hits = 0
    for r in pool.primary[:400:4]:
        s = r[1:1 + rl].tobytes().decode()
        hits += any(s in t or synth.revcomp(s) in t for t in rep.clones)
    assert hits > 70
    # per-record read info as add_read_info registers it
    assert np.array_equal(pool.read_num[:8], [1, 1, 2, 2, 1, 1, 2, 2]) and np.array_equal(pool.is_rc[:4], [0, 1, 0, 1])
    assert np.array_equal(np.sort(pool.reg_rank), np.arange(a.shape[0]))


def test_stream_is_pinned():
    rep = synth.make_repertoire(30, seed=20261002)
    pool = synth.make_reads_cb(rep, 3000, seed=20261002)
    h = hashlib.sha256(pool.primary.tobytes() + pool.secondary.tobytes()).hexdigest()
    assert h == PINNED, h


PINNED = "edd474b44ffc7e213ddec6652616fb46bc2cf2aa9325ec9a81dabab31afdb068"
