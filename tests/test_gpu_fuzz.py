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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _draw(rng, mode):
    rl = int(rng.choice([36, 40, 50, 50, 50, 64]))
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


@pytest.mark.parametrize("mode,seed,n", [("gt32", 101, 40), ("le32", 102, 40), ("any", 103, 40)])
def test_kmer_build_random_configurations_vs_oracle(ctx, mode, seed, n):
    from vdjer_amd import synth
    import test_gpu_parity as T
    rng = np.random.default_rng(seed)
    for it in range(n):
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
