#!/usr/bin/env python3
"""vdjx_bucket_mix (vdjx_common.h) against vdjx_mix, restated in numpy: bucket occupancies of the distinct k-mers of a clone-like
read set (k = 25, 35, 50; 2^10 / 2^15 / 2^20 buckets: variance / mean = 1 for a Poisson fill) and the avalanche of the 32 result
bits under every single-bit change of the key.  CPU only, ~1 min."""
import numpy as np

np.seterr(over="ignore")
U = np.uint64


def mix(lo, hi):
    x = lo ^ (hi * U(0x9E3779B97F4A7C15)) ^ U(0x2545F4914F6CDD1D)
    for _ in range(2):
        x ^= x >> U(32)
        x = x * U(0xD6E8FEB86659FD93)
    return x ^ (x >> U(32))


def bucket_mix(lo, hi):
    a = (lo & U(0xFFFFFFFF)).astype(np.uint32)
    b = (lo >> U(32)).astype(np.uint32)
    c = (hi & U(0xFFFFFFFF)).astype(np.uint32) ^ ((hi >> U(32)).astype(np.uint32) << np.uint32(26))
    h = (c ^ (c >> np.uint32(15))) * np.uint32(0x9E3779B1)
    h ^= h >> np.uint32(15)
    h = (h ^ a) * np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h = (h ^ b ^ (b >> np.uint32(16))) * np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    h = h * np.uint32(0x27D4EB2F)
    h ^= h >> np.uint32(15)
    return h.astype(U) << U(32)


def kmers(rng, n, k, rl):
    T = rng.integers(0, 4, size=(200, 400), dtype=np.uint8)
    idx, st = rng.integers(0, 200, size=n), rng.integers(0, 400 - rl, size=n)
    reads = np.stack([T[i, s:s + rl] for i, s in zip(idx, st)])
    reads = np.where(rng.random(reads.shape) < 0.01, rng.integers(0, 4, size=reads.shape, dtype=np.uint8), reads)
    lo, hi = [], []
    for o in range(rl - k + 1):
        vl, vh = np.zeros(n, dtype=U), np.zeros(n, dtype=U)
        for j in range(k):
            vh = (vh << U(2)) | (vl >> U(62))
            vl = (vl << U(2)) | reads[:, o + j].astype(U)
        lo.append(vl)
        hi.append(vh & U((1 << max(0, 2 * k - 64)) - 1))
    u = np.unique(np.stack([np.concatenate(lo), np.concatenate(hi)], 1), axis=0)
    return u[:, 0].copy(), u[:, 1].copy()


def main():
    rng = np.random.default_rng(1)
    for k in (25, 35, 50):
        lo, hi = kmers(rng, 200000, k, 50 if k < 50 else 64)
        for name, f in (("vdjx_mix", mix), ("vdjx_bucket_mix", bucket_mix)):
            h = f(lo, hi)
            row = []
            for nb in (10, 15, 20):
                c = np.bincount((h >> U(64 - nb)).astype(np.int64), minlength=1 << nb)
                row.append(f"2^{nb}: max {c.max()} var/mean {c.var() / c.mean():.3f}")
            print(f"k {k} {len(lo)} distinct  {name:16s} " + "  ".join(row))
    lo = rng.integers(0, 1 << 63, size=100000, dtype=U)
    hi = rng.integers(0, 1 << 36, size=100000, dtype=U)
    for name, f in (("vdjx_mix", mix), ("vdjx_bucket_mix", bucket_mix)):
        h0, worst = f(lo, hi), (1.0, 0.0)
        for bit in range(100):
            h1 = f(lo ^ U(1 << bit), hi) if bit < 64 else f(lo, hi ^ U(1 << (bit - 64)))
            d = (h0 ^ h1) >> U(32)
            p = np.array([((d >> U(j)) & U(1)).mean() for j in range(12, 32)])        # (the buckets use the 20 leading bits)
            worst = (min(worst[0], p.min()), max(worst[1], p.max()))
        print(f"{name}: a key bit flips a leading result bit with probability {worst[0]:.3f} .. {worst[1]:.3f}")


if __name__ == "__main__":
    main()
