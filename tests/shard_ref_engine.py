"""Pure-Python engine with the same phase interface as vdjer_amd.shard.HipShardEngine -- TEST INFRASTRUCTURE.

Lets the multi-rank driver (vdjer_amd/shard.py: the all_gather / all_to_all / all_reduce choreography and
all the split arithmetic) run under `gloo` on CPU, world_size 2, where there is no GPU.  The per-k-mer
semantics restate SURVEY Appendix A.1 / A.5 with dictionaries; small inputs only.
"""
import zlib

import numpy as np
import torch

from vdjer_amd import synth
from vdjer_amd.api import Graph

CODE = {"A": 0, "T": 1, "C": 2, "G": 3}
GATED = 1 << 40


class RefShardEngine:
    def __init__(self, v_codes, j_codes):
        self.vs, self.js = set(int(x) for x in v_codes), set(int(x) for x in j_codes)

    def begin(self, pool, k, mf, mq, rank, world, stride):
        self.pool, self.k, self.mf, self.mq = pool, k, mf, min(mq, 254)
        self.rank, self.world, self.stride = rank, world, stride
        self.rl = pool.rl
        self.P = self.rl - k + 1
        self.obits = world.bit_length() - 1
        self.local = np.concatenate([pool.primary, pool.secondary], axis=0)

    def _owner(self, key: bytes) -> int:
        return (zlib.crc32(key) >> (32 - self.obits)) if self.obits else 0

    def pool_export(self):
        return [torch.from_numpy(self.local.copy())]

    def set_pool(self, glob):
        self.glob = glob[0].numpy()

    def _instances(self):
        rl, k = self.rl, self.k
        for r in range(self.local.shape[0]):
            seq = self.local[r, 1:1 + rl].tobytes()
            q = self.local[r, 1 + rl:1 + 2 * rl]
            for o in range(self.P):
                km = seq[o:o + k]
                if any(c not in b"ACGT" for c in km):
                    continue
                gated = all(((int(x) - 33) & 0xFF) >= 20 for x in q[o:o + k])
                yield km, ((self.rank * self.stride + r) * self.P + o) | (GATED if gated else 0)

    def partition_count(self):
        self.out = [[] for _ in range(self.world)]
        for km, inst in self._instances():
            self.out[self._owner(km)].append((km, inst))
        return np.array([len(x) for x in self.out], dtype=np.int64)

    def partition_fill(self, n):
        flat = [x for part in self.out for x in part]
        keys = np.frombuffer(b"".join(x[0] for x in flat), dtype=np.uint8).reshape(-1, self.k).copy() if flat else np.zeros((0, self.k), np.uint8)
        inst = np.array([x[1] for x in flat], dtype=np.int64)
        return [torch.from_numpy(keys), torch.from_numpy(inst)]

    def recv_like(self, n):
        return [torch.empty((n, self.k), dtype=torch.uint8), torch.empty(n, dtype=torch.int64)]

    def reduce(self, recv):
        keys, insts = recv[0].numpy(), recv[1].numpy()
        rl, k, P = self.rl, self.k, self.P
        tab = {}
        for i in range(keys.shape[0]):
            tab.setdefault(keys[i].tobytes(), []).append(int(insts[i]))
        self.surv = []
        ndist = 0
        for km, lst in tab.items():
            gated = sorted(x & (GATED - 1) for x in lst if x & GATED)
            if not gated:
                continue
            ndist += 1
            first = gated[0]
            frec = self.glob[first // P]
            fseq = frec[1:1 + rl].tobytes()
            S = [(int(frec[1 + rl + j]) - 33) & 0xFF for j in range(k)]          # A2:337-339
            multi = False
            for x in gated[1:]:
                rec = self.glob[x // P]
                off = x % P
                if rec[1:1 + rl].tobytes() != fseq:
                    multi = True
                for j in range(k):
                    qv = (int(rec[1 + rl + off + j]) - 33) & 0xFF
                    S[j] = S[j] + qv if S[j] + qv < 214 else 255               # A2:354-361
            cnt = min(len(gated), 32765)
            if cnt >= self.mf and multi and all(s >= self.mq for s in S):
                every = sorted(x & (GATED - 1) for x in lst)
                self.surv.append((km, cnt, first, min(len(every), 32765), every[0]))
        return len(self.surv), ndist

    def survivors(self, ns):
        w = self.k + 16
        out = np.zeros((ns, w), np.uint8)
        for i, (km, gc, gf, uc, uf) in enumerate(self.surv):
            out[i, :self.k] = np.frombuffer(km, np.uint8)
            out[i, self.k:] = np.array([gc, gf, uc, uf], dtype="<u4").view(np.uint8)
        return torch.from_numpy(out)

    def edges(self, surv_all):
        a = surv_all.numpy()
        self.all = [(a[i, :self.k].tobytes(), *[int(x) for x in a[i, self.k:].view("<u4")]) for i in range(a.shape[0])]
        idx = {s[0]: i for i, s in enumerate(self.all)}
        n = len(self.all)
        ef = np.full(n * 4, -1, np.int32)
        et = np.full(n * 4, -1, np.int32)
        rl, k = self.rl, self.k
        for r in range(self.local.shape[0]):
            seq = self.local[r, 1:1 + rl].tobytes()
            prev = -1
            for o in range(self.P):
                cur = idx.get(seq[o:o + k], -1)
                if cur >= 0 and prev >= 0:
                    e = prev * 4 + CODE[chr(seq[o + k - 1])]
                    inst = (self.rank * self.stride + r) * self.P + o
                    if ef[e] == -1 or inst < ef[e]:
                        ef[e] = inst
                    et[e] = cur
                prev = cur
        return torch.from_numpy(ef), torch.from_numpy(et)

    def finish(self, ef, et, pre_total):
        ef, et = ef.numpy(), et.numpy()
        n, k, P = len(self.all), self.k, self.P
        order = sorted(range(n), key=lambda i: self.all[i][4])
        rank = {s: r for r, s in enumerate(order)}
        g = Graph(k, n, pre_total, np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8),
                  np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32), np.zeros(n, np.uint8),
                  np.zeros((n, 4), np.uint32), np.zeros((n, k), np.uint8))
        edges = []
        for r, s in enumerate(order):
            km, gc, gf, uc, uf = self.all[s]
            g.first_inst[r] = (uf // P) * 64 + uf % P
            g.gated_count[r], g.freq[r] = gc, uc
            g.kmers[r] = np.frombuffer(km, np.uint8)
            if k > 16:
                code = synth.seq_to_int(km[:16].decode())
                g.has_v[r], g.has_j[r] = int(code in self.vs and code != 0), int(code in self.js and code != 0)
            else:
                g.has_v[r] = g.has_j[r] = 1
            for b in range(4):
                if ef[s * 4 + b] != -1:
                    edges.append((int(ef[s * 4 + b]), r, rank[int(et[s * 4 + b])]))
        for first, u, v in sorted(edges, reverse=True):
            g.to_ids[u, g.to_deg[u]] = v + 1
            g.to_deg[u] += 1
            g.from_ids[v, g.from_deg[v]] = u + 1
            g.from_deg[v] += 1
        return g

    def end(self):
        pass
