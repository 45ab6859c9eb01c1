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
NONE = 0xFFFFFFFF
NONE64 = -1            # all-ones as int64


class RefShardEngine:
    def __init__(self, v_codes, j_codes):
        self.vs, self.js = set(int(x) for x in v_codes), set(int(x) for x in j_codes)

    def begin(self, pool, k, mf, mq, rank, world, stride):
        self.pool, self.k, self.mf, self.mq = pool, k, mf, min(mq, 254)
        self.rank, self.world, self.stride = rank, world, stride
        self.rl = pool.rl
        self.P = self.rl - k + 1
        self.recs = np.concatenate([pool.primary, pool.secondary], axis=0)

    def _owner(self, key: bytes) -> int:
        return (zlib.crc32(key) * self.world) >> 32                 # any number of ranks

    def _instances(self):
        rl, k = self.rl, self.k
        for r in range(self.recs.shape[0]):
            seq = self.recs[r, 1:1 + rl].tobytes()
            q = self.recs[r, 1 + rl:1 + 2 * rl]
            for o in range(self.P):
                km = seq[o:o + k]
                if any(c not in b"ACGT" for c in km):
                    continue
                gated = all(((int(x) - 33) & 0xFF) >= 20 for x in q[o:o + k])
                yield km, ((self.rank * self.stride + r) << 6) | o, gated

    def _rec(self, inst):
        return self.recs[(inst >> 6) - self.rank * self.stride]

    # ---- phase 1: this rank's partial aggregates (A.1 restated per rank)
    def local(self):
        k, rl = self.k, self.rl
        self.tab = {}
        for km, inst, gated in self._instances():
            e = self.tab.setdefault(km, {"g": [], "a": []})
            e["a"].append(inst)
            if gated:
                e["g"].append(inst)
        self.sent = [[] for _ in range(self.world)]
        for km, e in self.tab.items():
            g = sorted(e["g"])
            fl = 0
            if g:
                fseq = self._rec(g[0])[1:1 + rl].tobytes()
                fl = int(any(self._rec(x)[1:1 + rl].tobytes() != fseq for x in g[1:]))
            if g:                   # only k-mers with a gated instance travel
                self.sent[self._owner(km)].append((km, min(len(g), 32765), g[0], fl))
        counts = np.array([len(x) for x in self.sent], dtype=np.int64)
        rows = np.zeros((int(counts.sum()), k + 24), np.uint8)
        for i, (km, cg, mg, fl) in enumerate(x for part in self.sent for x in part):
            rows[i, :k] = np.frombuffer(km, np.uint8)
            rows[i, k:] = np.array([cg, mg, fl], dtype="<u8").view(np.uint8)
        return counts, torch.from_numpy(counts.astype(np.int32)), torch.from_numpy(rows)

    # ---- phase 2 (owner): merge, decide, ask
    def merge(self, recv_dir, recv_parts, recv_counts):
        k = self.k
        rows = recv_parts.numpy()
        mqq = min(self.mq, 214)
        self.tlow = 1 + (mqq + 19) // 20
        merged = {}
        at = 0
        for src, n in enumerate(int(v) for v in recv_counts):
            for i in range(n):
                row = rows[at + i]
                cg, mg, fl = (int(x) for x in row[k:].copy().view("<u8"))
                m = merged.setdefault(row[:k].tobytes(), {"cg": 0, "mg": 2 ** 63, "fl": 0, "src": []})
                m["cg"] += cg
                m["mg"] = min(m["mg"], mg)
                m["fl"] |= fl
                m["src"].append((src, i))
            at += n
        self.surv, self.pend = [], []
        self.ndist = len(merged)
        self.q = [[] for _ in range(self.world)]
        for km, m in merged.items():
            if m["cg"] < max(self.mf, 2):
                continue
            need = 0
            if not m["fl"]:
                if len(m["src"]) < 2:
                    continue
                need |= 1
            if m["cg"] < self.tlow:
                need |= 2
            rec = (km, min(m["cg"], 32765), m["mg"], 0, NONE64)      # the recount comes with the edge pass
            if not need:
                self.surv.append(rec)
                continue
            pid = len(self.pend)
            self.pend.append({"rec": rec, "need": need, "cg": m["cg"], "mg": m["mg"], "fl": 0, "S": [0] * k, "seq0": None, "seqs": []})
            for src, i in m["src"]:
                self.q[src].append((i, pid, need))
        self.stats = {"open_flag": sum(1 for p_ in self.pend if p_["need"] & 1), "low_count": sum(1 for p_ in self.pend if p_["need"] & 2)}
        return np.array([len(x) for x in self.q], dtype=np.int64)

    def queries(self):
        flat = [x for part in self.q for x in part]
        return torch.from_numpy(np.array(flat, dtype="<u8").reshape(-1, 3).view(np.uint8).reshape(-1, 24).copy())

    # ---- phase 3 (every rank): answer with per-read data
    def reply(self, queries, counts):
        k, rl, P = self.k, self.rl, self.P
        qs = queries.numpy().copy().view("<u8").reshape(-1, 3)
        out = np.zeros((qs.shape[0], 24 + rl + 3 * k), np.uint8)
        at = 0
        for owner, n in enumerate(int(v) for v in counts):
            for j in range(n):
                i, pid, need = (int(x) for x in qs[at + j])
                km, cg, mg, fl = self.sent[owner][i]
                g = sorted(self.tab[km]["g"])
                frec = self._rec(mg)
                QS = [0] * k
                for x in g[1:]:
                    rec, off = self._rec(x), x & 63
                    for c in range(k):
                        QS[c] += (int(rec[1 + rl + off + c]) - 33) & 0xFF
                row = out[at + j]
                row[:24] = np.array([pid, need, mg], dtype="<u8").view(np.uint8)
                row[24:24 + rl] = frec[1:1 + rl]
                row[24 + rl:24 + rl + k] = [min(v, 255) for v in QS]
                row[24 + rl + k:24 + rl + 2 * k] = [(int(frec[1 + rl + (mg & 63) + c]) - 33) & 0xFF for c in range(k)]
                row[24 + rl + 2 * k:] = [(int(frec[1 + rl + c]) - 33) & 0xFF for c in range(k)]       # A2:337-339
            at += n
        return torch.from_numpy(out)

    # ---- phase 4 (owner)
    def resolve(self, replies):
        k, rl = self.k, self.rl
        rows = replies.numpy()
        mqq = min(self.mq, 214)
        for row in rows:
            pid, need, mg = (int(x) for x in row[:24].copy().view("<u8"))
            p = self.pend[pid]
            seq = row[24:24 + rl].tobytes()
            first = mg == p["mg"]
            if first:
                p["seq0"] = seq
            p["seqs"].append(seq)
            own = row[24 + rl + 2 * k:] if first else row[24 + rl + k:24 + rl + 2 * k]
            for c in range(k):
                p["S"][c] += int(row[24 + rl + c]) + int(own[c])
        self.stats["flag_set_by_answers"] = 0
        for p in self.pend:
            ok = True
            if p["need"] & 1:
                ok = any(sq != p["seq0"] for sq in p["seqs"])
                self.stats["flag_set_by_answers"] += int(ok)
            if ok and p["cg"] < self.tlow:
                ok = all(v >= mqq for v in p["S"])
            if ok and p["rec"][1] >= self.mf:
                self.surv.append(p["rec"])
        return len(self.surv), self.ndist

    def survivors(self, ns):
        w = self.k + 16
        out = np.zeros((ns, w), np.uint8)
        for i, (km, gc, gf, uc, uf) in enumerate(self.surv):
            out[i, :self.k] = np.frombuffer(km, np.uint8)
            out[i, self.k:] = np.array([gc, gf], dtype="<u8").view(np.uint8)
        return torch.from_numpy(out)

    def edges(self, surv_all):
        """this rank's share of add_to_graph (A2:261-320) for all survivors: in-edge first sights [4n] | node first sights [n]
        (int64, -1 = none) and instance counts [n]"""
        a = surv_all.numpy()
        self.all = [(a[i, :self.k].tobytes(), *[int(x) for x in a[i, self.k:].copy().view("<u8")]) for i in range(a.shape[0])]
        idx = {s[0]: i for i, s in enumerate(self.all)}
        n = len(self.all)
        inf = np.full(n * 4, -1, np.int64)
        rl, k = self.rl, self.k
        for r in range(self.recs.shape[0]):
            seq = self.recs[r, 1:1 + rl].tobytes()
            prev = -1
            for o in range(self.P):
                cur = idx.get(seq[o:o + k], -1)
                if cur >= 0 and prev >= 0:
                    e = cur * 4 + CODE[chr(seq[o - 1])]                 # (head, first base of the tail k-mer)
                    inst = ((self.rank * self.stride + r) << 6) | o
                    if inf[e] == -1 or inst < inf[e]:
                        inf[e] = inst
                prev = cur
        ucnt = np.zeros(n, np.int32)
        ufirst = np.full(n, -1, np.int64)
        for km, e in self.tab.items():
            i = idx.get(km, -1)
            if i >= 0:
                ucnt[i] = len(e["a"])
                ufirst[i] = min(e["a"])
        return torch.from_numpy(np.concatenate([inf, ufirst])), torch.from_numpy(ucnt)

    def finish(self, mins, ucnt, pre_total):
        n, k = len(self.all), self.k
        mins, ucnt = mins.numpy(), ucnt.numpy()
        inf, ufirst = mins[:4 * n], mins[4 * n:]
        self.all = [(km, gc, gf, min(int(ucnt[i]), 32765), int(ufirst[i])) for i, (km, gc, gf) in enumerate(self.all)]
        order = sorted(range(n), key=lambda i: self.all[i][4])
        rank = {s: r for r, s in enumerate(order)}
        idx = {s[0]: i for i, s in enumerate(self.all)}
        g = Graph(k, n, pre_total, np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8),
                  np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32), np.zeros(n, np.uint8),
                  np.zeros((n, 4), np.uint32), np.zeros((n, k), np.uint8))
        edges = []
        for r, s in enumerate(order):
            km, gc, gf, uc, uf = self.all[s]
            g.first_inst[r] = uf
            g.gated_count[r], g.freq[r] = gc, uc
            g.kmers[r] = np.frombuffer(km, np.uint8)
            if k > 16:
                code = synth.seq_to_int(km[:16].decode())
                g.has_v[r], g.has_j[r] = int(code in self.vs and code != 0), int(code in self.js and code != 0)
            else:
                g.has_v[r] = g.has_j[r] = 1
            for b in range(4):
                if inf[s * 4 + b] != -1:
                    u = idx["ATCG"[b].encode() + km[:-1]]             # the tail of the in-edge (v, first base of u)
                    edges.append((int(inf[s * 4 + b]), rank[u], r))
        for first, u, v in sorted(edges, reverse=True):
            g.to_ids[u, g.to_deg[u]] = v + 1
            g.to_deg[u] += 1
            g.from_ids[v, g.from_deg[v]] = u + 1
            g.from_deg[v] += 1
        return g

    def end(self):
        pass


class RefScorerEngine:
    """CPU stand-in for vdjer_amd.shard.HipScorerEngine over the oracle's read index of THIS rank's pool -- TEST INFRASTRUCTURE."""

    def __init__(self, pool):
        from oracle import oracle
        self.ix = oracle.ReadIndex(pool)
        self.rl = pool.rl

    def window_pairs(self, windows):
        self._ln = len(windows[0]) if windows else 0
        self.lists = []
        for w in windows:
            pairs, _ = self.ix.quick_map(w)
            self.lists.append(np.array([(1 << 32) | (int(q["pos1"]) << 16) | int(q["pos2"]) for q in pairs], dtype=np.int64))
        ent = np.array([x.shape[0] for x in self.lists], np.uint32)
        return ent, ent.copy()

    def window_fetch(self, window_ids, total):
        parts = [self.lists[int(i)] for i in window_ids]
        return torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.int64))

    def window_cover(self, n, lists, nsrc, counts, ins, **cov):
        a = lists.numpy()
        counts = np.asarray(counts).reshape(nsrc, n)
        per = [[] for _ in range(n)]
        at = 0
        for s in range(nsrc):
            for w in range(n):
                c = int(counts[s, w])
                per[w].append(a[at:at + c])
                at += c
        out = np.zeros(n, np.uint8)
        for w in range(n):
            x = np.concatenate(per[w]) if per[w] else np.zeros(0, np.int64)
            st = []
            for e in x:
                m, p1, p2 = int(e) >> 32, (int(e) >> 16) & 0xFFFF, int(e) & 0xFFFF
                st += [(p1, p2), (p2, p1)] * m
            st = np.array(sorted(st), dtype=np.int32).reshape(-1, 2)
            out[w] = self.ix.coverage_is_valid(st, self._ln, ins, **cov)
        return out
