"""Multi-GPU k-mer build: hash-prefix sharding, partial aggregates merged by the owner.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Every rank holds its own slice of the read pool.  Record numbering is rank-major with a common
stride: rank r's records are [r*stride, r*stride + R_r), which is also the scan order (A2:1388-1390) of the
union pool, so "first instance" means the same thing as in a single-process run over the concatenation.

Phases (the engine does the compute, this module only moves bytes):
  1. every rank aggregates ITS gated instances per distinct k-mer: count, first instance, "saw two different reads"
     (add_to_table A2:322-367 restated per rank)                                     -- no communication
  2. ONE bulk all-to-all: the partial aggregates (32 B per distinct gated k-mer per rank, NOT per instance) go to the
     owner of the k-mer = its hash bucket / buckets per owner (any number of ranks), with a per-bucket directory
  3. the owner merges (counts add, firsts take the minimum, flags OR) and decides almost every k-mer; the few whose
     verdict needs per-read data (flag still open although several ranks hold the k-mer; count so low that the
     quality sums matter) cost a question (8 B) to the holders and an answer (200 B): two tiny all-to-alls
  4. all_gather of the survivors (32 B each); every rank walks ITS records for add_to_graph's bookkeeping of all
     survivors (A2:261-320: node frequency, first sights of nodes and in-edges); all_reduce MIN (first sights, 64-bit
     instance ids) and SUM (counts)
  5. every rank numbers the nodes and orders the edge lists (identical result on all ranks)
The read pool itself never leaves its rank.  The serial de Bruijn traversal then runs on rank 0 (north star:
host-side, not sharded).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .api import Graph, _p, check

SURV_BYTES = 32


def _bind(L):
    if getattr(L, "_shard_bound", False):
        return
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.vdjx_shard_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.vdjx_shard_begin_share.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(vp)]
    L.vdjx_shard_free.argtypes = [vp]
    L.vdjx_shard_free.restype = None
    L.vdjx_shard_record_bytes.argtypes = [C.c_int]
    L.vdjx_shard_record_bytes.restype = C.c_size_t
    L.vdjx_shard_count.argtypes = [vp, u64p]
    L.vdjx_shard_geometry.argtypes = [vp, C.c_uint64]
    L.vdjx_shard_symmetric.argtypes = [vp]
    L.vdjx_shard_geometry2.argtypes = [vp, C.c_uint64, C.c_int]
    L.vdjx_shard_local.argtypes = [vp, u64p, C.POINTER(C.c_uint32)]
    L.vdjx_shard_local_fill.argtypes = [vp, vp, vp]
    L.vdjx_shard_merge.argtypes = [vp, vp, vp, u64p, u64p]
    L.vdjx_shard_queries.argtypes = [vp, vp]
    L.vdjx_shard_reply.argtypes = [vp, vp, u64p, vp]
    L.vdjx_shard_resolve.argtypes = [vp, vp, C.c_uint64, u64p, u64p]
    L.vdjx_shard_survivors.argtypes = [vp, vp]
    L.vdjx_shard_edges.argtypes = [vp, vp, C.c_uint64, vp, vp, vp]
    L.vdjx_shard_finish.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L._shard_bound = True


class HipShardEngine:
    """The phases of one sharded build on this rank's GPU (libvdjx.so, device tensors owned by torch).
    Exchanged records are opaque rows of bytes: [n, vdjx_shard_record_bytes(kind)] uint8."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.L = ctx.L
        _bind(self.L)
        self.dev = device
        self.h = None
        self.W = [int(self.L.vdjx_shard_record_bytes(i)) for i in range(3)]

    def begin(self, pool, k, mf, mq, rank, world, stride, scan_index=None, total_records=0):
        """scan_index (device int32/uint32 tensor [pool.n_records], ascending): the pool is a SHARE of a pool of total_records records
        and record i sits at scan position scan_index[i] (vdjx_shard_begin_share); None: rank r holds records [r*stride, ...)"""
        h = C.c_void_p()
        self._keep = []
        if scan_index is not None:
            self._keep.append(scan_index)
            check(self.L.vdjx_shard_begin_share(self.ctx.h, pool.h, k, mf, mq, rank, world, self._dp(scan_index), int(total_records), C.byref(h)),
                  "vdjx_shard_begin_share")
        else:
            check(self.L.vdjx_shard_begin(self.ctx.h, pool.h, k, mf, mq, rank, world, stride, C.byref(h)), "vdjx_shard_begin")
        self.h, self.pool, self.k, self.world = h, pool, k, world

    def _dp(self, t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None

    def _u64(self, values):
        return (C.c_uint64 * self.world)(*[int(v) for v in values])

    def count(self) -> int:
        """this rank's gated k-mer instances (the ranks agree on the largest: geometry())"""
        n = C.c_uint64()
        check(self.L.vdjx_shard_count(self.h, C.byref(n)), "vdjx_shard_count")
        return int(n.value)

    def symmetric(self) -> int:
        """1: this rank's pool is made of couples (record, its reverse complement) and k is odd: its local phase could move half the tuples"""
        return int(self.L.vdjx_shard_symmetric(self.h))

    def geometry(self, agreed: int, all_symmetric: int = 0) -> None:
        check(self.L.vdjx_shard_geometry2(self.h, int(agreed), int(all_symmetric)), "vdjx_shard_geometry2")

    def local(self):
        """-> (partials per owner [G], directory int32 [G*dir_len], partial aggregates [n, 32] uint8; the engine keeps reading the
        aggregates from that tensor until reply() has returned)"""
        t = self.torch
        cnt, dl = (C.c_uint64 * self.world)(), C.c_uint32()
        check(self.L.vdjx_shard_local(self.h, cnt, C.byref(dl)), "vdjx_shard_local")
        counts = np.array(list(cnt), dtype=np.int64)
        d = t.empty(self.world * int(dl.value), dtype=t.int32, device=self.dev)
        parts = t.empty((int(counts.sum()), self.W[0]), dtype=t.uint8, device=self.dev)
        check(self.L.vdjx_shard_local_fill(self.h, self._dp(d), self._dp(parts)), "vdjx_shard_local_fill")
        self._keep.append(parts)
        return counts, d, parts

    def merge(self, recv_dir, recv_parts, recv_counts):
        """-> questions per rank [G]"""
        qc = (C.c_uint64 * self.world)()
        self._keep += [recv_dir, recv_parts]
        check(self.L.vdjx_shard_merge(self.h, self._dp(recv_dir), self._dp(recv_parts), self._u64(recv_counts), qc), "vdjx_shard_merge")
        self._nq = np.array(list(qc), dtype=np.int64)
        return self._nq

    def queries(self):
        t = self.torch
        q = t.empty((int(self._nq.sum()), self.W[1]), dtype=t.uint8, device=self.dev)
        check(self.L.vdjx_shard_queries(self.h, self._dp(q)), "vdjx_shard_queries")
        return q

    def reply(self, queries, counts):
        t = self.torch
        out = t.empty((queries.shape[0], self.W[2]), dtype=t.uint8, device=self.dev)
        check(self.L.vdjx_shard_reply(self.h, self._dp(queries), self._u64(counts), self._dp(out)), "vdjx_shard_reply")
        return out

    def resolve(self, replies):
        ns, nd = C.c_uint64(), C.c_uint64()
        check(self.L.vdjx_shard_resolve(self.h, self._dp(replies), replies.shape[0], C.byref(ns), C.byref(nd)), "vdjx_shard_resolve")
        return int(ns.value), int(nd.value)

    def survivors(self, ns):
        t = self.torch
        out = t.empty((ns, SURV_BYTES), dtype=t.uint8, device=self.dev)
        check(self.L.vdjx_shard_survivors(self.h, self._dp(out)), "vdjx_shard_survivors")
        return out

    def edges(self, surv_all):
        """-> (mins, ucnt): `mins` = first sight of every in-edge slot [4n] | first sight of every survivor [n] on this rank, one
        int64 tensor (global instance ids, all-ones = none) to be MIN-reduced in unsigned order; ucnt int32 [n] = this rank's
        instances per survivor, to be SUMmed"""
        t = self.torch
        n = surv_all.shape[0]
        mins = t.empty(n * 5, dtype=t.int64, device=self.dev)
        ucnt = t.empty(n, dtype=t.int32, device=self.dev)
        self._keep += [surv_all, mins, ucnt]
        check(self.L.vdjx_shard_edges(self.h, self._dp(surv_all), n, self._dp(mins[:4 * n]), self._dp(ucnt), self._dp(mins[4 * n:])),
              "vdjx_shard_edges")
        return mins, ucnt

    def finish(self, mins, ucnt, pre_total, keep_device: bool = False, async_export: bool = False):
        g = C.c_void_p()
        n = ucnt.shape[0]
        check(self.L.vdjx_shard_finish(self.h, self._dp(mins[:4 * n]), self._dp(ucnt), self._dp(mins[4 * n:]), pre_total, C.byref(g)),
              "vdjx_shard_finish")
        return self.ctx._export_graph(g, self.k, keep_device or async_export, async_export)

    def end(self):
        if self.h:
            self.L.vdjx_shard_free(self.h)
            self.h = None
        self._keep = []


class HipScorerEngine:
    """The halves of the window scorer on this rank's GPU (vdjx_window_pairs / _fetch / vdjx_window_cover): this rank's read index
    holds only ITS pairs (both mates of a pair on one rank)."""

    def __init__(self, ctx, device, rl: int):
        import torch
        self.torch, self.ctx, self.dev, self.rl = torch, ctx, device, rl

    def window_pairs(self, windows):
        """windows: list of equally long strings, or Context.pack_strings / pin_strings output"""
        self._ln = windows[2] if isinstance(windows, tuple) else (len(windows[0]) if windows else 0)
        return self.ctx.window_pairs(windows)

    def window_fetch(self, window_ids, total: int):
        t = self.torch
        out = t.empty(max(total, 1), dtype=t.int64, device=self.dev)
        self.ctx.window_pairs_fetch(window_ids, out.data_ptr())
        return out[:total]

    def window_cover(self, n: int, lists, nsrc: int, counts, ins: int, **cov):
        return self.ctx.window_cover(n, self._ln, self.rl, lists.data_ptr() if lists.numel() else 0, nsrc, counts, ins, **cov)


class Comm:
    """The four collectives the sharded path needs, on RCCL ("nccl") natively and on "gloo" through host copies
    (gloo has no all_to_all_single / all_gather_into_tensor; used by the CPU tests and single-device dry runs)."""

    def __init__(self, dist, device):
        import torch
        self.t, self.dist, self.dev = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        backend = dist.get_backend() if hasattr(dist, "get_backend") else "native"
        self.native = backend != "gloo"
        self.async_ok = backend == "nccl"          # real RCCL: collectives may overlap with the library's own stream
        self.on_gpu = device.type == "cuda"
        self.bytes = 0

    def sync(self):
        if self.on_gpu:
            self.t.cuda.synchronize(self.dev)

    def _host(self, x):
        return x if self.native else x.cpu()

    def all_reduce(self, x, op):
        if self.native:
            flat = x.view(-1)
            cap = max(1, self.CHUNK_BYTES // max(1, x.element_size()))
            for a in range(0, max(flat.numel(), 1), cap):        # (see CHUNK_BYTES)
                self.dist.all_reduce(flat[a:a + cap], op=op)
        else:
            h = x.cpu()
            self.dist.all_reduce(h, op=op)
            x.copy_(h)
        return x

    def all_gather_cat(self, x):
        """equal shapes on every rank -> concatenation along dim 0 (rank order)"""
        t = self.t
        self.bytes += x.numel() * x.element_size() * self.world
        if self.native:
            x = x.contiguous()
            n = x.shape[0]
            out = t.empty((self.world * n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            row = max(1, (x[0].numel() if n else 1) * x.element_size())
            cap = max(1, self.CHUNK_BYTES // row)
            if n <= cap:
                self.dist.all_gather_into_tensor(out, x)
                return out
            ov = out.view((self.world, n) + tuple(x.shape[1:]))
            for a in range(0, n, cap):                           # (see CHUNK_BYTES)
                b = min(n, a + cap)
                piece = t.empty((self.world * (b - a),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                self.dist.all_gather_into_tensor(piece, x[a:b].contiguous())
                ov[:, a:b].copy_(piece.view((self.world, b - a) + tuple(x.shape[1:])))
            return out
        h = x.cpu().contiguous()
        outs = [t.empty_like(h) for _ in range(self.world)]
        self.dist.all_gather(outs, h)
        return t.cat(outs, dim=0).to(x.device)

    def all_gather_var(self, x, counts):
        """first dimension differs per rank (counts known everywhere)"""
        t = self.t
        mx = int(max(counts)) if len(counts) else 0
        pad = t.zeros((max(mx, 1),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if x.shape[0]:
            pad[:x.shape[0]] = x
        full = self.all_gather_cat(pad)
        if not sum(counts):
            return x[:0]
        return t.cat([full[i * pad.shape[0]:i * pad.shape[0] + int(c)] for i, c in enumerate(counts)], dim=0)

    # RCCL moved the second half of a 1.09 GB all_to_all_single wrongly on this stack (measured: ROCm 7.0.2 / RCCL 2.26.6, one rank
    # sending 34 M partials to itself: bytes beyond 2^29 differ, silently): no single transfer is larger than CHUNK_BYTES here, and
    # what a rank sends to itself is a device copy
    CHUNK_BYTES = 128 << 20

    def all_to_all_v(self, send, in_splits, recv, out_splits):
        self.bytes += send.numel() * send.element_size()
        if self.native and not hasattr(self.dist, "P2POp"):      # (in-process stand-ins of the tests: tensor copies, no RCCL)
            self.dist.all_to_all_single(recv, send, out_splits, in_splits)
            return
        if self.native:
            row = max(1, (send[0].numel() if send.shape[0] else 1) * send.element_size())
            cap = max(1, self.CHUNK_BYTES // row)                 # rows per transfer
            s_off = [0]
            for v in in_splits:
                s_off.append(s_off[-1] + int(v))
            r_off = [0]
            for v in out_splits:
                r_off.append(r_off[-1] + int(v))
            me = self.rank
            n_self = min(int(in_splits[me]), int(out_splits[me]))
            if n_self:
                recv[r_off[me]:r_off[me] + n_self].copy_(send[s_off[me]:s_off[me] + n_self])
            rounds = 0
            for peer in range(self.world):
                if peer != me:
                    rounds = max(rounds, -(-int(in_splits[peer]) // cap), -(-int(out_splits[peer]) // cap))
            # every rank runs the same number of rounds per peer pair (both sides know the counts of their pair)
            for rd in range(rounds):
                ops = []
                for peer in range(self.world):
                    if peer == me:
                        continue
                    a, b = rd * cap, min((rd + 1) * cap, int(in_splits[peer]))
                    if a < b:
                        ops.append(self.dist.P2POp(self.dist.isend, send[s_off[peer] + a:s_off[peer] + b], peer))
                    a, b = rd * cap, min((rd + 1) * cap, int(out_splits[peer]))
                    if a < b:
                        ops.append(self.dist.P2POp(self.dist.irecv, recv[r_off[peer] + a:r_off[peer] + b], peer))
                if ops:
                    for w in self.dist.batch_isend_irecv(ops):
                        w.wait()
            return
        hs = send.cpu()
        hr = self.t.empty(recv.shape, dtype=recv.dtype)
        ins, outs = list(hs.split(in_splits)), list(hr.split(out_splits))
        reqs = []
        for peer in range(self.world):
            if peer == self.rank:
                outs[peer].copy_(ins[peer])
            else:
                reqs.append(self.dist.isend(ins[peer].contiguous(), peer))
                reqs.append(self.dist.irecv(outs[peer], peer))
        for r in reqs:
            r.wait()
        recv.copy_(hr)


class ShardedHotPath:
    """Drives one sharded k-mer build over a torch.distributed process group."""

    def __init__(self, ctx, dist, device, engine=None, stride=None):
        """stride: records per rank in the global numbering (default: the largest local pool, agreed by all_reduce)"""
        import torch
        self.torch, self.dist, self.dev = torch, dist, device
        self.comm = Comm(dist, device)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.engine = engine if engine is not None else HipShardEngine(ctx, device)
        self.stride = stride
        self.laps = {}               # seconds per phase of kmer_build, summed over calls (host clock, diagnostic)

    @property
    def bytes_exchanged(self):
        return self.comm.bytes

    def window_score(self, scorer, windows, ins: int, **cov):
        """quick_map_process_contig + coverage_is_valid (A2:841-847) over a pool sharded BY PAIR: every rank maps ALL windows
        against its own reads (`scorer`: HipScorerEngine or a stand-in), window w's pair lists meet on rank w % G (one
        all-to-all of 8-byte entries: identical read pairs travel once, with a multiplicity), which tests their union.
        Every rank passes the same windows and gets (valid[n], npairs[n])."""
        t, dist, cm = self.torch, self.dist, self.comm
        G, r = self.world, self.rank
        n = windows[1] if isinstance(windows, tuple) else len(windows)
        if n == 0:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        ent, npairs = scorer.window_pairs(windows)
        if G == 1 and not os.environ.get("VDJX_SHARD_SELF_COLLECTIVES"):               # (one rank: its lists are the union already; nothing to gather, move or reduce)
            ids = np.arange(n, dtype=np.uint32)
            lists = scorer.window_fetch(ids, int(ent.sum()))
            return scorer.window_cover(n, lists, 1, ent.astype(np.int64).reshape(1, n), ins, **cov), npairs.astype(np.uint32)
        own = [np.arange(o, n, G) for o in range(G)]
        all_ent = cm.all_gather_cat(t.from_numpy(ent.astype(np.int64)).to(self.dev).view(1, n)).cpu().numpy()      # [G, n]
        send_ids = np.concatenate(own).astype(np.uint32)
        send_counts = [int(ent[own[o]].sum()) for o in range(G)]
        counts = all_ent[:, own[r]]                                         # what every source holds for MY windows
        recv_counts = [int(v) for v in counts.sum(axis=1)]
        send = scorer.window_fetch(send_ids, int(sum(send_counts)))
        recv = t.empty(int(sum(recv_counts)), dtype=t.int64, device=send.device)
        cm.all_to_all_v(send, send_counts, recv, recv_counts)
        cm.sync()
        valid_mine = scorer.window_cover(len(own[r]), recv, G, counts, ins, **cov)
        tot = t.from_numpy(npairs.astype(np.int64)).to(self.dev)
        cm.all_reduce(tot, dist.ReduceOp.SUM)
        parts = cm.all_gather_var(t.from_numpy(np.ascontiguousarray(valid_mine)).to(self.dev), [len(own[o]) for o in range(G)]).cpu().numpy()
        valid = np.zeros(n, np.uint8)
        at = 0
        for o in range(G):
            valid[own[o]] = parts[at:at + len(own[o])]
            at += len(own[o])
        return valid, tot.cpu().numpy().astype(np.uint32)

    def sam_body(self, ctx, contigs, contig_ids, d_reg_rank: int):
        """output_mapping (quick_map3.c:152-181) over a pool sharded BY PAIR: every rank formats the SAM records of ITS pairs for these
        contigs on its device (vdjx_sam_blocks: text, bytes per pair, a 64-bit key contig << 44 | read-1 position << 32 | global
        registration rank), the blocks travel to rank 0 (three all-to-alls with one receiver) and vdjx_sam_merge lays them out in key
        order = the order quick_map_process_contig walks its lists in.  -> the text on rank 0 (b"" elsewhere).  The driver of
        vdjx_mgpu.c:do_sam in Python (one run: the caller bounds the contigs)."""
        t, cm = self.torch, self.comm
        G, r = self.world, self.rank
        nb, nby, dk, dl, dt = ctx.sam_blocks(contigs, contig_ids, d_reg_rank)
        keys = t.empty(max(nb, 1), dtype=t.int64, device=self.dev)
        lens = t.empty(max(nb, 1), dtype=t.int32, device=self.dev)
        text = t.empty(max(nby, 1), dtype=t.uint8, device=self.dev)
        ctx.device_copy(keys.data_ptr(), dk, nb * 8)
        ctx.device_copy(lens.data_ptr(), dl, nb * 4)
        ctx.device_copy(text.data_ptr(), dt, nby)
        if G == 1:
            return ctx.sam_merge(nb, nby, keys.data_ptr(), lens.data_ptr(), text.data_ptr())
        meta = cm.all_gather_cat(t.tensor([[nb, nby]], dtype=t.int64, device=self.dev)).cpu().numpy()
        NB, NBY = int(meta[:, 0].sum()), int(meta[:, 1].sum())
        to0 = lambda n_: [int(n_)] + [0] * (G - 1)            # noqa: E731
        from0 = lambda col: [int(v) for v in meta[:, col]] if r == 0 else [0] * G      # noqa: E731
        rk = t.empty(NB if r == 0 else 0, dtype=t.int64, device=self.dev)
        rl_ = t.empty(NB if r == 0 else 0, dtype=t.int32, device=self.dev)
        rt = t.empty(NBY if r == 0 else 0, dtype=t.uint8, device=self.dev)
        cm.all_to_all_v(keys[:nb], to0(nb), rk, from0(0))
        cm.all_to_all_v(lens[:nb], to0(nb), rl_, from0(0))
        cm.all_to_all_v(text[:nby], to0(nby), rt, from0(1))
        cm.sync()
        if r != 0 or NB == 0:
            return b""
        return ctx.sam_merge(NB, NBY, rk.data_ptr(), rl_.data_ptr(), rt.data_ptr())

    def kmer_build(self, pool, k: int = 35, mf: int = 3, mq: int = 90, keep_device: bool = False, async_export: bool = False,
                   scan_index=None, total_records: int = 0):
        """scan_index / total_records: the pool is this rank's SHARE of the whole pool (by pair, any dealing that keeps the scan order),
        record i at scan position scan_index[i]; without them rank r holds the r-th slice of the scan order"""
        t, dist, eng, cm = self.torch, self.dist, self.engine, self.comm
        G, r = self.world, self.rank
        if scan_index is None and self.stride is None:
            s = t.tensor([pool.n_records], dtype=t.int64, device=self.dev)
            cm.all_reduce(s, dist.ReduceOp.MAX)
            self.stride = int(s.item())
        stride = self.stride
        alone = G == 1 and not os.environ.get("VDJX_SHARD_SELF_COLLECTIVES")      # (the variable: a lone rank calls the collectives all the same -- the one-rank RCCL test)
        import time
        clock = [time.perf_counter()]

        def lap(name):
            now = time.perf_counter()
            self.laps[name] = self.laps.get(name, 0.0) + now - clock[0]
            clock[0] = now

        if scan_index is not None:
            eng.begin(pool, k, mf, mq, r, G, 0, scan_index=scan_index, total_records=total_records)
        else:
            eng.begin(pool, k, mf, mq, r, G, stride)
        try:
            def exchange(send, ins, outs):
                if alone:             # nothing to move: what a rank keeps for itself is its send buffer (with peers, a 1/G-th of it is copied)
                    return send
                recv = t.empty((int(sum(outs)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
                cm.all_to_all_v(send, [int(v) for v in ins], recv, [int(v) for v in outs])
                return recv

            def counts_of(x):            # what every peer will send me, given what I send every peer
                if alone:
                    return np.asarray([int(v) for v in x], dtype=np.int64)
                got = exchange(t.tensor([int(v) for v in x], dtype=t.int64, device=self.dev), [1] * G, [1] * G)
                cm.sync()
                return got.cpu().numpy()

            # 0. the ranks agree on the bucket geometry: the largest number of gated instances any of them holds
            if hasattr(eng, "count") and not os.environ.get("VDJX_SHARD_NO_AGREE"):      # (the variable: the stride's bound instead, for comparison)
                # (one reduction for both: the largest count, and -- as the largest of the negated flags -- whether EVERY rank's pool is made
                # of couples, in which case all cut their buckets by the smaller of a k-mer and its reverse complement)
                sym = eng.symmetric() if hasattr(eng, "symmetric") else 0
                nmax = t.tensor([eng.count(), 1 - sym], dtype=t.int64, device=self.dev)
                if G > 1:
                    cm.all_reduce(nmax, dist.ReduceOp.MAX)
                if hasattr(eng, "symmetric"):
                    eng.geometry(int(nmax[0].item()), 1 - int(nmax[1].item()))
                else:
                    eng.geometry(int(nmax[0].item()))
                lap("count")
            # 1. local aggregation; 2. the bulk exchange: per-bucket directory (its sums are the receive counts), then the
            #    partial aggregates themselves
            send_counts, sdir, sparts = eng.local()
            lap("local")
            dl = sdir.numel() // G
            rdir = exchange(sdir, [dl] * G, [dl] * G)
            cm.sync()
            recv_counts = rdir.view(G, dl).sum(dim=1, dtype=t.int64).cpu().numpy()
            rparts = exchange(sparts, send_counts, recv_counts)
            cm.sync()
            lap("exchange_partials")
            # 3. owners merge and decide; questions and answers for the few open k-mers
            q_out = eng.merge(rdir, rparts, recv_counts)
            lap("merge")
            q_in = counts_of(q_out)
            rq = exchange(eng.queries(), q_out, q_in)
            cm.sync()
            answers = exchange(eng.reply(rq, q_in), q_in, q_out)
            cm.sync()
            lap("questions_answers")
            ns, ndist = eng.resolve(answers)
            del sparts, rq
            lap("resolve")
            meta = np.asarray([[ns, ndist]], dtype=np.int64) if alone else cm.all_gather_cat(t.tensor([[ns, ndist]], dtype=t.int64, device=self.dev)).cpu().numpy()
            ns_all = [int(v) for v in meta[:, 0]]
            pre_total = int(meta[:, 1].sum())
            # 4. survivors everywhere, local edges, MIN over ranks
            surv_all = eng.survivors(ns) if alone else cm.all_gather_var(eng.survivors(ns), ns_all)
            if not alone:
                cm.sync()
            lap("gather_survivors")
            mins, ucnt = eng.edges(surv_all)
            lap("edges")
            if ucnt.numel() and not alone:
                flip = -2 ** 63       # unsigned order on int64 tensors: flip the sign bit around the MIN (all-ones = none stays largest)
                mins.bitwise_xor_(flip)
                cm.all_reduce(mins, dist.ReduceOp.MIN)
                mins.bitwise_xor_(flip)
                cm.all_reduce(ucnt, dist.ReduceOp.SUM)
            if not alone:
                cm.sync()
            lap("reduce_edges")
            # 5. node numbering + list order
            if async_export:
                return eng.finish(mins, ucnt, pre_total, async_export=True)
            return eng.finish(mins, ucnt, pre_total, keep_device=True) if keep_device else eng.finish(mins, ucnt, pre_total)
        finally:
            eng.end()
