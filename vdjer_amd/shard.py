"""Multi-GPU k-mer build: hash-prefix sharding with one exchange step.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Every rank holds its own slice of the read pool.  Record numbering is rank-major with a common
stride: rank r's records are [r*stride, r*stride + R_r), which is also the scan order (A2:1388-1390) of the
union pool, so "first instance" means the same thing as in a single-process run over the concatenation.

Phases (the engine does the compute, this module only moves bytes):
  1. all_gather of the packed pool (16 B bases + 8 B N mask + qstride B qualities per record) -- the prune
     needs the first instance's record and the low-count instances' qualities, wherever they live
  2. every rank partitions its k-mer instances by owner = top log2(G) bits of the k-mer hash
  3. ONE all-to-all (counts, then the three tuple columns): every instance reaches its owner
  4. owners reduce (count / first / distinct-read flag / quality sums / ungated recount) and prune
  5. all_gather of the survivors (32 B each), local edge pass, all_reduce(MIN) of the edge first-sight arrays
  6. every rank numbers the nodes and orders the edge lists (identical result on all ranks)
The serial de Bruijn traversal then runs on rank 0 (north star: host-side, not sharded).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .api import Graph, _p, check

SURV_BYTES = 32


def _bind(L):
    if getattr(L, "_shard_bound", False):
        return
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.vdjx_shard_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.vdjx_shard_free.argtypes = [vp]
    L.vdjx_shard_free.restype = None
    L.vdjx_shard_qstride.argtypes = [vp]
    L.vdjx_shard_key_hi_bytes.argtypes = [vp]
    L.vdjx_shard_pool_export.argtypes = [vp, vp, vp, vp]
    L.vdjx_shard_set_pool.argtypes = [vp, vp, vp, vp]
    L.vdjx_shard_partition_count.argtypes = [vp, u64p]
    L.vdjx_shard_partition_fill.argtypes = [vp, vp, vp, vp]
    L.vdjx_shard_reduce.argtypes = [vp, vp, vp, vp, C.c_uint64, u64p, u64p]
    L.vdjx_shard_survivors.argtypes = [vp, vp]
    L.vdjx_shard_edges.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.vdjx_shard_finish.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L._shard_bound = True


class HipShardEngine:
    """The phases of one sharded build on this rank's GPU (libvdjx.so, device tensors owned by torch)."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.L = ctx.L
        _bind(self.L)
        self.dev = device
        self.h = None

    def begin(self, pool, k, mf, mq, rank, world, stride):
        h = C.c_void_p()
        check(self.L.vdjx_shard_begin(self.ctx.h, pool.h, k, mf, mq, rank, world, stride, C.byref(h)), "vdjx_shard_begin")
        self.h, self.pool, self.k, self.world = h, pool, k, world
        self.hi_bytes = self.L.vdjx_shard_key_hi_bytes(h)
        self.qstride = self.L.vdjx_shard_qstride(h)

    def _dp(self, t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None

    def pool_export(self):
        t, R = self.torch, self.pool.n_records
        out = [t.empty((R, 16), dtype=t.uint8, device=self.dev), t.empty((R, 8), dtype=t.uint8, device=self.dev),
               t.empty((R, self.qstride), dtype=t.uint8, device=self.dev)]
        if R:
            check(self.L.vdjx_shard_pool_export(self.h, *[self._dp(x) for x in out]), "vdjx_shard_pool_export")
        return out

    def set_pool(self, tensors):
        self._gpool = tensors
        check(self.L.vdjx_shard_set_pool(self.h, *[C.c_void_p(x.data_ptr()) for x in tensors]), "vdjx_shard_set_pool")

    def partition_count(self):
        cnt = (C.c_uint64 * self.world)()
        check(self.L.vdjx_shard_partition_count(self.h, cnt), "vdjx_shard_partition_count")
        return np.array(list(cnt), dtype=np.int64)

    def partition_fill(self, n):
        t = self.torch
        lo = t.empty(n, dtype=t.int64, device=self.dev)
        hi = t.empty(n, dtype=t.int64 if self.hi_bytes == 8 else t.int32, device=self.dev)
        inst = t.empty(n, dtype=t.int32, device=self.dev)
        check(self.L.vdjx_shard_partition_fill(self.h, self._dp(lo), self._dp(hi), self._dp(inst)), "vdjx_shard_partition_fill")
        return [lo, hi, inst]

    def recv_like(self, n):
        t = self.torch
        return [t.empty(n, dtype=t.int64, device=self.dev),
                t.empty(n, dtype=t.int64 if self.hi_bytes == 8 else t.int32, device=self.dev),
                t.empty(n, dtype=t.int32, device=self.dev)]

    def reduce(self, recv):
        ns, nd = C.c_uint64(), C.c_uint64()
        self._recv = recv
        check(self.L.vdjx_shard_reduce(self.h, self._dp(recv[0]), self._dp(recv[1]), self._dp(recv[2]), recv[0].numel(),
                                       C.byref(ns), C.byref(nd)), "vdjx_shard_reduce")
        return int(ns.value), int(nd.value)

    def survivors(self, ns):
        t = self.torch
        out = t.empty((ns, SURV_BYTES), dtype=t.uint8, device=self.dev)
        check(self.L.vdjx_shard_survivors(self.h, self._dp(out)), "vdjx_shard_survivors")
        return out

    def edges(self, surv_all):
        t = self.torch
        n = surv_all.shape[0]
        ef = t.empty(n * 4, dtype=t.int32, device=self.dev)
        et = t.empty(n * 4, dtype=t.int32, device=self.dev)
        self._surv_all = surv_all
        check(self.L.vdjx_shard_edges(self.h, self._dp(surv_all), n, self._dp(ef), self._dp(et)), "vdjx_shard_edges")
        return ef, et

    def finish(self, ef, et, pre_total, keep_device: bool = False):
        g = C.c_void_p()
        check(self.L.vdjx_shard_finish(self.h, self._dp(ef), self._dp(et), pre_total, C.byref(g)), "vdjx_shard_finish")
        return self.ctx._export_graph(g, self.k, keep_device)

    def end(self):
        if self.h:
            self.L.vdjx_shard_free(self.h)
            self.h = None
        self._gpool = self._recv = self._surv_all = None


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
            self.dist.all_reduce(x, op=op)
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
            out = t.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            self.dist.all_gather_into_tensor(out, x.contiguous())
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

    def all_to_all_v(self, send, in_splits, recv, out_splits):
        self.bytes += send.numel() * send.element_size()
        if self.native:
            self.dist.all_to_all_single(recv, send, out_splits, in_splits)
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

    def __init__(self, ctx, dist, device, engine=None):
        import torch
        self.torch, self.dist, self.dev = torch, dist, device
        self.comm = Comm(dist, device)
        self.rank, self.world = self.comm.rank, self.comm.world
        if self.world & (self.world - 1):
            raise ValueError("the number of ranks must be a power of two (ownership = hash-prefix bits)")
        self.engine = engine if engine is not None else HipShardEngine(ctx, device)
        self.stride = None

    @property
    def bytes_exchanged(self):
        return self.comm.bytes

    def kmer_build(self, pool, k: int = 35, mf: int = 3, mq: int = 90, keep_device: bool = False):
        t, dist, eng, cm = self.torch, self.dist, self.engine, self.comm
        G, r = self.world, self.rank
        if self.stride is None:
            s = t.tensor([pool.n_records], dtype=t.int64, device=self.dev)
            cm.all_reduce(s, dist.ReduceOp.MAX)
            self.stride = int(s.item())
        stride = self.stride
        eng.begin(pool, k, mf, mq, r, G, stride)
        try:
            # 1. replicate the packed pool (rank-major, common stride).  On RCCL the all_gathers are issued
            #    asynchronously: they run on the communicator's stream while this rank partitions its k-mers (2.)
            glob, pending = [], []
            for x in eng.pool_export():
                pad = t.zeros((stride,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                pad[:x.shape[0]] = x
                if cm.async_ok:
                    full = t.empty((G * stride,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                    cm.sync()                     # `pad` was written on torch's stream by the library-exported copy
                    pending.append((dist.all_gather_into_tensor(full, pad, async_op=True), pad))
                    cm.bytes += full.numel() * full.element_size()
                    glob.append(full)
                else:
                    glob.append(cm.all_gather_cat(pad))
            if not pending:
                cm.sync()
            eng.set_pool(glob)        # pointers only; the first consumer is reduce() in step 4
            # 2./3. partition by owner, then the exchange step: counts, then the tuple columns
            send_counts = eng.partition_count()
            sc = t.tensor(send_counts, dtype=t.int64, device=self.dev)
            rc = t.empty_like(sc)
            cm.all_to_all_v(sc, [1] * G, rc, [1] * G)
            cm.sync()
            recv_counts = rc.cpu().numpy()
            send = eng.partition_fill(int(send_counts.sum()))
            recv = eng.recv_like(int(recv_counts.sum()))
            ins, outs = [int(v) for v in send_counts], [int(v) for v in recv_counts]
            for s_, r_ in zip(send, recv):
                cm.all_to_all_v(s_, ins, r_, outs)
            for h_, _pad in pending:
                h_.wait()
            cm.sync()
            del send, pending
            # 4. owners reduce and prune
            ns, ndist = eng.reduce(recv)
            meta = cm.all_gather_cat(t.tensor([[ns, ndist]], dtype=t.int64, device=self.dev)).cpu().numpy()
            ns_all = [int(v) for v in meta[:, 0]]
            pre_total = int(meta[:, 1].sum())
            # 5. survivors everywhere, local edges, MIN over ranks
            surv_all = cm.all_gather_var(eng.survivors(ns), ns_all)
            cm.sync()
            ef, et = eng.edges(surv_all)
            if ef.numel():
                flip = -2 ** 31       # unsigned order on int32 tensors: flip the sign bit around the MIN
                for x in (ef, et):
                    x.bitwise_xor_(flip)
                    cm.all_reduce(x, dist.ReduceOp.MIN)
                    x.bitwise_xor_(flip)
            cm.sync()
            # 6. node numbering + list order
            return eng.finish(ef, et, pre_total, keep_device=True) if keep_device else eng.finish(ef, et, pre_total)
        finally:
            eng.end()
