"""Host-side Python mirror of the C ABI in include/vdjx.h (one object per opaque handle).

The names follow the reference's call sites: a `Context` owns the V/J anchor sets
(vjf_init), the V-region index (score_seq_init) and the read index (add_read_info); `kmer_build`
stands where build_pre_graph/prune_pre_graph/build_graph2 stand in assemble() (A2:1388-1408);
`root_score` is score_seq (A2:1103); `window_score` is quick_map_process_contig + coverage_is_valid
(A2:841-847); `map_emit` is quick_map_process_contig_file (A2:912).
Everything computes on the GPU through libvdjx.so; nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import CovParams, VdjxError, check

PAIR_DTYPE = np.dtype([("pair_id", "<u4"), ("rec1", "<u4"), ("rec2", "<u4"), ("pos1", "<i2"), ("pos2", "<i2"),
                       ("insert", "<i2"), ("rc1", "u1"), ("rc2", "u1")])
assert PAIR_DTYPE.itemsize == C.sizeof(_lib.Pair)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


@dataclass
class Graph:
    """Result of kmer_build: nodes in creation order + ordered edge lists (1-based ids, list order)."""
    k: int
    n: int
    pre_nodes: int
    first_inst: np.ndarray
    gated_count: np.ndarray
    freq: np.ndarray
    has_v: np.ndarray
    has_j: np.ndarray
    to_deg: np.ndarray
    to_ids: np.ndarray
    from_deg: np.ndarray
    from_ids: np.ndarray
    kmers: np.ndarray        # [n, k] uint8 ASCII
    n_roots: int = 0
    handle: object = None    # the device-resident graph (kept only with keep_device=True); free() releases it
    ctx: object = None

    def kmer(self, i: int) -> str:
        return self.kmers[i].tobytes().decode()

    def wait(self):
        """after kmer_build(..., async_export=True): the host arrays are valid once this returns"""
        if self.handle is not None and getattr(self, "_pending", False):
            check(_lib.lib().vdjx_graph_export_end(self.handle), "vdjx_graph_export_end")
            self._pending = False

    def free(self):
        if self.handle is not None:
            self.wait()
            _lib.lib().vdjx_graph_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pool:
    def __init__(self, ctx: "Context", handle, rl: int, n_records: int):
        self.ctx, self.h, self.rl, self.n_records = ctx, handle, rl, n_records
        self._src = None           # host buffers an asynchronous load still reads

    def wait(self):
        """after pool_load_forward(..., wait=False): the pool is loaded (and well-formed) once this returns"""
        try:
            check(_lib.lib().vdjx_pool_wait(self.h), "vdjx_pool_wait")
        finally:
            self._src = None

    def free(self):
        if self.h:
            _lib.lib().vdjx_pool_free(self.h)      # (waits for a load that is still running)
            self.h = None
            self._src = None
            if self in getattr(self.ctx, "_pools", []):
                self.ctx._pools.remove(self)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """pinned_results=True: result arrays (graph export, mapped pairs) are views of page-locked buffers owned by the context
    and REUSED by the next call of the same kind — the way a C caller keeps one set of result buffers per stage."""

    def __init__(self, device: int = 0, pinned_results: bool = False):
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.vdjx_init(device, C.byref(h)), "vdjx_init")
        self.h = h
        self.device = device
        self._pools = []
        self.pinned_results = pinned_results
        self._pinned = {}            # tag -> (ptr, capacity bytes)
        self._retired = []           # outgrown pinned buffers, freed with the context

    def _result_arrays(self, tag: str, specs):
        """specs: [(shape, dtype)] -> zero-copy numpy arrays; pinned and recycled per `tag` when pinned_results is set"""
        if not self.pinned_results:
            return [np.zeros(shape, dt) for shape, dt in specs]
        sizes = [int(np.prod(shape)) * np.dtype(dt).itemsize for shape, dt in specs]
        need = sum((b + 255) & ~255 for b in sizes) + 256
        ptr, cap = self._pinned.get(tag, (None, 0))
        if cap < need:
            if ptr:
                # arrays of an earlier result may still be views of the old buffer (and an asynchronous export may still be
                # writing into it): it is retired, not freed, until the context closes
                self._retired.append(ptr)
            new = C.c_void_p()
            cap = need + need // 4
            check(self.L.vdjx_host_alloc(self.h, cap, C.byref(new)), "vdjx_host_alloc")
            ptr = new.value
            self._pinned[tag] = (ptr, cap)
        # (the views of the last call with the same shapes over the same buffer are the same arrays: building ten of them costs ~0.1 ms,
        # a tenth of a whole step on a small pool)
        key = (ptr, tuple((tuple(shape), np.dtype(dt).str) for shape, dt in specs))
        views = getattr(self, "_views", None)
        if views is None:
            views = self._views = {}
        hit = views.get(tag)
        if hit is not None and hit[0] == key:
            return list(hit[1])
        out, at = [], ptr
        for (shape, dt), b in zip(specs, sizes):
            raw = (C.c_uint8 * max(b, 1)).from_address(at)
            out.append(np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape))
            at += (b + 255) & ~255
        views[tag] = (key, list(out))
        return out

    def close(self):
        for p in list(getattr(self, "_pools", [])):
            p.free()
        self._pools = []
        if getattr(self, "h", None):
            for ptr, _ in getattr(self, "_pinned", {}).values():
                self.L.vdjx_host_free(self.h, ptr)
            for ptr in getattr(self, "_retired", []):
                self.L.vdjx_host_free(self.h, ptr)
            self._pinned, self._retired = {}, []
            self.L.vdjx_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.L.vdjx_sync(self.h), "vdjx_sync")

    def device_copy(self, d_dst: int, d_src: int, nbytes: int) -> None:
        check(self.L.vdjx_device_copy(self.h, C.c_void_p(d_dst), C.c_void_p(d_src), nbytes), "vdjx_device_copy")

    def read_index_drop(self):
        check(self.L.vdjx_read_index_drop(self.h), "vdjx_read_index_drop")

    def trim(self):
        """hand the workspaces' device memory back (vdjx_trim): a context keeps the peak of its calls otherwise"""
        check(self.L.vdjx_trim(self.h), "vdjx_trim")

    # ---- a-0
    def pool_load(self, primary: np.ndarray, secondary: np.ndarray, rl: int) -> Pool:
        pri = _c(primary, np.uint8).reshape(-1, 2 * rl + 1)
        sec = _c(secondary, np.uint8).reshape(-1, 2 * rl + 1)
        h = C.c_void_p()
        check(self.L.vdjx_pool_load(self.h, _p(pri), pri.shape[0], _p(sec), sec.shape[0], rl, C.byref(h)), "vdjx_pool_load")
        p = Pool(self, h, rl, pri.shape[0] + sec.shape[0])
        self._pools.append(p)
        return p

    @staticmethod
    def packed_read_bytes(rl: int) -> int:
        return int(_lib.lib().vdjx_packed_read_bytes(rl))

    @staticmethod
    def pack_reads(ascii_reads: np.ndarray, rl: int) -> np.ndarray:
        """[n, 2*rl+1] ASCII records of reads ('0' + bases + qualities) -> [n, packed_read_bytes(rl)] in the packed host format
        (vdjx_pack_reads: 2-bit bases, then the quality bytes with bit 7 = not ACGT)"""
        a = _c(ascii_reads, np.uint8).reshape(-1, 2 * rl + 1)
        S = Context.packed_read_bytes(rl)
        out = np.zeros((a.shape[0], S), np.uint8)
        check(_lib.lib().vdjx_pack_reads(_p(a), a.shape[0], rl, _p(out)), "vdjx_pack_reads")
        return out

    def pool_load_packed(self, primary_reads, secondary_reads, rl: int, n_primary: int = None, n_secondary: int = None, wait: bool = True) -> Pool:
        """pool_load_forward for reads in the packed host format ([n, packed_read_bytes(rl)] uint8 arrays, or raw host addresses with counts)"""
        S = self.packed_read_bytes(rl)
        if isinstance(primary_reads, int):
            pp, ps, npri, nsec = C.c_void_p(primary_reads), C.c_void_p(secondary_reads), n_primary, n_secondary
        else:
            pri = _c(primary_reads, np.uint8).reshape(-1, S)
            sec = _c(secondary_reads, np.uint8).reshape(-1, S)
            pp, ps, npri, nsec = _p(pri), _p(sec), pri.shape[0], sec.shape[0]
        h = C.c_void_p()
        fn = self.L.vdjx_pool_load_packed if wait else self.L.vdjx_pool_load_packed_begin
        check(fn(self.h, pp, npri, ps, nsec, rl, C.byref(h)), "vdjx_pool_load_packed")
        p = Pool(self, h, rl, 2 * (npri + nsec))
        if not wait and not isinstance(primary_reads, int):
            p._src = (pri, sec)
        self._pools.append(p)
        return p

    def pool_load_forward(self, primary_reads, secondary_reads, rl: int, n_primary: int = None, n_secondary: int = None,
                          wait: bool = True) -> Pool:
        """the reads as extracted only ([n, 2*rl+1] uint8 arrays or raw host addresses with counts): every read's reverse-complement
        record is derived on the device (record 2i = read i, 2i+1 = its reverse complement)"""
        if isinstance(primary_reads, int):
            pp, ps, npri, nsec = C.c_void_p(primary_reads), C.c_void_p(secondary_reads), n_primary, n_secondary
        else:
            pri = _c(primary_reads, np.uint8).reshape(-1, 2 * rl + 1)
            sec = _c(secondary_reads, np.uint8).reshape(-1, 2 * rl + 1)
            pp, ps, npri, nsec = _p(pri), _p(sec), pri.shape[0], sec.shape[0]
        h = C.c_void_p()
        fn = self.L.vdjx_pool_load_forward if wait else self.L.vdjx_pool_load_forward_begin      # wait=False: Pool.wait() before use
        check(fn(self.h, pp, npri, ps, nsec, rl, C.byref(h)), "vdjx_pool_load_forward")
        p = Pool(self, h, rl, 2 * (npri + nsec))
        if not wait and not isinstance(primary_reads, int):
            # the upload is still reading these buffers (they may be contiguous COPIES made above): the pool keeps them alive
            # until Pool.wait() / free()
            p._src = (pri, sec)
        self._pools.append(p)
        return p

    def pool_load_device(self, d_primary, n_primary: int = None, d_secondary=None, n_secondary: int = None, rl: int = None, keepalive=None) -> Pool:
        """ASCII pools already resident in device memory (16-byte aligned).

        LIFETIME (include/vdjx.h, vdjx_pool_load_device): bases and masks are packed, but the quality characters are NOT copied --
        the pool keeps pointers into the two buffers, which must stay valid and unchanged until Pool.free().  Pass the buffers as
        torch tensors ([records, 2*rl+1] uint8, contiguous) and the Pool holds references to them until it is freed; with raw
        device pointers (ints) the caller answers for the lifetime, or hands the owning objects over in `keepalive`.  A buffer
        that is released or overwritten while the pool lives gives silently wrong quality sums and SAM qualities."""
        owners = [keepalive] if keepalive is not None else []

        def as_ptr(x, n):
            if hasattr(x, "data_ptr"):
                if not x.is_contiguous():
                    raise VdjxError("vdjx_pool_load_device: the buffer must be contiguous")
                owners.append(x)
                return x.data_ptr(), (int(x.shape[0]) if n is None else n), (int(x.shape[1] - 1) // 2 if x.dim() == 2 else None)
            return (int(x) if x is not None else 0), (n or 0), None
        pp, np_, rl_p = as_ptr(d_primary, n_primary)
        ps, ns_, rl_s = as_ptr(d_secondary, n_secondary)
        if rl is None:
            rl = rl_p if rl_p is not None else rl_s
        if rl is None:
            raise VdjxError("vdjx_pool_load_device: rl is needed with raw pointers")
        h = C.c_void_p()
        check(self.L.vdjx_pool_load_device(self.h, C.c_void_p(pp), np_, C.c_void_p(ps), ns_, rl, C.byref(h)), "vdjx_pool_load_device")
        p = Pool(self, h, rl, np_ + ns_)
        p._src = owners or None      # dropped in Pool.free()
        self._pools.append(p)
        return p

    # ---- a-6
    def anchor_sets_load(self, v_codes, j_codes) -> None:
        v = _c(v_codes, np.uint32)
        j = _c(j_codes, np.uint32)
        check(self.L.vdjx_anchor_sets_load(self.h, _p(v), v.shape[0], _p(j), j.shape[0]), "vdjx_anchor_sets_load")

    def anchor_sets_from_anchors(self, v_anchor_codes, j_anchor_codes, am: int = 4) -> None:
        """the sets of anchor_sets_load, built on the device from the anchors themselves (Hamming balls of radius min(am, 5))"""
        v, j = _c(v_anchor_codes, np.uint32), _c(j_anchor_codes, np.uint32)
        check(self.L.vdjx_anchor_sets_from_anchors(self.h, _p(v), v.shape[0], _p(j), j.shape[0], am), "vdjx_anchor_sets_from_anchors")

    def index_generate(self, anchor_codes, start: int = 0, end: int = 2 ** 32 - 1, max_dist: int = 5):
        """rows of a v_index / j_index file (seq_dist.c:49-71) for the codes [start, end]: (codes uint32, distances uint8)"""
        a = _c(anchor_codes, np.uint32)
        n = C.c_uint64()
        check(self.L.vdjx_index_generate(self.h, _p(a), a.shape[0], start, end, max_dist, 0, C.byref(n), None, None), "vdjx_index_generate(count)")
        codes, dists = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint8)
        if n.value:
            check(self.L.vdjx_index_generate(self.h, _p(a), a.shape[0], start, end, max_dist, n.value, C.byref(n), _p(codes), _p(dists)),
                  "vdjx_index_generate")
        return codes, dists

    def anchor_probe(self, contig: str):
        n = max(0, len(contig) - 16)
        ov = np.zeros(n, np.uint8)
        oj = np.zeros(n, np.uint8)
        check(self.L.vdjx_anchor_probe(self.h, contig.encode(), len(contig), _p(ov), _p(oj)), "vdjx_anchor_probe")
        return ov, oj

    # ---- a-1..a-3
    def kmer_build(self, pool: Pool, k: int = 35, mf: int = 3, mq: int = 90, export: bool = True, keep_device: bool = False,
                   async_export: bool = False):
        """async_export (implies keep_device): the copy of the graph into the host arrays runs on a second stream beside whatever
        is done next with the context; call Graph.wait() before reading the arrays"""
        g = C.c_void_p()
        check(self.L.vdjx_kmer_build(self.h, pool.h, k, mf, mq, C.byref(g)), "vdjx_kmer_build")
        if not export:
            try:
                return int(self.L.vdjx_graph_nodes(g)), int(self.L.vdjx_graph_pre_nodes(g))
            finally:
                self.L.vdjx_graph_free(g)
        return self._export_graph(g, k, keep_device or async_export, async_export)

    def _export_graph(self, g, k: int, keep_device: bool = False, async_export: bool = False) -> Graph:
        """copy a finished graph into host arrays; the handle is released unless keep_device (then Graph.free() does it)"""
        keep = False
        try:
            n = int(self.L.vdjx_graph_nodes(g))
            if self.pinned_results and n:
                # ONE transfer of the graph's device block into a page-locked block of the same layout; the arrays are views of it
                # (ten transfers cost a small pool's step a tenth of its time in calls alone)
                offs, nb = (C.c_uint64 * 10)(), C.c_uint64()
                check(self.L.vdjx_graph_block_layout(g, offs, C.byref(nb)), "vdjx_graph_block_layout")
                blk = self._result_arrays("graphblock", [((int(nb.value),), np.uint8)])[0]
                key = (blk.ctypes.data, n, k, tuple(offs))
                hit = getattr(self, "_graph_views", None)
                if hit is None or hit[0] != key:
                    specs = [(np.uint64, (n,)), (np.uint32, (n,)), (np.uint32, (n,)), (np.uint8, (n,)), (np.uint8, (n,)), (np.uint8, (n,)),
                             (np.uint32, (n, 4)), (np.uint8, (n,)), (np.uint32, (n, 4)), (np.uint8, (n, k))]
                    views = [blk[int(o):int(o) + int(np.prod(sh)) * np.dtype(dt).itemsize].view(dt).reshape(sh) for o, (dt, sh) in zip(offs, specs)]
                    hit = self._graph_views = (key, views)
                out = Graph(k, n, int(self.L.vdjx_graph_pre_nodes(g)), *hit[1], n_roots=int(self.L.vdjx_graph_roots(g)))
                fn = self.L.vdjx_graph_export_block_begin if async_export else self.L.vdjx_graph_export_block
                check(fn(g, _p(blk)), "vdjx_graph_export_block")
                if keep_device:
                    out.handle, out.ctx, keep = g, self, True
                    out._pending = async_export
                return out
            out = Graph(k, n, int(self.L.vdjx_graph_pre_nodes(g)), *self._result_arrays("graph", [
                ((n,), np.uint64), ((n,), np.uint32), ((n,), np.uint32), ((n,), np.uint8), ((n,), np.uint8), ((n,), np.uint8),
                ((n, 4), np.uint32), ((n,), np.uint8), ((n, 4), np.uint32), ((n, k), np.uint8)]),
                n_roots=int(self.L.vdjx_graph_roots(g)))
            fn = self.L.vdjx_graph_export_begin if async_export else self.L.vdjx_graph_export
            check(fn(g, _p(out.first_inst), _p(out.gated_count), _p(out.freq), _p(out.has_v), _p(out.has_j), _p(out.to_deg),
                     _p(out.to_ids), _p(out.from_deg), _p(out.from_ids), _p(out.kmers)), "vdjx_graph_export")
            if keep_device:
                out.handle, out.ctx, keep = g, self, True
                out._pending = async_export
            return out
        finally:
            if not keep:
                self.L.vdjx_graph_free(g)

    # ---- a-7
    def vregion_load(self, lines, vk: int = 15) -> None:
        arr = (C.c_char_p * len(lines))(*[s.encode() for s in lines])
        check(self.L.vdjx_vregion_load(self.h, arr, len(lines), vk), "vdjx_vregion_load")

    def root_score(self, kmers, k: int, threshold: int) -> np.ndarray:
        if isinstance(kmers, np.ndarray):
            buf = _c(kmers, np.uint8).reshape(-1, k)
            n = buf.shape[0]
            raw = buf.tobytes()
        else:
            n = len(kmers)
            raw = "".join(kmers).encode()
            assert len(raw) == n * k
        out = np.zeros(n, np.uint8)
        check(self.L.vdjx_root_score(self.h, raw, n, k, threshold, _p(out)), "vdjx_root_score")
        return out

    def root_score_graph(self, graph: Graph, threshold: int, first: int = 0, stride: int = 1, wait: bool = True):
        """scores the roots of a device-resident graph (kmer_build(..., keep_device=True)): (1-based node ids, verdicts).
        On a context made with pinned_results=True the two arrays are views of ONE recycled page-locked buffer (like every pinned
        result of this class): they are valid until the next root_score_graph call of the context -- a second call of the same size
        returns the very same arrays with new contents.  Copy them (np.array(x)) to hold two results at once."""
        if graph.handle is None:
            raise VdjxError("root_score_graph: the graph was not kept on the device (keep_device=True)")
        n = int(self.L.vdjx_root_part(graph.handle, first, stride))
        ids, out = self._result_arrays("roots", [((n,), np.uint32), ((n,), np.uint8)])
        if not wait and self.pinned_results:
            # queued, not waited for: the arrays are valid after root_score_wait() (vdjx_root_score_graph_begin / _end)
            check(self.L.vdjx_root_score_graph_begin(self.h, graph.handle, threshold, first, stride, _p(ids), _p(out)), "vdjx_root_score_graph_begin")
            self._roots_pending = True
            return ids, out
        check(self.L.vdjx_root_score_graph(self.h, graph.handle, threshold, first, stride, _p(ids), _p(out)), "vdjx_root_score_graph")
        return ids, out

    def root_score_wait(self):
        if getattr(self, "_roots_pending", False):
            self._roots_pending = False
            check(self.L.vdjx_root_score_graph_end(self.h), "vdjx_root_score_graph_end")

    # ---- a-8..a-10
    def read_index_build(self, pool: Pool, pair_id, read_num, is_rc, reg_rank, n_pairs: int) -> None:
        a, b, c_, d = _c(pair_id, np.uint32), _c(read_num, np.uint8), _c(is_rc, np.uint8), _c(reg_rank, np.uint32)
        assert a.shape[0] == pool.n_records
        check(self.L.vdjx_read_index_build(self.h, pool.h, _p(a), _p(b), _p(c_), _p(d), n_pairs), "vdjx_read_index_build")

    def read_index_build_device(self, pool: Pool, d_pair_id: int, d_read_num: int, d_is_rc: int, d_reg_rank: int, n_pairs: int, wait: bool = True) -> None:
        """the per-record arrays as raw device pointers (uint32, uint8, uint8, uint32; one entry per pool record).
        wait=False: begun on the index stream, beside the calls that follow (the k-mer build of the same pool); read_index_wait() --
        or the first scorer call -- ends it (vdjx_read_index_build_device_begin / _end)"""
        if not wait:
            check(self.L.vdjx_read_index_build_device_begin(self.h, pool.h, C.c_void_p(d_pair_id), C.c_void_p(d_read_num), C.c_void_p(d_is_rc),
                                                            C.c_void_p(d_reg_rank), n_pairs), "vdjx_read_index_build_device_begin")
            return
        check(self.L.vdjx_read_index_build_device(self.h, pool.h, C.c_void_p(d_pair_id), C.c_void_p(d_read_num), C.c_void_p(d_is_rc),
                                                  C.c_void_p(d_reg_rank), n_pairs), "vdjx_read_index_build_device")

    def read_index_build_begin(self, pool: Pool, pair_id, read_num, is_rc, reg_rank, n_pairs: int) -> None:
        """host arrays, begun (vdjx_read_index_build_begin): they are kept alive here until read_index_wait()"""
        a, b, c_, d = _c(pair_id, np.uint32), _c(read_num, np.uint8), _c(is_rc, np.uint8), _c(reg_rank, np.uint32)
        assert a.shape[0] == pool.n_records
        self._ri_keep = (a, b, c_, d, pool)
        check(self.L.vdjx_read_index_build_begin(self.h, pool.h, _p(a), _p(b), _p(c_), _p(d), n_pairs), "vdjx_read_index_build_begin")

    def read_index_wait(self) -> None:
        try:
            check(self.L.vdjx_read_index_build_end(self.h), "vdjx_read_index_build_end")
        finally:
            self._ri_keep = None

    @staticmethod
    def pack_strings(strings):
        """(raw bytes, n, len) of equally long strings: pass it instead of the list to skip the per-call join/encode"""
        n = len(strings)
        ln = len(strings[0]) if n else 0
        assert all(len(w) == ln for w in strings)
        return ("".join(strings).encode(), n, ln)

    def pin_strings(self, strings, tag: str):
        """pack_strings into a page-locked buffer owned by the context (one per `tag`): the scorers' input then crosses PCIe as a
        DMA instead of being staged by the runtime (a C caller builds its windows in vdjx_host_alloc memory to the same effect)"""
        raw, n, ln = self.pack_strings(strings)
        need = max(len(raw), 1)
        ptr, cap = self._pinned.get("in:" + tag, (None, 0))
        if cap < need:
            if ptr:
                self._retired.append(ptr)
            new = C.c_void_p()
            check(self.L.vdjx_host_alloc(self.h, need, C.byref(new)), "vdjx_host_alloc")
            ptr, cap = new.value, need
            self._pinned["in:" + tag] = (ptr, cap)
        C.memmove(ptr, raw, len(raw))
        return (C.cast(ptr, C.c_char_p), n, ln)

    def pin_array(self, rows: np.ndarray, tag: str):
        """pin_strings for strings that already are the rows of a C-contiguous uint8 matrix (no join / encode)"""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, ln = rows.shape
        need = max(rows.nbytes, 1)
        ptr, cap = self._pinned.get("in:" + tag, (None, 0))
        if cap < need:
            if ptr:
                self._retired.append(ptr)
            new = C.c_void_p()
            check(self.L.vdjx_host_alloc(self.h, need + need // 4, C.byref(new)), "vdjx_host_alloc")
            ptr, cap = new.value, need + need // 4
            self._pinned["in:" + tag] = (ptr, cap)
        C.memmove(ptr, rows.ctypes.data, rows.nbytes)
        return (C.cast(ptr, C.c_char_p), n, ln)

    def pin_rows_take(self, rows: np.ndarray, idx: np.ndarray, tag: str, first: int = 0, length: int | None = None):
        """pin_array(rows[idx, first:first + length]) without intermediate arrays: the chosen rows' slices are copied straight into
        the page-locked buffer (vdjx_host_take_rows)"""
        assert rows.dtype == np.uint8 and rows.ndim == 2 and rows.flags.c_contiguous
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        n = int(idx.shape[0])
        ln = int(rows.shape[1]) - first if length is None else int(length)
        need = max(n * ln, 1)
        ptr, cap = self._pinned.get("in:" + tag, (None, 0))
        if cap < need:
            if ptr:
                self._retired.append(ptr)
            new = C.c_void_p()
            check(self.L.vdjx_host_alloc(self.h, need + need // 4, C.byref(new)), "vdjx_host_alloc")
            ptr, cap = new.value, need + need // 4
            self._pinned["in:" + tag] = (ptr, cap)
        if n and int(idx.max()) >= rows.shape[0]:
            raise IndexError("pin_rows_take: row index out of range")
        check(self.L.vdjx_host_take_rows(ptr, rows.ctypes.data, rows.shape[1], first, ln, idx.ctypes.data, n), "vdjx_host_take_rows")
        return (C.cast(ptr, C.c_char_p), n, ln)

    def window_score(self, windows, ins: int, e0: int = 52, e1: int = 411, rs: int = 35, ms: int = 48, floor: int = 1):
        raw, n, ln = windows if isinstance(windows, tuple) else self.pack_strings(windows)
        if n == 0:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        cp = CovParams(e0, e1, rs, ms, ins, ins, floor)
        valid = np.zeros(n, np.uint8)
        npairs = np.zeros(n, np.uint32)
        check(self.L.vdjx_window_score(self.h, raw, n, ln, C.byref(cp), _p(valid), _p(npairs)), "vdjx_window_score")
        return valid, npairs

    # ---- the halves of window_score for a pool sharded by pair over several GPUs (vdjer_amd/shard.py drives them)
    def window_pairs(self, windows):
        """-> (entries per window's pair list, mapped pairs per window); the lists stay on the device for window_pairs_fetch"""
        raw, n, ln = windows if isinstance(windows, tuple) else self.pack_strings(windows)
        ent, npairs = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        if n:
            check(self.L.vdjx_window_pairs(self.h, raw, n, ln, _p(ent), _p(npairs)), "vdjx_window_pairs")
        return ent, npairs

    def window_pairs_fetch(self, window_ids, d_out: int):
        ids = _c(window_ids, np.uint32)
        check(self.L.vdjx_window_pairs_fetch(self.h, _p(ids), ids.shape[0], C.c_void_p(d_out)), "vdjx_window_pairs_fetch")

    def window_cover(self, n: int, ln: int, rl: int, d_lists: int, nsrc: int, counts, ins: int, e0: int = 52, e1: int = 411, rs: int = 35,
                     ms: int = 48, floor: int = 1) -> np.ndarray:
        """counts [nsrc, n] (source-major) entries per source and window; d_lists: device pointer to the lists in that order"""
        cnt = _c(counts, np.uint32).reshape(-1)
        assert cnt.shape[0] == nsrc * n
        cp = CovParams(e0, e1, rs, ms, ins, ins, floor)
        valid = np.zeros(n, np.uint8)
        if n:
            check(self.L.vdjx_window_cover(self.h, n, ln, rl, C.byref(cp), C.c_void_p(d_lists), nsrc, _p(cnt), _p(valid)), "vdjx_window_cover")
        return valid

    def map_emit(self, contigs, async_copy: bool = False):
        """async_copy: the pairs travel to the host on the copy stream while the caller goes on (two alternating pinned result
        buffers); they are valid after map_emit_wait(), which the next map_emit call also performs"""
        raw, n, ln = contigs if isinstance(contigs, tuple) else self.pack_strings(contigs)
        offs = np.zeros(n + 1, np.uint64)
        if n == 0:
            return offs, np.zeros(0, PAIR_DTYPE)
        self.map_emit_wait()
        check(self.L.vdjx_map_emit(self.h, raw, n, ln, _p(offs), None), "vdjx_map_emit(count)")
        self._pairs_flip = 1 - getattr(self, "_pairs_flip", 0)
        pairs, = self._result_arrays(f"pairs{self._pairs_flip}" if async_copy else "pairs", [((int(offs[n]),), PAIR_DTYPE)])
        if async_copy:
            check(self.L.vdjx_map_emit_begin(self.h, raw, n, ln, _p(offs), _p(pairs)), "vdjx_map_emit_begin")
            self._pairs_pending = True
        else:
            check(self.L.vdjx_map_emit(self.h, raw, n, ln, _p(offs), _p(pairs)), "vdjx_map_emit")
        return offs, pairs

    def map_emit_wait(self):
        if getattr(self, "_pairs_pending", False):
            check(self.L.vdjx_map_emit_end(self.h), "vdjx_map_emit_end")
            self._pairs_pending = False

    def sam_names_load(self, names) -> None:
        """read names by pair id (list of str) for sam_text_device"""
        enc = [s_.encode() for s_ in names]
        off = np.zeros(len(enc) + 1, np.uint64)
        off[1:] = np.cumsum([len(e) for e in enc])
        check(self.L.vdjx_sam_names_load(self.h, b"".join(enc), _p(off), len(enc)), "vdjx_sam_names_load")

    def sam_names_load_raw(self, cat: np.ndarray, off: np.ndarray) -> None:
        """the same from the names laid end to end (uint8) and their offsets (uint64 [n_pairs + 1]): no Python string per pair"""
        cat = np.ascontiguousarray(cat, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        check(self.L.vdjx_sam_names_load(self.h, C.cast(cat.ctypes.data, C.c_char_p), _p(off), off.shape[0] - 1), "vdjx_sam_names_load")

    @staticmethod
    def _ids(contig_ids, n):
        enc = [s_.encode() for s_ in contig_ids]
        off = np.zeros(n + 1, np.uint32)
        off[1:] = np.cumsum([len(e) for e in enc])
        return b"".join(enc), off

    def sam_blocks(self, contigs, contig_ids, d_reg_rank: int):
        """vdjx_sam_blocks: the SAM records of THIS context's pairs for these contigs (a pool sharded by pair), left on the device:
        -> (blocks, bytes, d_keys, d_lens, d_text) -- device pointers owned by the context, valid until the next sam_blocks / sam_text.
        d_reg_rank: device pointer, the GLOBAL registration rank of every record of the index's pool."""
        raw, n, ln = contigs if isinstance(contigs, tuple) else self.pack_strings(contigs)
        ids, off = self._ids(contig_ids, n)
        nb, nby = C.c_uint64(), C.c_uint64()
        dk, dl, dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.L.vdjx_sam_blocks(self.h, raw, n, ln, ids, _p(off), C.c_void_p(d_reg_rank), C.byref(nb), C.byref(nby), C.byref(dk), C.byref(dl), C.byref(dt)),
              "vdjx_sam_blocks")
        return int(nb.value), int(nby.value), dk.value or 0, dl.value or 0, dt.value or 0

    def sam_merge(self, n_blocks: int, n_bytes: int, d_keys: int, d_lens: int, d_text: int) -> bytes:
        """vdjx_sam_merge: the blocks of all sources (source after source) laid out in ascending key order = the reference's order"""
        txt, nb = C.c_char_p(), C.c_uint64()
        check(self.L.vdjx_sam_merge(self.h, n_blocks, n_bytes, C.c_void_p(d_keys), C.c_void_p(d_lens), C.c_void_p(d_text), C.byref(txt), C.byref(nb)), "vdjx_sam_merge")
        addr = C.cast(txt, C.c_void_p).value
        return bytes((C.c_char * nb.value).from_address(addr)) if nb.value else b""

    def sam_text_device(self, contigs, contig_ids) -> bytes:
        """the SAM records of output_mapping (quick_map3.c:152-181) for these contigs, formatted on the device"""
        raw, n, ln = contigs if isinstance(contigs, tuple) else self.pack_strings(contigs)
        enc = [s_.encode() for s_ in contig_ids]
        off = np.zeros(n + 1, np.uint32)
        off[1:] = np.cumsum([len(e) for e in enc])
        txt, nb = C.c_char_p(), C.c_uint64()
        check(self.L.vdjx_sam_text(self.h, raw, n, ln, b"".join(enc), _p(off), C.byref(txt), C.byref(nb)), "vdjx_sam_text")
        # (C.string_at takes a C int: the text of 10 M pairs is 3 GB)
        addr = C.cast(txt, C.c_void_p).value
        return bytes((C.c_char * nb.value).from_address(addr)) if nb.value else b""

    def stat(self, name: str) -> int:
        return int(self.L.vdjx_stat(self.h, name.encode()))

    # ---- profiling
    def profile(self, on: bool = True):
        check(self.L.vdjx_profile_enable(self.h, 1 if on else 0))

    def profile_only(self, name=None) -> None:
        """bracket only the launches of this scope from now on (None: all again)"""
        check(self.L.vdjx_profile_only(self.h, name.encode() if name else None), "vdjx_profile_only")

    def profile_reset(self):
        check(self.L.vdjx_profile_reset(self.h))

    def profile_get(self) -> dict:
        out = {}
        n = self.L.vdjx_profile_count(self.h)
        for i in range(n):
            name = C.c_char_p()
            ms = C.c_double()
            cnt = C.c_uint64()
            check(self.L.vdjx_profile_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = (ms.value, int(cnt.value))
        return out


def sam_text(pool_np, names, contig_ids, offsets, pairs, rl: int) -> str:
    """SAM records of quick_map3.c:152-181 from map_emit's pairs (host formatting; text only)."""
    pri, sec = pool_np
    npri = pri.shape[0]

    def rec(r):
        row = pri[r] if r < npri else sec[r - npri]
        return row[1:1 + rl].tobytes().decode(), row[1 + rl:1 + 2 * rl].tobytes().decode()

    out = []
    for ci, cid in enumerate(contig_ids):
        for q in pairs[int(offsets[ci]):int(offsets[ci + 1])]:
            name = names[int(q["pair_id"])]
            if name.startswith("@"):
                name = name[1:]
            f1 = 1 | 2 | (0x10 if q["rc1"] else 0x20) | 0x40
            f2 = 1 | 2 | (0x10 if q["rc2"] else 0x20) | 0x80
            s1, q1 = rec(int(q["rec1"]))
            s2, q2 = rec(int(q["rec2"]))
            out.append(f"{name}\t{f1}\t{cid}\t{q['pos1']}\t255\t{rl}M\t=\t{q['pos2']}\t{q['insert']}\t{s1}\t{q1}\n")
            out.append(f"{name}\t{f2}\t{cid}\t{q['pos2']}\t255\t{rl}M\t=\t{q['pos1']}\t{q['insert']}\t{s2}\t{q2}\n")
    return "".join(out)
