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

    def kmer(self, i: int) -> str:
        return self.kmers[i].tobytes().decode()


class Pool:
    def __init__(self, ctx: "Context", handle, rl: int, n_records: int):
        self.ctx, self.h, self.rl, self.n_records = ctx, handle, rl, n_records

    def free(self):
        if self.h:
            _lib.lib().vdjx_pool_free(self.h)
            self.h = None
            if self in getattr(self.ctx, "_pools", []):
                self.ctx._pools.remove(self)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device: int = 0):
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.vdjx_init(device, C.byref(h)), "vdjx_init")
        self.h = h
        self.device = device
        self._pools = []

    def close(self):
        for p in list(getattr(self, "_pools", [])):
            p.free()
        self._pools = []
        if getattr(self, "h", None):
            self.L.vdjx_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.L.vdjx_sync(self.h), "vdjx_sync")

    # ---- a-0
    def pool_load(self, primary: np.ndarray, secondary: np.ndarray, rl: int) -> Pool:
        pri = _c(primary, np.uint8).reshape(-1, 2 * rl + 1)
        sec = _c(secondary, np.uint8).reshape(-1, 2 * rl + 1)
        h = C.c_void_p()
        check(self.L.vdjx_pool_load(self.h, _p(pri), pri.shape[0], _p(sec), sec.shape[0], rl, C.byref(h)), "vdjx_pool_load")
        p = Pool(self, h, rl, pri.shape[0] + sec.shape[0])
        self._pools.append(p)
        return p

    def pool_load_device(self, d_primary: int, n_primary: int, d_secondary: int, n_secondary: int, rl: int) -> Pool:
        """ASCII pools already resident in device memory (raw device pointers, 16-byte aligned)."""
        h = C.c_void_p()
        check(self.L.vdjx_pool_load_device(self.h, C.c_void_p(d_primary), n_primary, C.c_void_p(d_secondary), n_secondary,
                                           rl, C.byref(h)), "vdjx_pool_load_device")
        p = Pool(self, h, rl, n_primary + n_secondary)
        self._pools.append(p)
        return p

    # ---- a-6
    def anchor_sets_load(self, v_codes, j_codes) -> None:
        v = _c(v_codes, np.uint32)
        j = _c(j_codes, np.uint32)
        check(self.L.vdjx_anchor_sets_load(self.h, _p(v), v.shape[0], _p(j), j.shape[0]), "vdjx_anchor_sets_load")

    def anchor_probe(self, contig: str):
        n = max(0, len(contig) - 16)
        ov = np.zeros(n, np.uint8)
        oj = np.zeros(n, np.uint8)
        check(self.L.vdjx_anchor_probe(self.h, contig.encode(), len(contig), _p(ov), _p(oj)), "vdjx_anchor_probe")
        return ov, oj

    # ---- a-1..a-3
    def kmer_build(self, pool: Pool, k: int = 35, mf: int = 3, mq: int = 90, export: bool = True):
        g = C.c_void_p()
        check(self.L.vdjx_kmer_build(self.h, pool.h, k, mf, mq, C.byref(g)), "vdjx_kmer_build")
        try:
            n = int(self.L.vdjx_graph_nodes(g))
            pre = int(self.L.vdjx_graph_pre_nodes(g))
            if not export:
                return n, pre
            out = Graph(k, n, pre, np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32),
                        np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32),
                        np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32), np.zeros((n, k), np.uint8))
            check(self.L.vdjx_graph_export(g, _p(out.first_inst), _p(out.gated_count), _p(out.freq), _p(out.has_v),
                                           _p(out.has_j), _p(out.to_deg), _p(out.to_ids), _p(out.from_deg),
                                           _p(out.from_ids), _p(out.kmers)), "vdjx_graph_export")
            return out
        finally:
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

    # ---- a-8..a-10
    def read_index_build(self, pool: Pool, pair_id, read_num, is_rc, reg_rank, n_pairs: int) -> None:
        a, b, c_, d = _c(pair_id, np.uint32), _c(read_num, np.uint8), _c(is_rc, np.uint8), _c(reg_rank, np.uint32)
        assert a.shape[0] == pool.n_records
        check(self.L.vdjx_read_index_build(self.h, pool.h, _p(a), _p(b), _p(c_), _p(d), n_pairs), "vdjx_read_index_build")

    @staticmethod
    def pack_strings(strings):
        """(raw bytes, n, len) of equally long strings: pass it instead of the list to skip the per-call join/encode"""
        n = len(strings)
        ln = len(strings[0]) if n else 0
        assert all(len(w) == ln for w in strings)
        return ("".join(strings).encode(), n, ln)

    def window_score(self, windows, ins: int, e0: int = 52, e1: int = 411, rs: int = 35, ms: int = 48, floor: int = 1):
        raw, n, ln = windows if isinstance(windows, tuple) else self.pack_strings(windows)
        if n == 0:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        cp = CovParams(e0, e1, rs, ms, ins, ins, floor)
        valid = np.zeros(n, np.uint8)
        npairs = np.zeros(n, np.uint32)
        check(self.L.vdjx_window_score(self.h, raw, n, ln, C.byref(cp), _p(valid), _p(npairs)), "vdjx_window_score")
        return valid, npairs

    def map_emit(self, contigs):
        raw, n, ln = contigs if isinstance(contigs, tuple) else self.pack_strings(contigs)
        offs = np.zeros(n + 1, np.uint64)
        if n == 0:
            return offs, np.zeros(0, PAIR_DTYPE)
        check(self.L.vdjx_map_emit(self.h, raw, n, ln, _p(offs), None), "vdjx_map_emit(count)")
        pairs = np.zeros(int(offs[n]), PAIR_DTYPE)
        check(self.L.vdjx_map_emit(self.h, raw, n, ln, _p(offs), _p(pairs)), "vdjx_map_emit")
        return offs, pairs

    def stat(self, name: str) -> int:
        return int(self.L.vdjx_stat(self.h, name.encode()))

    # ---- profiling
    def profile(self, on: bool = True):
        check(self.L.vdjx_profile_enable(self.h, 1 if on else 0))

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
