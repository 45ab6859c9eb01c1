"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/vdjx_oracle.h).  The product package `vdjer_amd` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(opt0: bool = False) -> str:
    target = "liboracle_O0.so" if opt0 else "liboracle.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


def lib(opt0: bool = False):
    global _LIB
    if opt0:
        return _bind(C.CDLL(build(True)))
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "vdjx_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = _bind(C.CDLL(path))
    return _LIB


class Pair(C.Structure):
    _fields_ = [("pair_id", C.c_uint32), ("rec1", C.c_uint32), ("rec2", C.c_uint32),
                ("pos1", C.c_int16), ("pos2", C.c_int16), ("insert", C.c_int16),
                ("rc1", C.c_uint8), ("rc2", C.c_uint8)]


PAIR_DTYPE = np.dtype([("pair_id", "<u4"), ("rec1", "<u4"), ("rec2", "<u4"), ("pos1", "<i2"), ("pos2", "<i2"),
                       ("insert", "<i2"), ("rc1", "u1"), ("rc2", "u1")])
assert PAIR_DTYPE.itemsize == C.sizeof(Pair)


def _bind(L):
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    L.vdjo_murmur64a.restype = C.c_uint64
    L.vdjo_murmur64a.argtypes = [C.c_char_p, i32, C.c_uint64]
    L.vdjo_seq_to_int.restype = C.c_uint32
    L.vdjo_seq_to_int.argtypes = [C.c_char_p, C.POINTER(i32)]
    L.vdjo_table_build.restype = vp
    L.vdjo_table_build.argtypes = [vp, sz, vp, sz, i32, i32]
    L.vdjo_table_size.restype = sz
    L.vdjo_table_size.argtypes = [vp]
    L.vdjo_table_prune.restype = sz
    L.vdjo_table_prune.argtypes = [vp, i32, i32]
    L.vdjo_table_export.restype = None
    L.vdjo_table_export.argtypes = [vp, vp, vp, vp, vp]
    L.vdjo_table_free.argtypes = [vp]
    L.vdjo_graph_build.restype = vp
    L.vdjo_graph_build.argtypes = [vp, vp, sz, vp, sz, i32, i32, vp, sz, vp, sz]
    L.vdjo_graph_nodes.restype = sz
    L.vdjo_graph_nodes.argtypes = [vp]
    L.vdjo_graph_export.restype = None
    L.vdjo_graph_export.argtypes = [vp] + [vp] * 8
    L.vdjo_graph_free.argtypes = [vp]
    L.vdjo_scorer_new.restype = vp
    L.vdjo_scorer_new.argtypes = [C.POINTER(C.c_char_p), sz, i32]
    L.vdjo_score_seq.restype = i32
    L.vdjo_score_seq.argtypes = [vp, C.c_char_p, i32, i32]
    L.vdjo_scorer_free.argtypes = [vp]
    L.vdjo_readidx_build.restype = vp
    L.vdjo_readidx_build.argtypes = [vp, sz, vp, sz, i32, vp, vp, vp, vp, C.c_uint32]
    L.vdjo_quick_map.restype = sz
    L.vdjo_quick_map.argtypes = [vp, C.c_char_p, i32, vp, vp, sz]
    L.vdjo_coverage_is_valid.restype = i32
    L.vdjo_coverage_is_valid.argtypes = [i32] * 8 + [vp, sz, i32]
    L.vdjo_sam_pair.restype = i32
    L.vdjo_sam_pair.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(Pair), C.c_char_p]
    L.vdjo_readidx_free.argtypes = [vp]
    L.vdjo_index_rows.restype = sz
    L.vdjo_index_rows.argtypes = [vp, sz, C.c_uint64, C.c_uint64, i32, vp, vp, sz]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def murmur64a(data: bytes, seed: int = 97) -> int:
    return int(lib().vdjo_murmur64a(data, len(data), seed))


def seq_to_int(s: str):
    ok = C.c_int(0)
    v = lib().vdjo_seq_to_int(s.encode(), C.byref(ok))
    return int(v) if ok.value else None


def index_rows(anchors, start: int, end: int, max_dist: int = 5):
    """process_kmers (seq_dist.c:49-71) over the codes [start, end]: (codes uint32, distances uint8), ascending"""
    a = _c(anchors, np.uint32)
    n = int(lib().vdjo_index_rows(_p(a), a.shape[0], start, end, max_dist, None, None, 0))
    codes, dists = np.zeros(n, np.uint32), np.zeros(n, np.uint8)
    lib().vdjo_index_rows(_p(a), a.shape[0], start, end, max_dist, _p(codes), _p(dists), n)
    return codes, dists


def inst_kmer(pool, inst: int, k: int) -> str:
    """ASCII k-mer of an instance index (record << 6 | offset; reads of more than 64 bases: record << 8 | offset) over
    primary-then-secondary records."""
    ob = 8 if pool.rl > 64 else 6
    rec, off = inst >> ob, inst & ((1 << ob) - 1)
    npri = pool.primary.shape[0]
    row = pool.primary[rec] if rec < npri else pool.secondary[rec - npri]
    return row[1 + off:1 + off + k].tobytes().decode()


class KmerTable:
    """a-1/a-2.  After prune(): survivors sorted by first gated instance."""

    def __init__(self, pool, k: int, L=None):
        self.L = L or lib()
        self.pool, self.k = pool, k
        self.pri = _c(pool.primary, np.uint8)
        self.sec = _c(pool.secondary, np.uint8)
        self.h = self.L.vdjo_table_build(_p(self.pri), self.pri.shape[0], _p(self.sec), self.sec.shape[0], pool.rl, k)
        if not self.h:
            raise ValueError("bad k")

    def size(self) -> int:
        return int(self.L.vdjo_table_size(self.h))

    def prune(self, mf: int, mq: int) -> int:
        return int(self.L.vdjo_table_prune(self.h, mf, mq))

    def export(self, with_qs: bool = False):
        n = self.size()
        first = np.zeros(n, np.uint64)
        count = np.zeros(n, np.uint32)
        multi = np.zeros(n, np.uint8)
        qs = np.zeros((n, 50), np.uint8) if with_qs else None
        self.L.vdjo_table_export(self.h, _p(first), _p(count), _p(multi), _p(qs))
        return first, count, multi, qs

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vdjo_table_free(self.h)
            self.h = None


class Graph:
    """a-3: nodes in creation order with ordered edge lists."""

    def __init__(self, table: KmerTable, v_codes, j_codes):
        L = self.L = table.L
        self.table = table
        vc = _c(v_codes, np.uint32)
        jc = _c(j_codes, np.uint32)
        t = table
        self.h = L.vdjo_graph_build(t.h, _p(t.pri), t.pri.shape[0], _p(t.sec), t.sec.shape[0], t.pool.rl, t.k,
                                    _p(vc), vc.shape[0], _p(jc), jc.shape[0])
        n = int(L.vdjo_graph_nodes(self.h))
        self.n = n
        self.first = np.zeros(n, np.uint64)
        self.freq = np.zeros(n, np.uint32)
        self.has_v = np.zeros(n, np.uint8)
        self.has_j = np.zeros(n, np.uint8)
        self.to_deg = np.zeros(n, np.uint8)
        self.to_ids = np.zeros((n, 4), np.uint32)
        self.from_deg = np.zeros(n, np.uint8)
        self.from_ids = np.zeros((n, 4), np.uint32)
        L.vdjo_graph_export(self.h, _p(self.first), _p(self.freq), _p(self.has_v), _p(self.has_j),
                            _p(self.to_deg), _p(self.to_ids), _p(self.from_deg), _p(self.from_ids))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vdjo_graph_free(self.h)
            self.h = None


class RootScorer:
    def __init__(self, lines, vk: int = 15, L=None):
        self.L = L or lib()
        arr = (C.c_char_p * len(lines))(*[s.encode() for s in lines])
        self.h = self.L.vdjo_scorer_new(arr, len(lines), vk)

    def score(self, kmer: str, thr: int) -> int:
        return int(self.L.vdjo_score_seq(self.h, kmer.encode(), len(kmer), thr))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vdjo_scorer_free(self.h)
            self.h = None


class ReadIndex:
    """a-8/a-9/a-10."""

    def __init__(self, pool, L=None):
        self.L = L or lib()
        self.pool = pool
        self.pri = _c(pool.primary, np.uint8)
        self.sec = _c(pool.secondary, np.uint8)
        self.pair_id = _c(pool.pair_id, np.uint32)
        self.read_num = _c(pool.read_num, np.uint8)
        self.is_rc = _c(pool.is_rc, np.uint8)
        self.reg_rank = _c(pool.reg_rank, np.uint32)
        self.h = self.L.vdjo_readidx_build(_p(self.pri), self.pri.shape[0], _p(self.sec), self.sec.shape[0], pool.rl,
                                           _p(self.pair_id), _p(self.read_num), _p(self.is_rc), _p(self.reg_rank),
                                           pool.n_pairs)

    def quick_map(self, contig: str, cap: int = 1 << 20):
        pairs = np.zeros(cap, PAIR_DTYPE)
        starts = np.zeros((2 * cap, 2), np.int32)
        n = int(self.L.vdjo_quick_map(self.h, contig.encode(), len(contig), _p(pairs), _p(starts), cap))
        if n > cap:
            return self.quick_map(contig, cap=n)
        return pairs[:n].copy(), starts[:2 * n].copy()

    def coverage_is_valid(self, starts, contig_len: int, ins: int, rl: int = None, e0: int = 52, e1: int = 411,
                          rs: int = 35, ms: int = 48, floor: int = 1) -> int:
        st = _c(starts, np.int32)
        return int(self.L.vdjo_coverage_is_valid(rl or self.pool.rl, contig_len, e0, e1, rs, ins, ins, floor,
                                                 _p(st), st.shape[0], ms))

    def sam_pair(self, contig_id: str, name: str, pair_row) -> str:
        p = Pair.from_buffer_copy(np.asarray(pair_row).tobytes())
        buf = C.create_string_buffer(2048)
        n = self.L.vdjo_sam_pair(self.h, contig_id.encode(), name.encode(), C.byref(p), buf)
        return buf.raw[:n].decode()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vdjo_readidx_free(self.h)
            self.h = None
