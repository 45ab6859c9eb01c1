"""ctypes mirror of the serial host stage (vdjer_amd/csrc/host/vdjh.h, libvdjhost.so): graph -> roots ->
condensation -> contig enumeration -> V/J windows -> acceptance -> vdj_contigs.fa / vdjer.dot / SAM.

`assemble()` takes the scorers as callables, so the same C code runs behind the GPU (`gpu_hooks`) in the product
and behind the CPU oracle in the tests.  The `vdjer` binary (vdjer_amd/csrc/host/vdjer_main.c) is the C front end
of the same library."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdjhost.so")
_lib = None
_libc = C.CDLL(None)
_libc.fwrite.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


class Params(C.Structure):
    _fields_ = [("k", C.c_int), ("min_node_freq", C.c_int), ("min_base_quality", C.c_int), ("min_contig_score", C.c_float),
                ("vj_min_win", C.c_int), ("vj_max_win", C.c_int), ("j_conserved", C.c_int), ("window_span", C.c_int),
                ("j_extension", C.c_int), ("read_filter_floor", C.c_int), ("min_source_homology_score", C.c_int),
                ("filter_read_span", C.c_int), ("filter_mate_span", C.c_int), ("eval_start", C.c_int), ("eval_stop", C.c_int),
                ("window_overlap_check_size", C.c_int), ("insert_len", C.c_int), ("vregion_kmer_size", C.c_int),
                ("read_length", C.c_int), ("threads", C.c_int)]


class GraphS(C.Structure):
    _fields_ = [("n", C.c_size_t), ("k", C.c_int), ("kmers", C.c_void_p), ("freq", C.c_void_p), ("has_v", C.c_void_p),
                ("has_j", C.c_void_p), ("to_deg", C.c_void_p), ("from_deg", C.c_void_p), ("to_ids", C.c_void_p),
                ("from_ids", C.c_void_p)]


ROOT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
WIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
SAM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
VJF_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p)
STATUS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)


class Hooks(C.Structure):
    _fields_ = [("ud", C.c_void_p), ("root_score", ROOT_FN), ("window_score", WIN_FN), ("sam_body", SAM_FN),
                ("v_codes", C.c_void_p), ("nv", C.c_size_t), ("j_codes", C.c_void_p), ("nj", C.c_size_t), ("status", STATUS_FN)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("n_roots", "n_roots_accepted", "n_contig_candidates", "n_windows_scored",
                                          "n_windows_valid", "n_contigs_out")]


class SphBucket(C.Structure):
    _fields_ = [("key", C.c_char_p), ("val", C.c_void_p), ("deleted", C.c_uint8)]


class SphTable(C.Structure):
    _fields_ = [("b", C.POINTER(SphBucket)), ("nbuckets", C.c_size_t), ("num_elements", C.c_size_t), ("num_deleted", C.c_size_t),
                ("keylen", C.c_int), ("enlarge", C.c_float), ("shrink", C.c_float), ("consider_shrink", C.c_int)]


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc", "host"), os.path.join("..", "..", "libvdjhost.so")])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: make -C vdjer_amd/csrc/host")
        L = C.CDLL(LIB_PATH)
        L.vdjh_default_params.argtypes = [C.POINTER(Params)]
        L.vdjh_default_params.restype = None
        L.vdjh_set_chain.argtypes = [C.POINTER(Params), C.c_char_p]
        L.vdjh_assemble.argtypes = [C.POINTER(Params), C.POINTER(GraphS), C.POINTER(Hooks), C.c_char_p, C.c_char_p, C.c_void_p,
                                    C.POINTER(Stats)]
        L.vdjh_node_order.argtypes = [C.POINTER(GraphS), C.c_void_p]
        L.vdjh_node_order.restype = None
        L.vdjh_vjf_search.argtypes = [C.POINTER(Params), C.POINTER(Hooks), C.c_char_p, VJF_CB, C.c_void_p]
        L.vdjh_last_error.restype = C.c_char_p
        L.sph_murmur64a.restype = C.c_uint64
        L.sph_murmur64a.argtypes = [C.c_char_p, C.c_int, C.c_uint64]
        for f in ("sph_init", "sph_free", "sph_resize0"):
            getattr(L, f).restype = None
        L.sph_init.argtypes = [C.POINTER(SphTable), C.c_int, C.c_int]
        L.sph_free.argtypes = [C.POINTER(SphTable)]
        L.sph_resize0.argtypes = [C.POINTER(SphTable)]
        L.sph_map_put.restype = C.c_size_t
        L.sph_map_put.argtypes = [C.POINTER(SphTable), C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]
        L.sph_erase.argtypes = [C.POINTER(SphTable), C.c_char_p]
        L.sph_next.restype = C.c_size_t
        L.sph_next.argtypes = [C.POINTER(SphTable), C.c_size_t]
        L.sph_size.restype = C.c_size_t
        L.sph_size.argtypes = [C.POINTER(SphTable)]
        _lib = L
    return _lib


def make_params(chain: str = "IGH", **kw) -> Params:
    """Defaults of params.c:53-73 + chain presets; keyword names are the CLI flags without dashes (k, mf, mq, mcs, ins ...)."""
    p = Params()
    lib().vdjh_default_params(C.byref(p))
    if lib().vdjh_set_chain(C.byref(p), chain.encode()):
        raise ValueError(lib().vdjh_last_error().decode())
    names = {"k": "k", "mf": "min_node_freq", "mq": "min_base_quality", "mcs": "min_contig_score", "miw": "vj_min_win",
             "maw": "vj_max_win", "ws": "window_span", "jext": "j_extension", "rf": "read_filter_floor",
             "mrs": "min_source_homology_score", "rs": "filter_read_span", "ms": "filter_mate_span", "e0": "eval_start",
             "e1": "eval_stop", "wo": "window_overlap_check_size", "ins": "insert_len", "vk": "vregion_kmer_size", "rl": "read_length", "t": "threads"}
    for k_, v in kw.items():
        setattr(p, names[k_], v)
    if p.min_base_quality >= 255:
        p.min_base_quality = 254
    return p


def _graph_struct(g):
    keep = dict(kmers=np.ascontiguousarray(g.kmers, np.uint8), freq=np.ascontiguousarray(g.freq, np.uint32),
                has_v=np.ascontiguousarray(g.has_v, np.uint8), has_j=np.ascontiguousarray(g.has_j, np.uint8),
                to_deg=np.ascontiguousarray(g.to_deg, np.uint8), from_deg=np.ascontiguousarray(g.from_deg, np.uint8),
                to_ids=np.ascontiguousarray(g.to_ids, np.uint32), from_ids=np.ascontiguousarray(g.from_ids, np.uint32))
    s = GraphS(g.n, g.k, *[keep[f].ctypes.data for f in ("kmers", "freq", "has_v", "has_j", "to_deg", "from_deg", "to_ids", "from_ids")])
    return s, keep


def node_order(g) -> np.ndarray:
    s, keep = _graph_struct(g)
    out = np.zeros(g.n, np.uint32)
    lib().vdjh_node_order(C.byref(s), out.ctypes.data)
    return out


def _hooks(root_score, window_score, sam_body, v_codes, j_codes, status=None):
    vc = np.ascontiguousarray(np.sort(np.asarray(v_codes, dtype=np.uint32)))
    jc = np.ascontiguousarray(np.sort(np.asarray(j_codes, dtype=np.uint32)))
    errs = []

    def c_root(ud, kmers, n, k, thr, out):
        try:
            km = np.ctypeslib.as_array(C.cast(kmers, C.POINTER(C.c_uint8)), shape=(n * k,)).reshape(n, k)
            res = np.asarray(root_score(km, k, thr), dtype=np.uint8)
            C.memmove(out, res.ctypes.data, n)
            return 0
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            return -1

    def c_win(ud, wins, n, ln, valid):
        try:
            buf = C.string_at(wins, n * ln).decode()
            res = np.asarray(window_score([buf[i * ln:(i + 1) * ln] for i in range(n)]), dtype=np.uint8)
            C.memmove(valid, res.ctypes.data, n)
            return 0
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            return -1

    def c_sam(ud, ids, contigs, n, ln, fp):
        try:
            buf = C.string_at(contigs, n * ln).decode()
            text = sam_body([ids[i].decode() for i in range(n)], [buf[i * ln:(i + 1) * ln] for i in range(n)]).encode()
            _libc.fwrite(text, 1, len(text), fp)
            return 0
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            return -1

    fr, fw, fs = ROOT_FN(c_root), WIN_FN(c_win), SAM_FN(c_sam)
    h = Hooks(None, fr, fw, fs, vc.ctypes.data, vc.shape[0], jc.ctypes.data, jc.shape[0])
    fst = None
    if status is not None:
        fst = STATUS_FN(lambda ud, desc: status(desc.decode()))
        h.status = fst
    return h, (vc, jc, fr, fw, fs, fst), errs


def vjf_search(p: Params, contig: str, v_codes, j_codes):
    h, keep, _ = _hooks(None, None, None, v_codes, j_codes)
    out = []
    cb = VJF_CB(lambda ud, w, c: out.append((w.decode(), c.decode())))
    if lib().vdjh_vjf_search(C.byref(p), C.byref(h), contig.encode(), cb, None):
        raise RuntimeError(lib().vdjh_last_error().decode())
    return out


def assemble(p: Params, g, root_score, window_score, sam_body, v_codes, j_codes, fasta_path=None, dot_path=None, sam_path=None,
             status=None):
    """root_score(kmers[n,k] uint8, k, thr) -> 0/1 array; window_score(list of str) -> 0/1 array;
    sam_body(ids, contigs) -> str (the SAM records after the header); status(name): the reference's stage markers."""
    s, keep = _graph_struct(g)
    h, keep2, errs = _hooks(root_score, window_score, sam_body, v_codes, j_codes, status)
    st = Stats()
    fp = _libc.fopen(sam_path.encode(), b"w") if sam_path else None
    try:
        rc = lib().vdjh_assemble(C.byref(p), C.byref(s), C.byref(h), fasta_path.encode() if fasta_path else None,
                                 dot_path.encode() if dot_path else None, fp, C.byref(st))
    finally:
        if fp:
            _libc.fclose(fp)
    if errs:
        raise errs[0]
    if rc:
        raise RuntimeError(lib().vdjh_last_error().decode())
    return {f[0]: getattr(st, f[0]) for f in Stats._fields_}


def gpu_hooks(ctx, pool_np, names, p: Params):
    """The three scorers bound to the GPU context (vdjer_amd.api.Context) -- the product wiring."""
    from . import api

    def root_score(km, k, thr):
        return ctx.root_score(km, k, thr)

    def window_score(wins):
        return ctx.window_score(wins, p.insert_len, e0=p.eval_start, e1=p.eval_stop, rs=p.filter_read_span,
                                ms=p.filter_mate_span, floor=p.read_filter_floor)[0]

    def sam_body(ids, contigs):
        offs, pairs = ctx.map_emit(contigs)
        return api.sam_text(pool_np, names, ids, offs, pairs, p.read_length)

    return root_score, window_score, sam_body
