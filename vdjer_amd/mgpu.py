"""ctypes binding of vdjer_amd/libvdjmgpu.so -- the multi-GPU driver of `vdjer --gpus N` (csrc/host/vdjx_mgpu.c + vdjx_comm.c: RCCL
over xGMI, or host sockets for ranks that share one device in the tests), so that bench.py --gpus N and the tests time and check the SAME
driver the command line ships (one multi-GPU driver, the product's).  vdjer_amd/shard.py (torch.distributed) stays as the
test-side model of the protocol only.

One process per GPU, started by a launcher (torch.distributed.run): the ranks meet in a directory (vdjx_comm_rendezvous), rank 0 hands
the RCCL id to the others over the control sockets -- exactly what vdjer_main.c does after its forks.  No process is ever re-exec'ed and
the rendezvous happens before or after the GPU is touched alike (sockets only)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import CovParams, VdjxError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdjmgpu.so")
ID_BYTES = 128
_L = None


def lib():
    global _L
    if _L is not None:
        return _L
    _lib.lib()                                   # libvdjx.so first (and torch's HIP runtime before it, see _lib.lib)
    if not os.path.exists(LIB_PATH):
        raise VdjxError(f"{LIB_PATH} is missing: build it with `make -C vdjer_amd/csrc/host`")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int, C.c_size_t
    L.vdjx_comm_last_error.restype = C.c_char_p
    L.vdjx_mgpu_last_error.restype = C.c_char_p
    L.vdjx_comm_rendezvous.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(i32)]
    L.vdjx_comm_unique_id.argtypes = [vp]
    L.vdjx_comm_init.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(i32), vp, C.POINTER(vp)]
    L.vdjx_comm_bytes_sent.argtypes = [vp]
    L.vdjx_comm_bytes_sent.restype = u64
    L.vdjx_mgpu_init.argtypes = [vp, i32, C.POINTER(vp)]
    L.vdjx_mgpu_free.argtypes = [vp]
    L.vdjx_mgpu_free.restype = None
    L.vdjx_mgpu_bytes_sent.argtypes = [vp]
    L.vdjx_mgpu_bytes_sent.restype = u64
    L.vdjx_mgpu_agree_max.argtypes = [vp, u64, C.POINTER(u64)]
    L.vdjx_mgpu_set_read_length.argtypes = [vp, i32]
    L.vdjx_mgpu_set_read_length.restype = None
    L.vdjx_mgpu_kmer_build_pool.argtypes = [vp, vp, vp, i32, i32, i32, u64, C.POINTER(vp)]
    L.vdjx_mgpu_kmer_build_share.argtypes = [vp, vp, vp, i32, i32, i32, vp, u64, C.POINTER(vp)]
    L.vdjx_mgpu_window_score2.argtypes = [vp, vp, vp, sz, i32, C.POINTER(CovParams), vp, vp]
    L.vdjx_mgpu_yield.argtypes = [vp]
    L.vdjx_mgpu_serve_step.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz), C.POINTER(i32)]
    L.vdjx_mgpu_finish.argtypes = [vp]
    _L = L
    return L


def _check(rc: int, what: str, L) -> None:
    if rc != 0:
        raise VdjxError(f"{what} failed ({rc}): {L.vdjx_mgpu_last_error().decode()} / {L.vdjx_comm_last_error().decode()}")


def _rw(fd: int, buf: bytes | None, n: int) -> bytes:
    if buf is not None:
        at = 0
        while at < n:
            at += os.write(fd, buf[at:])
        return buf
    out = b""
    while len(out) < n:
        got = os.read(fd, n - len(out))
        if not got:
            raise VdjxError("the control socket to rank 0 closed")
        out += got
    return out


class Driver:
    """One rank of a multi-GPU job over the C driver.  transport "rccl": every rank on its own device, bulk bytes over RCCL; "host": the
    ranks share one device (tests on the one-GPU box), bulk bytes through host sockets."""

    def __init__(self, ctx, rank: int, world: int, device: int, transport: str = "rccl", rdv_dir: str | None = None):
        self.L = L = lib()
        self.ctx, self.rank, self.world = ctx, rank, world
        rdv_dir = rdv_dir or os.environ.get("VDJX_RDV_DIR") or f"/tmp/vdjx_rdv_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}"
        fds = (C.c_int * max(world, 1))()
        _check(L.vdjx_comm_rendezvous(rdv_dir.encode(), rank, world, 1 if transport == "host" else 0, fds), "vdjx_comm_rendezvous", L)
        rid = (C.c_ubyte * ID_BYTES)()
        if world > 1:                              # the communicator's bootstrap id: rank 0 makes it, the control sockets carry it
            if rank == 0:
                if transport == "rccl":
                    _check(L.vdjx_comm_unique_id(rid), "vdjx_comm_unique_id", L)
                for r in range(1, world):
                    _rw(fds[r], bytes(rid), ID_BYTES)
            else:
                got = _rw(fds[0], None, ID_BYTES)
                C.memmove(rid, got, ID_BYTES)
        elif transport == "rccl":
            _check(L.vdjx_comm_unique_id(rid), "vdjx_comm_unique_id", L)
        cm = C.c_void_p()
        _check(L.vdjx_comm_init(transport.encode(), rank, world, device, fds, rid, C.byref(cm)), "vdjx_comm_init", L)
        self.m = C.c_void_p()
        _check(L.vdjx_mgpu_init(cm, device, C.byref(self.m)), "vdjx_mgpu_init", L)
        self.stride = None

    @property
    def bytes_exchanged(self) -> int:
        return int(self.L.vdjx_mgpu_bytes_sent(self.m))

    def agree_max(self, mine: int) -> int:
        most = C.c_uint64()
        _check(self.L.vdjx_mgpu_agree_max(self.m, int(mine), C.byref(most)), "vdjx_mgpu_agree_max", self.L)
        return int(most.value)

    def kmer_build(self, pool, k: int = 35, mf: int = 3, mq: int = 90, keep_device: bool = False, async_export: bool = False, scan_index=None,
                   total_records: int = 0):
        """collective.  scan_index (device int32/uint32 tensor, ascending) / total_records: the pool is this rank's share of the whole pool;
        without them rank r holds the r-th slice of the scan order (stride = the largest pool of any rank, agreed once)"""
        g = C.c_void_p()
        self.L.vdjx_mgpu_set_read_length(self.m, int(pool.rl))
        if scan_index is not None:
            _check(self.L.vdjx_mgpu_kmer_build_share(self.m, self.ctx.h, pool.h, k, mf, mq, C.c_void_p(scan_index.data_ptr()), int(total_records), C.byref(g)),
                   "vdjx_mgpu_kmer_build_share", self.L)
        else:
            if self.stride is None:
                self.stride = self.agree_max(pool.n_records)
            _check(self.L.vdjx_mgpu_kmer_build_pool(self.m, self.ctx.h, pool.h, k, mf, mq, self.stride, C.byref(g)), "vdjx_mgpu_kmer_build_pool", self.L)
        return self.ctx._export_graph(g, k, keep_device or async_export, async_export)

    def window_score(self, windows, ins: int, e0: int = 52, e1: int = 411, rs: int = 35, ms: int = 48, floor: int = 1):
        """rank 0's call (the others are in serve_step): -> (valid[n], mapped pairs per window over all shares)"""
        raw, n, ln = windows if isinstance(windows, tuple) else self.ctx.pack_strings(windows)
        valid, npairs = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        if n:
            cp = CovParams(e0, e1, rs, ms, ins, ins, floor)
            _check(self.L.vdjx_mgpu_window_score2(self.m, self.ctx.h, raw, n, ln, C.byref(cp), valid.ctypes.data, npairs.ctypes.data), "vdjx_mgpu_window_score2", self.L)
        return valid, npairs

    def yield_step(self) -> None:
        _check(self.L.vdjx_mgpu_yield(self.m), "vdjx_mgpu_yield", self.L)

    def serve_step(self, cap: int):
        """ranks other than 0: serves rank 0's scorer calls until it yields -> (valid, npairs) of the last window call (empty if none), released"""
        valid, npairs = np.zeros(max(cap, 1), np.uint8), np.zeros(max(cap, 1), np.uint32)
        n, rel = C.c_size_t(), C.c_int()
        _check(self.L.vdjx_mgpu_serve_step(self.m, self.ctx.h, valid.ctypes.data, npairs.ctypes.data, cap, C.byref(n), C.byref(rel)), "vdjx_mgpu_serve_step", self.L)
        k_ = min(int(n.value), cap)
        return valid[:k_], npairs[:k_], bool(rel.value)

    def finish(self) -> None:
        _check(self.L.vdjx_mgpu_finish(self.m), "vdjx_mgpu_finish", self.L)

    def close(self) -> None:
        if self.m:
            self.L.vdjx_mgpu_free(self.m)
            self.m = None
