"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/vdjx.h declares,
and the product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vdjer_amd import _lib
    header = open(os.path.join(ROOT, "include", "vdjx.h")).read()
    declared = sorted(set(re.findall(r"\b(vdjx_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.vdjx_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from vdjer_amd import api
    with pytest.raises(api.VdjxError):
        api.Context(0)


def test_product_never_imports_oracle():
    """nothing under vdjer_amd/ (nor bench-independent product code) imports, includes, links or dlopens oracle/"""
    pkg = os.path.join(ROOT, "vdjer_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|#\s*include\s*[<\"].*oracle|liboracle|vdjx_oracle|oracle/_ref|vdjer_ref", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                assert not bad.search(txt), os.path.join(dp, fn)


def test_struct_layouts_match_header():
    from vdjer_amd import _lib
    assert ctypes.sizeof(_lib.Pair) == 20
    assert ctypes.sizeof(_lib.CovParams) == 28


def test_pack_reads_host_format():
    """vdjx_pack_reads (plain C, no GPU): ASCII records -> the packed host format of vdjx_pool_load_packed: ceil(rl/4) bytes of 2-bit
    bases (A0 T1 C2 G3, seq_to_kmer.c:6-29; first base in the top bits of byte 0; code 0 where the base is not ACGT), then rl quality
    bytes (Phred+33, bit 7 = not ACGT), zero padding to a multiple of 16"""
    from vdjer_amd import api
    assert [api.Context.packed_read_bytes(rl) for rl in (50, 51, 52, 64, 36)] == [64, 64, 80, 80, 48]
    assert api.Context.packed_read_bytes(65) == 0                      # reads of more than 64 bases: vdjx_pool_load_forward
    rl = 10
    rec = np.frombuffer(b"0" + b"ACGTNRTTCA" + b"I5#II+IIII", np.uint8)
    out = api.Context.pack_reads(rec, rl)
    assert out.shape == (1, 16)
    # A0 C2 G3 T1 | N0 R0 T1 T1 | C2 A0 -- --
    assert out[0, :3].tolist() == [0b00101101, 0b00000101, 0b10000000]
    q = out[0, 3:13]
    assert (q & 0x7F).tobytes() == b"I5#II+IIII" and (q >> 7).tolist() == [0, 0, 0, 0, 1, 1, 0, 0, 0, 0]
    assert not out[0, 13:].any()
    with pytest.raises(Exception):
        api.Context.pack_reads(np.frombuffer(b"1" + b"A" * 10 + b"I" * 10, np.uint8), rl)


def test_mgpu_library_exports_its_headers():
    """libvdjmgpu.so (the multi-GPU driver of `vdjer --gpus N` as a library: bench.py --gpus N drives it through vdjer_amd/mgpu.py)
    exports every entry point vdjx_mgpu.h and vdjx_comm.h declare"""
    from vdjer_amd import mgpu
    L = mgpu.lib()
    for hdr in ("vdjx_mgpu.h", "vdjx_comm.h"):
        txt = open(os.path.join(ROOT, "vdjer_amd", "csrc", "host", hdr)).read()
        for s in sorted(set(re.findall(r"\b(vdjx_(?:mgpu|comm)_[a-z0-9_]+)\s*\(", txt))):
            assert hasattr(L, s), s


def _rdv_worker(d, me, G, mesh, q):
    from vdjer_amd import mgpu
    L = mgpu.lib()
    fds = (ctypes.c_int * G)()
    rc = L.vdjx_comm_rendezvous(d.encode(), me, G, mesh, fds)
    row = [int(x) for x in fds]
    got = {}
    if rc == 0:
        for j, fd in enumerate(row):               # every pair says hello both ways over its socket
            if fd >= 0:
                os.write(fd, bytes([me, j]))
        for j, fd in enumerate(row):
            if fd >= 0:
                got[j] = list(os.read(fd, 2))
    q.put((me, rc, [j for j, fd in enumerate(row) if fd >= 0], got))


@pytest.mark.parametrize("G,mesh", [(2, 0), (3, 1), (4, 0), (5, 1)])
def test_comm_rendezvous_of_started_ranks(G, mesh, tmp_path):
    """vdjx_comm_rendezvous: ranks that are processes already (bench.py under torch.distributed.run) meet in a directory and get the
    socket row vdjx_comm_sockets would have given them before a fork -- control pairs with rank 0 only (mesh 0: the RCCL transport) or
    the full mesh (the host transport).  No GPU involved."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdv_worker, args=(str(tmp_path / "rdv"), r, G, mesh, q)) for r in range(G)]
    for p in reversed(ps):                         # (the listeners start last: the callers retry until they are there)
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(G))
    for p in ps:
        p.join(30)
    for me, rc, peers, got in res:
        assert rc == 0
        want = [j for j in range(G) if j != me and (mesh or me == 0 or j == 0)]
        assert peers == want
        assert got == {j: [j, me] for j in want}
