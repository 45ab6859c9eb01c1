"""Two platform findings of round 2, as standalone reproducers (-m gpu).

Both led to workarounds that are still in the tree (vdjx_common.h: vdjx_kmer_at_lane; shard.Comm / vdjx_mgpu.c: transfers in
pieces of at most 128 MB).  A workaround proves nothing about the diagnosis, so each claim gets the smallest program that
would show it, marked xfail(strict=False) with the versions it was observed on:

  * if a reproducer FAILS here (= xfail), the platform still behaves as diagnosed;
  * if it PASSES (= xpass), that diagnosis was wrong or the platform was fixed -- and the real cause of the original symptom, if
    any, is still to be found.  The randomised differential tests (tests/test_gpu_fuzz.py), which found the first symptom, and
    the 10 M-pair sharded digest test (tests/test_gpu_fullsize.py), which found the second, keep watching the product paths.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SHIFT_SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned __int128 u128;
// out[4i .. 4i+3] = plain (hi, lo), lane-safe (hi, lo) of ((bhi:blo) >> sh[i]) masked to 2k bits
__global__ void k_shift(const u64* __restrict__ in, const unsigned* __restrict__ sh, int k, size_t n, u64* __restrict__ out) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 bhi = in[2 * i], blo = in[2 * i + 1];
	const unsigned s0 = sh[i];
	// (a) the plain 128-bit shift by a per-lane amount (vdjx_kmer_at)
	u128 v = (((u128) bhi << 64) | blo) >> s0;
	if (k < 64) v &= (((u128) 1) << (2 * k)) - 1;
	out[4 * i] = (u64) (v >> 64);
	out[4 * i + 1] = (u64) v;
	// (b) masks and double shifts only (vdjx_kmer_at_lane)
	const u64 big = 0ull - (u64) (s0 >> 6);
	const u64 x_lo = (bhi & big) | (blo & ~big), x_hi = bhi & ~big;
	const unsigned s = s0 & 63u;
	u64 lo = (x_lo >> s) | ((x_hi << 1) << (63u - s));
	u64 hi = x_hi >> s;
	if (k < 32) { lo &= (1ull << (2 * k)) - 1ull; hi = 0; }
	else if (k < 64) hi &= (1ull << (2 * k - 64)) - 1ull;
	out[4 * i + 2] = hi;
	out[4 * i + 3] = lo;
}
extern "C" int run_shift(const u64* in, const unsigned* sh, int k, size_t n, u64* out, int reps) {
	u64 *d_in, *d_out; unsigned* d_sh;
	if (hipMalloc(&d_in, n * 16) || hipMalloc(&d_sh, n * 4) || hipMalloc(&d_out, n * 32)) return 1;
	hipMemcpy(d_in, in, n * 16, hipMemcpyHostToDevice);
	hipMemcpy(d_sh, sh, n * 4, hipMemcpyHostToDevice);
	for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_shift, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, d_in, d_sh, k, n, d_out);
	if (hipDeviceSynchronize()) return 2;
	hipMemcpy(out, d_out, n * 32, hipMemcpyDeviceToHost);
	hipFree(d_in); hipFree(d_sh); hipFree(d_out);
	return 0;
}
'''


def _rocm_versions():
    out = {}
    try:
        import torch
        out["torch_hip"] = torch.version.hip
    except Exception:  # noqa: BLE001
        pass
    for f in ("/opt/rocm/.info/version", "/opt/rocm/.info/version-dev"):
        if os.path.exists(f):
            out["rocm"] = open(f).read().strip()
            break
    return out


@pytest.mark.xfail(strict=False, reason="round 2 (ROCm 7.0.2 / hipcc 7.2, gfx950): a 128-bit shift by a per-lane amount was blamed for wrong k-mers when a "
                                        "wave's amounts lie on both sides of 64; never isolated.  XPASS = the plain shift is fine in isolation.")
def test_plain_128_bit_shift_by_a_per_lane_amount():
    """random (hi, lo) and per-lane shift amounts straddling 64 inside every wave, as the k-mer extraction of reads with more
    than 32 offsets produces them: plain shift vs the host's integers.  The lane-safe form is checked as well (hard assert)."""
    with tempfile.TemporaryDirectory() as td:
        src, so = os.path.join(td, "shift.hip"), os.path.join(td, "libshift.so")
        open(src, "w").write(_SHIFT_SRC)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src])
        L = C.CDLL(so)
        L.run_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
        rng = np.random.default_rng(59093)
        n = 1 << 22
        bad_plain = bad_safe = 0
        for k in (25, 35, 50, 31, 33):
            vals = rng.integers(0, 1 << 63, size=(n, 2), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 2), dtype=np.uint64)
            # amounts 2*(rl-k-o): even, 0 .. 126, neighbours in a wave far apart
            sh = (rng.integers(0, 64, size=n, dtype=np.uint32) * np.uint32(2)).astype(np.uint32)
            out = np.zeros((n, 4), np.uint64)
            assert L.run_shift(vals.ctypes.data, sh.ctypes.data, k, n, out.ctypes.data, 20) == 0
            hi, lo, s = vals[:, 0].astype(object), vals[:, 1].astype(object), sh.astype(object)
            sel = np.arange(0, n, 37)                      # (Python integers are slow: a sample of 113 k lanes per k)
            full = [((int(hi[i]) << 64) | int(lo[i])) >> int(s[i]) & ((1 << (2 * k)) - 1) for i in sel]
            exp_hi = np.array([v >> 64 for v in full], dtype=np.uint64)
            exp_lo = np.array([v & ((1 << 64) - 1) for v in full], dtype=np.uint64)
            bad_plain += int(((out[sel, 0] != exp_hi) | (out[sel, 1] != exp_lo)).sum())
            bad_safe += int(((out[sel, 2] != exp_hi) | (out[sel, 3] != exp_lo)).sum())
        print("versions:", _rocm_versions(), "plain mismatches:", bad_plain, "lane-safe mismatches:", bad_safe)
        assert bad_safe == 0, "the lane-safe extraction itself is wrong"
        assert bad_plain == 0, f"{bad_plain} wrong results of the plain per-lane 128-bit shift"


@pytest.mark.xfail(strict=False, reason="round 2 (ROCm 7.0.2 / RCCL 2.26.6): a single 1.09 GB all_to_all_single on a one-rank communicator came back with its "
                                        "second half wrong.  XPASS = an unchunked transfer of that size is fine (the chunking then hid a caller-side bug).")
def test_unchunked_1088MB_all_to_all_single_one_rank():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dev = torch.device("cuda", 0)
    mine = not dist.is_initialized()
    if mine:
        dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
    try:
        for shape in (((1088 << 20) // 32, 4), ((1088 << 20) // 8,)):              # 32-byte records (the partials' shape), then flat
            x = torch.arange(0, (1088 << 20) // 8, dtype=torch.int64, device=dev).view(*shape)
            y = torch.zeros_like(x)
            dist.all_to_all_single(y, x, [x.shape[0]], [x.shape[0]])
            torch.cuda.synchronize()
            same = bool(torch.equal(x, y))
            if not same:
                first_bad = int(torch.nonzero((x != y).reshape(-1))[0])
                print("versions:", _rocm_versions(), f"shape {shape}: first wrong element {first_bad} of {x.numel()}")
            assert same, f"all_to_all_single of {x.numel() * 8 >> 20} MB in one piece: output differs"
            del x, y
    finally:
        if mine:
            dist.destroy_process_group()


_ARENA_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from vdjer_amd import api, synth
rep = synth.make_repertoire(6, seed=5)
vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
ctx = api.Context(0)
ctx.anchor_sets_load(vc, jc)
out = []
for pairs, k in ((3000, 35), (40000, 25), (3000, 35), (20000, 35)):     # calls of different sizes through one workspace
    pool = synth.make_reads(rep, pairs, noise_frac=0.3, seed=7 + pairs)
    p = ctx.pool_load(pool.primary, pool.secondary, pool.rl)
    g = ctx.kmer_build(p, k, 3, 90)
    out.append(hashlib.sha256(b"".join(np.ascontiguousarray(getattr(g, f)).tobytes() for f in ("first_inst", "freq", "to_ids", "from_ids"))).hexdigest())
    ctx.read_index_build(p, pool.pair_id, pool.read_num, pool.is_rc, pool.reg_rank, pool.n_pairs)
    w = [t[:486] for t in rep.clones if len(t) >= 486][:4]
    if w:
        v, n = ctx.window_score(w, 175)
        out.append(str(v.tolist()) + str(n.tolist()))
    p.free()
print("DIGEST", hashlib.sha256("|".join(out).encode()).hexdigest())
"""


@pytest.mark.gpu
def test_workspace_as_one_mapped_range_and_as_chunk_list_agree():
    """The workspace arena is one reserved address range backed piece by piece (hipMemAddressReserve / hipMemCreate / hipMemMap);
    where the runtime has no virtual memory management it falls back to a list of hipMalloc'ed chunks (VDJX_ARENA_CHUNKS=1 forces
    that).  Builds and scorer calls of different sizes through ONE context give the same results either way."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("range", "chunks"):
        env = dict(os.environ)
        env.pop("VDJX_ARENA_CHUNKS", None)
        if mode == "chunks":
            env["VDJX_ARENA_CHUNKS"] = "1"
        env["VDJX_ARENA_TRACE"] = "1"
        r = subprocess.run([os.sys.executable, "-c", _ARENA_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
        assert line, r.stdout[-500:]
        got[mode] = line[0]
        said = "one reserved address range" if mode == "range" else "list of chunks"
        assert said in r.stderr, r.stderr[-1500:]
    assert got["range"] == got["chunks"]
