"""Synthetic IgH repertoire / read-pool / ref-dir generator.

The reference ships no runnable inputs (demo BAM is a missing blob, the reference index bundle is a
download), so every workload this repo measures or tests is generated here from a fixed seed.

Conventions reproduced from the reference:
  * pool record  = '0' + rl bases + rl Phred+33 chars, two records per read (as-is, then
    reverse-complement with reversed qualities)           -- bam_read.c:206-244 (add_to_buffer)
  * per-record read info {pair id, read_num, is_rc}        -- quick_map3.c:126-149 (add_read_info)
  * ref-dir files v_index / j_index / ig_vdj.fa / v_region.fa -- params.c:37-51
  * anchor code: 16 bases, A=0 T=1 C=2 G=3, first base most significant -- seq_to_kmer.c:6-46

Transcript design (so that the reference's window finder, vj_filter.c:127-309, accepts a clone):
  V(300) = 99 random non-stop codons + TGT;   V anchor = V[277:293]  (Cys codon 4 nt after it)
  CDR3   = TGT + 3*U[8,20] core + TGG         (length multiple of 3 in [30,66])
  J tail = TGG + 119 random non-stop codons;  J anchor = J[8:24]     (5 nt after the Trp codon)
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_STOPS = {"TAG", "TAA", "TGA"}
_CODONS = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT"]
_NONSTOP = [c for c in _CODONS if c not in _STOPS]
_IDX = {c: i for i, c in enumerate("ACGT")}


def seq_to_int(s: str) -> int:
    """16-mer -> 32-bit anchor code (seq_to_kmer.c:32-46; A0 T1 C2 G3)."""
    val = 0
    code = {"A": 0, "T": 1, "C": 2, "G": 3}
    for ch in s[:16]:
        val = (val << 2) | code[ch]
    return val


def int_to_seq(code: int) -> str:
    """inverse of seq_to_int: 32-bit anchor code -> 16-mer"""
    return "".join("ATCG"[(code >> (2 * (15 - i))) & 3] for i in range(16))


def revcomp(s: str) -> str:
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def _rand_codons(rng: np.random.Generator, n: int) -> str:
    idx = rng.integers(0, len(_NONSTOP), size=n)
    return "".join(_NONSTOP[i] for i in idx)


@dataclass
class Repertoire:
    v_germ: list
    j_germ: list
    clones: list            # transcript strings
    clone_v: list
    clone_j: list
    weights: np.ndarray     # Zipf abundance, sums to 1
    seed: int
    # derived
    v_anchors: list = field(default_factory=list)
    j_anchors: list = field(default_factory=list)
    j_codon: str = "TGG"    # conserved J residue: W (IGH) or F (IGK/IGL: TTC), params.c:8-35

    @property
    def v_region(self) -> str:
        return "".join(self.v_germ)

    def windows(self, window_span: int = 486, j_extension: int = 162) -> list:
        """The 486-nt candidate window the reference derives for each clone (vj_filter.c:265-276)."""
        out = []
        for t, cdr3 in zip(self.clones, self.cdr3s()):
            c = t.find(cdr3)
            start = c - (window_span - (len(cdr3) + j_extension))
            out.append(t[start:start + window_span] if start >= 0 and len(t) - start > window_span else None)
        return out

    def cdr3s(self) -> list:
        out = []
        ja = set(self.j_anchors)
        for t in self.clones:
            c = 297
            w = t.find(self.j_codon, c + 3)
            # the designed Trp (Phe) codon is the in-frame one that starts the J tail
            while (w - c) % 3 != 0 or t[w + 8:w + 24] not in ja:
                w = t.find(self.j_codon, w + 1)
            out.append(t[c:w + 3])
        return out


def make_repertoire(n_clones: int, seed: int = 20261002, n_v: int = 60, n_j: int = 6,
                    zipf_s: float = 1.1, j_codons: int = 119, clone_seed: int | None = None, chain: str = "IGH",
                    private_v: bool = False, private_j: bool = False) -> Repertoire:
    """private_v: every clone has a germline V of its own (n_v is ignored: n_v = n_clones, clone i over V i).  With V segments SHARED
    by hundreds of clones (60 for all) the reference's contig enumeration (A2:939-1061) branches at every somatic mutation
    of every clone of the segment and does not end above ~1.5 M pairs; with private segments a root's paths are its clone's transcript
    and its surviving sequencing errors, and the serial traversal terminates at BASELINE size (10 M pairs / 20,000 clones: DESIGN 5).
    private_j: every clone has a J + constant tail of its own too (n_j = n_clones, clone i over J i).  Shared tails -- six for all clones --
    are covered by EVERY clone that uses them (hundreds of thousands of reads deep at 10 M pairs), so every sequencing error in them
    survives --mf 3 and every accepted root enumerates ~800 contig candidates through them (A2:939-1061: one per error branch within
    --mcs of the main path): 100,000 candidates in the reference's first two minutes of traversal at 10 M pairs / 2,500 clones.
    clone_seed: draw the clones from their own stream while the germline (and with it the ref-dir) stays the one of `seed`:
    several libraries of different clones over one reference (bench.py gives every GPU its own library).
    chain IGK / IGL: the J segment starts with the conserved Phe codon and the CDR3 is short enough for the light-chain
    window (J residue F, CDR3 of 0-60 nt: set_chain_info, params.c:20-31)."""
    if chain not in ("IGH", "IGK", "IGL"):
        raise ValueError(chain)
    j_codon = "TGG" if chain == "IGH" else "TTC"
    core_lo, core_hi = (8, 21) if chain == "IGH" else (7, 18)      # light chains: CDR3 of 27-57 nt (window start >= 0 needs >= 27)
    rng = np.random.default_rng(seed)
    if private_v:
        n_v = n_clones
    if private_j:
        n_j = n_clones
    v_germ = [_rand_codons(rng, 99) + "TGT" for _ in range(n_v)]
    j_germ = [j_codon + _rand_codons(rng, j_codons) for _ in range(n_j)]
    if clone_seed is not None:
        rng = np.random.default_rng(clone_seed)
    clones, cv, cj = [], [], []
    for ci_ in range(n_clones):
        g = ci_ if private_v else int(rng.integers(0, n_v))
        h = ci_ if private_j else int(rng.integers(0, n_j))
        v = list(v_germ[g])
        for _m in range(int(rng.integers(0, 7))):
            # somatic point mutation outside the anchor (277..292) and the Cys codon, never making a stop
            for _try in range(20):
                pos = int(rng.integers(0, 270))
                nb = "ACGT"[int(rng.integers(0, 4))]
                if nb == v[pos]:
                    continue
                c0 = pos - pos % 3
                cod = v[c0:c0 + 3]
                cod[pos % 3] = nb
                if "".join(cod) in _STOPS:
                    continue
                v[pos] = nb
                break
        core = _rand_codons(rng, int(rng.integers(core_lo, core_hi)))
        clones.append("".join(v) + core + j_germ[h])
        cv.append(g)
        cj.append(h)
    ranks = np.arange(1, n_clones + 1, dtype=np.float64)
    w = 1.0 / ranks ** zipf_s
    w /= w.sum()
    rep = Repertoire(v_germ, j_germ, clones, cv, cj, w, seed, j_codon=j_codon)
    rep.v_anchors = [v[277:293] for v in v_germ]
    rep.j_anchors = [j[8:24] for j in j_germ]
    return rep


def write_ref_dir(rep: Repertoire, path: str) -> str:
    """Synthetic --ref-dir (params.c:43-50): v_index, j_index, ig_vdj.fa, v_region.fa."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "v_region.fa"), "w") as f:
        f.write(">v_region\n" + rep.v_region + "\n")
    with open(os.path.join(path, "ig_vdj.fa"), "w") as f:
        for i, v in enumerate(rep.v_germ):
            f.write(f">V{i}\n{v}\n")
        for i, j in enumerate(rep.j_germ):
            f.write(f">J{i}\n{j}\n")
    with open(os.path.join(path, "v_index"), "w") as f:
        for a in sorted(set(rep.v_anchors)):
            f.write(f"{seq_to_int(a)}\t0\n")
    with open(os.path.join(path, "j_index"), "w") as f:
        for a in sorted(set(rep.j_anchors)):
            f.write(f"{seq_to_int(a)}\t0\n")
    return path


@dataclass
class ReadPool:
    """Pools in the reference's a-0 layout plus the per-record read info the quick_map index needs."""
    rl: int
    primary: np.ndarray       # uint8 [Rp, 2*rl+1]
    secondary: np.ndarray     # uint8 [Rs, 2*rl+1]
    # per-record info, primary records first then secondary (== scan order of build_pre_graph)
    pair_id: np.ndarray       # uint32 [R]
    read_num: np.ndarray      # uint8  [R]   1|2
    is_rc: np.ndarray         # uint8  [R]
    reg_rank: np.ndarray      # uint32 [R]   registration order (add_read_info call order)
    n_pairs: int

    @property
    def n_records(self) -> int:
        return self.primary.shape[0] + self.secondary.shape[0]

    def names(self):
        return [f"r{i}" for i in range(self.n_pairs)]

    def write_reads_file(self, path: str) -> None:
        """Extracted-reads text file (one read per line, registration order): the --in format of vdjer_amd/vdjer.
        `P|S r<pair id> <read_num> <is_rc> <SEQ> <QUAL>` per FORWARD record (the reverse-complement record of a read is derived by the
        reader).  Laid out with numpy, a field at a time over all lines (a Python loop over the 20 M reads of a 10 M-pair pool took two
        minutes; round 6's BASELINE-size tests and bench.py --cli-at-size write such files)."""
        rl = self.rl
        npri = self.primary.shape[0]
        order = np.argsort(self.reg_rank, kind="stable")
        order = order[self.reg_rank[order] % 2 == 0]              # forward records, in registration order
        n = int(order.shape[0])
        if n == 0:
            open(path, "wb").close()
            return
        pid = self.pair_id[order].astype(np.int64)
        nd = np.ones(n, np.int64)                                  # decimal digits of the pair id
        t = 10
        while True:
            more = pid >= t
            if not more.any():
                break
            nd += more
            t *= 10
        # P|S ' ' 'r' digits ' ' read_num ' ' is_rc ' ' SEQ ' ' QUAL '\n': a matrix of lines as wide as the longest, the digit field padded
        # with zero bytes on the right, which are dropped when the lines are laid end to end
        D = int(nd.max())
        width = 3 + D + 1 + 1 + 1 + 1 + 1 + rl + 1 + rl + 1
        chunk = 1_000_000
        p10 = 10 ** np.arange(D, dtype=np.int64)
        with open(path, "wb") as f:
            for a in range(0, n, chunk):
                b = min(n, a + chunk)
                rec_idx = order[a:b]
                in_p = rec_idx < npri
                M = np.full((b - a, width), ord(" "), np.uint8)
                M[:, 0] = np.where(in_p, ord("P"), ord("S"))
                M[:, 2] = ord("r")
                nd_, pid_ = nd[a:b], pid[a:b]
                for j in range(D):                                 # digit j from the left (zero byte where the number has fewer)
                    e = nd_ - 1 - j
                    dig = (pid_ // p10[np.clip(e, 0, D - 1)]) % 10
                    M[:, 3 + j] = np.where(e >= 0, ord("0") + dig, 0)
                c0 = 3 + D + 1
                M[:, c0] = ord("0") + self.read_num[rec_idx]
                M[:, c0 + 2] = ord("0") + self.is_rc[rec_idx]
                recs = np.empty((b - a, 2 * rl + 1), np.uint8)
                if in_p.any():
                    recs[in_p] = self.primary[rec_idx[in_p]]
                if (~in_p).any():
                    recs[~in_p] = self.secondary[rec_idx[~in_p] - npri]
                M[:, c0 + 4:c0 + 4 + rl] = recs[:, 1:1 + rl]
                M[:, c0 + 5 + rl:c0 + 5 + 2 * rl] = recs[:, 1 + rl:1 + 2 * rl]
                M[:, width - 1] = ord("\n")
                flat = M.reshape(-1)
                f.write(flat[flat != 0].tobytes())


_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)   # over ACGT indices
_QUAL_CHARS = np.frombuffer(bytes([40 + 33, 30 + 33, 12 + 33]), dtype=np.uint8)


def _qualities(rng, shape):
    u = rng.random(shape, dtype=np.float32)
    q = np.zeros(shape, dtype=np.uint8)
    q[u >= 0.85] = 1
    q[u >= 0.95] = 2
    return _QUAL_CHARS[q]


def make_reads(rep: Repertoire, n_pairs: int, noise_frac: float = 0.3, rl: int = 50, seed: int | None = None,
               ins_mean: float = 175.0, ins_sd: float = 10.0, ins_lo: int = 120, ins_hi: int = 240,
               err: float = 0.002, n_rate: float = 0.001, swap: float = 0.5,
               clean: bool = False, chunk: int = 1 << 20) -> ReadPool:
    """Generate n_pairs paired reads.  Clone pairs go to the primary pool, noise pairs to the secondary
    pool (the reference routes reads with a V/D/J 15-mer to primary and other unmapped reads to
    secondary, bam_read.c:355-380).  clean=True: no errors / Ns / low qualities (tiny goldens)."""
    rng = np.random.default_rng(rep.seed + 7919 if seed is None else seed)
    C = len(rep.clones)
    lens = np.array([len(t) for t in rep.clones], dtype=np.int64)
    lmax = int(lens.max())
    T = np.zeros((C, lmax), dtype=np.uint8)
    for i, t in enumerate(rep.clones):
        T[i, :len(t)] = np.array([_IDX[c] for c in t], dtype=np.uint8)
    Tflat = T.reshape(-1)
    cdf = np.cumsum(rep.weights)
    cdf[-1] = 1.0

    rec_len = 2 * rl + 1
    is_noise_all = rng.random(n_pairs) < noise_frac
    n_noise = int(is_noise_all.sum())
    n_clone = n_pairs - n_noise
    primary = np.empty((4 * n_clone, rec_len), dtype=np.uint8)
    secondary = np.empty((4 * n_noise, rec_len), dtype=np.uint8)
    ar = np.arange(rl, dtype=np.int64)

    pcur = scur = 0
    for c0 in range(0, n_pairs, chunk):
        c1 = min(n_pairs, c0 + chunk)
        m = c1 - c0
        noise = is_noise_all[c0:c1]
        # --- bases (ACGT indices) for both mates, forward orientation of the fragment
        r1 = rng.integers(0, 4, size=(m, rl), dtype=np.uint8)
        r2 = rng.integers(0, 4, size=(m, rl), dtype=np.uint8)
        ci = np.flatnonzero(~noise)
        if ci.size:
            clone = np.searchsorted(cdf, rng.random(ci.size), side="right").astype(np.int64)
            clone = np.minimum(clone, C - 1)
            ins = np.clip(np.rint(rng.normal(ins_mean, ins_sd, ci.size)), max(ins_lo, rl), max(ins_hi, rl)).astype(np.int64)      # (a fragment holds a whole read)
            ins = np.minimum(ins, lens[clone])
            start = (rng.random(ci.size) * (lens[clone] - ins + 1)).astype(np.int64)
            base = clone * lmax + start
            f1 = Tflat[base[:, None] + ar[None, :]]
            f2 = Tflat[(base + ins - rl)[:, None] + ar[None, :]]
            r1[ci] = f1
            r2[ci] = _COMP[f2[:, ::-1]]
            sw = ci[rng.random(ci.size) < swap]
            tmp = r1[sw].copy()
            r1[sw] = r2[sw]
            r2[sw] = tmp
            if not clean and err > 0:
                for r in (r1, r2):
                    e = rng.random((ci.size, rl), dtype=np.float32) < err
                    sub = rng.integers(1, 4, size=(ci.size, rl), dtype=np.uint8)
                    blk = r[ci]
                    blk[e] = (blk[e] + sub[e]) & 3
                    r[ci] = blk
        a1 = ACGT[r1]
        a2 = ACGT[r2]
        if clean:
            q1 = np.full((m, rl), 40 + 33, dtype=np.uint8)
            q2 = np.full((m, rl), 40 + 33, dtype=np.uint8)
        else:
            q1 = _qualities(rng, (m, rl))
            q2 = _qualities(rng, (m, rl))
            if n_rate > 0:
                a1[rng.random((m, rl), dtype=np.float32) < n_rate] = ord("N")
                a2[rng.random((m, rl), dtype=np.float32) < n_rate] = ord("N")
        # --- 4 records per pair: R1, rc(R1), R2, rc(R2)
        recs = np.empty((m, 4, rec_len), dtype=np.uint8)
        recs[:, :, 0] = ord("0")
        comp_ascii = np.arange(256, dtype=np.uint8)
        for a, b in (("A", "T"), ("T", "A"), ("C", "G"), ("G", "C")):
            comp_ascii[ord(a)] = ord(b)
        recs[:, 0, 1:1 + rl] = a1
        recs[:, 0, 1 + rl:] = q1
        recs[:, 1, 1:1 + rl] = comp_ascii[a1[:, ::-1]]
        recs[:, 1, 1 + rl:] = q1[:, ::-1]
        recs[:, 2, 1:1 + rl] = a2
        recs[:, 2, 1 + rl:] = q2
        recs[:, 3, 1:1 + rl] = comp_ascii[a2[:, ::-1]]
        recs[:, 3, 1 + rl:] = q2[:, ::-1]
        pc = recs[~noise].reshape(-1, rec_len)
        sc = recs[noise].reshape(-1, rec_len)
        primary[pcur:pcur + pc.shape[0]] = pc
        secondary[scur:scur + sc.shape[0]] = sc
        pcur += pc.shape[0]
        scur += sc.shape[0]

    # --- per-record info in scan order (primary records then secondary)
    pair_idx = np.arange(n_pairs, dtype=np.uint32)
    pid_p = np.repeat(pair_idx[~is_noise_all], 4)
    pid_s = np.repeat(pair_idx[is_noise_all], 4)
    pair_id = np.concatenate([pid_p, pid_s])
    R = pair_id.shape[0]
    within = np.tile(np.arange(4, dtype=np.uint32), R // 4)
    read_num = (1 + within // 2).astype(np.uint8)
    is_rc = (within & 1).astype(np.uint8)            # unmapped BAM reads: as-is record is_rc=0, rc record 1
    reg_rank = (pair_id.astype(np.uint64) * 4 + within).astype(np.uint32)
    return ReadPool(rl, primary, secondary, pair_id, read_num, is_rc, reg_rank, n_pairs)


def tile_reads(rep: Repertoire, clone_ids, ins: int = 175, copies: int = 3, rl: int = 50, step: int = 1) -> ReadPool:
    """Deterministic dense tiling (every start, fixed insert, `copies` copies, quality 'I'):
    the SURVEY §8c recipe that makes a clone pass the coverage validator."""
    p_recs = []
    pid = []
    n = 0
    comp = str.maketrans("ACGT", "TGCA")
    for c in clone_ids:
        t = rep.clones[c]
        for s in range(0, len(t) - ins + 1, step):
            a = t[s:s + rl]
            b = t[s + ins - rl:s + ins][::-1].translate(comp)
            for _ in range(copies):
                q = "I" * rl
                for seq in (a, a[::-1].translate(comp), b, b[::-1].translate(comp)):
                    p_recs.append(("0" + seq + q).encode())
                    pid.append(n)
                n += 1
    primary = np.frombuffer(b"".join(p_recs), dtype=np.uint8).reshape(-1, 2 * rl + 1).copy()
    secondary = np.zeros((0, 2 * rl + 1), dtype=np.uint8)
    pair_id = np.array(pid, dtype=np.uint32)
    R = pair_id.shape[0]
    within = np.tile(np.arange(4, dtype=np.uint32), R // 4)
    return ReadPool(rl, primary, secondary, pair_id, (1 + within // 2).astype(np.uint8), (within & 1).astype(np.uint8),
                    (pair_id * 4 + within).astype(np.uint32), n)


# ----------------------------------------------------------------------------------------------
# Counter-based generator (BASELINE-sized workloads).
#
# make_reads() above draws from one sequential numpy stream: 15 s per million pairs on one core, and only on the
# host.  The 10 M-pair configurations are generated by the function below instead: every random decision is a
# 64-bit hash of (seed, pair index, slot), evaluated with integer tensor arithmetic only, so the SAME pool comes
# out of torch on the CPU (build container: oracle digests, CPU baseline sample) and on the GPU (bench.py and the
# full-size parity tests generate 10 M pairs in HBM in well under a second).  Same read model as make_reads
# (SURVEY §8d): Zipf clone choice, insert ~ 175 +- 10 clipped to [120, 240], 50 % orientation swap, 0.2 %
# substitutions, Phred {40: 85 %, 30: 10 %, 12: 5 %}, 0.1 % N, `noise_frac` uniform random pairs in the secondary pool.
# ----------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _s64(x: int) -> int:
    """two's-complement int64 view of a 64-bit constant"""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


_SM_A, _SM_B, _GOLD = _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB), _s64(0x9E3779B97F4A7C15)


def _lsr(x, s: int):
    """logical right shift of int64 tensors"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _sm64(x):
    """splitmix64 finalizer on int64 tensors (wrapping multiplication)"""
    x = (x ^ _lsr(x, 30)) * _SM_A
    x = (x ^ _lsr(x, 27)) * _SM_B
    return x ^ _lsr(x, 31)


def _sm64_py(x: int) -> int:
    """the same on Python integers (tests pin the tensor arithmetic against it)"""
    x &= _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


@dataclass
class DeviceReadPool:
    """make_reads_cb(..., device=cuda): the pools stay in HBM (torch uint8 tensors [R, 2*rl+1]); the per-record read
    info is small and lives on the host like ReadPool's."""
    rl: int
    primary: object
    secondary: object
    pair_id: np.ndarray
    read_num: np.ndarray
    is_rc: np.ndarray
    reg_rank: np.ndarray
    n_pairs: int

    @property
    def n_records(self) -> int:
        return int(self.primary.shape[0] + self.secondary.shape[0])

    def to_host(self) -> ReadPool:
        return ReadPool(self.rl, self.primary.cpu().numpy(), self.secondary.cpu().numpy(), self.pair_id, self.read_num,
                        self.is_rc, self.reg_rank, self.n_pairs)


def make_reads_cb(rep: Repertoire, n_pairs: int, noise_frac: float = 0.3, rl: int = 50, seed: int = 20261002,
                  device: str = "cpu", chunk: int | None = None, pair0: int = 0):
    """Counter-based pool of pairs [pair0, pair0 + n_pairs) of the stream `seed`: ReadPool (device='cpu') or
    DeviceReadPool.  Bit-identical on every device; a sub-range equals the same pairs of a larger call."""
    import torch
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"
    if chunk is None:
        chunk = (1 << 20) if on_gpu else (1 << 17)
    C = len(rep.clones)
    lens_np = np.array([len(t) for t in rep.clones], dtype=np.int64)
    lmax = int(lens_np.max())
    T = np.zeros((C, lmax), dtype=np.uint8)
    for i, t in enumerate(rep.clones):
        T[i, :len(t)] = np.frombuffer(t.encode(), dtype=np.uint8)
    lut = np.zeros(256, np.uint8)
    for ch, v in _IDX.items():
        lut[ord(ch)] = v
    Tflat = torch.from_numpy(lut[T].reshape(-1)).to(dev)
    lens = torch.from_numpy(lens_np).to(dev)
    cdf = np.cumsum(rep.weights)
    cdf_i = np.minimum(np.floor(cdf * float(1 << 53)), float((1 << 53) - 1)).astype(np.int64)
    cdf_i[-1] = (1 << 53) - 1
    cdf_t = torch.from_numpy(cdf_i).to(dev)
    acgt = torch.from_numpy(ACGT.copy()).to(dev)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
    qchars = torch.tensor([40 + 33, 30 + 33, 12 + 33], dtype=torch.uint8, device=dev)
    noise_thr = int(round(noise_frac * 65536))
    seedc = _s64(_sm64_py(seed * 0x9E3779B97F4A7C15 + 0x1234567))
    rec_len = 2 * rl + 1
    mate_stride = 64 if rl <= 56 else 256              # (slots of the two mates must not meet; 64 is what the committed digests were made with)
    slot = (torch.arange(2, device=dev, dtype=torch.int64)[:, None] * mate_stride + torch.arange(rl, device=dev, dtype=torch.int64)[None, :] + 8)
    slot = (slot * _GOLD)[None, :, :]                                   # [1, 2, rl]
    ar = torch.arange(rl, device=dev, dtype=torch.int64)
    pri_parts, sec_parts, noise_parts = [], [], []
    for c0 in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - c0)
        p = torch.arange(pair0 + c0, pair0 + c0 + m, device=dev, dtype=torch.int64)
        hp = _sm64(p * _GOLD + seedc)
        noise = (hp & 0xFFFF) < noise_thr
        swap = (_lsr(hp, 16) & 1) == 1
        h2, h3, h4 = _sm64(hp + _GOLD), _sm64(hp + _s64(2 * 0x9E3779B97F4A7C15)), _sm64(hp + _s64(3 * 0x9E3779B97F4A7C15))
        clone = torch.clamp(torch.searchsorted(cdf_t, _lsr(h2, 11), right=True), max=C - 1)
        s4 = (h3 & 0xFFFF) + (_lsr(h3, 16) & 0xFFFF) + (_lsr(h3, 32) & 0xFFFF) + (_lsr(h3, 48) & 0xFFFF)
        # sum of four uniforms: sd = 65536 * sqrt(4/12); scaled to sd 10 around 175 with integer arithmetic
        ins = 175 + torch.div((s4 - 131070) * 17321 + 32768 * 1000, 65536 * 1000, rounding_mode="floor")
        ins = torch.minimum(torch.clamp(ins, max(120, rl), max(240, rl)), lens[clone])
        start = _lsr(_lsr(h4, 32) * (lens[clone] - ins + 1), 32)
        hb = _sm64(hp[:, None, None] + slot)                           # [m, 2, rl]
        base = (hb & 3).to(torch.uint8)                                # noise pairs: uniform bases
        gb = clone * lmax + start
        f1 = Tflat[gb[:, None] + ar[None, :]]
        f2 = comp[Tflat[(gb + ins - rl)[:, None] + (rl - 1 - ar)[None, :]].long()]
        ca = torch.where(swap[:, None], f2, f1)
        cb = torch.where(swap[:, None], f1, f2)
        cl = torch.stack([ca, cb], dim=1)                              # [m, 2, rl]
        err = (_lsr(hb, 26) & 0x1FF) == 0                              # 1/512 = 0.195 %
        sub = (((_lsr(hb, 35) & 0xFFFF) * 3) >> 16).to(torch.uint8) + 1
        cl = torch.where(err, (cl + sub) & 3, cl)
        base = torch.where(noise[:, None, None], base, cl)
        a = acgt[base.long()]
        a = torch.where((_lsr(hb, 16) & 0x3FF) == 0, torch.full_like(a, ord("N")), a)       # 1/1024 = 0.098 %
        qu = _lsr(hb, 8) & 0xFF
        q = qchars[((qu >= 218).to(torch.int64) + (qu >= 243).to(torch.int64))]             # 85.2 / 9.8 / 5.1 %
        recs = torch.empty((m, 4, rec_len), dtype=torch.uint8, device=dev)
        recs[:, :, 0] = ord("0")
        ca_ = torch.full((256,), 0, dtype=torch.uint8, device=dev)
        ca_[:] = torch.arange(256, device=dev, dtype=torch.uint8)
        for x, y in (("A", "T"), ("T", "A"), ("C", "G"), ("G", "C")):
            ca_[ord(x)] = ord(y)
        for mate in (0, 1):
            recs[:, 2 * mate, 1:1 + rl] = a[:, mate]
            recs[:, 2 * mate, 1 + rl:] = q[:, mate]
            recs[:, 2 * mate + 1, 1:1 + rl] = ca_[a[:, mate].flip(1).long()]
            recs[:, 2 * mate + 1, 1 + rl:] = q[:, mate].flip(1)
        pri_parts.append(recs[~noise].reshape(-1, rec_len))
        sec_parts.append(recs[noise].reshape(-1, rec_len))
        noise_parts.append(noise.cpu().numpy())
        del hb, base, cl, a, q, qu, recs, f1, f2, ca, cb, err, sub
    z = torch.zeros((0, rec_len), dtype=torch.uint8, device=dev)
    primary = torch.cat(pri_parts) if pri_parts else z
    secondary = torch.cat(sec_parts) if sec_parts else z
    del pri_parts, sec_parts
    is_noise = np.concatenate(noise_parts) if noise_parts else np.zeros(0, bool)
    pair_idx = np.arange(n_pairs, dtype=np.uint32)
    pair_id = np.concatenate([np.repeat(pair_idx[~is_noise], 4), np.repeat(pair_idx[is_noise], 4)])
    R = pair_id.shape[0]
    within = np.tile(np.arange(4, dtype=np.uint32), R // 4)
    read_num = (1 + within // 2).astype(np.uint8)
    is_rc = (within & 1).astype(np.uint8)
    reg_rank = (pair_id.astype(np.uint64) * 4 + within).astype(np.uint32)
    if on_gpu:
        return DeviceReadPool(rl, primary, secondary, pair_id, read_num, is_rc, reg_rank, n_pairs)
    return ReadPool(rl, primary.numpy(), secondary.numpy(), pair_id, read_num, is_rc, reg_rank, n_pairs)
