"""SURVEY §8f-2 on the CPU: the BAM/BGZF/BAI reader and the extraction rules of vdjer_amd/csrc/host/bamx.c.
Decoding is pinned on files written by real samtools (tests/golden/bam, from the reference tree's samtools-1.2/test/mpileup);
the extraction order rules are checked against the independent model of tests/bam_model.py (the reference side of this row
cannot be compiled here: parity of the rules is a restatement, not a pinned oracle)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import bam_model as B

GB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bam")


def test_decode_matches_the_sam_written_by_samtools():
    refs, recs = B.read_all(os.path.join(GB, "ce5b.bam"))
    sam = [l.rstrip("\n").split("\t") for l in open(os.path.join(GB, "ce5b.sam")) if not l.startswith("@")]
    sq = [l.split("\t")[1][3:] for l in open(os.path.join(GB, "ce5b.sam")) if l.startswith("@SQ")]
    assert refs == sq
    assert len(recs) == len(sam) > 5
    for r, s in zip(recs, sam):
        assert r["qname"] == s[0] and r["flag"] == int(s[1]) and r["pos"] == int(s[3]) - 1 and r["mapq"] == int(s[4])
        assert (refs[r["tid"]] if r["tid"] >= 0 else "*") == s[2]
        assert r["seq"] == ("" if s[9] == "*" else s[9])
        if s[10] != "*":
            assert r["qual"] == s[10]
        # reference length of the CIGAR
        import re
        rl = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", s[5]) if op in "MDN=X")
        assert r["end"] == r["pos"] + (rl if s[5] != "*" else 1)


@pytest.mark.parametrize("name", ["ce5b", "mpileup1"])
def test_region_iterator_on_samtools_made_indexes(name):
    """what the iterator hands out == the records that overlap the region, for indexes written by real samtools"""
    path = os.path.join(GB, name + ".bam")
    refs, recs = B.read_all(path)
    assert len(recs) > 5
    rng = np.random.default_rng(1)
    spans = {}
    for r in recs:
        if r["tid"] >= 0:
            lo, hi = spans.get(r["tid"], (1 << 30, 0))
            spans[r["tid"]] = (min(lo, r["pos"]), max(hi, r["end"]))
    regions = []
    for tid, (lo, hi) in spans.items():
        regions += [refs[tid], f"{refs[tid]}:{lo + 1}-{hi}", f"{refs[tid]}:{lo + 1}-{lo + 1}", f"{refs[tid]}:{hi + 10}-{hi + 500}",
                    f"{refs[tid]}:1-{max(1, lo)}"]
        for _ in range(12):
            a = int(rng.integers(max(0, lo - 50), hi + 50))
            b = a + 1 + int(rng.integers(0, max(2, (hi - lo) // 3)))
            regions.append(f"{refs[tid]}:{a + 1:,}-{b}")
    for reg in regions:
        got, _tell = B.query(path, reg)
        nm, beg, end = B._parse_region(reg)
        tid = refs.index(nm)
        want = [(r["qname"], r["pos"]) for r in recs if r["tid"] == tid and r["end"] > beg and end > r["pos"]]
        assert got == want, reg
    with pytest.raises(RuntimeError):
        B.query(path, "no_such_chromosome:1-100")


def _mk(qname, flag, tid, pos, seq, qual=None, cigar=None):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, seq=seq, qual=qual or "I" * len(seq),
                cigar=cigar if cigar is not None else ([("M", len(seq))] if tid >= 0 and not flag & 4 else []))


def _case(tmp_path, records, vdj_text, refs, block):
    bam = str(tmp_path / "x.bam")
    voffs = B.write_bam(bam, refs, records, block=block)
    B.write_bai(bam + ".bai", len(refs), records, voffs)
    fa = str(tmp_path / "ig_vdj.fa")
    open(fa, "w").write(vdj_text)
    return bam, fa, voffs


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


@pytest.mark.parametrize("block", [0xff00, 700, 233])
def test_writer_reader_round_trip(tmp_path, block):
    rng = np.random.default_rng(2)
    refs = [("chr1", 100000), ("chr14", 107043718)]
    recs = [_mk(f"r{i}", 0x41 if i % 2 == 0 else 0x81, 1, 5000 + 37 * i, _rand_seq(rng, 50)) for i in range(60)]
    recs += [_mk(f"u{i}", 0x4d if i % 2 == 0 else 0x8d, -1, -1, _rand_seq(rng, 50).replace("A", "N", 1)) for i in range(40)]
    bam, _fa, voffs = _case(tmp_path, recs, ">x\nACGT\n", refs, block)
    names, got = B.read_all(bam)
    assert names == [r[0] for r in refs] and len(got) == len(recs)
    for th in (2, 5):             # the same records, offsets included, with the blocks inflated ahead by other threads
        assert B.read_all(bam, threads=th) == (names, got)
    for g, r, (u, v) in zip(got, recs, voffs):
        assert (g["qname"], g["flag"], g["tid"], g["pos"], g["seq"], g["qual"]) == (r["qname"], r["flag"], r["tid"], r["pos"], r["seq"], r["qual"])
        assert g["voff"] == u and g["tell"] == v and g["end"] == B.rec_end(r)
        assert g["bin"] == (B.reg2bin(r["pos"], B.rec_end(r)) if r["pos"] >= 0 else 4680)


VR, CR = "chr14:105566277-106879844", "chr14:105566277-105939754"          # set_chain_info, params.c:13-15


def _extraction_case(seed, n_other, n_v, n_c, n_unmapped, dup=True):
    """reads around the IGH loci, reads elsewhere (some carrying a V/D/J 15-mer), unmapped pairs, secondary/supplementary lines"""
    rng = np.random.default_rng(seed)
    vdj = [_rand_seq(rng, 300) for _ in range(3)]
    vdj_text = "".join(f">g{i}\n{s}\n" for i, s in enumerate(vdj)) + ">short\nACGTACGTAC\n"

    def seq_with_kmer():
        g = vdj[int(rng.integers(0, 3))]
        st = int(rng.integers(0, 250))
        s = g[st:st + 50]
        return s if rng.random() < 0.5 else "".join(B._complement(c) for c in reversed(s))
    recs = []

    def pair(name, tid, pos, kmer=False, unmapped=False, rev=False):
        for num, bit in ((1, 0x40), (2, 0x80)):
            s = seq_with_kmer() if kmer and num == 1 else _rand_seq(rng, 50)
            fl = 1 | bit | (4 if unmapped else 0) | (16 if rev and num == 1 else 0)
            recs.append(_mk(name, fl, tid, pos + (0 if num == 1 else 120), s))
    for i in range(n_other):
        pair(f"o{i}", 0, int(rng.integers(1000, 90000)), kmer=(i % 5 == 0))
    for i in range(n_other):
        pair(f"b{i}", 1, int(rng.integers(1000, 105000000)), kmer=(i % 7 == 0), rev=(i % 2 == 0))
    for i in range(n_c):
        pair(f"c{i}", 1, int(rng.integers(105566277 - 200, 105939754)), rev=(i % 3 == 0))
    for i in range(n_v):
        pair(f"v{i}", 1, int(rng.integers(105939754, 106879844)), kmer=(i % 4 == 0))
    for i in range(n_other // 2):
        pair(f"a{i}", 1, int(rng.integers(106879844 + 10, 107000000)), kmer=(i % 3 == 0))
    if dup:      # secondary / supplementary lines and a repeated first-in-pair line: the first qualifying record wins
        recs.append(_mk("v0", 1 | 0x40 | 0x100, 1, 106000000, _rand_seq(rng, 50)))
        recs.append(_mk("v1", 1 | 0x40 | 0x800, 1, 106000010, _rand_seq(rng, 50)))
        recs.append(_mk("v2", 1 | 0x40, 1, 106000020, _rand_seq(rng, 50)))
        recs.append(_mk("c0", 1 | 0x40 | 0x80, 1, 105700000, _rand_seq(rng, 50)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    for i in range(n_unmapped):
        pair(f"u{i}", -1, -1, kmer=(i % 3 == 0), unmapped=True)
    return recs, vdj_text


@pytest.mark.parametrize("seed,n_other,n_v,n_c,n_u,block", [(1, 40, 30, 20, 30, 0xff00), (2, 60, 50, 0, 25, 1500), (3, 30, 0, 25, 20, 800),
                                                              (4, 50, 0, 0, 40, 0xff00), (5, 200, 150, 90, 120, 4096), (6, 10, 3, 2, 0, 300)])
def test_extraction_rules_against_the_model(tmp_path, seed, n_other, n_v, n_c, n_u, block):
    refs = [("chr1", 100000), ("chr14", 107043718)]
    recs, vdj_text = _extraction_case(seed, n_other, n_v, n_c, n_u)
    bam, fa, voffs = _case(tmp_path, recs, vdj_text, refs, block)
    got, info = B.extract(bam, fa, VR, CR)
    want, winfo = B.model_extract(recs, voffs, B.index_of(len(refs), recs, voffs), [r[0] for r in refs], vdj_text, VR, CR)
    assert info == winfo
    assert got == want
    assert len(got) > 0
    assert B.extract(bam, fa, VR, CR, threads=4) == (got, info)          # (`vdjer --t 4`: blocks inflated ahead of the parser in the sequential passes)
    pools = {p for p, *_ in got}
    assert "P" in pools or n_v + n_other == 0
    # the sequential pass starts where the iterators stopped: reads that lie before that point and outside both loci are not
    # scanned for V/D/J 15-mers (bam_read.c:346-374 after :316-342); with hits in the constant region some "o"/"b" pairs that
    # carry a 15-mer are therefore NOT extracted, although they would be if the pass started at the top of the file
    names = {n for _p, n, *_ in got}
    if n_c or n_v:
        assert not any(n.startswith("o") for n in names)


def test_all_unmapped_with_the_locus_in_the_header_scans_the_whole_file(tmp_path):
    """a header that names chr14 and only unmapped reads: both region queries are empty and nothing is skipped"""
    rng = np.random.default_rng(9)
    g = _rand_seq(rng, 200)
    recs = []
    for i in range(30):
        s1 = g[i:i + 50] if i % 3 == 0 else _rand_seq(rng, 50)
        recs.append(_mk(f"@q{i}", 77, -1, -1, s1))
        recs.append(_mk(f"@q{i}", 141, -1, -1, _rand_seq(rng, 50)))
    bam, fa, voffs = _case(tmp_path, recs, f">g\n{g}\n", [("chr14", 107043718)], 0xff00)
    got, info = B.extract(bam, fa, VR, CR)
    assert info["read_len"] == info["max_len"] == 50
    assert [x[1] for x in got] == [r["qname"] for r in recs]                       # file order, both mates
    assert [x[0] for x in got[:2]] == ["P", "P"] and {x[0] for x in got} == {"P", "S"}
    assert [x[2] for x in got[:4]] == [1, 2, 1, 2]


def test_errors(tmp_path):
    rng = np.random.default_rng(3)
    recs = [_mk("a", 77, -1, -1, _rand_seq(rng, 50)), _mk("a", 141, -1, -1, _rand_seq(rng, 50))]
    bam, fa, _ = _case(tmp_path, recs, ">g\nACGT\n", [("chr7", 1000)], 0xff00)
    with pytest.raises(RuntimeError, match="not in the BAM header"):
        B.extract(bam, fa, VR, CR)
    os.remove(bam + ".bai")
    with pytest.raises(RuntimeError, match="cannot open"):
        B.extract(bam, fa, "chr7:1-10", "chr7:1-10")
    assert B.lib().bamx_is_bam(bam.encode()) == 1 and B.lib().bamx_is_bam(fa.encode()) == 0


def _raw_bam(path, stream: bytes):
    with open(path, "wb") as f:
        f.write(B.bgzf_block(stream) + B.bgzf_block(b""))


def test_corrupt_sizes_fail_cleanly(tmp_path):
    """sizes taken from the file are bounded before they reach an allocator (a hostile or damaged BAM / BAI must not crash the
    reader): absurd reference count, reference-name length, record length, and index bin count"""
    import struct
    L = B.lib()
    text = b"@HD\tVN:1.4\n"
    head = b"BAM\1" + struct.pack("<I", len(text)) + text
    cases = {
        "refs": head + struct.pack("<I", 0xFFFFFFF0),
        "name": head + struct.pack("<I", 1) + struct.pack("<I", 0x7FFFFFFF) + b"chr1\0",
    }
    for tag, stream in cases.items():
        p = str(tmp_path / f"{tag}.bam")
        _raw_bam(p, stream)
        assert not L.bamx_open(p.encode()), tag
        assert L.bamx_last_error()
    # a record that claims 3 GB
    good = head + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chr1\0" + struct.pack("<I", 1000)
    p = str(tmp_path / "rec.bam")
    _raw_bam(p, good + struct.pack("<I", 0xC0000000) + b"\0" * 64)
    f = L.bamx_open(p.encode())
    assert f
    r = B.Rec()
    assert L.bamx_read1(f, C.byref(r)) < -1 and b"sane" in L.bamx_last_error()
    L.bamx_close(f)
    # an index that claims 2^31 bins
    pb = str(tmp_path / "x.bam")
    with open(pb + ".bai", "wb") as fh:
        fh.write(b"BAI\1" + struct.pack("<I", 1) + struct.pack("<I", 0x7FFFFFFF))
    assert not L.bamx_index_load(pb.encode())
    assert b"cannot open" not in L.bamx_last_error()
