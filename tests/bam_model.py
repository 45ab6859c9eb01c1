"""BAM/BGZF/BAI writer, ctypes view of bamx (libvdjhost.so) and an independent model of V'DJer's read extraction
-- TEST INFRASTRUCTURE for SURVEY §8f-2.

The reference side of the extraction row (bam_read.c over htslib 1.2.1) cannot be compiled here, so the C restatement
(vdjer_amd/csrc/host/bamx.c) is checked two ways:
  * decoding: against BAM/BAI/SAM files written by real samtools (tests/golden/bam, from samtools-1.2/test/mpileup);
  * the order rules of extract (region iterators, where the sequential pass starts, first-0x40/0x80-wins, set precedence):
    against `model_extract` below, written from the same description (bam_read.c:294-446, hts.c:1372-1580)
    but over a list of records with virtual offsets instead of a byte stream -- it shares no code with bamx.c.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
CIGAR_OPS = "MIDNSHP=X"
BLOCK = 0xff00


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_len_of(cigar) -> int:
    return sum(n for op, n in cigar if op in "MDN=X")


def rec_end(r) -> int:
    return r["pos"] + (ref_len_of(r["cigar"]) if r["cigar"] else 1)            # bam_readrec, sam.c:458-467


def encode_record(r) -> bytes:
    name = r["qname"].encode() + b"\0"
    seq, qual = r["seq"], r["qual"]
    cig = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in r["cigar"])
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= SEQ_CODE[ch] << (4 if i % 2 == 0 else 0)
    q = bytes((ord(c) - 33) & 0xFF for c in qual)
    end = rec_end(r)
    b = reg2bin(r["pos"], end) if r["pos"] >= 0 else 4680
    body = struct.pack("<iiIIiiii", r["tid"], r["pos"], (b << 16) | (r.get("mapq", 0) << 8) | len(name), (r["flag"] << 16) | len(r["cigar"]),
                       len(seq), r.get("mtid", -1), r.get("mpos", -1), 0) + name + cig + bytes(packed) + q
    return struct.pack("<I", len(body)) + body


def bgzf_block(data: bytes, level: int = 6) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bam(path: str, refs, records, block: int = BLOCK):
    """refs: [(name, length)]; records: dicts (qname, flag, tid, pos, cigar [(op, n)], seq, qual).  Records are packed into
    BGZF blocks of `block` inflated bytes and may straddle blocks, as bgzf_write does.
    Returns [(start voffset, end voffset)] per record with bgzf_tell's convention (a used-up block reports the next one)."""
    text = "@HD\tVN:1.4\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    head = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for n, l in refs:
        head += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", l)
    stream = bytearray(head)
    spans = []
    for r in records:
        s = len(stream)
        stream += encode_record(r)
        spans.append((s, len(stream)))
    addr = []                                   # file address of every compressed block
    out = bytearray()
    for o in range(0, len(stream), block):
        addr.append(len(out))
        out += bgzf_block(bytes(stream[o:o + block]))
    eof_addr = len(out)
    out += bgzf_block(b"")
    with open(path, "wb") as f:
        f.write(out)

    def voff(u: int) -> int:                     # inflated offset -> virtual offset as bgzf_tell reports it after reading up to u
        bi, off = divmod(u, block)
        if bi >= len(addr) or u >= len(stream):          # a used-up block reports the address of the next one
            return eof_addr << 16
        return (addr[bi] << 16) | off
    return [(voff(s), voff(e)) for s, e in spans]


def write_bai(path: str, n_ref: int, records, voffs):
    """a valid BAI for coordinate-sorted records: one chunk per run of consecutive records of a bin, 16 kb linear index
    (first record overlapping each window), no bin compression, metadata pseudo-bin omitted"""
    out = bytearray(b"BAI\1" + struct.pack("<I", n_ref))
    for tid in range(n_ref):
        bins, lin = {}, {}
        last_bin = None
        for r, (u, v) in zip(records, voffs):
            if r["tid"] != tid or r["pos"] < 0:
                if r["tid"] != tid:
                    last_bin = None
                continue
            end = rec_end(r)
            b = reg2bin(r["pos"], end)
            if b == last_bin:
                bins[b][-1][1] = v
            else:
                bins.setdefault(b, []).append([u, v])
            last_bin = b
            for w in range(r["pos"] >> 14, ((end - 1) >> 14) + 1):
                lin.setdefault(w, u)
        out += struct.pack("<I", len(bins))
        for b in sorted(bins):
            out += struct.pack("<II", b, len(bins[b]))
            for u, v in bins[b]:
                out += struct.pack("<QQ", u, v)
        n_intv = max(lin) + 1 if lin else 0
        out += struct.pack("<I", n_intv)
        for w in range(n_intv):
            out += struct.pack("<Q", lin.get(w, 0))
    with open(path, "wb") as f:
        f.write(out)


# ------------------------------------------------------------------------------------------------------------------
# ctypes view of bamx
# ------------------------------------------------------------------------------------------------------------------
class Rec(C.Structure):
    _fields_ = [("tid", C.c_int32), ("pos", C.c_int32), ("l_qseq", C.c_int32), ("n_cigar", C.c_int32), ("end", C.c_int32),
                ("flag", C.c_uint16), ("bin", C.c_uint16), ("mapq", C.c_uint8), ("qname", C.c_char * 256), ("seq", C.c_char * 1024),
                ("qual", C.c_char * 1024), ("voff", C.c_uint64)]


class Read(C.Structure):
    _fields_ = [("pool", C.c_char), ("name", C.c_char_p), ("read_num", C.c_int), ("is_rev", C.c_int), ("seq", C.c_char_p), ("qual", C.c_char_p),
                ("seq_no", C.c_uint64), ("pool_no", C.c_uint64)]


class Reads(C.Structure):
    _fields_ = [("v", C.POINTER(Read)), ("n", C.c_size_t), ("read_len", C.c_int), ("max_len", C.c_int), ("n_primary_names", C.c_size_t),
                ("n_secondary_names", C.c_size_t), ("n_primary_reads", C.c_uint64), ("n_secondary_reads", C.c_uint64), ("arena", C.c_void_p)]


_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(os.path.join(ROOT, "vdjer_amd", "libvdjhost.so"))
        L.bamx_last_error.restype = C.c_char_p
        L.bamx_open.restype = C.c_void_p
        L.bamx_open.argtypes = [C.c_char_p]
        L.bamx_close.argtypes = [C.c_void_p]
        L.bamx_read1.argtypes = [C.c_void_p, C.POINTER(Rec)]
        L.bamx_tell.restype = C.c_uint64
        L.bamx_tell.argtypes = [C.c_void_p]
        L.bamx_n_ref.argtypes = [C.c_void_p]
        L.bamx_ref_name.restype = C.c_char_p
        L.bamx_ref_name.argtypes = [C.c_void_p, C.c_int]
        L.bamx_index_load.restype = C.c_void_p
        L.bamx_index_load.argtypes = [C.c_char_p]
        L.bamx_index_free.argtypes = [C.c_void_p]
        L.bamx_query.restype = C.c_long
        L.bamx_query.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]
        L.bamx_extract.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Reads)]
        L.bamx_free.argtypes = [C.POINTER(Reads)]
        L.bamx_is_bam.argtypes = [C.c_char_p]
        L.bamx_threads.argtypes = [C.c_int]
        L.bamx_threads.restype = None
        L.bamx_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.bamx_seek.argtypes = [C.c_void_p, C.c_uint64]
        _L = L
    return _L


def read_all(path: str, threads: int = 1):
    """[(dict of the decoded fields)] for every record, plus the reference names (threads > 1: through the read-ahead)"""
    L = lib()
    f = L.bamx_open(path.encode())
    assert f, L.bamx_last_error()
    if threads > 1:
        assert L.bamx_set_threads(f, threads) == 0, L.bamx_last_error()
    refs = [L.bamx_ref_name(f, i).decode() for i in range(L.bamx_n_ref(f))]
    out, r = [], Rec()
    while True:
        rc = L.bamx_read1(f, C.byref(r))
        if rc < 0:
            assert rc == -1, L.bamx_last_error()
            break
        out.append(dict(qname=r.qname.decode(), flag=r.flag, tid=r.tid, pos=r.pos, end=r.end, mapq=r.mapq, seq=r.seq.decode(),
                        qual=r.qual.decode(), voff=r.voff, n_cigar=r.n_cigar, bin=r.bin, tell=int(L.bamx_tell(f))))
    L.bamx_close(f)
    return refs, out


CB = C.CFUNCTYPE(None, C.POINTER(Rec), C.c_void_p)


def query(path: str, region: str):
    """names of the records the region iterator returns, and the virtual offset the file is left at"""
    L = lib()
    f = L.bamx_open(path.encode())
    ix = L.bamx_index_load(path.encode())
    assert f and ix, L.bamx_last_error()
    got = []
    cb = CB(lambda r, _ud: got.append((r.contents.qname.decode(), r.contents.pos)))
    n = L.bamx_query(f, ix, region.encode(), C.cast(cb, C.c_void_p), None)
    tell = int(L.bamx_tell(f))
    L.bamx_index_free(ix)
    L.bamx_close(f)
    if n < 0:
        raise RuntimeError(L.bamx_last_error().decode())
    assert n == len(got)
    return got, tell


def extract(bam: str, vdj_fasta: str, v_region: str, c_region: str, threads: int = 1):
    L = lib()
    L.bamx_threads(threads)
    rs = Reads()
    rc = L.bamx_extract(bam.encode(), vdj_fasta.encode(), v_region.encode(), c_region.encode(), C.byref(rs))
    if rc:
        raise RuntimeError(L.bamx_last_error().decode())
    out = [(rs.v[i].pool.decode(), rs.v[i].name.decode(), rs.v[i].read_num, rs.v[i].is_rev, rs.v[i].seq.decode(), rs.v[i].qual.decode())
           for i in range(rs.n)]
    info = dict(read_len=rs.read_len, max_len=rs.max_len, n_primary_names=rs.n_primary_names, n_secondary_names=rs.n_secondary_names)
    L.bamx_free(C.byref(rs))
    L.bamx_threads(1)
    return out, info


# ------------------------------------------------------------------------------------------------------------------
# the independent model
# ------------------------------------------------------------------------------------------------------------------
def _complement(ch):
    return {"A": "T", "T": "A", "C": "G", "G": "C"}.get(ch, ch)


def model_vdj_kmers(fasta_text: str):
    """load_kmers (bam_read.c:180-204) on the text of ig_vdj.fa; fgets chunks of 1023 characters"""
    kmers = set()
    for line in fasta_text.splitlines(keepends=True):
        chunks = [line[i:i + 1023] for i in range(0, len(line), 1023)]
        for buf in chunks:
            if buf[0] == ">" or len(buf) < 15:
                continue
            buf = buf[:-1]                                  # "Remove newline": whatever the last character is
            rc = "".join(_complement(c) for c in reversed(buf))
            for i in range(0, max(0, len(buf) - 15)):
                kmers.add(buf[i:i + 15])
                kmers.add(rc[i:i + 15])
    return kmers


def _parse_region(s: str):
    name, beg, end = s, 0, 2 ** 31 - 1
    if ":" in s:
        nm, rng = s.rsplit(":", 1)
        rng = rng.replace(",", "")
        parts = rng.split("-")
        if all(p.isdigit() for p in parts if p) and len(parts) <= 2 and parts[0]:
            name = nm
            beg = max(0, int(parts[0]) - 1)
            end = int(parts[1]) if len(parts) == 2 and parts[1] else 2 ** 31 - 1
    return name, beg, end


def _reg2bins(beg, end):
    out = []
    if beg >= end:
        return out
    end = min(end, 1 << 29) - 1
    t, s = 0, 29
    for l in range(6):
        out.extend(range(t + (beg >> s), t + (end >> s) + 1))
        t += 1 << (3 * l)
        s -= 3
    return out


def model_query(records, voffs, index, refs, region, state):
    """the iterator over a record list: `index` = {tid: (bins {bin: [(u, v)]}, linear [voffset])}; `state["pos"]` = index of the next
    record a sequential read would return (the file position); returns the names the iterator hands out"""
    name, beg, end = _parse_region(region)
    tid = refs.index(name)
    bins, lin = index.get(tid, ({}, []))
    lin = list(lin)
    for j in range(1, len(lin)):
        if lin[j] == 0:
            lin[j] = lin[j - 1]

    def loff(b):
        lvl, x = 0, b
        while x:
            lvl, x = lvl + 1, (x - 1) >> 3
        bot = (b - ((1 << (3 * lvl)) - 1) // 7) << ((5 - lvl) * 3)
        return lin[bot] if bot < len(lin) else 0
    b = 4681 + (beg >> 14)
    hit = None
    while True:
        if b in bins:
            hit = b
            break
        first = (((b - 1) >> 3) << 3) + 1
        b = b - 1 if b > first else (b - 1) >> 3
        if b == 0:
            break
    if hit is None and 0 in bins:
        hit = 0
    min_off = loff(hit) if hit is not None else 0
    off = sorted((u, v) for bb in _reg2bins(beg, end) for (u, v) in bins.get(bb, []) if v > min_off)
    names = []
    if not off:
        return names
    kept = [list(off[0])]
    for u, v in off[1:]:
        if kept[-1][1] < v:
            kept.append([u, v])
    for i in range(1, len(kept)):
        if kept[i - 1][1] >= kept[i][0]:
            kept[i - 1][1] = kept[i][0]
    merged = [kept[0]]
    for u, v in kept[1:]:
        if merged[-1][1] >> 16 == u >> 16:
            merged[-1][1] = v
        else:
            merged.append([u, v])
    start_of = {u: i for i, (u, _v) in enumerate(voffs)}
    i_chunk, curr = -1, 0
    pos = state["pos"]
    while True:
        if curr == 0 or curr >= merged[i_chunk][1]:
            if i_chunk == len(merged) - 1:
                break
            if i_chunk < 0 or merged[i_chunk][1] != merged[i_chunk + 1][0]:
                pos = start_of[merged[i_chunk + 1][0]]
                curr = merged[i_chunk + 1][0]
            i_chunk += 1
        if pos >= len(records):
            break
        r = records[pos]
        curr = voffs[pos][1]
        pos += 1
        if r["tid"] != tid or r["pos"] >= end:
            break
        if rec_end(r) > beg and end > r["pos"]:
            names.append(r["qname"])
    state["pos"] = pos
    return names


def model_extract(records, voffs, index, refs, vdj_fasta_text, v_region, c_region):
    kmers = model_vdj_kmers(vdj_fasta_text)
    state = {"pos": 0}
    primary, secondary = set(), set()
    primary.update(model_query(records, voffs, index, refs, v_region, state))
    secondary.update(model_query(records, voffs, index, refs, c_region, state))
    read_len = 0
    for r in records[state["pos"]:]:
        if read_len == 0:
            read_len = len(r["seq"])
        s = r["seq"]
        for i in range(0, max(0, len(s) - 15)):
            if s[i:i + 15] in kmers:
                primary.add(r["qname"])
            elif r["flag"] & 4:
                secondary.add(r["qname"])
    out, seen = [], set()
    for r in records:
        if r["flag"] & 0x900:
            continue
        nm = r["qname"]
        pool = "P" if nm in primary else "S" if nm in secondary else None
        if pool is None:
            continue
        num = 1 if (r["flag"] & 0x40 and (pool, nm, 1) not in seen) else 2 if (r["flag"] & 0x80 and (pool, nm, 2) not in seen) else 0
        if num:
            seen.add((pool, nm, num))
            out.append((pool, nm, num, int(bool(r["flag"] & 16)), r["seq"][:read_len], r["qual"][:read_len]))
    return out, dict(read_len=read_len, max_len=max(len(r["seq"]) for r in records), n_primary_names=len(primary),
                     n_secondary_names=len(secondary))


def index_of(n_ref, records, voffs):
    """the index write_bai writes, as the structure model_query reads"""
    out = {}
    for tid in range(n_ref):
        bins, lin, last = {}, {}, None
        for r, (u, v) in zip(records, voffs):
            if r["tid"] != tid or r["pos"] < 0:
                if r["tid"] != tid:
                    last = None
                continue
            end = rec_end(r)
            b = reg2bin(r["pos"], end)
            if b == last:
                bins[b][-1] = (bins[b][-1][0], v)
            else:
                bins.setdefault(b, []).append((u, v))
            last = b
            for w in range(r["pos"] >> 14, ((end - 1) >> 14) + 1):
                lin.setdefault(w, u)
        n = max(lin) + 1 if lin else 0
        out[tid] = (bins, [lin.get(w, 0) for w in range(n)])
    return out
