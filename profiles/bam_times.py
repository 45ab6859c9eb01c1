#!/usr/bin/env python3
"""bamx_extract (vdjer --in x.bam: bam_read.c:294-446 restated in csrc/host/bamx.c) on a synthetic coordinate-sorted BAM of ~1 M
reads, by number of inflating threads.  CPU only.  usage: python profiles/bam_times.py [n_other_pairs] [dir]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bam_model as B
from test_bam_cpu import CR, VR, _extraction_case

n_other = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/bamt"
os.makedirs(d, exist_ok=True)
bam, fa = os.path.join(d, f"x{n_other}.bam"), os.path.join(d, "ig_vdj.fa")
refs = [("chr1", 100000), ("chr14", 107043718)]
if not os.path.exists(bam):
    t = time.perf_counter()
    recs, vdj_text = _extraction_case(11, n_other, n_other // 10, n_other // 20, n_other // 4, dup=False)
    voffs = B.write_bam(bam, refs, recs, block=0xff00)
    B.write_bai(bam + ".bai", len(refs), recs, voffs)
    open(fa, "w").write(vdj_text)
    print("made", len(recs), "records in", round(time.perf_counter() - t, 1), "s,", os.path.getsize(bam) >> 20, "MiB")
n_reads = None
for th in (1, 2, 4, 8):
    best = 1e9
    for _ in range(2):
        t = time.perf_counter()
        got, info = B.extract(bam, fa, VR, CR, threads=th)
        best = min(best, time.perf_counter() - t)
    print(f"threads {th}: {best:.3f} s, {len(got)} reads extracted, info {info}")
