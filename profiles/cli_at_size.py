"""The `vdjer` command line of this build on a larger extracted-reads file (no reference beside it: does it run, how long do its
stages take): python profiles/cli_at_size.py [pairs] [rl]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdjer_amd import synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rep = synth.make_repertoire(max(4, pairs // 500), seed=777)
t0 = time.perf_counter()
pool = synth.make_reads_cb(rep, pairs, noise_frac=0.3, rl=rl, seed=778)
with tempfile.TemporaryDirectory() as td:
    pool.write_reads_file(os.path.join(td, "reads.txt"))
    synth.write_ref_dir(rep, os.path.join(td, "ref"))
    print(f"{pairs} pairs of {rl} bases written in {time.perf_counter() - t0:.1f} s ({os.path.getsize(os.path.join(td, 'reads.txt')) / 1e6:.0f} MB)")
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    t1 = time.perf_counter()
    marks, last = [], t1
    with open(os.path.join(td, "sam.out"), "wb") as so:
        pr = subprocess.Popen([exe, "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", str(max(175, rl + 40)), "--t", "8"],
                              cwd=td, stdout=so, stderr=subprocess.PIPE, text=True, errors="replace")
        tail = []
        for line in pr.stderr:
            tail.append(line.rstrip())
            if line.startswith("ELAPSED_SECS\t"):
                now = time.perf_counter()
                marks.append((line.split("\t")[1], round(now - last, 3)))
                last = now
        rc = pr.wait()
    print("exit", rc, "wall %.2f s" % (time.perf_counter() - t1))
    print(marks)
    fa = os.path.join(td, "vdj_contigs.fa")
    print("contigs", sum(1 for l in open(fa) if l.startswith(">")) if os.path.exists(fa) else None, "SAM bytes", os.path.getsize(os.path.join(td, "sam.out")))
    if rc:
        print("\n".join(tail[-8:]))
