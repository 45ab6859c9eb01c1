#!/usr/bin/env python3
"""Reads longer than 64 bases through the whole CLI (round 3 addition to tests/golden; same rules as make_golden.py: runs only
in the build container, needs oracle/_ref/vdjer_ref = the reference's own sources compiled by oracle/Makefile).

  e2e_rl100 / e2e_rl151   2x100 and 2x151 bp libraries (the reference takes reads of up to 255 bases, bam_read.c:208): vdj_contigs.fa,
                          SAM, vdjer.dot of complete runs of the compiled reference

Adds its entries to MANIFEST.json under "e2e_chains" (chain IGH).
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from vdjer_amd import synth  # noqa: E402
from make_golden import REF, gz_write, save_pool  # noqa: E402
from make_golden_chains import complete_run  # noqa: E402


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    work = tempfile.mkdtemp(prefix="vdjx_golden_long_")
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    for tag, rl, ins, seed, pairs in (("e2e_rl100", 100, 220, 171, 9000), ("e2e_rl151", 151, 260, 191, 8000)):
        rp = synth.make_repertoire(3, seed=seed, zipf_s=0.2)
        pl = synth.make_reads(rp, pairs, noise_frac=0.2, seed=seed + 10, rl=rl, ins_mean=float(ins), ins_hi=ins + 60)
        wd = os.path.join(work, tag)
        os.makedirs(wd)
        synth.write_ref_dir(rp, os.path.join(wd, "ref"))
        pl.write_reads_file(os.path.join(wd, "reads.txt"))
        flags = ["--ins", str(ins)]
        (fa, sam, dot), _, nroots = complete_run(wd, ["run", "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--t", "1"] + flags)
        print(tag, "contigs", fa.count(">"), "sam lines", sam.count("\n"), "roots", nroots, flush=True)
        assert fa.count(">") >= 2, f"{tag}: the reference found {fa.count('>')} contigs; pick another seed"
        save_pool(f"{tag}.npz", rp, pl)
        gz_write(f"{tag}.contigs.fa.gz", fa)
        gz_write(f"{tag}.sam.gz", sam)
        gz_write(f"{tag}.dot.gz", dot)
        man["e2e_chains"][tag] = {"chain": "IGH", "flags": [], "ins": ins, "rl": rl, "contigs": fa.count(">"), "sam_lines": sam.count("\n"), "roots": nroots}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
