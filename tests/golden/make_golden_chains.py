#!/usr/bin/env python3
"""Light-chain presets and the stage log (round 2 additions to tests/golden; same rules as make_golden.py: runs only in
the build container, needs oracle/_ref/vdjer_ref = the reference's own sources compiled by oracle/Makefile).

  e2e_igk / e2e_igl   the full CLI with --chain IGK / IGL (set_chain_info, params.c:20-31: conserved J residue F, CDR3
                      window 0-60) on a light-chain-shaped repertoire: vdj_contigs.fa, SAM, vdjer.dot of complete runs
  stage_markers.json  the ELAPSED_SECS marker NAMES the reference prints, in order (status.c:22-32; call sites
                      A2:1387-1473, 1336, 1511-1543), for the e2e_mixed input

Adds its entries to MANIFEST.json (key "e2e_chains", "stage_markers").
"""
from __future__ import annotations

import json
import os
import re
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from vdjer_amd import synth  # noqa: E402
from make_golden import REF, gz_write, run_ref, save_pool  # noqa: E402


def complete_run(wd, args, tries=12):
    """outputs of >= 2 mutually identical runs that scored every root, plus the stderr of the last one"""
    complete, err = [], ""
    for _ in range(tries):
        with open(os.path.join(wd, "out.sam"), "w") as so:
            r = run_ref(args, wd, stdout=so)
        nroots = int(re.search(r"num root nodes: (\d+)", r.stderr).group(1))
        scored = int(re.search(r"HARNESS_ROOTS_SCORED\t(\d+)", r.stderr).group(1))
        if scored == nroots:
            complete.append((open(os.path.join(wd, "vdj_contigs.fa")).read(), open(os.path.join(wd, "out.sam")).read(),
                             open(os.path.join(wd, "vdjer.dot")).read()))
            err = r.stderr
        if len(complete) >= 3:
            break
    assert len(complete) >= 2 and all(c == complete[0] for c in complete), "not enough identical complete runs"
    return complete[0], err, nroots


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    work = tempfile.mkdtemp(prefix="vdjx_golden_chains_")
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    chains = {}
    for tag, chain, seed in (("e2e_igk", "IGK", 131), ("e2e_igl", "IGL", 151)):
        rp = synth.make_repertoire(3, seed=seed, chain=chain, zipf_s=0.2)        # flat abundances: every clone reaches a contig
        pl = synth.make_reads(rp, 14000, noise_frac=0.2, seed=seed + 10)
        wd = os.path.join(work, tag)
        os.makedirs(wd)
        synth.write_ref_dir(rp, os.path.join(wd, "ref"))
        pl.write_reads_file(os.path.join(wd, "reads.txt"))
        save_pool(f"{tag}.npz", rp, pl)
        (fa, sam, dot), _, nroots = complete_run(wd, ["run", "--in", "reads.txt", "--chain", chain, "--ref-dir", "ref", "--ins", "175", "--t", "1"])
        assert fa.count(">") >= 2, f"{tag}: the reference found {fa.count('>')} contigs; pick another seed"
        gz_write(f"{tag}.contigs.fa.gz", fa)
        gz_write(f"{tag}.sam.gz", sam)
        gz_write(f"{tag}.dot.gz", dot)
        chains[tag] = {"chain": chain, "flags": [], "contigs": fa.count(">"), "sam_lines": sam.count("\n"), "roots": nroots}
    man["e2e_chains"] = chains

    # stage log of the e2e_mixed input (the same pool and flags make_golden.py used)
    rp = synth.make_repertoire(6, seed=31)
    pl = synth.make_reads(rp, 9000, noise_frac=0.2, seed=41)
    wd = os.path.join(work, "stage")
    os.makedirs(wd)
    synth.write_ref_dir(rp, os.path.join(wd, "ref"))
    pl.write_reads_file(os.path.join(wd, "reads.txt"))
    _, err, _ = complete_run(wd, ["run", "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--t", "1"])
    names = [l.split("\t")[1] for l in err.splitlines() if l.startswith("ELAPSED_SECS\t")]
    with open(os.path.join(HERE, "stage_markers.json"), "w") as f:
        json.dump({"input": "e2e_mixed", "markers": names}, f, indent=1)
        f.write("\n")
    man["stage_markers"] = {"n": len(names)}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print(json.dumps({"e2e_chains": chains, "markers": names}, indent=1))
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
