#!/usr/bin/env python3
"""Mid-scale end-to-end goldens: digests of what the compiled reference (oracle/_ref/vdjer_ref, --t 1) writes for pools of
hundreds of thousands of pairs -- tens of contigs, thousands of roots, hundreds of candidate windows, several growth steps of
the order-defining sparsehash tables (A2:775-914, vj_filter.c:209-309) -- where tests/golden/e2e_* stop at 9 k pairs.

Runs ONLY in the build container (needs `make -C oracle ref`).  The inputs are not stored: they regenerate from the counter-based
generator (vdjer_amd/synth.py: make_repertoire + make_reads_cb, bit-identical on CPU and GPU), so a case is its generator
parameters, its flags and the SHA-256 / sizes of vdj_contigs.fa, the SAM on stdout, vdjer.dot and the per-root verdict log.

Complete-run rule (as make_golden.py): the reference loses its last root(s) to a wake-up race even at --t 1; a run
counts only if the harness saw as many roots scored as the reference reported, and the case is accepted only when at least two
complete runs exist and ALL complete runs agree byte for byte.

    python tests/golden/make_golden_midscale.py [case ...]      # -> tests/golden/midscale.json
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vdjer_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "vdjer_ref")
OUT = os.path.join(HERE, "midscale.json")

# name -> generator + flags.  mid_400k is the sample bench.py times the reference on (cpu_baseline / cli_end_to_end);
# mid_cfg1 is BASELINE.json configs[1] whole (1 M pairs, 2,000 clones); mid_k25 is the sensitive mode of configs[3] at a size the
# reference's serial traversal still finishes.
CASES = {
    "mid_400k": dict(pairs=400_000, clones=800, seed=20261002, noise=0.3, chain="IGH", ins=175,
                     flags=["--k", "35", "--mf", "3", "--mq", "90", "--mrs", "30"]),
    "mid_cfg1": dict(pairs=1_000_000, clones=2000, seed=20261002, noise=0.3, chain="IGH", ins=175,
                     flags=["--k", "35", "--mf", "3", "--mq", "90", "--mrs", "30"]),
    "mid_k25": dict(pairs=200_000, clones=400, seed=20261002, noise=0.3, chain="IGH", ins=175,
                    flags=["--k", "25", "--mf", "2", "--mq", "60", "--mcs", "-5.5", "--mrs", "20"]),
    # BASELINE.json configs[3] AS WRITTEN (README's sensitive mode, no --mrs: the default 30 stands): at k = 25 the root DP is bounded by
    # 25 < 30, so no root can pass (seq_score.c:92-116, params.c:66, A2:1103): empty FASTA, header-less SAM, every verdict 0
    # BASELINE.json configs[2]'s SIZE on a repertoire the reference's serial traversal finishes (VERDICT r5 missing #4): 10 M pairs, every clone
    # with a germline V AND a J + constant tail of its own (no contig enumeration across clones that share a segment, A2:939-1061) and an
    # abundance flat enough (Zipf 0.25 over 2,500 clones: 400x mean coverage) that the sequencing errors of a clone rarely reach --mf 3:
    # thousands of contigs, the reference's whole run ~20 min at --t 1.  What does not end here, measured: SURVEY 8d's C3 repertoire (20,000
    # clones over 60 shared V segments, Zipf 1.1) -- 200 of ~55,000 accepted roots in the first three minutes of traversal, the deep clones'
    # error branches each a window to map; and private V segments over the six SHARED J + constant tails -- every clone covers its tail, so
    # every sequencing error in a tail survives and every root enumerates ~800 candidates through it: 100,000 in two minutes
    "cfg2_pv": dict(pairs=10_000_000, clones=2500, seed=20261002, noise=0.3, chain="IGH", ins=175, private_v=True, private_j=True, zipf_s=0.25,
                    flags=["--k", "35", "--mf", "3", "--mq", "90", "--mrs", "30"], attempts=4, parallel=2),
    # the same pool in the sensitive mode of BASELINE.json configs[3] (--mrs 20 so that roots can pass): --mf 2 lets a clone's
    # sequencing errors survive (tens of error branches per clone), so the traversal enumerates and maps many more windows than cfg2_pv
    "cfg3_pv": dict(pairs=10_000_000, clones=2500, seed=20261002, noise=0.3, chain="IGH", ins=175, private_v=True, private_j=True, zipf_s=0.25,
                    flags=["--k", "25", "--mf", "2", "--mq", "60", "--mcs", "-5.5", "--mrs", "20"], attempts=3, parallel=1),
    # BASELINE.json configs[4]'s other two chains at its per-GPU size (12.5 M pairs): the light-chain presets of set_chain_info (params.c:20-31:
    # J residue F, CDR3 window 0-60), private repertoire as above
    "cfg4_igk_pv": dict(pairs=12_500_000, clones=3125, seed=20261002, noise=0.3, chain="IGK", ins=175, private_v=True, private_j=True, zipf_s=0.25,
                        flags=["--k", "35", "--mf", "3", "--mq", "90", "--mrs", "30"], attempts=4, parallel=2),
    "cfg4_igl_pv": dict(pairs=12_500_000, clones=3125, seed=20261003, noise=0.3, chain="IGL", ins=175, private_v=True, private_j=True, zipf_s=0.25,
                        flags=["--k", "35", "--mf", "3", "--mq", "90", "--mrs", "30"], attempts=4, parallel=2),
    "mid_k25_mrs30": dict(pairs=200_000, clones=400, seed=20261002, noise=0.3, chain="IGH", ins=175,
                          flags=["--k", "25", "--mf", "2", "--mq", "60", "--mcs", "-5.5"]),
}


def make_rep(case: dict):
    """the case's repertoire (private_v / zipf_s: the second parameterisation of the generator, see cfg2_pv)"""
    return synth.make_repertoire(case["clones"], seed=case["seed"], private_v=bool(case.get("private_v", False)), private_j=bool(case.get("private_j", False)),
                                 zipf_s=float(case.get("zipf_s", 1.1)), chain=case.get("chain", "IGH"))


def write_inputs(case: dict, d: str):
    """The files a case's command line reads (also used by the tests, which is why it lives beside the digests)."""
    rep = make_rep(case)
    dev = "cpu"
    try:                                    # (the counter-based generator makes the same bytes on either device; on the GPU box 10 M pairs take a second)
        import torch
        if torch.cuda.is_available():
            dev = "cuda:0"
    except Exception:  # noqa: BLE001
        pass
    pool = synth.make_reads_cb(rep, case["pairs"], noise_frac=case["noise"], seed=case["seed"] + 13, device=dev)
    if hasattr(pool, "to_host"):
        pool = pool.to_host()
    pool.write_reads_file(os.path.join(d, "reads.txt"))
    synth.write_ref_dir(rep, os.path.join(d, "ref"))
    return rep, pool


def argv_of(case: dict, threads: int = 1):
    return ["--in", "reads.txt", "--chain", case["chain"], "--ref-dir", "ref", "--ins", str(case["ins"]), "--t", str(threads)] + case["flags"]


def digest(path: str) -> dict:
    h, n, lines = hashlib.sha256(), 0, 0
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 22)
            if not b:
                break
            h.update(b)
            n += len(b)
            lines += b.count(b"\n")
    return {"sha256": h.hexdigest(), "bytes": n, "lines": lines}


def run_once(case: dict, work: str, wd: str):
    """one --t 1 run of the compiled reference in wd (inputs symlinked from work); -> the process"""
    os.makedirs(wd)
    os.symlink(os.path.join(work, "reads.txt"), os.path.join(wd, "reads.txt"))
    os.symlink(os.path.join(work, "ref"), os.path.join(wd, "ref"))
    return subprocess.Popen([REF, "run"] + argv_of(case), cwd=wd, stdout=open(os.path.join(wd, "out.sam"), "wb"),
                            stderr=open(os.path.join(wd, "err.txt"), "wb"), env=dict(os.environ, VDJX_REF_ROOT_LOG="roots.log"))


def run_is_complete(name: str, wd: str):
    """the complete-run rule: the reference reported as many roots as the harness saw scored -> (complete, roots), or None if it did not finish"""
    err = open(os.path.join(wd, "err.txt"), errors="replace").read()
    m1, m2 = re.search(r"num root nodes: (\d+)", err), re.search(r"HARNESS_ROOTS_SCORED\t(\d+)", err)
    if not (m1 and m2 and "FINIS" in err):
        print(f"{name}: a run did not finish: {err[-300:]}", flush=True)
        return None
    return int(m1.group(1)) == int(m2.group(1)), int(m1.group(1)), int(m2.group(1))


def collect(name: str, case: dict, run_dirs, wall_s: float) -> dict:
    """run directories of finished runs -> the case's entry: at least two complete runs, ALL complete runs byte-identical"""
    complete = []
    for wd in run_dirs:
        r = run_is_complete(name, wd)
        if r is None:
            continue
        ok, nroots, nscored = r
        print(f"{name}: run {os.path.basename(wd)} scored {nscored} of {nroots} roots{'' if ok else ' (incomplete: dropped)'}", flush=True)
        if ok:
            complete.append((wd, nroots))
    assert len(complete) >= 2, f"{name}: fewer than two complete runs among {len(list(run_dirs))}"
    dg = [{f: digest(os.path.join(wd, f)) for f in ("vdj_contigs.fa", "out.sam", "vdjer.dot", "roots.log")} for wd, _ in complete]
    assert all(d == dg[0] for d in dg), f"{name}: complete runs disagree"
    wd, nroots = complete[0]
    log = [l.split("\t") for l in open(os.path.join(wd, "roots.log")).read().splitlines()]
    fa = open(os.path.join(wd, "vdj_contigs.fa")).read()
    res = {k_: v_ for k_, v_ in case.items() if k_ not in ("attempts", "parallel")}
    res.update(roots=nroots, roots_accepted=sum(int(v) for _, v in log), contigs=fa.count(">"), complete_runs=len(complete), runs=len(list(run_dirs)),
               fasta=dg[0]["vdj_contigs.fa"], sam=dg[0]["out.sam"], dot=dg[0]["vdjer.dot"], root_log=dg[0]["roots.log"],
               root_log_format="<root k-mer>\\t<score_seq verdict>\\n per root in dispatch order (A2:1305-1318 at --t 1)",
               reference_wall_s=round(wall_s))
    return res


def one_case(name: str, case: dict, attempts: int = 8, parallel: int = 4) -> dict:
    attempts, parallel = int(case.get("attempts", attempts)), int(case.get("parallel", parallel))
    work = tempfile.mkdtemp(prefix=f"vdjx_{name}_")
    t0 = time.time()
    write_inputs(case, work)
    print(f"{name}: inputs in {time.time() - t0:.0f}s", flush=True)
    dirs, tried, n_complete = [], 0, 0
    while tried < attempts and n_complete < 2:
        procs = []
        for _ in range(min(parallel, attempts - tried)):
            wd = os.path.join(work, f"run{tried}")
            procs.append((wd, run_once(case, work, wd)))
            tried += 1
        for wd, pr in procs:
            pr.wait()
            dirs.append(wd)
            r = run_is_complete(name, wd)
            n_complete += bool(r and r[0])
    res = collect(name, case, dirs, (time.time() - t0) / max(1, (tried + parallel - 1) // parallel))
    subprocess.run(["rm", "-rf", work])
    return res


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    if len(sys.argv) >= 4 and sys.argv[1] == "--collect":
        # runs started by hand, one at a time (a 10 M-pair run at k = 25 peaks at 33 GB: two at once do not fit this container's 62 GB):
        #   make_golden_midscale.py --collect <name> <wall seconds of one run> <run dir> <run dir> ...
        name, wall = sys.argv[2], float(sys.argv[3])
        out = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
        out["cases"][name] = collect(name, CASES[name], sys.argv[4:], wall)
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
        print(json.dumps(out["cases"][name]), flush=True)
        return
    names = sys.argv[1:] or list(CASES)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
    out["note"] = ("digests of complete --t 1 runs of the compiled reference; inputs regenerate from vdjer_amd/synth.py "
                   "(make_golden_midscale.write_inputs)")
    for n in names:
        out["cases"][n] = one_case(n, CASES[n])
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
        print(json.dumps(out["cases"][n]), flush=True)


if __name__ == "__main__":
    main()
