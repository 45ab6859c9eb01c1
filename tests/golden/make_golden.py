#!/usr/bin/env python3
"""Generate tests/golden/* by running the compiled reference (oracle/_ref/vdjer_ref).

Runs ONLY in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`).  The fixtures written here are data: seeded synthetic inputs and what the
reference's own functions returned for them.  No reference source is stored.

    python tests/golden/make_golden.py            # regenerate everything

Lost-root race: `run` outputs are accepted only from runs in which the harness
counted as many scored roots as the reference reported ("num root nodes"), and only if all such
complete runs agree byte for byte.
"""
from __future__ import annotations

import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vdjer_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "vdjer_ref")


def gz_write(name: str, text: str) -> None:
    with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as f:
        f.write(text.encode())


def save_pool(name: str, rep: synth.Repertoire, pool: synth.ReadPool) -> None:
    np.savez_compressed(
        os.path.join(HERE, name),
        rl=np.int32(pool.rl), n_pairs=np.int64(pool.n_pairs),
        primary=pool.primary, secondary=pool.secondary, pair_id=pool.pair_id, read_num=pool.read_num,
        is_rc=pool.is_rc, reg_rank=pool.reg_rank,
        v_region=np.array([rep.v_region]), clones=np.array(rep.clones),
        v_codes=np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32),
        j_codes=np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32))


def run_ref(args, cwd, stdout=None):
    return subprocess.run([REF] + args, cwd=cwd, stdout=stdout, stderr=subprocess.PIPE, text=True)


def graph_case(tag, work, flags):
    out = os.path.join(work, tag)
    r = run_ref(["graph", out] + flags, work)
    m = re.search(r"HARNESS_GRAPH\tpre=(\d+)\tsurvivors=(\d+)\tnodes=(\d+)", r.stderr)
    assert m, r.stderr[-2000:]
    res = {"pre": int(m.group(1)), "survivors": int(m.group(2)), "nodes": int(m.group(3))}
    for part in ("survivors", "nodes", "node_order", "roots", "condensed"):
        gz_write(f"{tag}.{part}.tsv.gz", open(f"{out}.{part}.tsv").read())
    return res, out


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    work = tempfile.mkdtemp(prefix="vdjx_golden_")
    manifest = {}

    # ------------------------------------------------------------------ case "noisy": a-1/a-2/a-3
    rep = synth.make_repertoire(4, seed=5)
    pool = synth.make_reads(rep, 2500, noise_frac=0.3, seed=77)
    synth.write_ref_dir(rep, os.path.join(work, "ref_noisy"))
    pool.write_reads_file(os.path.join(work, "noisy.reads"))
    save_pool("noisy.npz", rep, pool)
    base = ["--in", "noisy.reads", "--chain", "IGH", "--ref-dir", "ref_noisy", "--ins", "175"]
    manifest["noisy"] = {}
    for tag, extra in (("noisy_k35", []),
                       ("noisy_k25", ["--k", "25", "--mf", "2", "--mq", "60", "--mrs", "20"]),
                       ("noisy_mq230", ["--mf", "1", "--mq", "230"]),
                       ("noisy_mq20", ["--mf", "2", "--mq", "20"])):
        res, _ = graph_case(tag, work, base + extra)
        manifest["noisy"][tag] = {"flags": extra, **res}

    # ------------------------------------------------------------------ case "pre": full pre-prune table
    rep2 = synth.make_repertoire(2, seed=9)
    pool2 = synth.make_reads(rep2, 260, noise_frac=0.25, seed=3, err=0.01, n_rate=0.004)
    synth.write_ref_dir(rep2, os.path.join(work, "ref_pre"))
    pool2.write_reads_file(os.path.join(work, "pre.reads"))
    save_pool("pre.npz", rep2, pool2)
    res, out = graph_case("pre_k35", work, ["--in", "pre.reads", "--chain", "IGH", "--ref-dir", "ref_pre", "--ins", "175",
                                            "--mf", "2", "--mq", "60"])
    gz_write("pre_k35.pre.tsv.gz", open(out + ".pre.tsv").read())
    manifest["pre"] = {"pre_k35": {"flags": ["--mf", "2", "--mq", "60"], **res}}

    # ------------------------------------------------------------------ case "score": a-7
    rng = np.random.default_rng(123)
    queries = []
    t0 = rep.clones[0]
    for c in rep.clones:
        queries.append(c[:35])
    for s in range(0, len(t0) - 35, 7):
        queries.append(t0[s:s + 35])
    for s in range(0, 280, 11):
        q = list(rep.v_germ[1][s:s + 35])
        if len(q) < 35:
            continue
        for _ in range(int(rng.integers(1, 8))):
            q[int(rng.integers(0, 35))] = "ACGT"[int(rng.integers(0, 4))]
        queries.append("".join(q))
        g = rep.v_germ[2][s:s + 36]
        if len(g) == 36:
            cut = int(rng.integers(5, 30))
            queries.append(g[:cut] + g[cut + 1:])                                   # deletion
            queries.append((g[:cut] + "ACGT"[int(rng.integers(0, 4))] + g[cut:])[:35])  # insertion
    vr = rep.v_region
    queries.append(vr[:35])
    queries.append(vr[-35:])
    queries.append(vr[-40:-5])
    for _ in range(20):
        queries.append("".join("ACGT"[i] for i in rng.integers(0, 4, 35)))
    with open(os.path.join(work, "queries35.txt"), "w") as f:
        f.write("\n".join(queries) + "\n")
    with open(os.path.join(work, "queries25.txt"), "w") as f:
        f.write("\n".join(q[3:28] for q in queries) + "\n")
    sc = {}
    for tag, k, thr, qf in (("k35_t30", 35, 30, "queries35.txt"), ("k35_t25", 35, 25, "queries35.txt"),
                            ("k35_t34", 35, 34, "queries35.txt"), ("k25_t20", 25, 20, "queries25.txt")):
        r = run_ref(["score", "ref_noisy/v_region.fa", str(k), "15", str(thr), qf], work, stdout=subprocess.PIPE)
        gz_write(f"score_{tag}.tsv.gz", r.stdout)
        sc[tag] = {"k": k, "thr": thr, "n": r.stdout.count("\n"), "ones": sum(l.endswith("\t1") for l in r.stdout.splitlines())}
    manifest["score"] = sc

    # ------------------------------------------------------------------ case "map": a-8/a-9
    rep3 = synth.make_repertoire(3, seed=21)
    pool3 = synth.make_reads(rep3, 6000, noise_frac=0.1, seed=8)
    synth.write_ref_dir(rep3, os.path.join(work, "ref_map"))
    pool3.write_reads_file(os.path.join(work, "map.reads"))
    save_pool("map.npz", rep3, pool3)
    wins = []
    for w in rep3.windows():
        wins.append(w)
        wins.append(w[:411])
        wins.append(w[51:411])
        m = list(w)
        m[200] = "A" if m[200] != "A" else "C"
        wins.append("".join(m))
    for t in rep3.clones:
        wins.append(t[1:487])
        wins.append(synth.revcomp(t)[100:586])
    with open(os.path.join(work, "windows.txt"), "w") as f:
        f.write("\n".join(wins) + "\n")
    mp = {}
    for tag, extra in (("ins175", []), ("ins150_rf2", ["--rf", "2"]), ("ins200_ms20", ["--ms", "20", "--rs", "40"])):
        ins = {"ins175": "175", "ins150_rf2": "150", "ins200_ms20": "200"}[tag]
        r = run_ref(["map", "windows.txt", "--in", "map.reads", "--chain", "IGH", "--ref-dir", "ref_map", "--ins", ins] + extra,
                    work, stdout=subprocess.PIPE)
        gz_write(f"map_{tag}.txt.gz", r.stdout)
        mp[tag] = {"ins": int(ins), "flags": extra,
                   "valid": [int(l.split("\t")[2]) for l in r.stdout.splitlines() if l.startswith("W\t")]}
    gz_write("map_windows.txt.gz", "\n".join(wins) + "\n")
    manifest["map"] = mp

    # ------------------------------------------------------------------ case "hash": a-4/a-5
    strs = ["ACGTACGTACGTACGTACGTACGTACGTACGTACG", "ACTTCTGGGGCCAGGG", "A", "AC", "ACG", "ACGTACG", "ACGTACGT", "ACGTACGTA"]
    for n in (15, 16, 25, 35, 50, 63, 64, 65, 360):
        strs.append("".join("ACGT"[i] for i in rng.integers(0, 4, n)))
    with open(os.path.join(work, "hash.txt"), "w") as f:
        f.write("\n".join(strs) + "\n")
    r = run_ref(["hash", "hash.txt"], work, stdout=subprocess.PIPE)
    gz_write("hash.tsv.gz", r.stdout)
    manifest["hash"] = {"n": len(strs)}

    # ------------------------------------------------------------------ case "vjf": window discovery
    contigs = [t[:650] for t in rep.clones] + [t[:650] for t in rep3.clones] + [t[3:640] for t in rep3.clones]
    contigs += [synth.revcomp(rep.clones[0])[:650]]
    with open(os.path.join(work, "contigs.txt"), "w") as f:
        f.write("\n".join(contigs) + "\n")
    # one ref-dir holding both repertoires' anchors
    os.makedirs(os.path.join(work, "ref_vjf"), exist_ok=True)
    for fn in ("v_index", "j_index"):
        with open(os.path.join(work, "ref_vjf", fn), "w") as f:
            f.write(open(os.path.join(work, "ref_noisy", fn)).read() + open(os.path.join(work, "ref_map", fn)).read())
    for fn in ("ig_vdj.fa", "v_region.fa"):
        shutil.copy(os.path.join(work, "ref_noisy", fn), os.path.join(work, "ref_vjf", fn))
    open(os.path.join(work, "dummy.in"), "w").write("x")
    r = run_ref(["vjf", "contigs.txt", "--in", "dummy.in", "--chain", "IGH", "--ref-dir", "ref_vjf", "--ins", "175"], work,
                stdout=subprocess.PIPE)
    gz_write("vjf_out.txt.gz", r.stdout)
    gz_write("vjf_contigs.txt.gz", "\n".join(contigs) + "\n")
    np.savez_compressed(os.path.join(HERE, "vjf_codes.npz"),
                        v_codes=np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors + rep3.v_anchors}), dtype=np.uint32),
                        j_codes=np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors + rep3.j_anchors}), dtype=np.uint32))
    manifest["vjf"] = {"n": len(contigs)}

    # ------------------------------------------------------------------ case "order": dense_hash_map iteration order
    ops = []
    keys = ["".join("ACGT"[i] for i in rng.integers(0, 4, 360)) for _ in range(150)]
    for i, kx in enumerate(keys):
        ops.append("I " + kx)
        if i in (3, 17, 40, 99):
            ops.append("D")
    ops.append("D")
    for i in rng.permutation(150)[:120]:
        ops.append("E " + keys[int(i)])
    ops.append("D")
    ops.append("I " + keys[0][::-1])
    ops.append("D")
    ops.append("R")
    ops.append("D")
    with open(os.path.join(work, "ops.txt"), "w") as f:
        f.write("\n".join(ops) + "\n")
    r = run_ref(["order", "360", "ops.txt"], work, stdout=subprocess.PIPE)
    gz_write("order_ops.txt.gz", "\n".join(ops) + "\n")
    gz_write("order_out.txt.gz", r.stdout)

    # ------------------------------------------------------------------ case "e2e": full CLI, FASTA + SAM
    e2e = {}
    for tag, rp, pl, flags in (
            ("e2e_tiled", synth.make_repertoire(3, seed=11), None, []),
            ("e2e_mixed", synth.make_repertoire(6, seed=31), "reads", []),
            ("e2e_k25", synth.make_repertoire(6, seed=31), "reads", ["--k", "25", "--mf", "2", "--mq", "60", "--mcs", "-5.5", "--mrs", "20"])):
        if pl is None:
            pl = synth.tile_reads(rp, [0, 1, 2], copies=3)
        else:
            pl = synth.make_reads(rp, 9000, noise_frac=0.2, seed=41)
        wd = os.path.join(work, tag)
        os.makedirs(wd)
        synth.write_ref_dir(rp, os.path.join(wd, "ref"))
        pl.write_reads_file(os.path.join(wd, "reads.txt"))
        save_pool(f"{tag}.npz", rp, pl)
        complete = []
        for attempt in range(12):
            with open(os.path.join(wd, "out.sam"), "w") as so:
                r = run_ref(["run", "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--t", "1"] + flags,
                            wd, stdout=so)
            nroots = int(re.search(r"num root nodes: (\d+)", r.stderr).group(1))
            scored = int(re.search(r"HARNESS_ROOTS_SCORED\t(\d+)", r.stderr).group(1))
            if scored == nroots:
                complete.append((open(os.path.join(wd, "vdj_contigs.fa")).read(), open(os.path.join(wd, "out.sam")).read(),
                                 open(os.path.join(wd, "vdjer.dot")).read()))
            if len(complete) >= 3:
                break
        assert len(complete) >= 2, f"{tag}: not enough complete runs"
        assert all(c == complete[0] for c in complete), f"{tag}: complete runs disagree"
        fa, sam, dot = complete[0]
        gz_write(f"{tag}.contigs.fa.gz", fa)
        gz_write(f"{tag}.sam.gz", sam)
        gz_write(f"{tag}.dot.gz", dot)
        e2e[tag] = {"flags": flags, "contigs": fa.count(">"), "sam_lines": sam.count("\n"), "roots": nroots}
    manifest["e2e"] = e2e

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True)[:3000])
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
