"""The C driver of the sharded path (`vdjer --gpus N`, vdjer_amd/csrc/host/vdjx_mgpu.c) AT SIZE on the one GPU of the box:
N process-ranks on device 0 (VDJX_MGPU_ONE_DEVICE=1: the bytes move through host sockets, RCCL refuses two ranks on a device),
millions of pairs per rank, per-phase wall times and bytes sent (VDJX_TIMES -> VDJX_MGPU_PHASE lines), and the same file through
`--gpus 1` beside it: the outputs must be the same bytes.

    python profiles/mgpu_at_size.py [pairs_per_rank=2000000] [ranks=8] [ig_pairs=1000000] [chain=IGH]

The pool is noise-dominated (SURVEY §6 probe A: 92 % of an extracted pool are unmapped non-Ig reads): ig_pairs pairs from ig_pairs/500
clones, the rest uniform random reads -- the serial traversal (the reference's, on rank 0) explodes combinatorially on THIS generator's
repertoire above ~1.5 M Ig pairs, and noise is the worst case for the exchange (every k-mer instance its own partial).
Not a scaling measurement: N ranks share one device and wait for each other at every collective."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdjer_amd import synth  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ig = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
chain = sys.argv[4] if len(sys.argv) > 4 else "IGH"
pairs = per * ranks
noise = max(0.0, 1.0 - ig / pairs)


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 22), b""):
            h.update(b)
    return h.hexdigest()


def run(td, name, gpus):
    d = os.path.join(td, name)
    os.makedirs(d)
    for f in ("reads.txt", "ref"):
        os.symlink(os.path.join(td, f), os.path.join(d, f))
    cmd = [os.path.join(ROOT, "vdjer_amd", "vdjer"), "--in", "reads.txt", "--chain", chain, "--ref-dir", "ref", "--ins", "175", "--t", "8", "--gpus", str(gpus)]
    env = dict(os.environ, VDJX_MGPU_ONE_DEVICE="1", VDJX_TIMES="1", VDJX_MGPU_TIMEOUT_S="900", VDJX_REPORT_SHARE="1")
    t0 = time.perf_counter()
    phases, times, shares, tail = [], {}, [], []
    with open(os.path.join(d, "sam.out"), "wb") as so:
        pr = subprocess.Popen(cmd, cwd=d, stdout=so, stderr=subprocess.PIPE, text=True, errors="replace", env=env)
        for line in pr.stderr:
            f = line.rstrip("\n").split("\t")
            tail.append(line.rstrip())
            if f[0] == "VDJX_MGPU_PHASE":
                phases.append({"phase": f[3], "ms": float(f[5]), "calls": int(f[7]), "bytes_sent_rank0": int(f[9])})
            elif f[0] == "VDJX_TIMES":
                times.setdefault(f[1], float(f[2]))
            elif f[0] == "share":
                shares.append(line.rstrip())
        rc = pr.wait()
    wall = time.perf_counter() - t0
    out = {"gpus": gpus, "exit": rc, "wall_s": round(wall, 2), "stage_ms_since_start": times, "phases_rank0": phases}
    if rc:
        out["stderr_tail"] = tail[-12:]
        return out, None
    dg = {f: sha(os.path.join(d, f)) for f in ("vdj_contigs.fa", "sam.out", "vdjer.dot")}
    out["contigs"] = sum(1 for l in open(os.path.join(d, "vdj_contigs.fa")) if l.startswith(">"))
    out["sam_bytes"] = os.path.getsize(os.path.join(d, "sam.out"))
    out["summary"] = [l for l in tail if l.startswith(("k-mer table sharded", "Pre Num nodes", "Num nodes", "num root nodes", "contig_candidates", "windows scored")) or "bytes sent" in l]
    out["shares"] = shares[:ranks]
    return out, dg


t0 = time.perf_counter()
rep = synth.make_repertoire(max(4, ig // 500), seed=4242, chain=chain)
pool = synth.make_reads_cb(rep, pairs, noise_frac=noise, seed=4243)
with tempfile.TemporaryDirectory(dir=os.environ.get("VDJX_TMP")) as td:
    pool.write_reads_file(os.path.join(td, "reads.txt"))
    synth.write_ref_dir(rep, os.path.join(td, "ref"))
    del pool
    res = {"pairs": pairs, "pairs_per_rank": per, "ranks": ranks, "ig_pairs": ig, "clones": len(rep.clones), "noise": round(noise, 4), "chain": chain,
           "reads_file_MB": round(os.path.getsize(os.path.join(td, "reads.txt")) / 1e6), "generate_s": round(time.perf_counter() - t0, 1)}
    a, da = run(td, "n", ranks)
    res["sharded"] = a
    if da is not None and os.environ.get("VDJX_AT_SIZE_NO_ONE") is None:
        b, db = run(td, "one", 1)
        res["one_gpu"] = {k_: b[k_] for k_ in ("exit", "wall_s", "stage_ms_since_start", "contigs", "sam_bytes") if k_ in b}
        res["outputs_identical_to_one_gpu_run"] = da == db
    print(json.dumps(res))
    sys.exit(0 if a["exit"] == 0 and res.get("outputs_identical_to_one_gpu_run", True) else 1)
