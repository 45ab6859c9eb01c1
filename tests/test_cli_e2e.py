"""End-to-end on the GPU: the `vdjer` command line (C host over libvdjx) and the Python wiring of the same stages
must reproduce the reference's vdj_contigs.fa, SAM (stdout) and vdjer.dot byte for byte (tests/golden/e2e_*:
outputs of complete runs of the compiled reference)."""
import os
import subprocess

import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_inputs(c, d):
    os.makedirs(os.path.join(d, "ref"), exist_ok=True)
    c.pool.write_reads_file(os.path.join(d, "reads.txt"))
    with open(os.path.join(d, "ref", "v_region.fa"), "w") as f:
        f.write(">v_region\n" + c.v_region + "\n")
    for fn, codes in (("v_index", c.v_codes), ("j_index", c.j_codes)):
        with open(os.path.join(d, "ref", fn), "w") as f:
            f.write("".join(f"{int(x)}\t0\n" for x in codes))
    open(os.path.join(d, "ref", "ig_vdj.fa"), "w").write(">x\nACGT\n")


@pytest.mark.parametrize("tag", ["e2e_tiled", "e2e_mixed", "e2e_k25"])
def test_vdjer_cli_matches_reference(tag, tmp_path):
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    assert os.path.exists(exe), "build it: make -C vdjer_amd/csrc/host"
    c = G.Case(tag)
    info = G.manifest()["e2e"][tag]
    _write_inputs(c, str(tmp_path))
    cmd = [exe, "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--t", "1"] + info["flags"]
    r = subprocess.run(cmd, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "vdj_contigs.fa").read_text() == G.text(f"{tag}.contigs.fa.gz")
    assert r.stdout == G.text(f"{tag}.sam.gz")
    assert (tmp_path / "vdjer.dot").read_text() == G.text(f"{tag}.dot.gz")


def test_cli_rejects_bad_input(tmp_path):
    exe = os.path.join(ROOT, "vdjer_amd", "vdjer")
    r = subprocess.run([exe, "--in", "nope", "--chain", "IGH", "--ref-dir", ".", "--ins", "175"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "Could not find input file" in r.stderr
    r = subprocess.run([exe, "--in", "x", "--chain", "IGX"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "Invalid chain" in r.stderr


@pytest.mark.parametrize("tag", ["e2e_mixed"])
def test_python_wiring_matches_reference(tag, tmp_path):
    from vdjer_amd import api, host
    c = G.Case(tag)
    info = G.manifest()["e2e"][tag]
    fl = G.flags_to_params(info["flags"])
    p = host.make_params("IGH", ins=175, k=fl["k"], mf=fl["mf"], mq=fl["mq"], mcs=fl["mcs"], mrs=fl["mrs"], rl=c.pool.rl)
    ctx = api.Context(0)
    ctx.anchor_sets_load(c.v_codes, c.j_codes)
    ctx.vregion_load([c.v_region], 15)
    pool = ctx.pool_load(c.pool.primary, c.pool.secondary, c.pool.rl)
    ctx.read_index_build(pool, c.pool.pair_id, c.pool.read_num, c.pool.is_rc, c.pool.reg_rank, c.pool.n_pairs)
    g = ctx.kmer_build(pool, fl["k"], fl["mf"], fl["mq"])
    fa, sam = tmp_path / "c.fa", tmp_path / "o.sam"
    host.assemble(p, g, *host.gpu_hooks(ctx, (c.pool.primary, c.pool.secondary), c.pool.names(), p), c.v_codes, c.j_codes,
                  str(fa), None, str(sam))
    assert fa.read_text() == G.text(f"{tag}.contigs.fa.gz")
    assert sam.read_text() == G.text(f"{tag}.sam.gz")
    ctx.close()


def test_vdjx_index_cli_prints_the_reference_rows():
    """the index generator as a command: same arguments and the same stdout bytes as the reference's process_kmers"""
    exe = os.path.join(ROOT, "vdjer_amd", "vdjx_index")
    assert os.path.exists(exe), "build it: make -C vdjer_amd/csrc/host"
    _, ranges = G.index_case()
    for s, e, codes, dists in ranges[:4] + ranges[-1:]:
        r = subprocess.run([exe, os.path.join(G.GOLD, "index_anchors.txt"), str(s), str(e)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout == "".join(f"{c}\t{d}\n" for c, d in zip(codes.tolist(), dists.tolist()))
    bad = subprocess.run([exe, os.path.join(G.GOLD, "index_anchors.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bad.returncode != 0 and "Usage" in bad.stderr
