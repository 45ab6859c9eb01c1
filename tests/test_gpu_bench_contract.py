"""bench.py as the driver runs it: ONE JSON line on stdout with the contract's keys, at N = 1 and -- two and three ranks on this one
GPU (VDJX_BENCH_ONE_DEVICE=1: every rank on device 0, the exchange over gloo) -- through `python -m torch.distributed.run`."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(cmd, env=None):
    e = dict(os.environ)
    e.pop("VDJX_HIT_CHUNK", None)                 # (conftest's small slices are for the tests, not for the bench)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                 # ONE line, nothing else on stdout
    return json.loads(lines[0])


def _check(d, n, steps, warmup):
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1
    # whole-job throughput: the pairs of all ranks over the slowest rank's time
    pairs = d["config"]["pairs_per_gpu"] * n
    assert abs(d["value"] - pairs / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]


def test_bench_line_one_gpu():
    d = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--pairs", "200000"])
    _check(d, 1, 3, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_line_ranks_on_one_device(n):
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--pairs", "150000", "--no-cpu", "--no-e2e"],
              env={"VDJX_BENCH_ONE_DEVICE": "1"})
    _check(d, n, 2, 1)
    assert d["config"]["parallelism"].endswith(f"x{n}")


def test_bench_line_through_the_c_driver_one_rank():
    """--force-shard: the step through libvdjmgpu.so (vdjx_mgpu.c, the driver of `vdjer --gpus N`) with one rank; the line names the driver"""
    d = _line([sys.executable, "bench.py", "--force-shard", "--steps", "2", "--warmup", "1", "--pairs", "200000", "--no-cpu", "--no-e2e"])
    _check(d, 1, 2, 1)
    assert "libvdjmgpu.so" in d["multi_gpu_driver"] and d["counts"]["nodes"] > 1000 and d["counts"]["windows"] > 0


def test_bench_line_private_repertoire_takes_the_traversal_windows():
    """--repertoire private: the workload on which the reference's serial traversal ends at BASELINE size (tests/golden/midscale.json cfg2_pv);
    here small: the scorer inputs are the windows the host traversal asks for, whatever the size"""
    d = _line([sys.executable, "bench.py", "--repertoire", "private", "--steps", "2", "--warmup", "1", "--pairs", "400000", "--no-cpu", "--no-e2e", "--no-index-leg"])
    _check(d, 1, 2, 1)
    assert d["config"]["windows"] == "traversal" and d["config"]["clones_per_gpu"] == 100 and "private" in d["config"]["workload"]
    assert d["counts"]["contigs"] >= 50
