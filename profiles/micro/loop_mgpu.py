import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from tests import golden_util as G
from tests.test_cli_e2e import _write_inputs
c = G.Case("e2e_mixed")
info = G.manifest()["e2e"]["e2e_mixed"]
td = tempfile.mkdtemp()
_write_inputs(c, td)
exe = os.path.join(os.getcwd(), "vdjer_amd", "vdjer")
bad = 0
for i in range(int(sys.argv[1])):
    for g in (2, 4):
        r = subprocess.run([exe, "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--t", "2", "--gpus", str(g)] + info["flags"], cwd=td,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, VDJX_MGPU_ONE_DEVICE="1", VDJX_MGPU_TIMEOUT_S="120"))
        if r.returncode != 0:
            bad += 1
            print("iteration", i, "gpus", g, "rc", r.returncode)
            print("\n".join(r.stderr.splitlines()[-12:]))
print("bad", bad)
