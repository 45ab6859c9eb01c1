import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from vdjer_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
rep = synth.make_repertoire(max(4, n // 500), seed=20261002)
pool = synth.make_reads_cb(rep, n, noise_frac=0.3, seed=20261002 + 13)
td = tempfile.mkdtemp()
pool.write_reads_file(os.path.join(td, "reads.txt"))
synth.write_ref_dir(rep, os.path.join(td, "ref"))
for i in range(2):
    t = time.perf_counter()
    r = subprocess.run([os.path.join(os.getcwd(), "vdjer_amd", "vdjer"), "--in", "reads.txt", "--chain", "IGH", "--ref-dir", "ref", "--ins", "175", "--t", "8"], cwd=td,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, VDJX_TIMES="1"))
    print("run", i, "wall", round(time.perf_counter() - t, 3), "rc", r.returncode)
    for l in r.stderr.splitlines():
        if l.startswith("VDJX_TIMES"):
            print("  ", l)
