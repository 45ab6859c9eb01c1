#!/usr/bin/env python3
"""One step of bench.py, dispatch by dispatch, from a `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR` run:
start (us from the step's first dispatch), duration, the gap to whatever ended last before it, name.  Kernels, fills and copies of
all streams in one list (a copy on the copy stream overlaps the compute stream: negative gaps).
usage: python profiles/timeline.py DIR [step_from_the_end=3] [min_gap_us=0]"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
K = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]) for r in K]
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"][12:]))
ev.sort()
packs = [i for i, e in enumerate(ev) if "k_pool_pack" in e[2]]        # two launches per step (per chain with --config4)
c = len(packs) // 2 - back
seg = ev[packs[2 * c]:packs[2 * c + 2]]
t0, prev = seg[0][0], seg[0][0]
busy = 0
print("span %.1f us, %d dispatches" % ((max(e[1] for e in seg) - t0) / 1e3, len(seg)))
for s, e, n in seg:
    gap = (s - prev) / 1e3
    if gap >= min_gap or min_gap == 0:
        print("%9.1f +%8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev = max(prev, e)
