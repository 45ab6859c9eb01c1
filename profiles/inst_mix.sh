#!/bin/bash
# instruction mix of the k-mer build's kernels (rocprofv3 PMC, two passes over a short bench.py run): per kernel the executed
# VALU / SALU / LDS / vector-memory instructions per wave.   profiles/inst_mix.sh <tag>   -> gpurun_out/inst_<tag>.json
TAG=$1
CMD="bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --parity-sample 0"
export TMPDIR=/tmp
OUT=gpurun_out/inst_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/a -- python3 $CMD > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/b -- python3 $CMD > /dev/null 2> $OUT/b.err
python3 - $OUT <<'PY' > gpurun_out/inst_$TAG.json
import collections, csv, glob, json, os, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in per.items():
    d = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    w = d.get("SQ_WAVES")
    if w:
        for c in list(d):
            if c.startswith("SQ_INSTS_"):
                d[c + "_per_wave"] = round(d[c] / w, 1)
    res[k] = d
json.dump(res, sys.stdout, indent=1, sort_keys=True)
PY
find $OUT -name "*.csv" -delete
tail -c 200 $OUT/a.err $OUT/b.err
