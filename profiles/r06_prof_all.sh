#!/bin/bash
# round 6: the rocprofv3 stats + PMC passes (profiles/prof_step.sh) for the four lines the bench reports -- the default configs[2] step, the
# same step through the sharded phases, configs[3] (k = 25) and the configs[4] geometry on one GPU -- run from the repo root on the GPU box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r06a}
bash profiles/prof_step.sh ${T} > gpurun_out/prof_${T}.log 2>&1
bash profiles/prof_step.sh ${T}_shard --force-shard > gpurun_out/prof_${T}_shard.log 2>&1
bash profiles/prof_step.sh ${T}_k25 --k 25 --mf 2 --mq 60 --mrs 20 > gpurun_out/prof_${T}_k25.log 2>&1
PROF_CHAINS=3 bash profiles/prof_step.sh ${T}_config4 --config4 --gpus 1 > gpurun_out/prof_${T}_config4.log 2>&1
for v in "" _shard _k25 _config4; do
  d=gpurun_out/prof_${T}$v
  ls $d/summary.json $d/bench.json > /dev/null 2>&1 || echo "missing $d"
  python3 - <<PY
import json
try:
    s=json.load(open("$d/summary.json")); b=json.loads(open("$d/bench.json").read().strip().splitlines()[-1])
    print("$v", b["value"], b["ms_per_step"], (s.get("step") or {}).get("bytes_per_step"), len(s["kernels"]))
except Exception as e:
    print("$v", "ERR", e)
PY
done
