# one-knob sweeps of the default bench step (run on the GPU box from the repo root): bash profiles/sweep.sh KNOB v1 v2 ...
K=$1; shift
for v in "$@"; do
  env $K=$v python3 bench.py --steps 30 --warmup 3 --no-cpu --no-e2e --parity-sample 0 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/sw.json"))
k=d["kernels_ms_per_step"]
print("$K=$v", d["value"], d["ms_per_step"], "walk", k.get("k_walk_items"), "hist", k.get("k_gated_hist"), "partr", k.get("k_part_records"), "partt", k.get("k_part_tuples"), "reduce", k.get("k_gated_reduce"), "items", k.get("k_part_items"), "recount", k.get("k_recount"))
PY
done
