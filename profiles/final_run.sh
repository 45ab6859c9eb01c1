cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r05a}
bash profiles/prof_step.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; tail -2 gpurun_out/prof_$TAG.log
mkdir -p gpurun_out/r05m
python bench.py > gpurun_out/r05m/default.json 2> gpurun_out/r05m/default.err
python bench.py --pairs 1000000 --steps 200 > gpurun_out/r05m/1M.json 2> gpurun_out/r05m/1M.err
python bench.py --k 25 --mf 2 --mq 60 --mrs 20 --steps 30 > gpurun_out/r05m/k25.json 2> gpurun_out/r05m/k25.err
python bench.py --force-shard --steps 30 > gpurun_out/r05m/shard.json 2> gpurun_out/r05m/shard.err
python bench.py --pairs 100000 --steps 200 --no-cpu > gpurun_out/r05m/100k.json 2>/dev/null
python bench.py --config4 --gpus 1 --steps 5 --warmup 1 --no-cpu > gpurun_out/r05m/config4_1gpu.json 2> gpurun_out/r05m/config4_1gpu.err
python - <<PY
import json
for n in ("default","1M","k25","shard","100k","config4_1gpu"):
    try:
        d=json.load(open(f"gpurun_out/r05m/{n}.json"))
    except Exception as e:
        print(n, "ERR", e); continue
    print(n, d["value"], d["ms_per_step"], d["device_busy_frac"], (d.get("value_with_read_index") or {}).get("value"), d["parity_gate_timed_step"] and d["parity_gate_timed_step"]["ok"], d["cpu_baseline"] and round(d["cpu_baseline"]["value"],4), d["host_side"].get("read_index_build_s"), d.get("first_step_ms"))
d=json.load(open("gpurun_out/r05m/default.json"))
print(d["roofline"]["frac"], d["roofline"]["frac_on_traffic"], d["roofline"]["hot_path_frac"], d["roofline"]["hot_path_frac_gated"], d["value_end_to_end"]["value"], d["cli_end_to_end"]["wall_s_process"], d["cli_end_to_end"]["stages_s"], d["cli_end_to_end"]["outputs_identical_to_reference"])
PY
