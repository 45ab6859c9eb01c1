cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash profiles/prof_step.sh r04h > gpurun_out/prof_r04h.log 2>&1; tail -2 gpurun_out/prof_r04h.log
mkdir -p gpurun_out/r04m
python bench.py > gpurun_out/r04m/default.json 2> gpurun_out/r04m/default.err
python bench.py --pairs 1000000 --steps 200 > gpurun_out/r04m/1M.json 2> gpurun_out/r04m/1M.err
python bench.py --k 25 --mf 2 --mq 60 --mrs 20 --steps 30 > gpurun_out/r04m/k25.json 2> gpurun_out/r04m/k25.err
python bench.py --force-shard --steps 30 > gpurun_out/r04m/shard.json 2> gpurun_out/r04m/shard.err
python bench.py --pairs 100000 --steps 200 --no-cpu > gpurun_out/r04m/100k.json 2>/dev/null
python - <<PY
import json
for n in ("default","1M","k25","shard","100k"):
    d=json.load(open(f"gpurun_out/r04m/{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["device_busy_frac"], (d.get("value_with_read_index") or {}).get("value"), d["parity_gate_timed_step"] and d["parity_gate_timed_step"]["ok"], d["cpu_baseline"] and round(d["cpu_baseline"]["value"],4), d["host_side"].get("read_index_build_s"), d.get("first_step_ms"))
d=json.load(open("gpurun_out/r04m/default.json"))
print(d["roofline"]["frac"], d["roofline"]["frac_on_traffic"], d["roofline"]["hot_path_frac"], d["roofline"]["hot_path_frac_gated"], d["value_end_to_end"]["value"], d["cli_end_to_end"]["wall_s_process"], d["cli_end_to_end"]["stages_s"])
PY
