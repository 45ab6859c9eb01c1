# the round's lines of record in ONE call on the GPU box (run from the repo root): profiles/final_run.sh <tag>
#   the rocprofv3 stats + PMC passes of the four variants (profiles/r06_prof_all.sh), then the bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06f}
bash profiles/r06_prof_all.sh $TAG > gpurun_out/prof_all_$TAG.log 2>&1; tail -4 gpurun_out/prof_all_$TAG.log
M=gpurun_out/${TAG}m
mkdir -p $M
python bench.py > $M/default.json 2> $M/default.err
python bench.py --pairs 1000000 --steps 200 > $M/1M.json 2> $M/1M.err
python bench.py --k 25 --mf 2 --mq 60 --mrs 20 --steps 30 > $M/k25.json 2> $M/k25.err
python bench.py --force-shard --steps 30 > $M/shard.json 2> $M/shard.err
python bench.py --pairs 100000 --steps 200 --no-cpu > $M/100k.json 2>/dev/null
python bench.py --config4 --gpus 1 --steps 5 --warmup 1 --no-cpu > $M/config4_1gpu.json 2> $M/config4_1gpu.err
python bench.py --repertoire private --cli-at-size --steps 30 --no-cpu --no-e2e > $M/private_10M.json 2> $M/private_10M.err
python - <<PY
import json
for n in ("default","1M","k25","shard","100k","config4_1gpu","private_10M"):
    try:
        d=json.load(open(f"$M/{n}.json"))
    except Exception as e:
        print(n, "ERR", e); continue
    print(n, d["value"], d["ms_per_step"], d["device_busy_frac"], (d.get("value_with_read_index") or {}).get("value"), d["parity_gate_timed_step"] and d["parity_gate_timed_step"]["ok"], d["cpu_baseline"] and round(d["cpu_baseline"]["value"],4), d.get("first_step_ms"))
    if d.get("cli_at_size"):
        c=d["cli_at_size"]; print("   cli_at_size", {k: c.get(k) for k in ("outputs_identical_to_reference","root_verdicts_identical","contigs","sam_lines","wall_s_process","differs")})
d=json.load(open("$M/default.json"))
r=d["roofline"]
print({k: r.get(k) for k in ("kernel","frac","frac_on_traffic","frac_model_all_records","step_traffic_frac","hot_path_frac","hot_path_frac_gated")}, d["value_end_to_end"]["value"], d["cli_end_to_end"]["wall_s_process"], d["cli_end_to_end"]["stages_s"], d["cli_end_to_end"]["outputs_identical_to_reference"])
PY
