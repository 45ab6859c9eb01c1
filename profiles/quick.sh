# quick look: parity of the k-mer build + the default line's kernel times (gpurun -- 'bash profiles/quick.sh [bench args]')
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu --steps 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['device_busy_frac'], d['parity_gate'], (d['parity_gate_timed_step'] or {}).get('ok'))
print({k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
"
python bench.py --no-cpu --pairs 1000000 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('1M', d['value'], d['ms_per_step'], d['device_busy_frac'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.05})"
python bench.py --no-cpu --no-e2e --pairs 100000 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('100k', d['value'], d['ms_per_step'], d['device_busy_frac'], d['wall_ms_per_step'])"
