# ablation of k_walk_items (VDJX_WALK_DBG bits: 1 no item stores, 2 no range histogram, 4 successor = index+1 without the load,
# 8 no filter/table/key loads at run starts); the ablated kernel runs into scratch buffers and is timed as k_walk_dbg beside the
# real one.  Round 2, 10 M pairs: whole 7.5 ms; 1: 7.0, 2: 7.5, 4: 4.6, 8: 3.4, 12: 2.5, 15: 2.0 ms.
# Tried on that evidence and dropped (slower or no gain): survivors renumbered in chain order with a one-byte successor code
# (walk 7.4 -> 8.5 ms plus 3.1 ms of renumbering), two records per lane with up-front independent loads (-> 8.6 ms), a minimizer
# filter in front of the k-mer filter (-> 9.2 ms: 24 m-mer hashes per run start make the kernel ALU-bound), the k-mer filter of all
# 16 offsets probed up front as independent loads (-> 9.3 ms: 640 M probes instead of 250 M; the filter's 2 MB do not stay in the
# XCD's 4 MB L2 beside 4.5 GB of streamed records and items, so the kernel is bound by the NUMBER of random line fills, 35 GB of
# fabric traffic at 4.7 TB/s, not by their latency).
for d in 1 2 4 8 12 15; do
  VDJX_WALK_DBG=$d timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --parity-sample 0 --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('dbg', $d, 'walk_dbg', k.get('k_walk_dbg'), 'walk', k.get('k_walk_items'))"
done
