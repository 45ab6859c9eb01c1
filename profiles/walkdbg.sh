# ablation of k_walk_items (VDJX_WALK_DBG bits: 1 no item stores, 2 no range histogram, 8 no filter/table/key loads at run
# starts); the ablated kernel runs into scratch buffers and is timed as k_walk_dbg beside the real one.
# Round 2, 10 M pairs, per-instance items and one 4-byte successor link per offset (r02a-r02d): whole 7.5 ms; 1: 7.0, 2: 7.5,
# 4 (successor = index + 1 without the load): 4.6, 8: 3.4, 12: 2.5, 15: 2.0 ms.
# Tried on that evidence and dropped (slower or no gain): two records per lane with up-front independent loads (-> 8.6 ms), a
# minimizer filter in front of the k-mer filter (-> 9.2 ms: 24 m-mer hashes per run start make the kernel ALU-bound), the k-mer
# filter of all 16 offsets probed up front as independent loads (-> 9.3 ms: 640 M probes instead of 250 M), and a first chain-order
# renumbering over the SINGLE-successor links (8.5 ms: in deep clones every k-mer has surviving error branches, so those chains
# are one or two nodes long where the instances are).  The heavy-path chains + run items of r02e are what that evidence led to.
# (the switches live in the ablation build only: make -C vdjer_amd/csrc ablate)
for d in 1 2 8 11; do
  VDJX_LIB_PATH=vdjer_amd/libvdjx_ablate.so VDJX_BENCH_ABLATION=1 VDJX_WALK_DBG=$d timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --parity-sample 0 --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('dbg', $d, 'walk_dbg', k.get('k_walk_dbg'), 'walk', k.get('k_walk_items'))"
done
