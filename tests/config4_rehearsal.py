#!/usr/bin/env python3
"""BASELINE.json configs[4] rehearsed on ONE GPU (test infrastructure; run by tests/test_gpu_fullsize.py in a fresh process).

configs[4] = 100 M 50 bp pairs, hash-prefix sharded over 8 GPUs, IGH then IGK then IGL back to back.  No 8-GPU node has been
available to any round, so its GEOMETRY runs here: 8 ranks as threads of one process (tests/fake_dist.py: the real driver
vdjer_amd/shard.py and the real HIP phases, collectives as tensor copies), each with its own context and 12.5 M pairs
(50 M records) of a 100 M-pair library, global instance ids record << 6 | offset running up to 2.56e10 (past 2^32 from the
second rank on), 8 x 1.3 GB of partial aggregates exchanged -- and the three chain presets (set_chain_info, params.c:13-30) one
after the other in the SAME process and contexts (arenas and exchange buffers reused, a new repertoire / ref-dir / pool each).
Checked: every rank ends with the same graph; and that graph is
  - at 8 x 8 M pairs (the most one context takes: records x offsets < 2^32) the graph the ONE-GPU build makes of the union pool
    (itself pinned to the oracle at 10 M and 40 M pairs by the tests around this one),
  - at the full 8 x 12.5 M (the union does not fit one context) the graph 4 ranks of 25 M pairs each make of the same union
    (another partition of the same scan order: any dependence on where the slices are cut shows).
Round 5: the rest of the path on the same shares.  After the build every rank hands its workspace back (vdjx_trim), builds the read
index of ITS pairs and takes part in the sharded window scorer (quick_map_process_contig + coverage_is_valid, A2:841-847: 20,000
candidate windows in batches of 5,000, every rank against its own reads, the pair lists meeting on the window's owner) and in the
SAM records of a few hundred accepted contigs (output_mapping, quick_map3.c:152-181: formatted per rank, merged by key on rank 0).
Checked: verdicts, pair counts and the SAM text of 8 ranks == those of 4 ranks of twice the size (every chain).
Prints one JSON line per chain with the phases' wall times and the bytes a rank exchanged.

usage: config4_rehearsal.py [pairs_per_rank=12500000] [ranks=8] [chains=IGH,IGK,IGL] [score|build]
"""
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests.fake_dist import ThreadDist  # noqa: E402
from vdjer_amd import api, shard, synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


FIELDS = ("first_inst", "freq", "gated_count", "has_v", "has_j", "to_ids", "from_ids")


def names_of(first_pair: int, n: int):
    """read names "r%09d" of pairs first_pair .. first_pair + n - 1, laid end to end, and their offsets"""
    g = np.arange(first_pair, first_pair + n, dtype=np.int64)
    cat = np.empty((n, 10), np.uint8)
    cat[:, 0] = ord("r")
    for j in range(9):
        cat[:, 1 + j] = (g // 10 ** (8 - j)) % 10 + 48
    return cat.reshape(-1), np.arange(n + 1, dtype=np.uint64) * 10


def rank_inputs(metas, first_rank: int, per: int):
    """per-record read info of a rank that holds the pools of consecutive ranks first_rank, first_rank + 1, ... (metas: their
    (pair_id, read_num, is_rc)): pair ids numbered over the block, registration ranks GLOBAL (the order of add_read_info over the
    whole pool, bam_read.c:228,243), names by global pair"""
    pid = np.concatenate([m[0].astype(np.uint32) + np.uint32(i * per) for i, m in enumerate(metas)])
    rn = np.concatenate([m[1] for m in metas])
    rc = np.concatenate([m[2] for m in metas])
    within = np.concatenate([(2 * (m[1].astype(np.uint32) - 1) + m[2]) for m in metas])
    reg = ((np.uint64(first_rank * per) + pid.astype(np.uint64)) * np.uint64(4) + within).astype(np.uint32)
    cat, off = names_of(first_rank * per, len(metas) * per)
    return dict(pair_id=pid, read_num=rn, is_rc=rc, reg=reg, n_pairs=len(metas) * per, names=(cat, off))


def sharded_build(ctxs, dist, dev, pools, stride, vc, jc, k, mf, mq, keep_graph=False, score=None):
    """one sharded k-mer build with len(pools) ranks as threads; -> (digests of rank 0's graph [+ the graph], phase laps of rank 0, bytes per rank).
    score = dict(inputs=[rank_inputs per rank], batches=[[windows]], ins, v_lines, n_contigs): the sharded scorers on the same shares afterwards"""
    world = len(pools)
    out, errs, laps, moved = [None] * world, [], [None] * world, [0] * world

    def work(r):
        try:
            dist.set_rank(r)
            c = ctxs[r]
            c.anchor_sets_load(vc, jc)
            p = c.pool_load_device(pools[r][0], d_secondary=pools[r][1])
            drv = shard.ShardedHotPath(c, dist, dev, stride=stride)
            g = drv.kmer_build(p, k, mf, mq)
            out[r] = {f: sha(getattr(g, f)) for f in FIELDS}
            out[r]["n"], out[r]["pre"] = g.n, g.pre_nodes
            if r == 0:
                out[r]["largest_instance_id"] = int(g.first_inst.max()) if g.n else 0
                if keep_graph:
                    out[r]["graph"] = g
            laps[r] = dict(drv.laps)
            moved[r] = drv.bytes_exchanged
            if score is not None:
                del g
                t0 = time.perf_counter()
                c.trim()                     # the build's workspace (tens of GB per rank) goes back to the device before the scorers start
                inp = score["inputs"][r]
                c.sam_names_load_raw(*inp["names"])
                ri = [torch.from_numpy(np.ascontiguousarray(inp[n_])).to(dev) for n_ in ("pair_id", "read_num", "is_rc", "reg")]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                c.read_index_build_device(p, ri[0].data_ptr(), ri[1].data_ptr(), ri[2].data_ptr(), ri[3].data_ptr(), inp["n_pairs"])
                t2 = time.perf_counter()
                sc = shard.HipScorerEngine(c, dev, 50)
                b0 = drv.bytes_exchanged
                vs, ns = [], []
                for batch in score["batches"]:
                    v_, n_ = drv.window_score(sc, batch, score["ins"])
                    vs.append(np.array(v_))
                    ns.append(np.array(n_))
                valid, npairs = np.concatenate(vs), np.concatenate(ns)
                t3 = time.perf_counter()
                b1 = drv.bytes_exchanged
                flat = [w for b_ in score["batches"] for w in b_]
                contigs = [flat[i][51:411] for i in np.flatnonzero(valid)[:score["n_contigs"]]]
                ids = [f"vjf_{i + 1}_CDR3" for i in range(len(contigs))]
                text = drv.sam_body(c, contigs, ids, ri[3].data_ptr()) if contigs else b""
                t4 = time.perf_counter()
                out[r].update(win_valid=sha(valid), win_npairs=sha(npairs.astype(np.uint32)), windows=int(valid.shape[0]), windows_valid=int(valid.sum()),
                              window_pairs=int(npairs.sum()))
                if r == 0:
                    out[r].update(sam_sha=hashlib.sha256(text).hexdigest(), sam_bytes=len(text), sam_lines=text.count(b"\n"), sam_contigs=len(contigs))
                laps[r].update(scorer_setup_upload=t1 - t0, read_index_of_the_share=t2 - t1, window_score_all_batches=t3 - t2, sam_blocks_and_merge=t4 - t3)
                moved[r] = {"build": moved[r], "window_lists": b1 - b0, "sam_blocks": drv.bytes_exchanged - b1}
                del ri, text
                # this chain's scorers are done: the index, the pair lists and the SAM buffers (about 10 GB per rank here) go back
                c.read_index_drop()
                c.trim()
            p.free()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            dist.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    if errs:
        raise RuntimeError("; ".join(errs[:2]))
    strip = lambda d: {k_: v for k_, v in d.items() if k_ not in ("graph", "largest_instance_id", "sam_sha", "sam_bytes", "sam_lines", "sam_contigs")}      # noqa: E731
    for r in range(1, world):
        assert strip(out[r]) == strip(out[0]), f"rank {r}'s graph or window verdicts differ from rank 0's"
    return out[0], laps[0], moved


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    chains = sys.argv[3].split(",") if len(sys.argv) > 3 else ["IGH", "IGK", "IGL"]
    do_score = (sys.argv[4] if len(sys.argv) > 4 else "score") == "score"
    k, mf, mq = 35, 3, 90
    P = 50 - k + 1
    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info()
    need = int(os.environ.get("VDJX_REHEARSAL_NEED_GIB", "0")) << 30 or world * per * 2300           # (measured: ~170 GB of workspace at 8 x 12.5 M pairs, 79 GB of pools in flight)
    if free < need:
        print(json.dumps({"skipped": f"needs about {need >> 30} GiB of device memory, {free >> 30} GiB free"}))
        return 0
    # one GPU holds records x offsets < 2^32 (include/vdjx.h): the union pool of the full configuration (400 M records x 16) does not
    # fit one context, 8 x 8 M pairs does.  Where it fits, the sharded graph is compared with the one-GPU build of the union pool;
    # where it does not, with a second sharded build of the same union by HALF as many ranks holding twice as much each.
    union_fits = per * world * 4 * P < (1 << 32) and per * world * 4 <= (1 << 29) and not os.environ.get("VDJX_REHEARSAL_HALF")      # (the variable: the half-world check at a small size)
    ctxs = [api.Context(0) for _ in range(world)]
    t_all = time.perf_counter()
    for ci, chain in enumerate(chains):
        t0 = time.perf_counter()
        rep = synth.make_repertoire(max(4, per * world // 1000), seed=20261002 + 7 * ci, chain=chain)
        vc = np.array(sorted({synth.seq_to_int(a) for a in rep.v_anchors}), dtype=np.uint32)
        jc = np.array(sorted({synth.seq_to_int(a) for a in rep.j_anchors}), dtype=np.uint32)
        # two ranks' pools lie in one block (the pool of one rank of the half-world cross-check: the same scan order)
        blocks, pools, metas = [], [], []
        for j in range(0, world, 2):
            two = [synth.make_reads_cb(rep, per, noise_frac=0.3, seed=20261002 + 1000 * ci, device="cuda:0", pair0=r * per) for r in (j, j + 1) if r < world]
            metas += [(p_.pair_id, p_.read_num, p_.is_rc) for p_ in two]
            blk = torch.cat([x for p_ in two for x in (p_.primary, p_.secondary)])
            at = 0
            for p_ in two:                 # (a rank's primary and secondary records as one block: the same scan order, and a 16-byte aligned base)
                n_ = p_.primary.shape[0] + p_.secondary.shape[0]
                assert (at * blk.shape[1]) % 16 == 0
                pools.append((blk[at:at + n_], None))
                at += n_
            blocks.append(blk)
            del two
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        stride = pools[0][0].shape[0]
        assert all(p_[0].shape[0] == stride for p_ in pools)
        # the scorers' inputs: 20,000 of the clones' candidate windows (the window the reference derives for a clone's transcript), in
        # batches of 5,000 like the traversal's calls; the SAM records of the first 400 accepted ones' contigs
        score = None
        if do_score:
            wins_all = [w for w in rep.windows() if w]
            wins = wins_all[::max(1, len(wins_all) // 20000)][:20000]
            batches = [wins[a:a + 5000] for a in range(0, len(wins), 5000)]
            score = dict(inputs=[rank_inputs([metas[r]], r, per) for r in range(world)], batches=batches, ins=175, n_contigs=400)
        t1 = time.perf_counter()
        d8, laps, moved = sharded_build(ctxs, ThreadDist(world), dev, pools, stride, vc, jc, k, mf, mq, keep_graph=union_fits, score=score)
        t_shard = time.perf_counter() - t1
        line = {"chain": chain, "ranks": world, "pairs_per_rank": per, "records": stride * world, "largest_instance_id": d8["largest_instance_id"],
                "instance_ids_pass_2_32": d8["largest_instance_id"] >= 1 << 32, "nodes": int(d8["n"]), "pre_nodes": int(d8["pre"]), "all_ranks_agree": True,
                "seconds": {"generate": round(t_gen, 2), "sharded_build_all_ranks_on_one_device": round(t_shard, 2)},
                "phase_wall_ms_rank0": {k_: round(v * 1e3, 2) for k_, v in laps.items()}, "bytes_exchanged_per_rank": moved,
                "shard_stats_rank0": {n_: ctxs[0].stat("shard_" + n_) for n_ in ("partials_received", "open_kmers", "questions", "decided_at_merge", "kept_after_answers")}}
        if score is not None:
            line["scorers"] = {n_: d8[n_] for n_ in ("windows", "windows_valid", "window_pairs", "sam_contigs", "sam_lines", "sam_bytes")}
        same = True
        if union_fits:
            # the union pool in ONE context: rank-major concatenation = the sharded build's scan order
            t2 = time.perf_counter()
            cat = torch.cat(blocks)
            torch.cuda.synchronize()           # (torch's stream made it; the library reads it on its own)
            del pools, blocks
            # (the ranks' contexts keep their workspaces between builds -- 8 x ~17 GB here; the one-GPU build of 64 M pairs wants
            # ~150 GB of its own: it gets a fresh context, the ranks' ones are closed.  This invocation has no further chain.)
            assert ci == len(chains) - 1, "the union check closes the ranks' contexts: last chain only"
            for c in ctxs:
                c.close()
            ctxs[:] = [api.Context(0)]
            ctxs[0].anchor_sets_load(vc, jc)
            pu = ctxs[0].pool_load_device(cat)
            gu = ctxs[0].kmer_build(pu, k, mf, mq)
            gs = d8["graph"]
            same = gu.n == gs.n and gu.pre_nodes == gs.pre_nodes and all(np.array_equal(getattr(gu, f), getattr(gs, f)) for f in FIELDS)
            line["sharded_equals_one_gpu_build_of_the_union_pool"] = bool(same)
            line["seconds"]["union_build_one_context"] = round(time.perf_counter() - t2, 2)
            pu.free()
            del cat, gu, gs
        elif (ci == len(chains) - 1 or do_score) and world % 2 == 0:
            # the same union through world/2 ranks of twice the size -- the last chain only without the scorers (the other contexts are
            # closed first: their arenas go back), every chain with them (the chain-specific code lives in the scorers' inputs)
            last = ci == len(chains) - 1
            if last:
                for c in ctxs[world // 2:]:
                    c.close()
                del ctxs[world // 2:]
            else:
                for c in ctxs:
                    c.trim()
            del pools
            t2 = time.perf_counter()
            score4 = None
            if score is not None:
                score4 = dict(score, inputs=[rank_inputs(metas[2 * j:2 * j + 2], 2 * j, per) for j in range(world // 2)])
            d4, laps4, moved4 = sharded_build(ctxs[:world // 2], ThreadDist(world // 2), dev, [(b, None) for b in blocks], 2 * stride, vc, jc, k, mf, mq, score=score4)
            strip = lambda d: {k_: v for k_, v in d.items() if k_ not in ("graph", "largest_instance_id")}      # noqa: E731
            same = strip(d4) == strip(d8)
            if score is not None:
                line["scorers_equal_those_of_half_as_many_ranks"] = bool(all(d4[n_] == d8[n_] for n_ in ("win_valid", "win_npairs", "sam_sha")))
                line["bytes_exchanged_per_rank_half_world"] = moved4
            line[f"equals_the_build_by_{world // 2}_ranks_of_twice_the_size"] = bool(same)
            line["seconds"]["half_world_build"] = round(time.perf_counter() - t2, 2)
            line["phase_wall_ms_rank0_half_world"] = {k_: round(v * 1e3, 2) for k_, v in laps4.items()}
            del blocks
        else:
            del pools, blocks
        line["device_free_GiB_after"] = round(torch.cuda.mem_get_info()[0] / 2 ** 30, 1)
        print(json.dumps(line), flush=True)
        del d8
        if not same:
            return 1
    print(json.dumps({"done": True, "seconds": round(time.perf_counter() - t_all, 1), "device_total_GiB": round(total / 2 ** 30, 1), "device_free_GiB_at_start": round(free / 2 ** 30, 1),
                      "torch_peak_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
