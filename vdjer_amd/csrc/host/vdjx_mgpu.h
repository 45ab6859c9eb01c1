/*
 * vdjx_mgpu.h -- `vdjer --gpus N` (one process per GPU): the hot path of include/vdjx.h over a read pool that is sharded across the
 * ranks, driven from C; the bytes between ranks move through vdjx_comm (RCCL over xGMI, or -- ranks sharing one device in the tests --
 * host sockets).  No counterpart in the reference (its only parallelism is pthreads over roots, A2:1287-1348).
 *
 * The pool is split BY PAIR (both mates of a pair, all four records, on one rank: the share of a rank).  A rank only ever holds its
 * share on the host.  On the device it holds
 *   - its share, packed, with the read index over it (the scorers: every rank maps every window / contig against ITS reads), and
 *   - for the duration of the k-mer build its SLICE of the scan order (records [rank*S, (rank+1)*S) of primary-then-secondary,
 *     A2:1388-1390), which the ranks deal out to each other from their shares with one all-to-all of the ASCII records
 *     (vdjx_mgpu_load): the k-mer build's instance ids are scan positions.
 * Rank 0 runs the serial traversal (north star: host-side) and calls the collective scorers; the other ranks wait for its commands
 * in vdjx_mgpu_serve.
 */
#ifndef VDJX_MGPU_H
#define VDJX_MGPU_H

#include <stdint.h>
#include <stdio.h>

#include "../../../include/vdjx.h"
#include "vdjx_comm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdjx_mgpu vdjx_mgpu;

/* takes over `cm` (freed with the handle) */
int vdjx_mgpu_init(vdjx_comm* cm, int device, vdjx_mgpu** out);
void vdjx_mgpu_free(vdjx_mgpu* m);
const char* vdjx_mgpu_last_error(void);
uint64_t vdjx_mgpu_bytes_sent(const vdjx_mgpu* m);
/* VDJX_TIMES set: this rank's wall milliseconds and bytes sent per phase, one VDJX_MGPU_PHASE line each (rank 0 prints them in vdjx_mgpu_finish) */
void vdjx_mgpu_print_times(const vdjx_mgpu* m, FILE* f);

/* This rank's share of the pool, as the extraction wrote it (bam_read.c:206-244): records of 2*rl+1 bytes, its primary-pool records
 * followed by its secondary-pool ones, in extraction order; per record
 *   scan_index   its position in the scan order of the WHOLE pool (ascending along the share)
 *   pair_id      the share's own numbering of the read names (0 .. n_pairs-1), read_num 1|2, is_rc (add_read_info, quick_map3.c:126-149)
 *   reg_rank     the order of the add_read_info calls over the WHOLE pool (global)
 * total_records = records of the whole pool (the same on every rank).  Collective.  Afterwards the context holds the read index of the
 * share (with vdjx_sam_names_load left to the caller) and the handle the share's packed pool and scan positions for vdjx_mgpu_kmer_build.
 * No record moves between ranks (round 4 dealt the ASCII pool out again into slices of the scan order: 5 GB per rank at configs[4]). */
int vdjx_mgpu_load(vdjx_mgpu* m, vdjx_ctx* ctx, const uint8_t* records, size_t n_records, int rl, const uint32_t* scan_index, const uint32_t* pair_id,
                   const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs, uint64_t total_records);
/* collective: the sharded k-mer build (vdjx_shard_begin_share ...) over the shares; the same graph on every rank. */
int vdjx_mgpu_kmer_build(vdjx_mgpu* m, vdjx_ctx* ctx, int k, int mf, int mq, vdjx_graph** out);
/* the same over pools the caller made itself (every rank passes its slice: records [rank*rec_stride, ...) of the scan order) */
int vdjx_mgpu_kmer_build_pool(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, uint64_t rec_stride, vdjx_graph** out);

/* the same over a pool the caller packed itself that is this rank's SHARE of a pool of total_records records: record i at scan position
 * d_scan_index[i] (device array, ascending; it must stay valid until the call returns) */
int vdjx_mgpu_kmer_build_share(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, const uint32_t* d_scan_index, uint64_t total_records, vdjx_graph** out);
/* the read length of the pools the scorer calls work on: set by vdjx_mgpu_load; a caller with pools of its own (vdjx_mgpu_kmer_build_pool /
 * _share) says so before the first vdjx_mgpu_window_score */
void vdjx_mgpu_set_read_length(vdjx_mgpu* m, int rl);
/* collective: the ranks' largest value of `mine` (e.g. the record stride of vdjx_mgpu_kmer_build_pool: the largest pool of any rank) */
int vdjx_mgpu_agree_max(vdjx_mgpu* m, uint64_t mine, uint64_t* most);

/* ---- rank 0 (the others are inside vdjx_mgpu_serve) ---- */
/* quick_map_process_contig + coverage_is_valid (A2:841-847) for n windows over the sharded pool: every rank maps every window against
 * its reads, window w's pair lists meet on rank w % N, which tests their union */
int vdjx_mgpu_window_score(vdjx_mgpu* m, vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p, uint8_t* out_valid);
/* the same, and out_npairs[w] (may be NULL) = the mapped pairs of window w over all shares */
int vdjx_mgpu_window_score2(vdjx_mgpu* m, vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p, uint8_t* out_valid, uint32_t* out_npairs);
/* the SAM records of the mapped pairs of n contigs (quick_map_process_contig_file -> output_mapping, quick_map3.c:152-181, 311-340) in
 * the reference's order: every rank formats its pairs' records (vdjx_sam_blocks), rank 0 merges them (vdjx_sam_merge).  The text
 * belongs to the context (valid until the next call). */
int vdjx_mgpu_sam_body(vdjx_mgpu* m, vdjx_ctx* ctx, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off,
                       const char** out_text, uint64_t* out_bytes);
/* releases the other ranks */
int vdjx_mgpu_finish(vdjx_mgpu* m);
/* ---- ranks other than 0: serves rank 0's calls until vdjx_mgpu_finish; 0 when released ---- */
int vdjx_mgpu_serve(vdjx_mgpu* m, vdjx_ctx* ctx);
/* A job of several steps (bench.py --gpus N: every step a collective build, then rank 0's scorer calls): rank 0 ends a step's scorer
 * calls with vdjx_mgpu_yield, the others' vdjx_mgpu_serve_step returns there (*released = 0) -- or when rank 0 finishes (*released = 1).
 * out_valid / out_npairs (cap entries each, may be NULL): verdicts and pair counts of the last window call served; *n_out its windows. */
int vdjx_mgpu_yield(vdjx_mgpu* m);
int vdjx_mgpu_serve_step(vdjx_mgpu* m, vdjx_ctx* ctx, uint8_t* out_valid, uint32_t* out_npairs, size_t cap, size_t* n_out, int* released);

#ifdef __cplusplus
}
#endif
#endif
