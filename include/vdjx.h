/*
 * include/vdjx.h -- C ABI of libvdjx.so, the MI355X (gfx950) implementation of V'DJer's hot path.
 *
 * The reference (mozack/vdjer) has no plugin/FFI layer: it is one C++ program whose hot path is a
 * set of internal call sites.  Each entry point below replaces one of those call
 * sites; the "replaces:" lines cite the reference interface under /root/reference/src/main/c
 * (A2 = assembler2_vdj.c).  INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions: plain C types only; every call returns 0 on success or a negative VDJX_E* code with a
 * message available from vdjx_last_error(); nothing ever calls exit().  One host thread drives one
 * context; calls are synchronous (internally they run on the context's HIP stream).  Host pointers
 * unless a parameter is named d_* (device pointer, same device as the context).
 */
#ifndef VDJX_H
#define VDJX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDJX_OK 0
#define VDJX_EINVAL (-1)   /* bad argument */
#define VDJX_EHIP (-2)     /* HIP runtime error */
#define VDJX_ELIMIT (-3)   /* input exceeds a documented limit: rl <= 160, k <= 50; 2^32 records in all (2^30 with reads of more than 64 bases); per GPU
                            * 2^29 records (2^27 with reads of more than 64 bases), 2^26 surviving k-mers, 2^26 - 1 distinct read sequences in the read index */
#define VDJX_ESTATE (-4)   /* call order violated (e.g. scorer used before its index was loaded) */

#define VDJX_MAX_READ_LEN 160   /* the reference takes up to 255 (bam_read.c:208 `char seq[256]`); reads of up to 64 bases run on the short-read kernels */
#define VDJX_SHORT_READ_LEN 64
#define VDJX_MAX_KMER 50    /* A2:70 MAX_KMER_LEN */

typedef struct vdjx_ctx vdjx_ctx;
typedef struct vdjx_pool vdjx_pool;
typedef struct vdjx_graph vdjx_graph;

const char* vdjx_last_error(void);
const char* vdjx_version(void);

/* One context per GPU (one process per GPU in multi-GPU runs). */
int vdjx_init(int device, vdjx_ctx** out);
void vdjx_shutdown(vdjx_ctx* ctx);
/* blocks until all work queued on the context's stream is done */
int vdjx_sync(vdjx_ctx* ctx);
/* gives the workspaces' device memory back (they hold the PEAK of the calls so far: tens of GB after a k-mer build of 10 M pairs) and with
 * them the scorers' grow-only result buffers (pair lists of the last window batch, mapped pairs and SAM records of the last contigs); the
 * next call maps again what it needs.  For a process that builds once and then serves scorer calls (a rank of `vdjer --gpus N`).  Not
 * between the two calls of a two-call protocol (vdjx_map_emit count / write, vdjx_window_pairs / _fetch), not during a sharded build. */
int vdjx_trim(vdjx_ctx* ctx);
/* drops the context's read index and frees its arrays (kept from build to build otherwise); the scorers need a new index afterwards */
int vdjx_read_index_drop(vdjx_ctx* ctx);
/* `bytes` bytes from one device buffer to another (synchronous): for drivers that move library-owned device results (vdjx_sam_blocks)
 * into exchange buffers of their own */
int vdjx_device_copy(vdjx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* ---- a-0: read pool -------------------------------------------------------------------------
 * replaces: the two NUL-terminated ASCII pools handed to assemble() (A2:1350-1358, 1545-1554),
 * produced by add_to_buffer (bam_read.c:206-244): records of 2*rl+1 bytes, '0' + rl bases + rl
 * Phred+33 characters, primary pool scanned before secondary (A2:1388-1390).
 * Packs the pool on the device (2-bit bases A0 T1 C2 G3 as seq_to_kmer.c:6-29, N mask, Phred<20
 * mask, quality bytes).  Bases other than ACGT are treated as 'N'.  The caller keeps the ASCII. */
int vdjx_pool_load(vdjx_ctx* ctx, const uint8_t* primary, size_t n_primary,
                   const uint8_t* secondary, size_t n_secondary, int rl, vdjx_pool** out);
/* The same pools given as the reads only, n records of 2*rl+1 bytes as add_to_buffer writes the FIRST of each read's two
 * records (bam_read.c:219-230); the reverse-complement record it writes next (bam_read.c:231-243: complemented bases in reverse
 * order, reversed qualities) is derived on the device: record 2i is read i, record 2i+1 its reverse complement, half the bytes
 * cross PCIe.  The resulting pool is identical to vdjx_pool_load's on the full buffers. */
int vdjx_pool_load_forward(vdjx_ctx* ctx, const uint8_t* primary_reads, size_t n_primary_reads,
                           const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out);
/* vdjx_pool_load_forward without waiting: upload and packing run on the context's copy stream beside whatever the main stream is
 * computing on another pool (page-locked buffers, vdjx_host_alloc, make the upload a DMA); the pool may be used after vdjx_pool_wait,
 * which also reports a malformed pool. */
int vdjx_pool_load_forward_begin(vdjx_ctx* ctx, const uint8_t* primary_reads, size_t n_primary_reads,
                                 const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out);
int vdjx_pool_wait(vdjx_pool* pool);
/* The reads in a PACKED host format, for callers that can produce it (an extraction that converts BAM's 4-bit bases itself): per read
 * vdjx_packed_read_bytes(rl) bytes -- ceil(rl/4) bytes of 2-bit bases (A0 T1 C2 G3 as seq_to_kmer.c:6-29, the first base in the top two
 * bits of byte 0, code 0 for a base that is not ACGT), then rl quality bytes (Phred+33, bit 7 set where the base is not ACGT), zero
 * padding to a multiple of 16: 64 bytes for a 50 bp read where add_to_buffer's record (bam_read.c:219-230) has 101 -- 128 bytes per
 * pair over PCIe instead of 202.  Reads of up to 64 bases.  Otherwise exactly vdjx_pool_load_forward[_begin]: record 2i is read i,
 * record 2i+1 its reverse complement with reversed qualities.  vdjx_pack_reads converts n ASCII records on the host (plain C). */
size_t vdjx_packed_read_bytes(int rl);
int vdjx_pack_reads(const uint8_t* ascii_reads, size_t n, int rl, uint8_t* out_packed);
int vdjx_pool_load_packed(vdjx_ctx* ctx, const uint8_t* primary_reads, size_t n_primary_reads,
                          const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out);
int vdjx_pool_load_packed_begin(vdjx_ctx* ctx, const uint8_t* primary_reads, size_t n_primary_reads,
                                const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out);
/* same, ASCII pools already resident in device memory (16-byte aligned).  The two buffers must stay valid and unchanged until
 * vdjx_pool_free: bases and masks are packed, but the quality characters are NOT copied -- the few k-mers whose quality sums
 * matter (count below 1 + ceil(mq/20), A2:454-465) read them from the records where they lie (a third of the packing's bytes). */
int vdjx_pool_load_device(vdjx_ctx* ctx, const uint8_t* d_primary, size_t n_primary,
                          const uint8_t* d_secondary, size_t n_secondary, int rl, vdjx_pool** out);
size_t vdjx_pool_records(const vdjx_pool* pool);
void vdjx_pool_free(vdjx_pool* pool);

/* ---- a-6: V/J anchor sets ---------------------------------------------------------------------
 * replaces: vjf_init -> load_kmers (vj_filter.c:56-68, 311-340): the codes whose distance column
 * passed --am.  Kept as two 2^32-bit bitmaps in HBM; code 0 is never a member (vj_filter.c:317-318). */
int vdjx_anchor_sets_load(vdjx_ctx* ctx, const uint32_t* v_codes, size_t nv, const uint32_t* j_codes, size_t nj);
/* replaces: seq_to_int + matches_vmer/matches_jmer over every offset of a contig
 * (vj_filter.c:221-238): out_v/out_j[i] for i in [0, len-16) */
int vdjx_anchor_probe(vdjx_ctx* ctx, const char* contig, int len, uint8_t* out_v, uint8_t* out_j);

/* ---- a-1, a-2, a-3: k-mer table, prune, graph build --------------------------------------------
 * replaces: build_pre_graph x2 + prune_pre_graph + build_graph2 x2 (A2:1388-1408; bodies
 * A2:240-259, 322-367, 454-484, 267-320, 190-237).  The result is everything the host-side
 * traversal needs without re-scanning the pool: nodes in creation order with their ordered edge
 * lists. */
int vdjx_kmer_build(vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, vdjx_graph** out);
size_t vdjx_graph_nodes(const vdjx_graph* g);
/* distinct gated k-mers before the prune ("Pre Num nodes", A2:407) */
size_t vdjx_graph_pre_nodes(const vdjx_graph* g);
/* Node i (0-based; reference node id = i+1, A2:188,200), in creation order:
 *   first_inst  record << 6 | offset of the first (ungated) occurrence, records counted over primary then secondary
 *               (pools of reads longer than 64 bases: record << 8 | offset)
 *   gated_count frequency of the pre_node (A2:130,345-347), saturated at 32765
 *   freq        node frequency (A2:119,261-265), saturated at 32765
 *   has_v/has_j A2:288-303
 *   to_deg/from_deg <= 4; to_ids/from_ids [n][4] 1-based node ids in list order (head of the
 *   reference's prepend list first, A2:223-237)
 *   kmers       n*k ASCII characters (may be NULL)                                              */
int vdjx_graph_export(const vdjx_graph* g, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq,
                      uint8_t* has_v, uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids,
                      uint8_t* from_deg, uint32_t* from_ids, char* kmers);
/* the same copies started on a second stream: they run beside whatever is done next with the context (e.g. vdjx_root_score_graph
 * on the device-resident graph); the arrays are valid after vdjx_graph_export_end.  Pinned arrays (vdjx_host_alloc) make it a DMA. */
int vdjx_graph_export_begin(const vdjx_graph* g, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq,
                            uint8_t* has_v, uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids,
                            uint8_t* from_deg, uint32_t* from_ids, char* kmers);
/* The ten arrays lie in one block on the device.  offsets[10]: where each starts in it, in the order of vdjx_graph_export's arguments
 * (every array holds vdjx_graph_nodes entries; ids in rows of 4, k-mers in rows of k); *bytes: the bytes that hold them all.
 * vdjx_graph_export_block copies those bytes into host_block (same layout) in ONE transfer; _begin does it on the second stream
 * (vdjx_graph_export_end waits).  For callers that export small graphs often: ten transfers cost more in calls than in bytes. */
int vdjx_graph_block_layout(const vdjx_graph* g, uint64_t* offsets, uint64_t* bytes);
int vdjx_graph_export_block(const vdjx_graph* g, void* host_block);
int vdjx_graph_export_block_begin(const vdjx_graph* g, void* host_block);
int vdjx_graph_export_end(const vdjx_graph* g);
void vdjx_graph_free(vdjx_graph* g);

/* ---- f-3: the v_index / j_index generator -------------------------------------------------------
 * replaces: process_kmers(anchors file, start, end) (seq_dist.c:49-71; its main() is commented out, :73-98) which
 * printed "<code>\t<min base distance to any anchor>" for every 16-base code of [start, end] (inclusive) whose
 * distance is <= MAX_DIST 5: the rows of <ref-dir>/v_index and j_index (load_kmers, vj_filter.c:56-68).
 * anchors: seq_to_int codes of the anchor 16-mers.  Rows come in ascending code order; *n_rows = how many
 * exist, the first min(cap, *n_rows) are written (call with cap 0 to count).                           */
int vdjx_index_generate(vdjx_ctx* ctx, const uint32_t* anchors, size_t n_anchors, uint64_t start, uint64_t end, int max_dist,
                        uint64_t cap, uint64_t* n_rows, uint32_t* codes, uint8_t* dists);
/* The two membership sets of a-6 straight from the anchors, without 10^7..10^8-row files in between: exactly the sets
 * vdjx_anchor_sets_load would hold after load_kmers(v_index, am) / load_kmers(j_index, am) on generated files
 * (distance <= min(am, 5); code 0 never a member).                                                   */
int vdjx_anchor_sets_from_anchors(vdjx_ctx* ctx, const uint32_t* v_anchors, size_t nv, const uint32_t* j_anchors, size_t nj, int am);

/* ---- result buffers ------------------------------------------------------------------------------
 * Every result pointer of this interface may be ordinary host memory.  Memory from vdjx_host_alloc is
 * page-locked: copies into it run at DMA speed (the graph of 1 M pairs is ~11 MB, the mapped pairs ~11 MB).
 * Windows / contigs handed to the scorers in such memory are not copied at all: the classifying kernel reads them where they lie
 * (each character once, over PCIe), and they must stay unchanged until the call returns -- as for any argument.
 * No counterpart in the reference (its tables are host memory throughout).                          */
int vdjx_host_alloc(vdjx_ctx* ctx, size_t bytes, void** out);
void vdjx_host_free(vdjx_ctx* ctx, void* p);
/* dst row i = bytes [first, first + len) of src row idx[i] (rows `stride` bytes apart): the contigs of the accepted windows, laid
 * end to end for vdjx_map_emit (output_windows hands the [51,411) slice of a window that passed coverage on, A2:841-847,872-914). */
int vdjx_host_take_rows(void* dst, const void* src, size_t stride, size_t first, size_t len, const uint32_t* idx, size_t n);

/* ---- a-7: root (V-region homology) scorer ------------------------------------------------------
 * replaces: score_seq_init(k, 1000, v_region.fa) (seq_score.c:50-70) and score_seq(kmer, thr)
 * (seq_score.c:118-158) as called per root by worker_thread (A2:1103).
 * lines: the non-header lines of v_region.fa, newline stripped; each must be longer than 2k.     */
int vdjx_vregion_load(vdjx_ctx* ctx, const char* const* lines, size_t n_lines, int vk);
/* kmers: n*k ASCII; out[i] = 0|1 */
int vdjx_root_score(vdjx_ctx* ctx, const char* kmers, size_t n, int k, int threshold, uint8_t* out);
/* The same scorer over the roots of a graph that is still on the device (identify_root_nodes, A2:653-676: the nodes
 * without predecessor), so that root k-mers never travel to the host and back.  vdjx_graph_roots = their number;
 * roots are taken in ascending node id, this call handles the ones at positions first, first+stride, ...
 * (vdjx_root_part of them: one call with (0,1) on one GPU, (rank,nranks) when sharded).
 * root_ids[i] = 1-based node id, out[i] = 0|1.                                                    */
size_t vdjx_graph_roots(const vdjx_graph* g);
size_t vdjx_root_part(const vdjx_graph* g, uint32_t first, uint32_t stride);
int vdjx_root_score_graph(vdjx_ctx* ctx, const vdjx_graph* g, int threshold, uint32_t first, uint32_t stride,
                          uint32_t* root_ids, uint8_t* out);
/* The same without waiting: everything is queued on the context's stream and the call returns; the two arrays (page-locked memory,
 * vdjx_host_alloc) are valid after vdjx_root_score_graph_end.  Calls made in between run behind it on the device, so a caller with
 * other work that does not need the verdicts (scoring windows it already has) keeps the device busy instead of waiting for a few
 * kilobytes.  The graph stays alive until _end; one call in flight per context. */
int vdjx_root_score_graph_begin(vdjx_ctx* ctx, const vdjx_graph* g, int threshold, uint32_t first, uint32_t stride,
                                uint32_t* root_ids, uint8_t* out);
int vdjx_root_score_graph_end(vdjx_ctx* ctx);

/* ---- a-8, a-9, a-10: read->contig mapper, coverage validator, SAM placements --------------------
 * replaces: add_read_info (quick_map3.c:126-149) for the index; quick_map_process_contig +
 * coverage_is_valid as called per candidate window by output_contig (A2:841-847; quick_map3.c:188-266,
 * coverage.c:10-130); quick_map_process_contig_file -> output_mapping for the final contigs
 * (quick_map3.c:152-181, 311-340).
 * Per pool record (scan order): pair id (identity of the read name), read_num 1|2, is_rc, and the
 * registration rank (order of the add_read_info calls).                                           */
int vdjx_read_index_build(vdjx_ctx* ctx, const vdjx_pool* pool, const uint32_t* pair_id,
                          const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs);
/* the same with the four per-record arrays already in device memory (e.g. written there by the extraction side) */
int vdjx_read_index_build_device(vdjx_ctx* ctx, const vdjx_pool* pool, const uint32_t* d_pair_id,
                                 const uint8_t* d_read_num, const uint8_t* d_is_rc, const uint32_t* d_reg_rank, uint32_t n_pairs);

/* The same two calls begun and ended: the index is built on a stream and out of a workspace of its own, by a thread of the library,
 * BESIDE whatever the caller does next with the context -- the k-mer build of the same pool above all: both only read the packed
 * records, and the reference orders them only by accident of its call sequence (add_read_info runs inside extract, bam_read.c:228,243,
 * before A2:1388; nothing reads the index before the first quick_map_process_contig, A2:841).  Everything queued on the context
 * before _begin comes first (the packing of `pool`; scorer calls that still read the index being replaced).  _end returns the
 * build's status; the first call that needs the index (vdjx_window_score, vdjx_window_pairs, vdjx_map_emit, vdjx_sam_text ...) ends a
 * build that was not.  _begin (host arrays): the four arrays must stay valid until _end.  One build in flight per context; `pool`
 * may not be freed before _end (vdjx_pool_free waits for it). */
int vdjx_read_index_build_begin(vdjx_ctx* ctx, const vdjx_pool* pool, const uint32_t* pair_id,
                                const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs);
int vdjx_read_index_build_device_begin(vdjx_ctx* ctx, const vdjx_pool* pool, const uint32_t* d_pair_id,
                                       const uint8_t* d_read_num, const uint8_t* d_is_rc, const uint32_t* d_reg_rank, uint32_t n_pairs);
int vdjx_read_index_build_end(vdjx_ctx* ctx);

typedef struct {
	int eval_start;    /* --e0 */
	int eval_stop;     /* --e1 */
	int read_span;     /* --rs */
	int mate_span;     /* --ms */
	int insert_low;    /* --ins */
	int insert_high;   /* --ins */
	int floor;         /* --rf; 0 disables the coverage check (A2:846) */
} vdjx_cov_params;

/* windows: n strings of `len` chars, stride `len`.  out_valid[i] = coverage_is_valid(...),
 * out_npairs[i] = mapped pairs.  Any n: the windows' pair lists (8 bytes per distinct hit) are built in device memory, and a call
 * whose lists would not fit takes its windows in slices (vdjx_stat "window_slices"); one window's lists have to fit. */
int vdjx_window_score(vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p,
                      uint8_t* out_valid, uint32_t* out_npairs);

/* The two halves of vdjx_window_score, for a read index that holds only this rank's share of the pairs (multi-GPU: the pool is
 * split BY PAIR, both mates of a pair on one rank).  Every rank maps every window against its own reads:
 *   vdjx_window_pairs        out_entries[i] = entries of window i's pair list (identical read pairs are one entry with a
 *                            multiplicity), out_npairs[i] = mapped pairs (quick_map3.c:223-245); the lists stay on the device
 *   vdjx_window_pairs_fetch  the lists of the m named windows laid end to end (8 bytes per entry) into d_out: what travels
 *                            to the ranks that own those windows
 * and the owner of a window tests the union of what all ranks found (coverage_is_valid only counts entries, coverage.c:64-130):
 *   vdjx_window_cover        d_lists = the lists as received: source after source, inside a source window after window;
 *                            counts[s*n + w] = entries of window w from source s                                       */
int vdjx_window_pairs(vdjx_ctx* ctx, const char* windows, size_t n, int len, uint32_t* out_entries, uint32_t* out_npairs);
int vdjx_window_pairs_fetch(vdjx_ctx* ctx, const uint32_t* window_ids, size_t m, void* d_out);
int vdjx_window_cover(vdjx_ctx* ctx, size_t n, int len, int rl, const vdjx_cov_params* p, const void* d_lists, size_t nsrc,
                      const uint32_t* counts, uint8_t* out_valid);

typedef struct {
	uint32_t pair_id;
	uint32_t rec1, rec2;       /* pool record (scan order) that matched for read 1 / read 2 */
	int16_t pos1, pos2, insert;
	uint8_t rc1, rc2;
} vdjx_pair;

/* Mapped pairs of each contig in the reference's output order.  Two-call protocol: first call with
 * pairs == NULL fills offsets[n+1]; second call with pairs sized offsets[n]. */
int vdjx_map_emit(vdjx_ctx* ctx, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs);
/* The writing call with the transfer to the host left running on the context's copy stream (page-locked `pairs`, vdjx_host_alloc,
 * make it a DMA beside the next kernels); the array is valid after vdjx_map_emit_end. */
int vdjx_map_emit_begin(vdjx_ctx* ctx, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs);
int vdjx_map_emit_end(vdjx_ctx* ctx);

/* The SAM records themselves, formatted on the device.
 * replaces: output_mapping (quick_map3.c:152-181) as called through quick_map_process_contig_file (:311-340): per mapped pair the two
 * lines "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n" (:168) -- read name without its leading '@', flag, contig id, position, read
 * length, mate position, insert, then the stored sequence and qualities of the record that matched -- contig after contig, in the order
 * of vdjx_map_emit.  (The header lines, output_header :274-309, are the caller's: they need only the contig ids.)
 *   vdjx_sam_names_load  the read names by pair id (the pair_id of vdjx_read_index_build): names[name_off[p] .. name_off[p+1])
 *   vdjx_sam_text        contig ids likewise (ids[id_off[c] .. id_off[c+1])); *out_text points at *out_bytes bytes (NUL after them) in a
 *                        page-locked buffer owned by the context, valid until the next vdjx_sam_text call
 * Bases that are not ACGT come out as N (they are stored as N: see vdjx_pool_load).                                                   */
int vdjx_sam_names_load(vdjx_ctx* ctx, const char* names, const uint64_t* name_off, uint32_t n_pairs);
int vdjx_sam_text(vdjx_ctx* ctx, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off,
                  const char** out_text, uint64_t* out_bytes);

/* ---- the same records for a pool that is sharded BY PAIR over several GPUs (row e; no counterpart in the reference) -----------------
 * Every rank formats the records of ITS pairs and leaves them on its device; the caller brings the ranks' results to one rank, which
 * lays them out in the order output_mapping would have written them (quick_map3.c:152-181 called per contig from :311-340; inside a
 * contig the order of quick_map_process_contig's lists, :199-245: offsets ascending, the instances of a read in registration order).
 *   vdjx_sam_blocks   maps `contigs` against this context's read index; per mapped pair ("block") the two SAM lines, their byte
 *                     count (u32) and a 64-bit key (contig << 44 | read-1 position << 32 | registration rank of the read-1 record;
 *                     d_reg_rank: the rank of every record of the index's pool, GLOBAL over all shards).  The three arrays stay on the
 *                     device, owned by the context, valid until the next vdjx_sam_blocks / vdjx_sam_text call.  Fewer than 2^20 contigs
 *                     of fewer than 4096 bases per call.
 *   vdjx_sam_merge    d_keys / d_lens / d_text = the blocks of all sources, source after source (every source's text the concatenation of
 *                     its blocks in its own order); *out_text = the n_bytes of text in ascending key order, in a page-locked buffer owned
 *                     by the context, valid until the next vdjx_sam_merge call.                                                   */
int vdjx_sam_blocks(vdjx_ctx* ctx, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off, const uint32_t* d_reg_rank,
                    uint64_t* n_blocks, uint64_t* n_bytes, const void** d_keys, const void** d_lens, const void** d_text);
int vdjx_sam_merge(vdjx_ctx* ctx, uint64_t n_blocks, uint64_t n_bytes, const void* d_keys, const void* d_lens, const void* d_text,
                   const char** out_text, uint64_t* out_bytes);
/* rows of `row` bytes on the device: row d_pos[i] of d_dst = row i of d_src.  (The records of a pool sharded by pair on their way to
 * the ranks that hold their slice of the scan order for the k-mer build, A2:1388-1390: every record arrives with its place.) */
int vdjx_rows_scatter(vdjx_ctx* ctx, void* d_dst, const void* d_src, const uint32_t* d_pos, size_t n, size_t row);

/* counters of the most recent scorer calls, by name: "window_hits" (read instances matched by the last
 * vdjx_window_score call, summed over windows), "window_hits_max", "window_pairs", "window_work_items",
 * "map_hits", "root_dp_items".  Unknown names return 0.  Used by bench.py to price the scorers' algorithmic bytes. */
uint64_t vdjx_stat(vdjx_ctx* ctx, const char* name);

/* ---- profiling hooks (HIP events on the context's stream) ---------------------------------------*/
int vdjx_profile_enable(vdjx_ctx* ctx, int on);
/* bracket only the launches of this scope name from now on (NULL: all again): two event records per scope and step cost a small pool's
 * step a tenth of its time; a caller that wants one kernel's duration from inside a region it times pays for that one */
int vdjx_profile_only(vdjx_ctx* ctx, const char* name);
int vdjx_profile_reset(vdjx_ctx* ctx);
/* number of distinct kernel names recorded since the last reset */
int vdjx_profile_count(vdjx_ctx* ctx);
/* idx-th entry: kernel name, summed milliseconds, launch count */
int vdjx_profile_get(vdjx_ctx* ctx, int idx, const char** name, double* total_ms, uint64_t* launches);

/* ---- multi-GPU k-mer build: hash-prefix sharding, partial aggregates merged by the owner ----
 * The reference has no counterpart (its only parallelism is pthreads over roots, A2:1287-1348); these
 * phases split vdjx_kmer_build so that the caller can move the bytes between ranks (one process per GPU;
 * vdjer_amd/shard.py does it with torch.distributed over RCCL, vdjer_amd/csrc/host/vdjx_mgpu.c with RCCL
 * directly).  Record numbering is rank-major with a common stride: rank r's records are
 * [r*rec_stride, r*rec_stride + R_r); instance ids are global, record << 6 | offset, so nranks * rec_stride
 * must stay below 2^32 records (record << 8 | offset and 2^30 records with reads of more than 64 bases).  Any 1 <= nranks <= 256; the owner of a k-mer is its hash bucket divided by
 * the buckets per owner (*dir_len of vdjx_shard_local: the quotient of the bucket count by nranks, rounded up).
 *   Every rank first aggregates ITS OWN gated instances per distinct k-mer (count, first instance, whether it
 * saw two different reads: add_to_table A2:322-367 restated per rank).  These partial aggregates (32 B per
 * distinct gated k-mer per rank, not per instance) are the one bulk exchange.  The owner merges them (counts
 * add, firsts take the minimum, flags OR) and decides almost every k-mer on the spot; only a k-mer whose
 * verdict needs per-read data -- no rank saw two different reads although several hold it, or its count is
 * below the level where the quality sums cannot fail -- costs a question to the ranks that hold it and a
 * fixed-size answer (the first record's bases, partial quality sums).  add_to_graph's bookkeeping (node
 * frequency, first sights of nodes and edges, A2:261-320) is computed by every rank over its own records for
 * ALL survivors and reduced: SUM for the counts, MIN for the first sights.
 * All pointers are device pointers owned by the caller.  Call order (brackets = the caller's collectives):
 *   begin -> (count, symmetric -> [all_reduce MAX, MIN] -> geometry2 ->) local -> local_fill -> [all_to_all: directories, counts, partial aggregates] -> merge
 *   -> queries -> [all_to_all: counts, questions] -> reply -> [all_to_all: answers] -> resolve
 *   -> survivors -> [all_gather] -> edges -> [all_reduce MIN, SUM] -> finish -> free                */
typedef struct vdjx_shard vdjx_shard;
int vdjx_shard_begin(vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, int rank, int nranks,
                     uint64_t rec_stride, vdjx_shard** out);
/* The same build over a SHARE of the pool (the way `vdjer --gpus N` deals the reads: by pair, both mates on one rank): record i of
 * `pool` is record d_scan_index[i] of the scan order of the whole pool (primary then secondary, A2:1388-1390; total_records records
 * over all ranks).  The positions ascend -- a share keeps the pool's order -- and every position belongs to exactly one rank.  The
 * kernels work on local record numbers; a first instance is translated through d_scan_index where it leaves the rank, so no record
 * ever moves between ranks.  d_scan_index is a device pointer that must stay valid until vdjx_shard_free. */
int vdjx_shard_begin_share(vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, int rank, int nranks,
                           const uint32_t* d_scan_index, uint64_t total_records, vdjx_shard** out);
void vdjx_shard_free(vdjx_shard* s);
/* bytes per exchanged record: kind 0 partial aggregate (32), 1 question (8), 2 answer (408: 240 for the k-mer -- the holder's first record,
 * its quality rows -- and, for builds over couples, 168 more for its reverse complement's; always ask, never assume), 3 survivor (32) */
size_t vdjx_shard_record_bytes(int kind);
/* optional, before vdjx_shard_local: this rank's gated k-mer instances (A2:240-259); the ranks compare them [all_reduce MAX] and pass
 * the largest to vdjx_shard_geometry, so that every rank cuts the hash buckets the one-GPU build would cut for the largest rank.
 * Without it the bucket count follows a bound from rec_stride (more, smaller buckets: a partition level more at 10 M pairs per rank). */
int vdjx_shard_count(vdjx_shard* s, uint64_t* gated_instances);
int vdjx_shard_geometry(vdjx_shard* s, uint64_t agreed_instances);
/* A pool as add_to_buffer writes it (bam_read.c:206-244) is made of couples: every read followed by its reverse complement with reversed
 * qualities.  The gated k-mer instances of the second record mirror those of the first, so the local phase can move ONE tuple per pair
 * of mirrored instances -- under the smaller of the k-mer and its reverse complement (k odd) -- and hand both aggregates to the same
 * owner.  That changes which bucket a k-mer falls into, so ALL ranks must do it or none: vdjx_shard_symmetric says whether this rank
 * could (its pool passed the packing's check, k is odd, reads of up to 64 bases); the caller ANDs the ranks' answers [all_reduce MIN]
 * and passes the result to vdjx_shard_geometry2 beside the agreed instance count.  (vdjx_shard_geometry = all_symmetric 0.) */
int vdjx_shard_symmetric(const vdjx_shard* s);
int vdjx_shard_geometry2(vdjx_shard* s, uint64_t agreed_instances, int all_symmetric);
/* this rank's partial aggregates, grouped by owner: send_counts[nranks]; *dir_len = hash buckets per owner */
int vdjx_shard_local(vdjx_shard* s, uint64_t* send_counts, uint32_t* dir_len);
/* d_dir: u32 [nranks*dir_len] partial aggregates per bucket (owner-major); d_partials: sum(send_counts) records, 16-byte aligned.
 * The aggregates are laid end to end IN d_partials, which must stay valid and unchanged until vdjx_shard_reply has returned (the
 * answers to the owners' questions are looked up in it: no second copy of a gigabyte per rank). */
int vdjx_shard_local_fill(vdjx_shard* s, void* d_dir, void* d_partials);
/* owner: directories and partial aggregates as received (source-major) -> decided k-mers and questions;
 * query_counts[r] = questions for rank r */
int vdjx_shard_merge(vdjx_shard* s, const void* d_recv_dir, const void* d_recv_partials, const uint64_t* recv_counts,
                     uint64_t* query_counts);
/* the questions, grouped by destination rank: sum(query_counts) records */
int vdjx_shard_queries(vdjx_shard* s, void* d_out);
/* every rank: the questions it received (counts[o] from owner o, in that order) -> answers in the same order */
int vdjx_shard_reply(vdjx_shard* s, const void* d_queries, const uint64_t* counts, void* d_replies);
/* owner: the answers (grouped by answering rank, each group in question order) -> this rank's survivors
 * (a-1/a-2 for the k-mers it owns); n_distinct = distinct gated k-mers it owns ("Pre Num nodes") */
int vdjx_shard_resolve(vdjx_shard* s, const void* d_replies, uint64_t n_replies, uint64_t* n_survivors, uint64_t* n_distinct);
/* n_survivors records of 32 B: {u64 key_lo, u64 key_hi, u32 gated count, u32 -, u64 first gated instance} */
int vdjx_shard_survivors(vdjx_shard* s, void* d_out);
/* all ranks' survivors (rank order) -> this rank's share of add_to_graph (A2:261-320) over ITS records:
 * d_in_first u64 [ns_total*4]: first sight (global instance id, all-ones = none) of the edge into survivor v whose tail
 * k-mer starts with base a, at [v*4+a]; d_ufirst u64 [ns_total]: first sight of the node; d_ucnt u32 [ns_total]:
 * its instances on this rank.  The caller reduces over ranks: MIN (unsigned order) for d_in_first and d_ufirst,
 * SUM for d_ucnt */
int vdjx_shard_edges(vdjx_shard* s, const void* d_surv_all, uint64_t ns_total, void* d_in_first, void* d_ucnt, void* d_ufirst);
/* reduced arrays -> the graph, identical on every rank */
int vdjx_shard_finish(vdjx_shard* s, const void* d_in_first, const void* d_ucnt, const void* d_ufirst,
                      uint64_t pre_nodes_total, vdjx_graph** out);

#ifdef __cplusplus
}
#endif
#endif
