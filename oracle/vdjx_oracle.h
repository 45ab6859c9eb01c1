/*
 * oracle/vdjx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's hot path (SURVEY.md §8a rows a-1 ... a-10), written
 * from the reference's behaviour, each function citing the reference file:line it follows
 * (paths under /root/reference/src/main/c; A2 = assembler2_vdj.c).
 *
 * Pinned against the compiled reference itself (oracle/_ref/vdjer_ref, built from the reference's
 * own sources by oracle/Makefile): tests/golden/ holds dumps of the reference's functions on seeded
 * inputs, and tests/test_oracle_vs_golden.py checks this restatement against them bit for bit.
 * The reference has no tests or golden vectors of its own for this path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libvdjx.so) never links, loads or calls it.
 */
#ifndef VDJX_ORACLE_H
#define VDJX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hash_utils.c:5-46 */
uint64_t vdjo_murmur64a(const void* key, int len, uint64_t seed);
/* seq_to_kmer.c:6-46; returns 0 and sets *ok=0 on a non-ACGT base (the reference exits) */
uint32_t vdjo_seq_to_int(const char* seq, int* ok);

/* ---- a-1/a-2: k-mer table + prune (A2:240-259, 322-409, 454-484) -------------------------- */
typedef struct vdjo_table vdjo_table;
/* pools are arrays of (2*rl+1)-byte records: '0' + rl bases + rl Phred+33 (bam_read.c:206-244) */
vdjo_table* vdjo_table_build(const uint8_t* primary, size_t n_primary,
                             const uint8_t* secondary, size_t n_secondary, int rl, int k);
size_t vdjo_table_size(const vdjo_table* t);
/* erases per A2:467-484; returns the number of survivors */
size_t vdjo_table_prune(vdjo_table* t, int mf, int mq);
/* entries sorted by first instance.  first_inst = record_index*64 + offset, records counted over
 * primary then secondary (scan order of A2:1388-1390).  qs may be NULL, else [n][50]. */
void vdjo_table_export(const vdjo_table* t, uint64_t* first_inst, uint32_t* count, uint8_t* multi, uint8_t* qs);
void vdjo_table_free(vdjo_table* t);

/* ---- a-3 (+a-5/a-6): graph build (A2:190-237, 261-320, 412-452) ---------------------------- */
typedef struct vdjo_graph vdjo_graph;
/* v_codes/j_codes: anchor codes that passed the distance filter of vj_filter.c:56-68 */
vdjo_graph* vdjo_graph_build(const vdjo_table* pruned,
                             const uint8_t* primary, size_t n_primary,
                             const uint8_t* secondary, size_t n_secondary, int rl, int k,
                             const uint32_t* v_codes, size_t nv, const uint32_t* j_codes, size_t nj);
size_t vdjo_graph_nodes(const vdjo_graph* g);
/* per node in creation order (id = index+1): first ungated instance, frequency (cap 32765), flags;
 * to_deg/from_deg in [0,4]; to_ids/from_ids [n][4] hold 1-based node ids in *list order*
 * (head of the reference's prepend-list first). */
void vdjo_graph_export(const vdjo_graph* g, uint64_t* first_inst, uint32_t* freq, uint8_t* has_v, uint8_t* has_j,
                       uint8_t* to_deg, uint32_t* to_ids, uint8_t* from_deg, uint32_t* from_ids);
void vdjo_graph_free(vdjo_graph* g);

/* ---- a-7: root scorer (seq_score.c:36-156) ------------------------------------------------- */
typedef struct vdjo_scorer vdjo_scorer;
/* lines: the non-header lines of v_region.fa, newline stripped */
vdjo_scorer* vdjo_scorer_new(const char* const* lines, size_t n_lines, int vk);
int vdjo_score_seq(const vdjo_scorer* s, const char* kmer, int k, int threshold);
void vdjo_scorer_free(vdjo_scorer* s);

/* ---- a-8/a-9/a-10: read->contig mapper, coverage validator, SAM text ------------------------ */
typedef struct vdjo_readidx vdjo_readidx;
typedef struct {
	uint32_t pair_id;
	uint32_t rec1, rec2;      /* matching record index (scan order) of read 1 / read 2 */
	int16_t pos1, pos2, insert;
	uint8_t rc1, rc2;
} vdjo_pair;
/* per-record info in scan order: pair id, read_num (1|2), is_rc, registration rank (add_read_info
 * call order, quick_map3.c:126-149) */
vdjo_readidx* vdjo_readidx_build(const uint8_t* primary, size_t n_primary,
                                 const uint8_t* secondary, size_t n_secondary, int rl,
                                 const uint32_t* pair_id, const uint8_t* read_num, const uint8_t* is_rc,
                                 const uint32_t* reg_rank, uint32_t n_pairs);
/* quick_map3.c:188-266.  pairs_out/starts_out sized by caller (cap entries / 2*cap int pairs);
 * returns the number of mapped pairs (may exceed cap: then outputs are truncated). */
size_t vdjo_quick_map(vdjo_readidx* ix, const char* contig, int len, vdjo_pair* pairs_out, int32_t* starts_out, size_t cap);
/* coverage.c:64-130 (+10-61).  starts = sorted (first,second) int pairs, n of them */
int vdjo_coverage_is_valid(int rl, int contig_len, int eval_start, int eval_stop, int read_span,
                           int insert_low, int insert_high, int floor_, const int32_t* starts, size_t n, int mate_span);
/* quick_map3.c:152-181: two SAM lines for one mapped pair; returns bytes written (buf >= 1024) */
int vdjo_sam_pair(const vdjo_readidx* ix, const char* contig_id, const char* read_name, const vdjo_pair* p, char* buf);
void vdjo_readidx_free(vdjo_readidx* ix);

/* f-3: process_kmers (seq_dist.c:49-71): rows (code, min base distance to any anchor) for codes in [start,end]
 * with distance <= max_dist, ascending; returns the row count, fills at most cap rows */
size_t vdjo_index_rows(const uint32_t* anchors, size_t n_anchors, uint64_t start, uint64_t end, int max_dist,
                       uint32_t* codes, uint8_t* dists, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
