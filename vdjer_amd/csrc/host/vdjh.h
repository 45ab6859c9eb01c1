/*
 * vdjh.h -- host-side (serial, plain C) part of the vdjer pipeline around the GPU hot path
 * (SURVEY §8f-1): graph reconstruction from vdjx_kmer_build's export, root identification, chain
 * condensation, DFS contig enumeration, V/J window discovery, window acceptance, overlap removal,
 * vdj_contigs.fa / vdjer.dot / SAM text.  Restated from the reference (A2 = assembler2_vdj.c):
 *   identify_root_nodes A2:653-676, condense_graph A2:598-650, build_contigs A2:939-1061,
 *   append_to_contig A2:916-937, output_contig A2:775-870, output_windows A2:872-914,
 *   dump_graph A2:1133-1198, print_windows vj_filter.c:209-309, find_conserved_aminos :127-194,
 *   output_header/output_mapping quick_map3.c:152-181,274-309.
 * The batched scorers are reached through callbacks: the CLI binds them to libvdjx (GPU); the CPU tests
 * bind them to the oracle.  Nothing here touches the GPU.
 */
#ifndef VDJH_H
#define VDJH_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int k;                 /* --k   */
	int min_node_freq;     /* --mf  */
	int min_base_quality;  /* --mq  */
	float min_contig_score;/* --mcs */
	int vj_min_win, vj_max_win;   /* --miw/--maw (chain presets params.c:13-30) */
	int j_conserved;       /* 'W' | 'F' */
	int window_span;       /* --ws 486 */
	int j_extension;       /* -jext 162 */
	int read_filter_floor; /* --rf */
	int min_source_homology_score; /* --mrs */
	int filter_read_span, filter_mate_span;   /* --rs --ms */
	int eval_start, eval_stop;                /* --e0 --e1 */
	int window_overlap_check_size;            /* --wo */
	int insert_len;        /* --ins */
	int vregion_kmer_size; /* --vk */
	int read_length;
	int threads;                  /* --t (A2:1287-1348: worker threads over the roots); results do not depend on it */
} vdjh_params;

void vdjh_default_params(vdjh_params* p);          /* params.c:53-73 */
int vdjh_set_chain(vdjh_params* p, const char* chain);   /* params.c:8-35; 0 ok */

/* the graph as vdjx_graph_export delivers it (node i has id i+1; lists hold 1-based ids in list order) */
typedef struct {
	size_t n;
	int k;
	const char* kmers;            /* n*k */
	const uint32_t* freq;
	const uint8_t *has_v, *has_j;
	const uint8_t *to_deg, *from_deg;
	const uint32_t *to_ids, *from_ids;   /* [n][4] */
} vdjh_graph;

typedef struct {
	void* ud;
	/* score_seq for n k-mers (A2:1103) */
	int (*root_score)(void* ud, const char* kmers, size_t n, int k, int threshold, uint8_t* out);
	/* quick_map_process_contig + coverage_is_valid for n windows of `len` chars (A2:841-847) */
	int (*window_score)(void* ud, const char* windows, size_t n, int len, uint8_t* valid);
	/* mapped pairs of n contigs in output order -> SAM body text appended to `out` (quick_map3.c:311-340) */
	int (*sam_body)(void* ud, const char* const* ids, const char* contigs, size_t n, int len, FILE* out);
	/* sorted anchor codes (vj_filter.c:56-78) */
	const uint32_t* v_codes; size_t nv;
	const uint32_t* j_codes; size_t nj;
	/* print_status (status.c:22-32) at the reference's stage boundaries inside assemble(): POST_GRAPH_BLOCK,
	 * POST_ROOT_TRACEBACK, POST_CONDENSE_GRAPH, STATUS_UPDATE, THREADS_DONE, PRE_CLEANUP, POST_CLEANUP
	 * (A2:1417-1464, 1336).  May be NULL. */
	void (*status)(void* ud, const char* desc);
} vdjh_hooks;

typedef struct {
	size_t n_roots, n_roots_accepted, n_contig_candidates, n_windows_scored, n_windows_valid, n_contigs_out;
} vdjh_stats;

/* Runs everything after the graph build.  fasta/dot may be NULL (skipped); sam may be NULL. */
int vdjh_assemble(const vdjh_params* p, const vdjh_graph* g, const vdjh_hooks* h,
                  const char* fasta_path, const char* dot_path, FILE* sam, vdjh_stats* st);

/* pieces exposed for tests */
/* iteration order of the `nodes` table (ids, n of them) */
void vdjh_node_order(const vdjh_graph* g, uint32_t* ids_out);
/* vjf_search on one contig: calls cb(window, cdr3) in table order */
int vdjh_vjf_search(const vdjh_params* p, const vdjh_hooks* h, const char* contig,
                    void (*cb)(void* ud, const char* window, const char* cdr3), void* ud);
const char* vdjh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
