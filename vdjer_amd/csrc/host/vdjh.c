/* vdjh.c -- see vdjh.h.  Serial host part of the pipeline, restated from the reference. */
#define _GNU_SOURCE
#include "vdjh.h"
#include <pthread.h>
#include "sph.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#define MIN_CONTIG_SIZE 550      /* A2:51 */
#define MAX_CONTIG_SIZE 650      /* A2:52 */
#define SEQ_LEN 16               /* seq_dist.h:4 */
#define ANCHOR_PADDING 16        /* vj_filter.c:31 */
#define MAX_PATHS_FROM_ROOT 500000000   /* A2:1107 */
#define MAX_CONTIGS 50000000            /* A2:1108 */

static __thread char g_err[512];
static void set_err(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}
const char* vdjh_last_error(void) { return g_err; }

/* params.c:53-73 */
void vdjh_default_params(vdjh_params* p) {
	memset(p, 0, sizeof *p);
	p->k = 35; p->min_node_freq = 3; p->min_base_quality = 90; p->min_contig_score = -5;
	p->window_span = 486; p->j_extension = 162; p->read_filter_floor = 1; p->vregion_kmer_size = 15;
	p->min_source_homology_score = 30; p->filter_read_span = 35; p->filter_mate_span = 48;
	p->eval_start = 52; p->eval_stop = 411; p->window_overlap_check_size = 320;
	p->threads = 1;               /* params.c:69 */
}

/* params.c:8-35 (the loci are only used by the BAM extraction, which is not part of this library) */
int vdjh_set_chain(vdjh_params* p, const char* chain) {
	if (!strcmp(chain, "IGH")) { p->j_conserved = 'W'; p->vj_min_win = 10; p->vj_max_win = 90; return 0; }
	if (!strcmp(chain, "IGL") || !strcmp(chain, "IGK")) { p->j_conserved = 'F'; p->vj_min_win = 0; p->vj_max_win = 60; return 0; }
	set_err("Invalid chain specified: %s.  Chain must be one of [IGH,IGL,IGK]", chain);
	return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* graph                                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct hnode {
	const char* kmer;
	char* seq;
	char kmer_seq[2];
	int to_n, from_n;
	struct hnode* to[4];
	struct hnode* from[4];
	int id;
	unsigned short frequency;
	char is_condensed, is_filtered, has_vmer, has_jmer;
} hnode;

typedef struct {
	hnode* nodes;
	size_t n;
	int k;
	sph_table table;          /* `nodes`: dense_hash_map<const char*, node*, my_hash, eqstr> (A2:1368-1369) */
} hgraph;

static int graph_build(hgraph* G, const vdjh_graph* g) {
	G->n = g->n;
	G->k = g->k;
	G->nodes = (hnode*) calloc(g->n + 1, sizeof(hnode));
	sph_init(&G->table, g->k, 0);
	for (size_t i = 0; i < g->n; i++) {
		hnode* nd = &G->nodes[i];
		nd->kmer = g->kmers + i * (size_t) g->k;
		nd->kmer_seq[0] = nd->kmer[0];              /* new_node, A2:198-201 */
		nd->id = (int) i + 1;
		nd->frequency = (unsigned short) g->freq[i];
		nd->has_vmer = (char) g->has_v[i];
		nd->has_jmer = (char) g->has_j[i];
		nd->to_n = g->to_deg[i];
		nd->from_n = g->from_deg[i];
		for (int e = 0; e < nd->to_n; e++) {
			uint32_t id = g->to_ids[i * 4 + e];
			if (id == 0 || id > g->n) { set_err("bad edge id"); return -1; }
			nd->to[e] = &G->nodes[id - 1];
		}
		for (int e = 0; e < nd->from_n; e++) {
			uint32_t id = g->from_ids[i * 4 + e];
			if (id == 0 || id > g->n) { set_err("bad edge id"); return -1; }
			nd->from[e] = &G->nodes[id - 1];
		}
		/* (*nodes)[kmer] = curr in creation order (A2:305) fixes the table's iteration order */
		sph_map_put(&G->table, nd->kmer, nd, NULL);
	}
	return 0;
}

static void graph_free(hgraph* G) {
	for (size_t i = 0; i < G->n; i++) free(G->nodes[i].seq);
	free(G->nodes);
	sph_free(&G->table);
}

void vdjh_node_order(const vdjh_graph* g, uint32_t* ids_out) {
	hgraph G;
	if (graph_build(&G, g)) return;
	size_t n = 0;
	for (size_t b = sph_next(&G.table, 0); b < G.table.nbuckets; b = sph_next(&G.table, b + 1))
		ids_out[n++] = (uint32_t) ((hnode*) G.table.b[b].val)->id;
	graph_free(&G);
}

/* A2:564-582 */
static int has_one_incoming(const hnode* n) { return n->from_n == 1; }
static int has_one_outgoing(const hnode* n) { return n->to_n == 1; }
static int prev_has_multiple_outgoing(const hnode* n) {
	if (has_one_incoming(n)) {
		const hnode* prev = n->from[0];
		if (prev->to_n >= 2) return 1;
	}
	return 0;
}

/* A2:598-650 */
static void condense_graph(hgraph* G) {
	for (size_t b = sph_next(&G->table, 0); b < G->table.nbuckets; b = sph_next(&G->table, b + 1)) {
		hnode* node = (hnode*) G->table.b[b].val;
		if ((!has_one_incoming(node) || prev_has_multiple_outgoing(node)) && has_one_outgoing(node)) {
			hnode* next = node->to[0];
			if (has_one_incoming(next)) {
				hnode* last = next;
				char* seq = (char*) calloc(MAX_CONTIG_SIZE + 2, 1);
				int idx = 0;
				seq[idx++] = node->kmer[0];
				int nodes_condensed = 1;
				char hv = node->has_vmer, hj = node->has_jmer;
				while (next != NULL && has_one_incoming(next) && nodes_condensed < MAX_CONTIG_SIZE) {
					last = next;
					seq[idx++] = next->kmer[0];
					hnode* temp = has_one_outgoing(next) ? next->to[0] : NULL;
					next->is_filtered = 1;
					hv = hv || next->has_vmer;
					hj = hj || next->has_jmer;
					next = temp;
					nodes_condensed += 1;
				}
				node->seq = seq;
				node->is_condensed = 1;
				/* node->toNodes = last->toNodes (A2:644).  `last` may be `node` itself on a cycle: copy first */
				int ln = last->to_n;
				hnode* lt[4];
				memcpy(lt, last->to, sizeof lt);
				node->to_n = ln;
				memcpy(node->to, lt, sizeof lt);
				node->has_vmer = hv;
				node->has_jmer = hj;
			}
		}
	}
}

/* A2:1133-1198 */
static int dump_graph(const hgraph* G, const char* path) {
	FILE* fp = fopen(path, "w");
	if (!fp) { set_err("cannot write %s", path); return -1; }
	fprintf(fp, "digraph vdjer {\n//\tEdges\n");
	for (size_t b = sph_next(&G->table, 0); b < G->table.nbuckets; b = sph_next(&G->table, b + 1)) {
		const hnode* n = (const hnode*) G->table.b[b].val;
		if (!n->is_filtered)
			for (int e = 0; e < n->to_n; e++) fprintf(fp, "\tv_%d -> v_%d\n", n->id, n->to[e]->id);
	}
	fprintf(fp, "//\tVertices\n");
	for (size_t b = sph_next(&G->table, 0); b < G->table.nbuckets; b = sph_next(&G->table, b + 1)) {
		const hnode* n = (const hnode*) G->table.b[b].val;
		if (n->is_filtered) continue;
		if (n->is_condensed) fprintf(fp, "\tv_%d [label=\"%s\",shape=box,color=blue]\n", n->id, n->seq);
		else fprintf(fp, "\tv_%d [label=\"%c\",shape=box]\n", n->id, n->kmer[0]);
	}
	fprintf(fp, "}\n");
	fclose(fp);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* V/J window discovery (vj_filter.c)                                                          */
/* ------------------------------------------------------------------------------------------ */
static int code_in(const uint32_t* codes, size_t n, uint32_t code) {
	if (code == 0) return 0;                 /* empty key of the reference's sets (vj_filter.c:317-318) */
	size_t lo = 0, hi = n;
	while (lo < hi) {
		size_t mid = (lo + hi) / 2;
		if (codes[mid] < code) lo = mid + 1; else hi = mid;
	}
	return lo < n && codes[lo] == code;
}

static int seq_to_int16(const char* s, uint32_t* out) {   /* seq_to_kmer.c:6-46 */
	uint32_t v = 0;
	for (int i = 0; i < SEQ_LEN; i++) {
		uint32_t b;
		switch (s[i]) {
		case 'A': b = 0; break;
		case 'T': b = 1; break;
		case 'C': b = 2; break;
		case 'G': b = 3; break;
		default: return -1;
		}
		v = (v << 2) + b;
	}
	*out = v;
	return 0;
}

static int is_stop_codon(const char* c) {
	return strncmp(c, "TAG", 3) == 0 || strncmp(c, "TAA", 3) == 0 || strncmp(c, "TGA", 3) == 0;
}

/* vj_filter.c:106-115 */
static int is_in_frame(const char* seq) {
	size_t n = strlen(seq);
	for (size_t i = 0; i + 2 < n; i += 3)
		if (is_stop_codon(seq + i)) return 0;
	return 1;
}

typedef struct { char** v; size_t n, cap; } strvec;
static void sv_push(strvec* s, char* x) {
	if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 16; s->v = (char**) realloc(s->v, s->cap * sizeof(char*)); }
	s->v[s->n++] = x;
}

/* vj_filter.c:127-194 */
static void find_conserved_aminos(const vdjh_params* p, int v_index, int j_index, const char* contig, int clen,
                                  sph_table* cdr3_seq, strvec* owned) {
	int vi[64], ji[64], nv = 0, nj = 0;
	for (int i = v_index; i < SEQ_LEN + ANCHOR_PADDING + v_index - 2; i++) {
		if (i < 0 || i >= clen) continue;
		if (strncmp(contig + i, "TGT", 3) == 0 || strncmp(contig + i, "TGC", 3) == 0) vi[nv++] = i;
	}
	for (int i = j_index - ANCHOR_PADDING; i < SEQ_LEN + j_index - 2; i++) {
		if (i < 0 || i >= clen) continue;          /* the reference reads outside the buffer here */
		if (p->j_conserved == 'W' && strncmp(contig + i, "TGG", 3) == 0) ji[nj++] = i;
		else if (p->j_conserved == 'F' && (strncmp(contig + i, "TTC", 3) == 0 || strncmp(contig + i, "TTT", 3) == 0)) ji[nj++] = i;
	}
	for (int a = 0; a < nv; a++) {
		for (int b = 0; b < nj; b++) {
			int window = ji[b] - vi[a] + 3;
			if (window % 3 == 0 && window >= p->vj_min_win && window <= p->vj_max_win && ji[b] >= vi[a]) {
				char* cdr3 = (char*) calloc((size_t) window + 1, 1);
				strncpy(cdr3, contig + vi[a], (size_t) window);
				int ins;
				sph_set_insert(cdr3_seq, cdr3, &ins);
				if (ins) sv_push(owned, cdr3); else free(cdr3);
			}
		}
	}
}

/* vj_filter.c:196-204: str is a proper substring of some member */
static int is_sub_string(const char* str, const sph_table* set) {
	for (size_t b = sph_next(set, 0); b < set->nbuckets; b = sph_next(set, b + 1))
		if (strstr(set->b[b].key, str) != NULL && strcmp(set->b[b].key, str) != 0) return 1;
	return 0;
}

/* vj_filter.c:209-309 with allow_cdr3_substrings = 1 (A2:802) */
int vdjh_vjf_search(const vdjh_params* p, const vdjh_hooks* h, const char* contig,
                    void (*cb)(void* ud, const char* window, const char* cdr3), void* ud) {
	const int clen = (int) strlen(contig);
	const int len = clen - SEQ_LEN;
	int* v_idx = (int*) malloc(sizeof(int) * (size_t) (clen + 1));
	int* j_idx = (int*) malloc(sizeof(int) * (size_t) (clen + 1));
	int nv = 0, nj = 0;
	for (int i = 0; i < len; i++) {
		uint32_t code;
		if (seq_to_int16(contig + i, &code)) { free(v_idx); free(j_idx); set_err("Error converting base in contig"); return -1; }
		if (code_in(h->v_codes, h->nv, code)) v_idx[nv++] = i;
		if (code_in(h->j_codes, h->nj, code)) j_idx[nj++] = i;
	}
	sph_table cdr3_seq, cdr3_temp, windows_temp, windows;
	sph_init(&cdr3_seq, 0, 1);          /* sparse_hash_set<const char*, vjf_hash, vjf_eqstr> */
	sph_init(&cdr3_temp, 0, 1);
	sph_init(&windows_temp, 0, 0);      /* dense_hash_map<..., vjf_hash, vjf_eqstr> */
	sph_init(&windows, 0, 0);
	strvec owned = {0};
	const int pad = SEQ_LEN * 2 + ANCHOR_PADDING * 2;
	for (int a = 0; a < nv; a++) {
		for (int b = 0; b < nj; b++) {
			int window = j_idx[b] - v_idx[a] + SEQ_LEN;
			if (window >= p->vj_min_win && window <= p->vj_max_win + pad && clen > window)
				find_conserved_aminos(p, v_idx[a], j_idx[b], contig, clen, &cdr3_seq, &owned);
		}
	}
	for (size_t b = sph_next(&cdr3_seq, 0); b < cdr3_seq.nbuckets; b = sph_next(&cdr3_seq, b + 1)) {
		const char* cdr3 = cdr3_seq.b[b].key;
		int window = (int) strlen(cdr3);
		const char* start = strstr(contig, cdr3);
		if (start == NULL) continue;
		int vpad = p->window_span - (window + p->j_extension);
		if (start - contig < vpad) continue;     /* the reference would read in front of its buffer here */
		start -= vpad;
		if ((int) strlen(start) > p->window_span) {
			char* win = (char*) calloc((size_t) p->window_span + 1, 1);
			strncpy(win, start, (size_t) p->window_span);
			if (is_in_frame(win) && sph_find(&windows_temp, win) == (size_t) -1) {
				char* fc = strdup(cdr3);
				sph_map_put(&windows_temp, win, fc, NULL);
				sph_set_insert(&cdr3_temp, fc, NULL);
				sv_push(&owned, win);
				sv_push(&owned, fc);
			} else {
				free(win);
			}
		}
	}
	for (size_t b = sph_next(&windows_temp, 0); b < windows_temp.nbuckets; b = sph_next(&windows_temp, b + 1)) {
		const char* cdr3 = (const char*) windows_temp.b[b].val;
		if (!is_sub_string(cdr3, &cdr3_temp)) sph_map_put(&windows, windows_temp.b[b].key, (void*) cdr3, NULL);
	}
	for (size_t b = sph_next(&windows, 0); b < windows.nbuckets; b = sph_next(&windows, b + 1))
		cb(ud, windows.b[b].key, (const char*) windows.b[b].val);
	for (size_t i = 0; i < owned.n; i++) free(owned.v[i]);
	free(owned.v);
	sph_free(&cdr3_seq); sph_free(&cdr3_temp); sph_free(&windows_temp); sph_free(&windows);
	free(v_idx); free(j_idx);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* contig enumeration                                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
	const char** frags;
	int nfr, cap;
	hnode* curr;
	double score;
	int real_size;
	char has_vmer, has_jmer;
} hcontig;

static hcontig* contig_new(void) { return (hcontig*) calloc(1, sizeof(hcontig)); }
static void contig_push(hcontig* c, const char* f) {
	if (c->nfr == c->cap) { c->cap = c->cap ? c->cap * 2 : 32; c->frags = (const char**) realloc(c->frags, sizeof(char*) * (size_t) c->cap); }
	c->frags[c->nfr++] = f;
}
static hcontig* contig_copy(const hcontig* o) {   /* A2:704-721 */
	hcontig* c = (hcontig*) calloc(1, sizeof(hcontig));
	c->cap = o->nfr + 8;
	c->frags = (const char**) malloc(sizeof(char*) * (size_t) c->cap);
	memcpy(c->frags, o->frags, sizeof(char*) * (size_t) o->nfr);
	c->nfr = o->nfr;
	c->real_size = o->real_size;
	c->score = o->score;
	c->has_vmer = o->has_vmer;
	c->has_jmer = o->has_jmer;
	return c;
}
static void contig_free(hcontig* c) { free(c->frags); free(c); }

/* distinct candidate windows in first-encounter order */
typedef struct {
	sph_table seen;             /* membership only (vjf_window_candidates, A2:93) */
	strvec win, cdr3;
	size_t n_candidates;
} wincoll;

typedef struct { wincoll* w; } vjf_ud;
static void on_window(void* ud, const char* window, const char* cdr3) {
	wincoll* w = ((vjf_ud*) ud)->w;
	if (sph_find(&w->seen, window) != (size_t) -1) return;
	char* wc = strdup(window);
	sph_set_insert(&w->seen, wc, NULL);
	sv_push(&w->win, wc);
	sv_push(&w->cdr3, strdup(cdr3));
}

/* A2:775-870 up to the point where windows are collected */
static int output_contig(const vdjh_params* p, const vdjh_hooks* h, const hcontig* c, wincoll* w) {
	if (!(c->real_size >= MIN_CONTIG_SIZE && c->has_vmer && c->has_jmer)) return 0;
	char buf[MAX_CONTIG_SIZE * 2 + 1];
	buf[0] = '\0';
	w->n_candidates++;
	for (int i = 0; i < c->nfr; i++) {
		int to_cat = MAX_CONTIG_SIZE - (int) strlen(buf);
		if (to_cat <= 0) break;
		strncat(buf, c->frags[i], (size_t) to_cat);
	}
	vjf_ud ud = {w};
	return vdjh_vjf_search(p, h, buf, on_window, &ud);
}

/* A2:916-937 */
static void append_to_contig(hcontig* c, int entire_kmer, int k) {
	c->has_vmer = c->has_vmer || c->curr->has_vmer;
	c->has_jmer = c->has_jmer || c->curr->has_jmer;
	if (c->curr->is_condensed) {
		contig_push(c, c->curr->seq);
		c->real_size += (int) strlen(c->curr->seq);
	} else if (!entire_kmer) {
		contig_push(c, c->curr->kmer_seq);
		c->real_size += 1;
	} else {
		c->real_size += k;           /* the terminal k-mer is counted but its text is never appended (A2:931-934) */
	}
}

/* A2:939-1061 */
static int build_contigs(const vdjh_params* p, const vdjh_hooks* h, hnode* root, wincoll* w) {
	hcontig** stack = NULL;
	size_t sn = 0, scap = 0;
#define PUSH(x) do { if (sn == scap) { scap = scap ? scap * 2 : 64; stack = (hcontig**) realloc(stack, scap * sizeof(hcontig*)); } stack[sn++] = (x); } while (0)
	hcontig* rc = contig_new();
	rc->curr = root;
	PUSH(rc);
	long paths = 1;
	long contig_count = 0;
	int status = 0;
	const int k = p->k;
	while (sn > 0 && status == 0) {
		hcontig* c = stack[sn - 1];
		if (c->curr->to_n == 0 || c->score < p->min_contig_score || c->real_size >= (MAX_CONTIG_SIZE - k - 1)) {
			append_to_contig(c, 1, k);
			if (c->real_size >= MIN_CONTIG_SIZE && c->has_vmer && c->has_jmer) contig_count++;
			if (output_contig(p, h, c, w)) status = -9;
			contig_free(c);
			sn--;
		} else {
			append_to_contig(c, 0, k);
			int total = 0;
			for (int e = 0; e < c->curr->to_n; e++) total += c->curr->to[e]->frequency;
			double l = log10(total);
			hnode* cur = c->curr;
			c->curr = cur->to[0];
			paths++;
			for (int e = 1; e < cur->to_n; e++) {
				hcontig* br = contig_copy(c);
				br->curr = cur->to[e];
				br->score = br->score + log10(br->curr->frequency) - l;
				PUSH(br);
				paths++;
			}
			c->score = c->score + log10(c->curr->frequency) - l;
		}
		if (contig_count >= MAX_CONTIGS) status = -2;
		if (paths >= MAX_PATHS_FROM_ROOT) status = -1;
	}
	while (sn > 0) contig_free(stack[--sn]);
	free(stack);
#undef PUSH
	if (status == -9) return -1;
	if (status) { set_err("Status: %d (too many paths/contigs from one root, A2:1115-1119)", status); return status; }
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* roots in parallel (process_roots, A2:1287-1348), merged in dispatch order                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
	const vdjh_params* p;
	const vdjh_hooks* h;
	hnode** roots;
	const uint8_t* accepted;
	size_t nroots;
	wincoll* per_root;            /* [nroots]: windows of root i in encounter order, distinct within the root */
	volatile long next;           /* next root to take */
	volatile int failed;
	int progress;                 /* VDJH_PROGRESS=1: the reference's per-100-roots progress lines, every 1000 roots */
} trav_job;

static void wincoll_free(wincoll* w) {
	for (size_t i = 0; i < w->win.n; i++) { free(w->win.v[i]); free(w->cdr3.v[i]); }
	free(w->win.v); free(w->cdr3.v);
	sph_free(&w->seen);
	memset(w, 0, sizeof *w);
}

static void* trav_worker(void* arg) {
	trav_job* j = (trav_job*) arg;
	for (;;) {
		const long i = __sync_fetch_and_add(&j->next, 1);
		if (i >= (long) j->nroots || j->failed) break;
		if (!j->accepted[i]) continue;
		wincoll* w = &j->per_root[i];
		sph_init(&w->seen, 0, 0);
		if (build_contigs(j->p, j->h, j->roots[i], w)) { j->failed = 1; break; }
		if (j->progress && (i % 1000) == 0) fprintf(stderr, "Processed %ld root nodes\n", i);       /* A2:1327-1331 */
		sph_free(&w->seen);                        /* membership inside the root is no longer needed */
		memset(&w->seen, 0, sizeof w->seen);
	}
	return NULL;
}

static int traverse_roots(const vdjh_params* p, const vdjh_hooks* h, hnode** roots, const uint8_t* accepted, size_t nroots, wincoll* out) {
	int nthreads = p->threads > 0 ? p->threads : 1;
	if (nthreads > 256) nthreads = 256;
	if (nthreads == 1) {
		for (size_t i = 0; i < nroots; i++) {
			if (!accepted[i]) continue;
			if (build_contigs(p, h, roots[i], out)) return -1;
		}
		return 0;
	}
	trav_job job;
	memset(&job, 0, sizeof job);
	job.p = p; job.h = h; job.roots = roots; job.accepted = accepted; job.nroots = nroots;
	job.progress = getenv("VDJH_PROGRESS") != NULL;
	job.per_root = (wincoll*) calloc(nroots + 1, sizeof(wincoll));
	pthread_t* th = (pthread_t*) calloc((size_t) nthreads, sizeof(pthread_t));
	int started = 0;
	for (int t = 0; t < nthreads; t++) {
		if (pthread_create(&th[t], NULL, trav_worker, &job)) break;
		started++;
	}
	if (started == 0) trav_worker(&job);
	for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
	free(th);
	int rc = job.failed ? -1 : 0;
	if (rc) set_err("contig enumeration failed in a worker thread (too many paths/contigs from one root, A2:1115-1119)");
	for (size_t i = 0; i < nroots && rc == 0; i++) {
		wincoll* w = &job.per_root[i];
		out->n_candidates += w->n_candidates;
		for (size_t q = 0; q < w->win.n; q++) {
			if (sph_find(&out->seen, w->win.v[q]) != (size_t) -1) continue;
			sph_set_insert(&out->seen, w->win.v[q], NULL);
			sv_push(&out->win, w->win.v[q]);            /* ownership moves to the merged list */
			sv_push(&out->cdr3, w->cdr3.v[q]);
			w->win.v[q] = NULL;
			w->cdr3.v[q] = NULL;
		}
	}
	for (size_t i = 0; i < nroots; i++) wincoll_free(&job.per_root[i]);
	free(job.per_root);
	return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* the whole host stage                                                                        */
/* ------------------------------------------------------------------------------------------ */
int vdjh_assemble(const vdjh_params* p, const vdjh_graph* g, const vdjh_hooks* h,
                  const char* fasta_path, const char* dot_path, FILE* sam, vdjh_stats* st) {
	vdjh_stats s0;
	memset(&s0, 0, sizeof s0);
	if (!st) st = &s0;
	memset(st, 0, sizeof *st);
	if (g->k != p->k) { set_err("graph k != params k"); return -1; }
	const int CONTIG_SIZE = p->eval_stop - p->eval_start + 1;      /* A2:1520 */
	hgraph G;
	if (graph_build(&G, g)) return -1;
	int rc = -1;

	/* identify_root_nodes (A2:653-676): table order, prepended => reversed */
	size_t nroots = 0;
	hnode** roots = (hnode**) malloc(sizeof(hnode*) * (G.n + 1));
	for (size_t b = sph_next(&G.table, 0); b < G.table.nbuckets; b = sph_next(&G.table, b + 1)) {
		hnode* n = (hnode*) G.table.b[b].val;
		if (n->from_n == 0) roots[nroots++] = n;
	}
	for (size_t i = 0; i < nroots / 2; i++) { hnode* t = roots[i]; roots[i] = roots[nroots - 1 - i]; roots[nroots - 1 - i] = t; }
	st->n_roots = nroots;
#define STAGE(name) do { if (h->status) h->status(h->ud, name); } while (0)
	STAGE("POST_GRAPH_BLOCK");                 /* A2:1417 */
	STAGE("POST_ROOT_TRACEBACK");              /* A2:1431 (traceback_roots is commented out in the reference) */

	condense_graph(&G);
	STAGE("POST_CONDENSE_GRAPH");              /* A2:1437 */
	if (dot_path && dump_graph(&G, dot_path)) { graph_free(&G); free(roots); return -1; }

	/* score_seq for every root (A2:1103), batched: a pure function of the k-mer */
	uint8_t* accepted_root = (uint8_t*) calloc(nroots + 1, 1);
	char* rk = (char*) malloc(nroots * (size_t) p->k + 1);
	for (size_t i = 0; i < nroots; i++) memcpy(rk + i * (size_t) p->k, roots[i]->kmer, (size_t) p->k);
	wincoll w;
	memset(&w, 0, sizeof w);
	sph_init(&w.seen, 0, 0);
	sph_table acc;                       /* vjf_windows: dense_hash_map<..., contig_hash, contig_eqstr> (A2:91) */
	sph_init(&acc, CONTIG_SIZE, 0);
	uint8_t* valid = NULL;
	char* wbuf = NULL;
	if (nroots && h->root_score(h->ud, rk, nroots, p->k, p->min_source_homology_score, accepted_root)) { set_err("root scorer failed"); goto done; }
	if (getenv("VDJH_ROOT_LOG")) {       /* (diagnostic: "<root k-mer>\t<verdict>" in dispatch order; bench.py sets it beside the reference's own log) */
		FILE* rl = fopen(getenv("VDJH_ROOT_LOG"), "w");
		if (rl) {
			for (size_t i = 0; i < nroots; i++) { fwrite(rk + i * (size_t) p->k, 1, (size_t) p->k, rl); fprintf(rl, "\t%d\n", (int) accepted_root[i]); }
			fclose(rl);
		}
	}
	/* process_roots prints STATUS_UPDATE when the first root is dispatched (`ts` starts at 0, A2:1301,1335) and then every 300 s */
	if (nroots) STAGE("STATUS_UPDATE");

	/* worker_thread/build_contigs per accepted root (A2:1305-1318, 1093-1131).  The reference hands roots to --t threads and
	 * its output then depends on their timing; here every root collects its windows on its own (any thread) and the lists are
	 * merged in dispatch order, which is exactly what one thread produces: a window counts where it is met first. */
	{
		size_t nacc = 0;
		for (size_t i = 0; i < nroots; i++) if (accepted_root[i]) nacc++;
		st->n_roots_accepted = nacc;
		if (traverse_roots(p, h, roots, accepted_root, nroots, &w)) goto done;
	}
	st->n_contig_candidates = w.n_candidates;

	/* score every distinct candidate window in one batch, then replay the acceptance in encounter order.
	 * A window is processed iff it was not seen before and its [e0-1, e0-1+CONTIG_SIZE) slice is not accepted yet
	 * (A2:824-828); an accepted window is cut at eval stop and keyed by that slice (A2:853-858). */
	{
		const size_t nw = w.win.n;
		const int ws = p->window_span;
		valid = (uint8_t*) calloc(nw + 1, 1);
		wbuf = (char*) malloc(nw * (size_t) ws + 1);
		for (size_t i = 0; i < nw; i++) memcpy(wbuf + i * (size_t) ws, w.win.v[i], (size_t) ws);
		st->n_windows_scored = nw;
		if (nw && h->window_score(h->ud, wbuf, nw, ws, valid)) { set_err("window scorer failed"); goto done; }
		for (size_t i = 0; i < nw; i++) {
			char* win = w.win.v[i];
			if (sph_find(&acc, win + (p->eval_start - 1)) != (size_t) -1) continue;
			if (!valid[i]) continue;
			st->n_windows_valid++;
			win[p->eval_start + CONTIG_SIZE - 1] = '\0';
			sph_map_put(&acc, win + p->eval_start - 1, w.cdr3.v[i], NULL);
		}
	}

	STAGE("THREADS_DONE");                     /* A2:1452 */
	/* output_windows (A2:872-914): overlap removal with erase-during-iteration, then the FASTA in table order */
	for (size_t b1 = sph_next(&acc, 0); b1 < acc.nbuckets; b1 = sph_next(&acc, b1 + 1)) {
		const char* w1 = acc.b[b1].key;
		int remove = 0;
		for (size_t b2 = sph_next(&acc, 0); b2 < acc.nbuckets && !remove; b2 = sph_next(&acc, b2 + 1)) {
			const char* w2 = acc.b[b2].key;
			for (int i = 1; i < CONTIG_SIZE - p->window_overlap_check_size; i++) {
				if (strncmp(w1 + i, w2, (size_t) p->window_overlap_check_size) == 0) { remove = 1; break; }
			}
		}
		if (remove) sph_erase_at(&acc, b1);
	}
	{
		size_t nc = sph_size(&acc);
		st->n_contigs_out = nc;
		char** ids = (char**) calloc(nc + 1, sizeof(char*));
		char* contigs = (char*) malloc(nc * (size_t) CONTIG_SIZE + 1);
		FILE* fp = fasta_path ? fopen(fasta_path, "w") : NULL;
		if (fasta_path && !fp) { set_err("cannot write %s", fasta_path); free(ids); free(contigs); goto done; }
		size_t i = 0;
		int contig_num = 1;
		for (size_t b = sph_next(&acc, 0); b < acc.nbuckets; b = sph_next(&acc, b + 1), i++) {
			const char* win = acc.b[b].key;
			const char* cdr3 = (const char*) acc.b[b].val;
			size_t need = strlen(cdr3) + 32;
			ids[i] = (char*) malloc(need);
			snprintf(ids[i], need, "vjf_%d_%s", contig_num++, cdr3);
			if (fp) fprintf(fp, ">%s\n%s\n", ids[i], win);
			memcpy(contigs + i * (size_t) CONTIG_SIZE, win, (size_t) CONTIG_SIZE);
		}
		if (fp) fclose(fp);
		int ok = 0;
		if (sam) {
			/* output_header (quick_map3.c:274-309) then quick_map_process_contig_file (:311-340) */
			fprintf(sam, "@HD\tVN:1.4\tSO:unsorted\n");
			for (size_t j = 0; j < nc; j++) {
				char id[256];
				strncpy(id, ids[j], 255);
				id[255] = 0;
				fprintf(sam, "@SQ\tSN:%s\tLN:%d\n", id, CONTIG_SIZE);
			}
			if (nc && h->sam_body(h->ud, (const char* const*) ids, contigs, nc, CONTIG_SIZE, sam)) { set_err("SAM mapper failed"); ok = -1; }
		}
		for (size_t j = 0; j < nc; j++) free(ids[j]);
		free(ids);
		free(contigs);
		if (ok) goto done;
	}
	STAGE("PRE_CLEANUP");                      /* A2:1461 */
	rc = 0;
done:
	for (size_t i = 0; i < w.win.n; i++) { free(w.win.v[i]); free(w.cdr3.v[i]); }
	free(w.win.v); free(w.cdr3.v);
	sph_free(&w.seen);
	sph_free(&acc);
	free(valid); free(wbuf); free(rk); free(accepted_root); free(roots);
	graph_free(&G);
	if (rc == 0) STAGE("POST_CLEANUP");        /* A2:1464 (`delete nodes`) */
#undef STAGE
	return rc;
}
