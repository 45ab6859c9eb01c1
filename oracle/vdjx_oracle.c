/*
 * oracle/vdjx_oracle.c -- TEST INFRASTRUCTURE ONLY (see vdjx_oracle.h).
 *
 * CPU restatement of the reference's hot path in plain C11.  Same data layout as the reference
 * (byte-per-base ASCII pool, string-keyed open-addressing tables hashed with MurmurHash64A seed 97,
 * single-threaded scans), so that it also serves as the "port" CPU baseline of bench.py.
 * Parity pinned by tests/test_oracle_vs_golden.py against dumps of the compiled reference.
 */
#define _GNU_SOURCE
#include "vdjx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_FREQUENCY 32766   /* A2:66 */
#define MAX_QUAL_SUM 255      /* A2:67 */
#define MIN_BASE_QUALITY 20   /* A2:76 */
#define MAX_KMER_LEN 50       /* A2:70 */
#define SEQ_LEN 16            /* seq_dist.h:4 */

/* ------------------------------------------------------------------------------------------ */
/* hash_utils.c:5-46                                                                          */
/* ------------------------------------------------------------------------------------------ */
uint64_t vdjo_murmur64a(const void* key, int len, uint64_t seed) {
	const uint64_t m = 0xc6a4a7935bd1e995ULL;
	const int r = 47;
	uint64_t h = seed ^ ((uint64_t) len * m);
	const unsigned char* p = (const unsigned char*) key;
	int nblocks = len / 8;
	for (int i = 0; i < nblocks; i++) {
		uint64_t k;
		memcpy(&k, p + 8 * i, 8);
		k *= m; k ^= k >> r; k *= m;
		h ^= k; h *= m;
	}
	const unsigned char* t = p + 8 * nblocks;
	switch (len & 7) {
	case 7: h ^= (uint64_t) t[6] << 48; /* fallthrough */
	case 6: h ^= (uint64_t) t[5] << 40; /* fallthrough */
	case 5: h ^= (uint64_t) t[4] << 32; /* fallthrough */
	case 4: h ^= (uint64_t) t[3] << 24; /* fallthrough */
	case 3: h ^= (uint64_t) t[2] << 16; /* fallthrough */
	case 2: h ^= (uint64_t) t[1] << 8;  /* fallthrough */
	case 1: h ^= (uint64_t) t[0]; h *= m;
	}
	h ^= h >> r; h *= m; h ^= h >> r;
	return h;
}

/* seq_to_kmer.c:6-46 */
uint32_t vdjo_seq_to_int(const char* seq, int* ok) {
	uint32_t val = 0;
	if (ok) *ok = 1;
	for (int i = 0; i < SEQ_LEN; i++) {
		uint32_t b;
		switch (seq[i]) {
		case 'A': b = 0; break;
		case 'T': b = 1; break;
		case 'C': b = 2; break;
		case 'G': b = 3; break;
		default: if (ok) *ok = 0; return 0;
		}
		val = (val << 2) + b;
	}
	return val;
}

/* ------------------------------------------------------------------------------------------ */
/* a-1 / a-2: k-mer table                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
	const char* key;               /* pointer into the pool (never copied), A2:341 */
	const char* first_read;        /* contributingRead, A2:333 */
	uint64_t first_inst;
	uint16_t frequency;
	uint8_t multi;
	uint8_t qual_sums[MAX_KMER_LEN];
} pre_entry;

struct vdjo_table {
	pre_entry* slots;
	size_t nslots, size;
	int rl, k;
};

static size_t table_find(const vdjo_table* t, const char* kmer) {
	size_t mask = t->nslots - 1;
	size_t b = (size_t) vdjo_murmur64a(kmer, t->k, 97) & mask;
	while (t->slots[b].key && strncmp(t->slots[b].key, kmer, (size_t) t->k) != 0) b = (b + 1) & mask;
	return b;
}

static void table_grow(vdjo_table* t) {
	pre_entry* old = t->slots;
	size_t on = t->nslots;
	t->nslots = on * 2;
	t->slots = (pre_entry*) calloc(t->nslots, sizeof(pre_entry));
	for (size_t i = 0; i < on; i++) {
		if (old[i].key) {
			size_t b = table_find(t, old[i].key);
			t->slots[b] = old[i];
		}
	}
	free(old);
}

/* A2:240-259 */
static int include_kmer(const char* seq, const char* qual, int idx, int k) {
	for (int i = idx; i < idx + k; i++) {
		if (seq[i] == 'N') return 0;
		if ((unsigned char) (qual[i] - '!') < MIN_BASE_QUALITY) return 0;
	}
	return 1;
}

/* A2:322-367 */
static void add_to_table(vdjo_table* t, const char* seq, const char* qual, uint64_t rec_index) {
	int rl = t->rl, k = t->k;
	for (int i = 0; i <= rl - k; i++) {
		if (!include_kmer(seq, qual, i, k)) continue;
		const char* kmer = seq + i;
		const char* kq = qual + i;
		size_t b = table_find(t, kmer);
		pre_entry* e = &t->slots[b];
		if (!e->key) {
			if ((t->size + 1) * 2 > t->nslots) {
				table_grow(t);
				b = table_find(t, kmer);
				e = &t->slots[b];
			}
			e->key = kmer;
			e->first_read = seq;
			e->first_inst = (rec_index << (rl > 64 ? 8 : 6)) + (uint64_t) i;      /* record << 6 | offset; reads of more than 64 bases: record << 8 | offset */
			e->frequency = 1;
			e->multi = 0;
			/* A2:337-339: the record's FIRST k qualities, not the k-mer's (shadowed loop index) */
			for (int j = 0; j < k; j++) e->qual_sums[j] = (unsigned char) (qual[j] - '!');
			t->size++;
		} else {
			if (e->frequency < MAX_FREQUENCY - 1) e->frequency++;
			if (!e->multi && !(e->first_read == seq || strncmp(e->first_read, seq, (size_t) rl) == 0)) e->multi = 1;
			for (int j = 0; j < k; j++) {
				unsigned char q = (unsigned char) (kq[j] - '!');
				if ((e->qual_sums[j] + q) < MAX_QUAL_SUM - 41) e->qual_sums[j] = (uint8_t) (e->qual_sums[j] + q);
				else e->qual_sums[j] = MAX_QUAL_SUM;
			}
		}
	}
}

vdjo_table* vdjo_table_build(const uint8_t* primary, size_t n_primary,
                             const uint8_t* secondary, size_t n_secondary, int rl, int k) {
	if (k > MAX_KMER_LEN || k > rl || k < 1) return NULL;
	vdjo_table* t = (vdjo_table*) calloc(1, sizeof *t);
	t->nslots = 1024;
	t->slots = (pre_entry*) calloc(t->nslots, sizeof(pre_entry));
	t->rl = rl; t->k = k;
	size_t rec = 2 * (size_t) rl + 1;
	/* A2:369-409, called for the primary pool then the secondary pool (A2:1388-1390) */
	for (size_t r = 0; r < n_primary; r++) {
		const char* p = (const char*) primary + r * rec;
		add_to_table(t, p + 1, p + 1 + rl, r);
	}
	for (size_t r = 0; r < n_secondary; r++) {
		const char* p = (const char*) secondary + r * rec;
		add_to_table(t, p + 1, p + 1 + rl, n_primary + r);
	}
	return t;
}

size_t vdjo_table_size(const vdjo_table* t) { return t->size; }

/* A2:454-484.  Erasure is by tombstone-free rebuild: membership is all that survives the prune. */
size_t vdjo_table_prune(vdjo_table* t, int mf, int mq) {
	if (mq >= MAX_QUAL_SUM) mq = MAX_QUAL_SUM - 1;        /* A2:1514-1516 */
	pre_entry* old = t->slots;
	size_t on = t->nslots;
	size_t keep = 0;
	for (size_t i = 0; i < on; i++) {
		pre_entry* e = &old[i];
		if (!e->key) continue;
		int good = 1;
		for (int j = 0; j < t->k; j++) if (e->qual_sums[j] < mq) { good = 0; break; }
		if (e->frequency < mf || !e->multi || !good) e->key = NULL; else keep++;
	}
	size_t ns = 1024;
	while (ns < keep * 2 + 2) ns *= 2;
	t->slots = (pre_entry*) calloc(ns, sizeof(pre_entry));
	t->nslots = ns;
	t->size = 0;
	for (size_t i = 0; i < on; i++) {
		if (old[i].key) {
			size_t b = table_find(t, old[i].key);
			t->slots[b] = old[i];
			t->size++;
		}
	}
	free(old);
	return t->size;
}

static int cmp_first_inst(const void* a, const void* b) {
	uint64_t x = ((const pre_entry*) a)->first_inst, y = ((const pre_entry*) b)->first_inst;
	return x < y ? -1 : x > y;
}

void vdjo_table_export(const vdjo_table* t, uint64_t* first_inst, uint32_t* count, uint8_t* multi, uint8_t* qs) {
	pre_entry* tmp = (pre_entry*) malloc((t->size + 1) * sizeof(pre_entry));
	size_t n = 0;
	for (size_t i = 0; i < t->nslots; i++) if (t->slots[i].key) tmp[n++] = t->slots[i];
	qsort(tmp, n, sizeof(pre_entry), cmp_first_inst);
	for (size_t i = 0; i < n; i++) {
		if (first_inst) first_inst[i] = tmp[i].first_inst;
		if (count) count[i] = tmp[i].frequency;
		if (multi) multi[i] = tmp[i].multi;
		if (qs) memcpy(qs + i * MAX_KMER_LEN, tmp[i].qual_sums, MAX_KMER_LEN);
	}
	free(tmp);
}

void vdjo_table_free(vdjo_table* t) {
	if (!t) return;
	free(t->slots);
	free(t);
}

/* ------------------------------------------------------------------------------------------ */
/* a-3: graph build                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
	const char* kmer;
	uint64_t first_inst;
	uint16_t frequency;
	uint8_t has_v, has_j;
	uint8_t to_n, from_n;
	uint32_t to[4], from[4];      /* insertion order; the reference list is this reversed (prepend) */
} gnode;

struct vdjo_graph {
	gnode* nodes;
	size_t n, cap;
	uint32_t* slots;              /* k-mer -> node index+1 */
	size_t nslots;
	int k;
};

static int code_in(const uint32_t* codes, size_t n, uint32_t code) {
	/* vj_filter.c:70-78; the sets use empty key 0, so code 0 is never a member (vj_filter.c:317-318) */
	if (code == 0) return 0;
	size_t lo = 0, hi = n;
	while (lo < hi) {
		size_t mid = (lo + hi) / 2;
		if (codes[mid] < code) lo = mid + 1; else hi = mid;
	}
	return lo < n && codes[lo] == code;
}

static int cmp_u32(const void* a, const void* b) {
	uint32_t x = *(const uint32_t*) a, y = *(const uint32_t*) b;
	return x < y ? -1 : x > y;
}

static size_t graph_find(const vdjo_graph* g, const char* kmer) {
	size_t mask = g->nslots - 1;
	size_t b = (size_t) vdjo_murmur64a(kmer, g->k, 97) & mask;
	while (g->slots[b] && strncmp(g->nodes[g->slots[b] - 1].kmer, kmer, (size_t) g->k) != 0) b = (b + 1) & mask;
	return b;
}

/* A2:210-237 */
static void link_nodes(vdjo_graph* g, uint32_t from, uint32_t to) {
	gnode* f = &g->nodes[from];
	gnode* t = &g->nodes[to];
	int seen = 0;
	for (int i = 0; i < f->to_n; i++) if (f->to[i] == to) seen = 1;
	if (!seen && f->to_n < 4) f->to[f->to_n++] = to;
	seen = 0;
	for (int i = 0; i < t->from_n; i++) if (t->from[i] == from) seen = 1;
	if (!seen && t->from_n < 4) t->from[t->from_n++] = from;
}

/* A2:267-320 */
static void add_to_graph(vdjo_graph* g, const vdjo_table* pruned, const char* seq, uint64_t rec_index, int rl,
                         const uint32_t* vc, size_t nv, const uint32_t* jc, size_t nj) {
	int k = g->k;
	long prev = -1;
	for (int i = 0; i <= rl - k; i++) {
		const char* kmer = seq + i;
		size_t pb = table_find(pruned, kmer);
		if (!pruned->slots[pb].key) { prev = -1; continue; }
		size_t b = graph_find(g, kmer);
		uint32_t curr;
		if (!g->slots[b]) {
			curr = (uint32_t) g->n;
			gnode* nd = &g->nodes[g->n++];
			memset(nd, 0, sizeof *nd);
			nd->kmer = kmer;
			nd->first_inst = (rec_index << (rl > 64 ? 8 : 6)) + (uint64_t) i;
			nd->frequency = 1;
			if (k > SEQ_LEN) {
				int ok;
				uint32_t code = vdjo_seq_to_int(kmer, &ok);
				nd->has_v = (uint8_t) (ok && code_in(vc, nv, code));
				nd->has_j = (uint8_t) (ok && code_in(jc, nj, code));
			} else {
				nd->has_v = nd->has_j = 1;
			}
			g->slots[b] = curr + 1;
		} else {
			curr = g->slots[b] - 1;
			if (g->nodes[curr].frequency < MAX_FREQUENCY - 1) g->nodes[curr].frequency++;
		}
		if (prev >= 0) link_nodes(g, (uint32_t) prev, curr);
		prev = (long) curr;
	}
}

vdjo_graph* vdjo_graph_build(const vdjo_table* pruned,
                             const uint8_t* primary, size_t n_primary,
                             const uint8_t* secondary, size_t n_secondary, int rl, int k,
                             const uint32_t* v_codes, size_t nv, const uint32_t* j_codes, size_t nj) {
	vdjo_graph* g = (vdjo_graph*) calloc(1, sizeof *g);
	g->k = k;
	g->cap = pruned->size + 1;
	g->nodes = (gnode*) calloc(g->cap, sizeof(gnode));
	g->nslots = 1024;
	while (g->nslots < g->cap * 2 + 2) g->nslots *= 2;
	g->slots = (uint32_t*) calloc(g->nslots, sizeof(uint32_t));
	uint32_t* vc = (uint32_t*) malloc((nv + 1) * sizeof(uint32_t));
	uint32_t* jc = (uint32_t*) malloc((nj + 1) * sizeof(uint32_t));
	memcpy(vc, v_codes, nv * sizeof(uint32_t));
	memcpy(jc, j_codes, nj * sizeof(uint32_t));
	qsort(vc, nv, sizeof(uint32_t), cmp_u32);
	qsort(jc, nj, sizeof(uint32_t), cmp_u32);
	size_t rec = 2 * (size_t) rl + 1;
	/* A2:412-452, primary then secondary (A2:1402,1408) */
	for (size_t r = 0; r < n_primary; r++)
		add_to_graph(g, pruned, (const char*) primary + r * rec + 1, r, rl, vc, nv, jc, nj);
	for (size_t r = 0; r < n_secondary; r++)
		add_to_graph(g, pruned, (const char*) secondary + r * rec + 1, n_primary + r, rl, vc, nv, jc, nj);
	free(vc);
	free(jc);
	return g;
}

size_t vdjo_graph_nodes(const vdjo_graph* g) { return g->n; }

void vdjo_graph_export(const vdjo_graph* g, uint64_t* first_inst, uint32_t* freq, uint8_t* has_v, uint8_t* has_j,
                       uint8_t* to_deg, uint32_t* to_ids, uint8_t* from_deg, uint32_t* from_ids) {
	for (size_t i = 0; i < g->n; i++) {
		const gnode* nd = &g->nodes[i];
		if (first_inst) first_inst[i] = nd->first_inst;
		if (freq) freq[i] = nd->frequency;
		if (has_v) has_v[i] = nd->has_v;
		if (has_j) has_j[i] = nd->has_j;
		if (to_deg) to_deg[i] = nd->to_n;
		if (from_deg) from_deg[i] = nd->from_n;
		for (int e = 0; e < 4; e++) {
			if (to_ids) to_ids[i * 4 + e] = e < nd->to_n ? nd->to[nd->to_n - 1 - e] + 1 : 0;
			if (from_ids) from_ids[i * 4 + e] = e < nd->from_n ? nd->from[nd->from_n - 1 - e] + 1 : 0;
		}
	}
}

void vdjo_graph_free(vdjo_graph* g) {
	if (!g) return;
	free(g->nodes);
	free(g->slots);
	free(g);
}

/* ------------------------------------------------------------------------------------------ */
/* a-7: root scorer                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const char* p; int pos; } vk_entry;

struct vdjo_scorer {
	char** lines;
	size_t* lens;
	size_t n_lines;
	vk_entry* idx;        /* every vk-mer start of every line, sorted by content */
	size_t n_idx;
	int vk;
};

static int g_vk_cmp_len;
static int cmp_vk(const void* a, const void* b) {
	const vk_entry* x = (const vk_entry*) a;
	const vk_entry* y = (const vk_entry*) b;
	int c = strncmp(x->p, y->p, (size_t) g_vk_cmp_len);
	if (c) return c;
	return x->pos < y->pos ? -1 : x->pos > y->pos;
}

/* seq_score.c:36-70: positions of all lines share one map keyed by the vk-mer's content */
vdjo_scorer* vdjo_scorer_new(const char* const* lines, size_t n_lines, int vk) {
	vdjo_scorer* s = (vdjo_scorer*) calloc(1, sizeof *s);
	s->vk = vk;
	s->n_lines = n_lines;
	s->lines = (char**) calloc(n_lines + 1, sizeof(char*));
	s->lens = (size_t*) calloc(n_lines + 1, sizeof(size_t));
	size_t total = 0;
	for (size_t i = 0; i < n_lines; i++) {
		s->lines[i] = strdup(lines[i]);
		s->lens[i] = strlen(lines[i]);
		if (s->lens[i] > (size_t) vk) total += s->lens[i] - (size_t) vk;
	}
	s->idx = (vk_entry*) malloc((total + 1) * sizeof(vk_entry));
	for (size_t i = 0; i < n_lines; i++) {
		long stop = (long) s->lens[i] - vk;        /* seq_score.c:37: i < strlen - vk */
		for (long j = 0; j < stop; j++) {
			s->idx[s->n_idx].p = s->lines[i] + j;
			s->idx[s->n_idx].pos = (int) j;
			s->n_idx++;
		}
	}
	g_vk_cmp_len = vk;
	qsort(s->idx, s->n_idx, sizeof(vk_entry), cmp_vk);
	return s;
}

/* seq_score.c:76-116: (len1+1) x (len2+1) char matrix, free ends, match +1 / mismatch 0 / gap -1 */
static int score_dp(const char* seq1, const char* seq2, int len1, int len2, int threshold) {
	signed char col_prev[MAX_KMER_LEN + 2], col_cur[MAX_KMER_LEN + 2];
	if (threshold <= 0) return 1;                 /* row/col 0 hold zeros and are tested too */
	memset(col_prev, 0, sizeof col_prev);
	for (int c = 1; c <= len2; c++) {
		col_cur[0] = 0;
		for (int r = 1; r <= len1; r++) {
			int v1 = col_prev[r] - 1;                                     /* (row, col-1) + GAP */
			int v2 = col_cur[r - 1] - 1;                                  /* (row-1, col) + GAP */
			int v3 = col_prev[r - 1] + (seq1[r - 1] == seq2[c - 1] ? 1 : 0);
			int m = v1 > v2 ? v1 : v2;
			m = m > v3 ? m : v3;
			col_cur[r] = (signed char) m;
			if (m >= threshold) return 1;
		}
		memcpy(col_prev, col_cur, sizeof col_cur);
	}
	return 0;
}

static int cmp_int(const void* a, const void* b) {
	int x = *(const int*) a, y = *(const int*) b;
	return x < y ? -1 : x > y;
}

/* seq_score.c:118-156 */
int vdjo_score_seq(const vdjo_scorer* s, const char* kmer, int k, int threshold) {
	int stop = k - s->vk;
	int* pos = NULL;
	size_t np = 0, cap = 0;
	for (int i = 0; i < stop; i++) {
		/* lower bound on content */
		size_t lo = 0, hi = s->n_idx;
		while (lo < hi) {
			size_t mid = (lo + hi) / 2;
			if (strncmp(s->idx[mid].p, kmer + i, (size_t) s->vk) < 0) lo = mid + 1; else hi = mid;
		}
		for (; lo < s->n_idx && strncmp(s->idx[lo].p, kmer + i, (size_t) s->vk) == 0; lo++) {
			if (np == cap) { cap = cap ? cap * 2 : 64; pos = (int*) realloc(pos, cap * sizeof(int)); }
			pos[np++] = s->idx[lo].pos;
		}
	}
	if (!np) { free(pos); return 0; }
	qsort(pos, np, sizeof(int), cmp_int);
	int res = 0;
	for (size_t li = 0; li < s->n_lines && !res; li++) {
		long len = (long) s->lens[li];
		for (size_t j = 0; j < np && !res; j++) {
			if (j && pos[j] == pos[j - 1]) continue;        /* std::set */
			long start = (long) pos[j] - k;
			if (start < 0) start = 0;
			if (len >= 2L * k && start >= len - 2L * k) start = len - 2L * k - 1;
			if (start < 0 || start + 2L * k > len) continue;  /* the reference reads out of bounds here */
			res = score_dp(kmer, s->lines[li] + start, k, 2 * k, threshold);
		}
	}
	free(pos);
	return res;
}

void vdjo_scorer_free(vdjo_scorer* s) {
	if (!s) return;
	for (size_t i = 0; i < s->n_lines; i++) free(s->lines[i]);
	free(s->lines);
	free(s->lens);
	free(s->idx);
	free(s);
}

/* ------------------------------------------------------------------------------------------ */
/* a-8 / a-9 / a-10                                                                            */
/* ------------------------------------------------------------------------------------------ */
struct vdjo_readidx {
	const uint8_t* primary;
	const uint8_t* secondary;
	size_t n_primary, n_records;
	int rl;
	const uint32_t* pair_id;
	const uint8_t* read_num;
	const uint8_t* is_rc;
	uint32_t* order;          /* record indices sorted by (sequence, registration rank) */
	uint32_t n_pairs;
	/* per-call R2 map keyed by pair id (quick_map3.c:197,214), epoch-stamped */
	uint32_t* r2_epoch;
	uint32_t* r2_rec;
	int16_t* r2_pos;
	uint32_t epoch;
};

static const char* rec_seq(const vdjo_readidx* ix, uint32_t r) {
	size_t rec = 2 * (size_t) ix->rl + 1;
	return r < ix->n_primary ? (const char*) ix->primary + r * rec + 1
	                         : (const char*) ix->secondary + (r - ix->n_primary) * rec + 1;
}

static const vdjo_readidx* g_sort_ix;
static const uint32_t* g_sort_rank;
static int cmp_rec(const void* a, const void* b) {
	uint32_t x = *(const uint32_t*) a, y = *(const uint32_t*) b;
	int c = strncmp(rec_seq(g_sort_ix, x), rec_seq(g_sort_ix, y), (size_t) g_sort_ix->rl);
	if (c) return c;
	return g_sort_rank[x] < g_sort_rank[y] ? -1 : g_sort_rank[x] > g_sort_rank[y];
}

/* quick_map3.c:126-149: sequence -> instances in registration order */
vdjo_readidx* vdjo_readidx_build(const uint8_t* primary, size_t n_primary,
                                 const uint8_t* secondary, size_t n_secondary, int rl,
                                 const uint32_t* pair_id, const uint8_t* read_num, const uint8_t* is_rc,
                                 const uint32_t* reg_rank, uint32_t n_pairs) {
	vdjo_readidx* ix = (vdjo_readidx*) calloc(1, sizeof *ix);
	ix->primary = primary; ix->secondary = secondary;
	ix->n_primary = n_primary; ix->n_records = n_primary + n_secondary;
	ix->rl = rl;
	ix->pair_id = pair_id; ix->read_num = read_num; ix->is_rc = is_rc;
	ix->n_pairs = n_pairs;
	ix->order = (uint32_t*) malloc((ix->n_records + 1) * sizeof(uint32_t));
	for (size_t i = 0; i < ix->n_records; i++) ix->order[i] = (uint32_t) i;
	g_sort_ix = ix; g_sort_rank = reg_rank;
	qsort(ix->order, ix->n_records, sizeof(uint32_t), cmp_rec);
	ix->r2_epoch = (uint32_t*) calloc((size_t) n_pairs + 1, sizeof(uint32_t));
	ix->r2_rec = (uint32_t*) calloc((size_t) n_pairs + 1, sizeof(uint32_t));
	ix->r2_pos = (int16_t*) calloc((size_t) n_pairs + 1, sizeof(int16_t));
	return ix;
}

typedef struct { uint32_t rec; int16_t pos; } r1_hit;

static int cmp_start(const void* a, const void* b) {
	const int32_t* x = (const int32_t*) a;
	const int32_t* y = (const int32_t*) b;
	if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
	return x[1] < y[1] ? -1 : x[1] > y[1];
}

/* quick_map3.c:188-266 */
size_t vdjo_quick_map(vdjo_readidx* ix, const char* contig, int len, vdjo_pair* pairs_out, int32_t* starts_out, size_t cap) {
	int rl = ix->rl;
	ix->epoch++;
	r1_hit* r1 = NULL;
	size_t n1 = 0, c1 = 0;
	for (int i = 0; i < len - rl; i++) {                    /* last offset excluded (quick_map3.c:200) */
		size_t lo = 0, hi = ix->n_records;
		while (lo < hi) {
			size_t mid = (lo + hi) / 2;
			if (strncmp(rec_seq(ix, ix->order[mid]), contig + i, (size_t) rl) < 0) lo = mid + 1; else hi = mid;
		}
		for (; lo < ix->n_records && strncmp(rec_seq(ix, ix->order[lo]), contig + i, (size_t) rl) == 0; lo++) {
			uint32_t r = ix->order[lo];
			if (ix->read_num[r] == 1) {
				if (n1 == c1) { c1 = c1 ? c1 * 2 : 1024; r1 = (r1_hit*) realloc(r1, c1 * sizeof(r1_hit)); }
				r1[n1].rec = r; r1[n1].pos = (int16_t) (i + 1); n1++;
			} else {
				uint32_t p = ix->pair_id[r];                /* read2[id] = m_info: last writer wins */
				ix->r2_epoch[p] = ix->epoch;
				ix->r2_rec[p] = r;
				ix->r2_pos[p] = (int16_t) (i + 1);
			}
		}
	}
	size_t np = 0;
	for (size_t i = 0; i < n1; i++) {
		uint32_t p = ix->pair_id[r1[i].rec];
		if (ix->r2_epoch[p] != ix->epoch) continue;
		uint32_t r2 = ix->r2_rec[p];
		if (ix->is_rc[r1[i].rec] == ix->is_rc[r2]) continue;
		int d = r1[i].pos - ix->r2_pos[p];
		int16_t insert = (int16_t) ((d < 0 ? -d : d) + rl);
		if (insert < 50 || insert > 400) continue;          /* quick_map3.c:23-24 */
		if (np < cap) {
			if (pairs_out) {
				vdjo_pair* o = &pairs_out[np];
				o->pair_id = p; o->rec1 = r1[i].rec; o->rec2 = r2;
				o->pos1 = r1[i].pos; o->pos2 = ix->r2_pos[p]; o->insert = insert;
				o->rc1 = ix->is_rc[r1[i].rec]; o->rc2 = ix->is_rc[r2];
			}
			if (starts_out) {
				starts_out[4 * np + 0] = r1[i].pos; starts_out[4 * np + 1] = ix->r2_pos[p];
				starts_out[4 * np + 2] = ix->r2_pos[p]; starts_out[4 * np + 3] = r1[i].pos;
			}
		}
		np++;
	}
	if (starts_out) qsort(starts_out, 2 * (np < cap ? np : cap), 2 * sizeof(int32_t), cmp_start);
	free(r1);
	return np;
}

/* coverage.c:10-61 */
static int mate_coverage_is_valid(int rl, int contig_len, int eval_start, int eval_stop, int mate_span,
                                  int insert_low, int insert_high, int floor_, const int32_t* st, size_t n) {
	long num = (long) n;
	long begin = 0;
	int pos = eval_start;
	int valid = 1;
	int mate_low = 0, mate_high = 0;
	int* coverage = (int*) malloc(((size_t) contig_len + 1 + 1024) * sizeof(int));
	while (mate_low < eval_stop && pos < eval_stop && valid) {
		while (begin < num && st[2 * begin] + rl - 1 < pos) begin++;
		mate_low = pos + insert_low - rl - mate_span / 2;
		mate_high = pos + insert_high - rl + mate_span / 2;
		if (mate_high > eval_stop) mate_high = eval_stop + 1;
		memset(coverage, 0, sizeof(int) * ((size_t) contig_len + 1 + 1024));
		long idx = begin;
		while (idx < num && st[2 * idx] <= pos) {
			for (int i = 0; i < rl; i++) coverage[st[2 * idx + 1] + i] += 1;
			idx++;
		}
		for (int j = mate_low; j < mate_high; j++) {
			if (j < 0 || j > contig_len + 1023 || coverage[j] < floor_) { valid = 0; break; }
		}
		pos++;
	}
	free(coverage);
	return valid;
}

/* coverage.c:64-130 */
int vdjo_coverage_is_valid(int rl, int contig_len, int eval_start, int eval_stop, int read_span,
                           int insert_low, int insert_high, int floor_, const int32_t* st, size_t n, int mate_span) {
	int valid = 1;
	int start_gap = rl - read_span;
	long num = (long) n;
	int first_in_range = -1;
	long i = 0;
	while (i < num && st[2 * i] <= (eval_stop - read_span) + 1) {
		if (st[2 * i] >= eval_start) {
			if (i < floor_) { valid = 0; break; }
			if (first_in_range < 0) {
				first_in_range = st[2 * i];
				if (first_in_range > eval_start + start_gap) { valid = 0; break; }
			}
			if (st[2 * (i - floor_)] < st[2 * i] - start_gap) { valid = 0; break; }
		}
		i++;
	}
	if (i >= num) i = num - 1;
	if (i > floor_ && st[2 * (i - floor_)] < st[2 * i] - start_gap) valid = 0;
	if (i > 0 && num > 0 && st[2 * i] > eval_stop - rl && valid) valid = 1; else valid = 0;
	if (valid) valid = mate_coverage_is_valid(rl, contig_len, eval_start, eval_stop, mate_span, insert_low, insert_high, floor_, st, n);
	return valid;
}

/* quick_map3.c:152-181 */
int vdjo_sam_pair(const vdjo_readidx* ix, const char* contig_id, const char* read_name, const vdjo_pair* p, char* buf) {
	int rl = ix->rl;
	const char* id = read_name[0] == '@' ? read_name + 1 : read_name;
	int flag1 = 0x1 | 0x2 | (p->rc1 ? 0x10 : 0x20) | 0x40;
	int flag2 = 0x1 | 0x2 | (p->rc2 ? 0x10 : 0x20) | 0x80;
	const char* s1 = rec_seq(ix, p->rec1);
	const char* s2 = rec_seq(ix, p->rec2);
	int n = sprintf(buf, "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%.*s\t%.*s\n", id, flag1, contig_id, (int) p->pos1, rl,
			(int) p->pos2, (int) p->insert, rl, s1, rl, s1 + rl);
	n += sprintf(buf + n, "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%.*s\t%.*s\n", id, flag2, contig_id, (int) p->pos2, rl,
			(int) p->pos1, (int) p->insert, rl, s2, rl, s2 + rl);
	return n;
}

void vdjo_readidx_free(vdjo_readidx* ix) {
	if (!ix) return;
	free(ix->order);
	free(ix->r2_epoch);
	free(ix->r2_rec);
	free(ix->r2_pos);
	free(ix);
}

/* ---- f-3: the v_index / j_index generator (seq_dist.c) -------------------------------------------
 * edit_dist (seq_dist.c:11-27): number of differing 2-bit bases of two 16-base codes.
 * process_kmers (seq_dist.c:49-71): for every code i in [start, end] (inclusive) the minimum over the
 * anchors; a row "i\tmin" when min <= MAX_DIST (5).  Returns the number of rows; fills at most cap. */
static int edit_dist16(uint32_t a, uint32_t b) {
	uint32_t v = a ^ b;
	int d = 0;
	while (v) {
		if (v & 3u) d++;
		v >>= 2;
	}
	return d;
}

size_t vdjo_index_rows(const uint32_t* anchors, size_t n_anchors, uint64_t start, uint64_t end, int max_dist,
                       uint32_t* codes, uint8_t* dists, size_t cap) {
	size_t n = 0;
	for (uint64_t i = start; i <= end && i <= 0xFFFFFFFFull; i++) {
		int best = 17;                                  /* SEQ_LEN + 1 */
		for (size_t a = 0; a < n_anchors; a++) {
			int d = edit_dist16((uint32_t) i, anchors[a]);
			if (d < best) best = d;
		}
		if (best <= max_dist) {
			if (n < cap) {
				if (codes) codes[n] = (uint32_t) i;
				if (dists) dists[n] = (uint8_t) best;
			}
			n++;
		}
	}
	return n;
}
