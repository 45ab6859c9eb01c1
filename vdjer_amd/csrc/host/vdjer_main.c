/*
 * vdjer_main.c -- the `vdjer` command line (params.c:74-101, 214-298) over libvdjx (GPU hot path) and
 * vdjh (serial host stage).  Plain C; talks to the GPU only through the C ABI of include/vdjx.h.
 *
 *   vdjer --in <reads> --chain IGH|IGK|IGL --ref-dir <dir> --ins <n> [--t --k --mf --mq --mcs --am --miw --maw
 *         --jc --ws -jext --rf --vk --mrs --rs --ms --e0 --e1 --wo --vf --jf --rms]
 * writes ./vdj_contigs.fa and ./vdjer.dot, SAM on stdout, log on stderr; exit 0 on success.
 *
 * --in: a BAM with its .bai (extraction as bam_read.c:264-446, restated over zlib in bamx.c), or -- recognised by its
 * content -- the extracted read pool as text, one read per line in extraction order:
 *     <pool:P|S> <name> <read_num:1|2> <is_rev:0|1> <SEQ> <QUAL>
 * from which the pool records (as-is + reverse complement, bam_read.c:206-244) are rebuilt.
 * stderr carries the reference's ELAPSED_SECS stage markers (status.c:22-32) in the reference's order; the k-mer table, prune and
 * graph build are ONE device call here, so the markers between PRE_PRE_GRAPH1 and POST_BUILD_GRAPH2 follow it back to back.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "../../../include/vdjx.h"
#include "sph.h"
#include "vdjh.h"
#include "bamx.h"
#include "vdjx_mgpu.h"
#include <errno.h>
#include <sys/wait.h>
#include <sys/resource.h>
#include <sys/prctl.h>
#include <signal.h>
#include <pthread.h>
#include <unistd.h>

typedef struct {
	vdjh_params hp;
	const char* in;
	char v_anchors[4096], j_anchors[4096], source_sim_file[4096], vdj_fasta[4096];
	char v_region[64], c_region[64];       /* set_chain_info, params.c:8-35 (hg38 coordinates); --vr / --cr override */
	int anchor_mismatches, threads;
	int gpus;                              /* --gpus N (not in the reference): the k-mer build sharded over N GPUs of this node */
	int have_chain, have_ref;
} cli;

static void usage(void) {
	fprintf(stderr, "vdjer \n\t--in <input BAM (with .bai), or the extracted reads as text (see vdjer_main.c)>\n\t--chain <IGH|IGK|IGL>\n\t--ref-dir </path/to/vdjer/ref/dir>\n"
	                "\t--mf <min node frequency (default: 3)>\n\t--mq <min base quality (default: 90)>\n\t--mcs <min contig score (default: -5)\n"
	                "\t--t <threads (default: 1)\n\t--am <anchor mismatches (default: 4)\n\t--miw/--maw <min/max window between conserved amino acids>\n"
	                "\t--jc <conserved J amino acid (W|F)\n\t--ws <window span (default: 486)\n\t--jext <J extension (default: 162)\n"
	                "\t--ins <expected / median insert length>\n\t--rf <read filter floor (default: 1)\n\t--k <kmer size (default: 35)>\n"
	                "\t--vk <vregion kmer size (default: 15)>\n\t--mrs <min source node homology score (default: 30)\n"
	                "\t--rs <read span distance (default: 35)>\n\t--ms <mate span distance (default: 48)>\n"
	                "\t--e0/--e1 <start/stop position for contig filtering (default: 52/411)>\n\t--wo <window overlap check size>\n"
	                "\t--gpus <GPUs of this node to shard the k-mer table over (default: 1)>\n");
}

static int file_exists(const char* f) { struct stat b; return stat(f, &b) == 0; }

/* params.c:214-298: positional "--flag value" pairs; unknown flags only warn */
static int parse(int argc, char** argv, cli* c) {
	memset(c, 0, sizeof *c);
	vdjh_default_params(&c->hp);
	c->anchor_mismatches = 4;
	c->threads = 1;
	c->gpus = 1;
	for (int i = 1; i < argc; i += 2) {
		const char* a = argv[i];
		if (!strcmp(a, "--help")) { usage(); exit(0); }
		if (i + 1 >= argc) { fprintf(stderr, "Missing value for param: %s\n", a); usage(); return -1; }
		const char* v = argv[i + 1];
		if (!strcmp(a, "--in")) c->in = v;
		else if (!strcmp(a, "--chain")) {
			if (vdjh_set_chain(&c->hp, v)) { fprintf(stderr, "%s\n", vdjh_last_error()); return -1; }
			c->have_chain = 1;
			const char* vr = !strcmp(v, "IGH") ? "chr14:105566277-106879844" : !strcmp(v, "IGL") ? "chr22:22026076-22922913" : "chr2:89851758-90235368";
			const char* cr = !strcmp(v, "IGH") ? "chr14:105566277-105939754" : !strcmp(v, "IGL") ? "chr22:22895375-22922913" : "chr2:88857361-88857683";
			snprintf(c->v_region, sizeof c->v_region, "%s", vr);
			snprintf(c->c_region, sizeof c->c_region, "%s", cr);
		}
		else if (!strcmp(a, "--ref-dir")) {
			snprintf(c->v_anchors, sizeof c->v_anchors, "%s/v_index", v);
			snprintf(c->j_anchors, sizeof c->j_anchors, "%s/j_index", v);
			snprintf(c->source_sim_file, sizeof c->source_sim_file, "%s/v_region.fa", v);
			snprintf(c->vdj_fasta, sizeof c->vdj_fasta, "%s/ig_vdj.fa", v);
			c->have_ref = 1;
		}
		else if (!strcmp(a, "--mf")) c->hp.min_node_freq = atoi(v);
		else if (!strcmp(a, "--mq")) c->hp.min_base_quality = atoi(v);
		else if (!strcmp(a, "--mcs")) c->hp.min_contig_score = (float) atof(v);
		else if (!strcmp(a, "--t")) c->threads = atoi(v);
		else if (!strcmp(a, "--gpus")) c->gpus = atoi(v);
		else if (!strcmp(a, "--vf")) snprintf(c->v_anchors, sizeof c->v_anchors, "%s", v);
		else if (!strcmp(a, "--jf")) snprintf(c->j_anchors, sizeof c->j_anchors, "%s", v);
		else if (!strcmp(a, "--am")) c->anchor_mismatches = atoi(v);
		else if (!strcmp(a, "--miw")) c->hp.vj_min_win = atoi(v);
		else if (!strcmp(a, "--maw")) c->hp.vj_max_win = atoi(v);
		else if (!strcmp(a, "--jc")) c->hp.j_conserved = v[0];
		else if (!strcmp(a, "--ws")) c->hp.window_span = atoi(v);
		else if (!strcmp(a, "-jext")) c->hp.j_extension = atoi(v);        /* sic: params.c:262 */
		else if (!strcmp(a, "--vdjf")) snprintf(c->vdj_fasta, sizeof c->vdj_fasta, "%s", v);
		else if (!strcmp(a, "--vr")) snprintf(c->v_region, sizeof c->v_region, "%s", v);
		else if (!strcmp(a, "--cr")) snprintf(c->c_region, sizeof c->c_region, "%s", v);
		else if (!strcmp(a, "--ins")) c->hp.insert_len = atoi(v);
		else if (!strcmp(a, "--rf")) c->hp.read_filter_floor = atoi(v);
		else if (!strcmp(a, "--k")) c->hp.k = atoi(v);
		else if (!strcmp(a, "--rms")) snprintf(c->source_sim_file, sizeof c->source_sim_file, "%s", v);
		else if (!strcmp(a, "--vk")) c->hp.vregion_kmer_size = atoi(v);
		else if (!strcmp(a, "--mrs")) c->hp.min_source_homology_score = atoi(v);
		else if (!strcmp(a, "--rs")) c->hp.filter_read_span = atoi(v);
		else if (!strcmp(a, "--ms")) c->hp.filter_mate_span = atoi(v);
		else if (!strcmp(a, "--e0")) c->hp.eval_start = atoi(v);
		else if (!strcmp(a, "--e1")) c->hp.eval_stop = atoi(v);
		else if (!strcmp(a, "--wo")) c->hp.window_overlap_check_size = atoi(v);
		else fprintf(stderr, "Invalid param: %s\n", a);
	}
	int ok = 1;       /* validate_params, params.c:143-212 */
	if (!c->in) { fprintf(stderr, "Input must be specified\n"); ok = 0; }
	else if (!file_exists(c->in)) { fprintf(stderr, "Could not find input file: %s\n", c->in); ok = 0; }
	if (!c->v_anchors[0]) { fprintf(stderr, "V anchor file must be specified\n"); ok = 0; }
	else if (!file_exists(c->v_anchors)) { fprintf(stderr, "Could not locate v_index file: %s\n", c->v_anchors); ok = 0; }
	if (!c->j_anchors[0]) { fprintf(stderr, "J anchor file must be specified\n"); ok = 0; }
	else if (!file_exists(c->j_anchors)) { fprintf(stderr, "Could not locate j_index file: %s\n", c->j_anchors); ok = 0; }
	if (!c->source_sim_file[0]) { fprintf(stderr, "source_sim_file file must be specified\n"); ok = 0; }
	if (c->hp.j_conserved != 'W' && c->hp.j_conserved != 'F') { fprintf(stderr, "Conserved J AA must be W or F: %c\n", c->hp.j_conserved); ok = 0; }
	if (c->hp.insert_len <= 0) { fprintf(stderr, "insert_len must be specified and > 0\n"); ok = 0; }
	if (!ok) { usage(); return -1; }
	if (c->hp.min_base_quality >= 255) c->hp.min_base_quality = 254;      /* A2:1514-1516 */
	return 0;
}

typedef struct { int device; vdjx_ctx* gx; int rc; char err[512]; } init_job;
static void* init_thread(void* p) {
	init_job* j = (init_job*) p;
	j->rc = vdjx_init(j->device, &j->gx);
	if (j->rc) snprintf(j->err, sizeof j->err, "%s", vdjx_last_error());      /* (the message is the calling thread's) */
	return NULL;
}

static time_t t_start, t_prev;
static int g_rank;                             /* --gpus N: only rank 0 writes the stage log */
static void status(const char* desc) {         /* status.c:22-32 without the /proc dumps */
	if (g_rank) return;
	if (getenv("VDJX_TIMES")) {                 /* (diagnostic: milliseconds since the process started, beside the reference's whole seconds) */
		static struct timespec t0;
		struct timespec t;
		clock_gettime(CLOCK_MONOTONIC, &t);
		if (!t0.tv_sec) t0 = t;
		fprintf(stderr, "VDJX_TIMES\t%s\t%.1f\n", desc, (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) / 1e6);
	}
	time_t now = time(NULL);
	fprintf(stderr, "ELAPSED_SECS\t%s\t%ld\t%ld\n", desc, (long) (now - t_start), (long) (now - t_prev));
	t_prev = now;
}

/* ------------------------------------------------------------------------------------------ */
/* inputs                                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* What one process holds of the read pool.  One GPU: all of it, the two pools as the reference lays them out (bam_read.c:376-379).
 * `--gpus N`: the process's SHARE -- the pairs whose read name hashes to its rank, both mates, all four records -- as ONE block
 * (`primary`: its primary-pool records followed by its secondary-pool ones, each group in extraction order), with every record's place
 * in the scan order of the whole pool and its registration rank over the whole pool beside it. */
typedef struct {
	int rl;
	uint8_t *primary, *secondary;
	size_t n_primary, n_secondary;      /* records */
	uint32_t *pair_id, *reg_rank;       /* per record of this process, block order */
	uint32_t* scan_index;               /* `--gpus N`: place in the whole pool's scan order (primary pool, then secondary) */
	uint8_t *read_num, *is_rc;
	char** names;                       /* by pair id (the process's own numbering); the characters live in `name_blocks` */
	void* name_blocks;
	uint32_t n_pairs;
	uint64_t total_records;             /* of the whole pool */
} reads_t;

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; } }
static uint8_t comp_lut[256];
static void comp_init(void) { if (!comp_lut['A']) for (int i = 0; i < 256; i++) comp_lut[i] = (uint8_t) comp((char) i); }

/* one extracted read before it becomes two pool records */
typedef struct {
	char pool; char* name; int rn, rev; char* seq; char* qual;
	uint64_t seq_no, pool_no;           /* among all reads of the extraction / among the reads of its pool (kept by this process or not) */
} read_in;
typedef struct str_blk { struct str_blk* next; size_t used, cap; char data[]; } str_blk;
typedef struct {
	read_in* v; size_t n, cap;
	uint64_t np, ns;                    /* reads of the primary / secondary pool, kept or not */
	int max_len;
	int rank, nranks;
	int one_block;                      /* lay the records out as a share (one block), whatever nranks: the `--gpus N` code path */
	str_blk *strs, *name_strs;          /* sequences and qualities / names of the kept reads: appended to megabyte blocks (a malloc per field
	                                     * cost more than reading it); the names' blocks go on to the pool, which points into them */
} collector;

static char* blk_str(str_blk** head, const char* s, size_t n) {
	if (!*head || (*head)->used + n + 1 > (*head)->cap) {
		const size_t cap = n + 1 > ((size_t) 4 << 20) ? n + 1 : ((size_t) 4 << 20);
		str_blk* b = (str_blk*) malloc(sizeof(str_blk) + cap);
		if (!b) return NULL;
		b->next = *head; b->used = 0; b->cap = cap;
		*head = b;
	}
	char* p = (*head)->data + (*head)->used;
	memcpy(p, s, n);
	p[n] = 0;
	(*head)->used += n + 1;
	return p;
}
static char* col_str(collector* c, const char* s, size_t n) { return blk_str(&c->strs, s, n); }
static char* col_name(collector* c, const char* s, size_t n) { return blk_str(&c->name_strs, s, n); }

static uint64_t name_hash(const char* s) {
	uint64_t h = 1469598103934665603ull;
	for (; *s; s++) { h ^= (unsigned char) *s; h *= 1099511628211ull; }
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
	return h;
}
/* `--gpus N`: which rank owns the pair of this name */
static int owner_of(const char* name, int nranks) { return nranks > 1 ? (int) (name_hash(name) % (uint64_t) nranks) : 0; }
static int keep_name(void* ud, const char* name) { const collector* c = (const collector*) ud; return owner_of(name, c->nranks) == c->rank; }

/* (name must be NUL-terminated at name[ln]) */
static int collect_n(collector* c, char pool, const char* name, size_t ln, int rn, int rev, const char* seq, size_t ls, const char* qual, size_t lq) {
	const uint64_t seq_no = c->np + c->ns, pool_no = pool == 'P' ? c->np++ : c->ns++;
	const int l = (int) ls;
	if (l > c->max_len) c->max_len = l;                       /* get_read_length: the maximum (bam_read.c:264-292) */
	if (owner_of(name, c->nranks) != c->rank) return 0;
	if (c->n == c->cap) {
		c->cap = c->cap ? c->cap * 2 : 4096;
		c->v = (read_in*) realloc(c->v, c->cap * sizeof(read_in));
		if (!c->v) return -1;
	}
	read_in* x = &c->v[c->n++];
	x->pool = pool; x->rn = rn; x->rev = rev; x->seq_no = seq_no; x->pool_no = pool_no;
	x->name = col_name(c, name, ln); x->seq = col_str(c, seq, ls); x->qual = col_str(c, qual, lq);
	return x->name && x->seq && x->qual ? 0 : -1;
}
static void collector_free(collector* c) {
	for (str_blk* b = c->strs; b;) { str_blk* nx = b->next; free(b); b = nx; }
	for (str_blk* b = c->name_strs; b;) { str_blk* nx = b->next; free(b); b = nx; }
	free(c->v);
	memset(c, 0, sizeof *c);
}

/* add_to_buffer (bam_read.c:206-244) for every collected read: forward record, reverse-complement record, both registered.
 * One GPU: two pools, pair ids in extraction order.  A share: one block, primary-pool reads first. */
static int build_reads(const collector* c, reads_t* r) {
	const int rl = c->max_len, multi = c->one_block;
	size_t np = 0, ns = 0;
	for (size_t i = 0; i < c->n; i++) { if (c->v[i].pool == 'P') np++; else ns++; }
	memset(r, 0, sizeof *r);
	r->rl = rl;
	r->total_records = 2 * (c->np + c->ns);
	if (r->total_records >= (1ull << 32)) { fprintf(stderr, "more than 2^32 pool records\n"); return -1; }
	const size_t rec = 2 * (size_t) rl + 1;
	const size_t R = 2 * (np + ns);
	r->n_primary = multi ? R : 2 * np;
	r->n_secondary = multi ? 0 : 2 * ns;
	r->primary = (uint8_t*) calloc(r->n_primary * rec + 1, 1);
	r->secondary = multi ? NULL : (uint8_t*) calloc(r->n_secondary * rec + 1, 1);
	r->pair_id = (uint32_t*) calloc(R + 1, 4);
	r->reg_rank = (uint32_t*) calloc(R + 1, 4);
	r->scan_index = (uint32_t*) calloc(R + 1, 4);
	r->read_num = (uint8_t*) calloc(R + 1, 1);
	r->is_rc = (uint8_t*) calloc(R + 1, 1);
	r->names = (char**) calloc(np + ns + 1, sizeof(char*));
	if (!r->primary || (!multi && !r->secondary) || !r->pair_id || !r->reg_rank || !r->scan_index || !r->read_num || !r->is_rc || !r->names) { fprintf(stderr, "out of memory\n"); return -1; }
	sph_table ids;                       /* read name -> pair id */
	sph_init(&ids, 0, 0);
	comp_init();
	size_t ip = 0, is = 0;
	for (size_t q = 0; q < c->n; q++) {
		const read_in* in = &c->v[q];
		if ((int) strlen(in->seq) != rl || (int) strlen(in->qual) != rl) { fprintf(stderr, "read %s: length != %d\n", in->name, rl); sph_free(&ids); return -1; }
		size_t b = sph_find(&ids, in->name);
		uint32_t pid;
		if (b == (size_t) -1) {
			pid = r->n_pairs++;
			r->names[pid] = in->name;                          /* (in the collector's string blocks, which this pool takes over) */
			sph_map_put(&ids, r->names[pid], (void*) (uintptr_t) (pid + 1), NULL);
		} else {
			pid = (uint32_t) (uintptr_t) ids.b[b].val - 1;
		}
		const int isp = in->pool == 'P';
		const size_t g = isp ? ip : 2 * np + is;                /* first record of the read in this process's block order */
		uint8_t* base = multi || isp ? r->primary + g * rec : r->secondary + is * rec;
		base[0] = '0';
		memcpy(base + 1, in->seq, (size_t) rl);
		memcpy(base + 1 + rl, in->qual, (size_t) rl);
		base[rec] = '0';
		for (int i = 0; i < rl; i++) {
			base[rec + 1 + i] = comp_lut[(uint8_t) in->seq[rl - 1 - i]];
			base[rec + 1 + rl + i] = (uint8_t) in->qual[rl - 1 - i];
		}
		for (int j = 0; j < 2; j++) {
			r->pair_id[g + j] = pid;
			r->read_num[g + j] = (uint8_t) in->rn;
			r->is_rc[g + j] = (uint8_t) (j ? !in->rev : (in->rev != 0));     /* add_read_info(..., bam_is_rev) then (!bam_is_rev) */
			r->reg_rank[g + j] = (uint32_t) (2 * in->seq_no + (uint64_t) j);
			r->scan_index[g + j] = (uint32_t) ((isp ? 0 : 2 * c->np) + 2 * in->pool_no + (uint64_t) j);
		}
		if (isp) ip += 2; else is += 2;
	}
	sph_free(&ids);
	return 0;
}

/* --in <bam>: get_read_length + extract (bam_read.c:264-446) through bamx; a rank of `--gpus N` runs the same passes and keeps its share */
static int load_bam(const cli* c, collector* col) {
	if (!c->vdj_fasta[0] || !c->v_region[0] || !c->c_region[0]) { fprintf(stderr, "BAM input needs --chain/--ref-dir (or --vdjf, --vr, --cr)\n"); return -1; }
	bamx_reads br;
	bamx_threads(c->threads);                              /* (--t: blocks are inflated ahead of the parser in the sequential passes) */
	if (bamx_extract_filtered(c->in, c->vdj_fasta, c->v_region, c->c_region, col->nranks > 1 ? keep_name : NULL, col, &br)) { fprintf(stderr, "%s\n", bamx_last_error()); return -1; }
	if (br.read_len != br.max_len) {
		fprintf(stderr, "reads of different lengths (%d and %d): the reference lays its pools out with one length\n", br.read_len, br.max_len);
		bamx_free(&br);
		return -1;
	}
	col->v = (read_in*) calloc(br.n + 1, sizeof(read_in));
	col->cap = br.n + 1;
	for (size_t i = 0; i < br.n; i++) {
		const bamx_read* x = &br.v[i];
		read_in* o = &col->v[col->n++];
		o->pool = x->pool; o->rn = x->read_num; o->rev = x->is_rev; o->seq_no = x->seq_no; o->pool_no = x->pool_no;
		o->name = col_name(col, x->name, strlen(x->name));
		o->seq = col_str(col, x->seq, (size_t) br.read_len);         /* (read_len characters, NUL padded by the extraction if the record was shorter) */
		o->qual = col_str(col, x->qual, (size_t) br.read_len);
	}
	col->np = br.n_primary_reads; col->ns = br.n_secondary_reads;
	col->max_len = br.max_len;
	bamx_free(&br);
	return 0;
}

/* the extracted read pool as text (see the head of this file): every process reads the whole file once and keeps its share.
 * Six whitespace-separated fields per read (fscanf's "%s %s %d %d %s %s": any run of blanks or newlines separates), a few MB at a time. */
static int is_ws(char ch) { return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v'; }
static int load_text(const char* path, collector* col) {
	FILE* fp = fopen(path, "r");
	const size_t CH = (size_t) 8 << 20;
	char* buf = (char*) malloc(CH + 1);
	if (!fp || !buf) { fprintf(stderr, "cannot open %s\n", path); if (fp) fclose(fp); free(buf); return -1; }
	size_t have = 0;
	int rc = 0, eof = 0, stop = 0;
	while (!stop && !rc) {
		if (!eof) {
			const size_t got = fread(buf + have, 1, CH - have, fp);
			have += got;
			if (got == 0) eof = 1;
		}
		buf[have] = 0;
		/* whole reads only: up to the last whitespace that ends a sixth field (at the end of the file: everything) */
		size_t at = 0;
		for (;;) {
			char* f[6];
			size_t fl[6];
			size_t p = at;
			int k = 0;
			for (; k < 6; k++) {
				while (p < have && is_ws(buf[p])) p++;
				if (p >= have) break;
				const size_t a0 = p;
				while (p < have && !is_ws(buf[p])) p++;
				if (p >= have && !eof) break;                       /* (the field may go on in the next piece) */
				f[k] = buf + a0; fl[k] = p - a0;
			}
			if (k < 6) { if (eof) stop = 1; break; }
			/* numbers as fscanf's %d reads them; a malformed line ends the input there, like the reference's format string would */
			char *e1, *e2;
			const char c2 = f[2][fl[2]], c3 = f[3][fl[3]];
			f[2][fl[2]] = 0; f[3][fl[3]] = 0;
			const long rn = strtol(f[2], &e1, 10), rv = strtol(f[3], &e2, 10);
			const int bad = *e1 || *e2 || fl[0] > 7 || fl[1] > 511 || fl[4] > 1023 || fl[5] > 1023;
			f[2][fl[2]] = c2; f[3][fl[3]] = c3;
			if (bad) { stop = 1; break; }
			f[1][fl[1]] = 0;                                        /* (the separator after the name: the name is hashed as a C string) */
			if ((rc = collect_n(col, f[0][0] == 'P' ? 'P' : 'S', f[1], fl[1], (int) rn, (int) rv, f[4], fl[4], f[5], fl[5])) != 0) { fprintf(stderr, "out of memory\n"); break; }
			at = p;
		}
		if (eof) break;
		memmove(buf, buf + at, have - at);
		have -= at;
		if (have == CH) { fprintf(stderr, "%s: a read of more than %zu bytes\n", path, CH); rc = -1; }
	}
	fclose(fp);
	free(buf);
	return rc;
}

static int load_reads(const cli* c, reads_t* r, int rank, int nranks, int as_share) {
	const int isbam = bamx_is_bam(c->in);
	if (isbam < 0) { fprintf(stderr, "cannot open %s\n", c->in); return -1; }
	collector col;
	memset(&col, 0, sizeof col);
	col.rank = rank; col.nranks = nranks; col.max_len = -1; col.one_block = as_share;
	int rc = isbam ? load_bam(c, &col) : load_text(c->in, &col);
	if (!rc && col.max_len <= 0) { fprintf(stderr, "Error retrieving read length from: %s\n", c->in); rc = -1; }
	if (!rc) rc = build_reads(&col, r);
	if (!rc) { r->name_blocks = col.name_strs; col.name_strs = NULL; }      /* (the names point into them) */
	collector_free(&col);
	return rc;
}

/* load_kmers, vj_filter.c:56-68 */
static int load_codes(const char* path, int max_dist, uint32_t** out, size_t* n) {
	FILE* in = fopen(path, "r");
	if (!in) { fprintf(stderr, "cannot open %s\n", path); return -1; }
	size_t cap = 1024, cnt = 0;
	uint32_t* v = (uint32_t*) malloc(cap * 4);
	unsigned long kmer;
	int freq;
	while (fscanf(in, "%lu\t%d\n", &kmer, &freq) == 2) {
		if (freq <= max_dist) {
			if (cnt == cap) { cap *= 2; v = (uint32_t*) realloc(v, cap * 4); }
			v[cnt++] = (uint32_t) kmer;
		}
	}
	fclose(in);
	*out = v;
	*n = cnt;
	return 0;
}

static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*) a, y = *(const uint32_t*) b; return x < y ? -1 : x > y; }

/* score_seq_init's reader (seq_score.c:50-70): fgets chunks of 999,999 chars, header lines skipped */
static int load_vregion(const char* path, char*** lines, size_t* n) {
	FILE* fp = fopen(path, "r");
	if (!fp) { fprintf(stderr, "cannot open %s\n", path); return -1; }
	char* buf = (char*) malloc(1000000);
	size_t cap = 4, cnt = 0;
	char** v = (char**) malloc(cap * sizeof(char*));
	while (fgets(buf, 1000000, fp) != NULL) {
		if (buf[0] == '>') continue;
		size_t l = strlen(buf);
		if (l) buf[l - 1] = '\0';                        /* "Get rid of newline" */
		if (cnt == cap) { cap *= 2; v = (char**) realloc(v, cap * sizeof(char*)); }
		v[cnt++] = strdup(buf);
	}
	fclose(fp);
	free(buf);
	*lines = v;
	*n = cnt;
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* hooks onto libvdjx                                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { vdjx_ctx* gx; const reads_t* r; const vdjh_params* p; vdjx_mgpu* mg; } hook_ud;

static int h_root_score(void* ud, const char* kmers, size_t n, int k, int thr, uint8_t* out) {
	hook_ud* u = (hook_ud*) ud;
	int rc = vdjx_root_score(u->gx, kmers, n, k, thr, out);
	if (rc) fprintf(stderr, "vdjx_root_score: %s\n", vdjx_last_error());
	return rc;
}

static int h_window_score(void* ud, const char* windows, size_t n, int len, uint8_t* valid) {
	hook_ud* u = (hook_ud*) ud;
	const vdjh_params* p = u->p;
	vdjx_cov_params cp = {p->eval_start, p->eval_stop, p->filter_read_span, p->filter_mate_span, p->insert_len, p->insert_len, p->read_filter_floor};
	if (u->mg) {                              /* `--gpus N`: every rank maps the windows against its share of the pool */
		const int rcm = vdjx_mgpu_window_score(u->mg, u->gx, windows, n, len, &cp, valid);
		if (rcm) fprintf(stderr, "vdjx_mgpu_window_score: %s\n", vdjx_mgpu_last_error());
		return rcm;
	}
	uint32_t* np = (uint32_t*) malloc((n + 1) * 4);
	int rc = vdjx_window_score(u->gx, windows, n, len, &cp, valid, np);
	if (rc) fprintf(stderr, "vdjx_window_score: %s\n", vdjx_last_error());
	free(np);
	return rc;
}

static const uint8_t* rec_ptr(const reads_t* r, uint32_t rec) {
	const size_t sz = 2 * (size_t) r->rl + 1;
	return rec < r->n_primary ? r->primary + rec * sz : r->secondary + (rec - r->n_primary) * sz;
}

/* output_mapping, quick_map3.c:152-181: mapped and formatted on the device (vdjx_sam_text); VDJX_SAM_HOST=1 formats the pairs of
 * vdjx_map_emit here instead (the same bytes; kept as the cross-check of the formatting kernel) */
static int h_sam_body(void* ud, const char* const* ids, const char* contigs, size_t n, int len, FILE* out) {
	hook_ud* u = (hook_ud*) ud;
	const reads_t* r = u->r;
	if (u->mg || !getenv("VDJX_SAM_HOST")) {
		uint32_t* id_off = (uint32_t*) calloc(n + 1, 4);
		size_t tot = 0;
		for (size_t c = 0; c < n; c++) { tot += strlen(ids[c]); id_off[c + 1] = (uint32_t) tot; }
		char* cat = (char*) malloc(tot + 1);
		for (size_t c = 0; c < n; c++) memcpy(cat + id_off[c], ids[c], id_off[c + 1] - id_off[c]);
		const char* text = NULL;
		uint64_t nb = 0;
		int rc = u->mg ? vdjx_mgpu_sam_body(u->mg, u->gx, contigs, n, len, cat, id_off, &text, &nb) : vdjx_sam_text(u->gx, contigs, n, len, cat, id_off, &text, &nb);
		free(cat);
		free(id_off);
		if (rc) { fprintf(stderr, "%s: %s\n", u->mg ? "vdjx_mgpu_sam_body" : "vdjx_sam_text", u->mg ? vdjx_mgpu_last_error() : vdjx_last_error()); return rc; }
		if (nb && fwrite(text, 1, (size_t) nb, out) != (size_t) nb) { fprintf(stderr, "short write of the SAM records\n"); return -1; }
		return 0;
	}
	uint64_t* offs = (uint64_t*) calloc(n + 1, 8);
	int rc = vdjx_map_emit(u->gx, contigs, n, len, offs, NULL);
	vdjx_pair* pairs = NULL;
	if (!rc) {
		pairs = (vdjx_pair*) malloc((offs[n] + 1) * sizeof(vdjx_pair));
		rc = vdjx_map_emit(u->gx, contigs, n, len, offs, pairs);
	}
	if (rc) { fprintf(stderr, "vdjx_map_emit: %s\n", vdjx_last_error()); free(offs); free(pairs); return rc; }
	const int rl = r->rl;
	for (size_t c = 0; c < n; c++) {
		char cid[256];
		strncpy(cid, ids[c], 255);
		cid[255] = 0;
		for (uint64_t i = offs[c]; i < offs[c + 1]; i++) {
			const vdjx_pair* q = &pairs[i];
			const char* name = r->names[q->pair_id];
			if (name[0] == '@') name++;
			const int flag1 = 0x1 | 0x2 | (q->rc1 ? 0x10 : 0x20) | 0x40;
			const int flag2 = 0x1 | 0x2 | (q->rc2 ? 0x10 : 0x20) | 0x80;
			const uint8_t* r1 = rec_ptr(r, q->rec1);
			const uint8_t* r2 = rec_ptr(r, q->rec2);
			fprintf(out, "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%.*s\t%.*s\n", name, flag1, cid, (int) q->pos1, rl, (int) q->pos2, (int) q->insert,
			        rl, (const char*) r1 + 1, rl, (const char*) r1 + 1 + rl);
			fprintf(out, "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%.*s\t%.*s\n", name, flag2, cid, (int) q->pos2, rl, (int) q->pos1, (int) q->insert,
			        rl, (const char*) r2 + 1, rl, (const char*) r2 + 1 + rl);
		}
	}
	free(offs);
	free(pairs);
	return 0;
}

static void h_status(void* ud, const char* desc) { (void) ud; status(desc); }

#define VX(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, vdjx_last_error()); return 1; } } while (0)

/* ---- `--gpus N`: the other ranks are child processes.  A rank that fails must never leave the others waiting inside a collective
 * (RCCL has no timeout of its own): a child dies with its parent (PR_SET_PDEATHSIG), the parent takes every child down when it
 * leaves for whatever reason (atexit), and a child that ends badly ends the whole run at once (SIGCHLD) instead of hanging it. */
static pid_t g_kids[256];
static volatile sig_atomic_t g_kid_state[256];      /* 0 running, 1 ended well, 2 reaped after being killed */
static int g_nkids = 0;

static volatile sig_atomic_t g_leaving = 0;       /* the parent is taking the ranks down itself: their end is not a failure to report */

static void kill_kids(void) {
	/* a deliberate teardown (rank 0 failed on its own, or is done): no SIGCHLD handler run from here on, so the exit status the
	 * caller chose survives and nobody blames a GPU rank for a host-side error */
	sigset_t chld;
	sigemptyset(&chld);
	sigaddset(&chld, SIGCHLD);
	sigprocmask(SIG_BLOCK, &chld, NULL);
	g_leaving = 1;
	for (int r = 1; r <= g_nkids; r++) if (g_kid_state[r] == 0) (void) kill(g_kids[r], SIGTERM);
	for (int r = 1; r <= g_nkids; r++) if (g_kid_state[r] == 0) { int st_; (void) waitpid(g_kids[r], &st_, 0); g_kid_state[r] = 2; }
}

static void on_sigchld(int sig) {
	(void) sig;
	if (g_leaving) return;
	for (int r = 1; r <= g_nkids; r++) {
		if (g_kid_state[r] != 0) continue;
		int st_ = 0;
		const pid_t p = waitpid(g_kids[r], &st_, WNOHANG);
		if (p != g_kids[r]) continue;
		if (WIFEXITED(st_) && WEXITSTATUS(st_) == 0) { g_kid_state[r] = 1; continue; }
		g_kid_state[r] = 2;
		static const char msg[] = "vdjer: a GPU rank failed; stopping the others\n";
		(void) !write(2, msg, sizeof msg - 1);
		for (int q = 1; q <= g_nkids; q++) if (g_kid_state[q] == 0) (void) kill(g_kids[q], SIGKILL);
		for (int q = 1; q <= g_nkids; q++) if (g_kid_state[q] == 0) { int s2; (void) waitpid(g_kids[q], &s2, 0); }      /* (no zombies left to whoever inherits them) */
		_exit(1);
	}
}

static void report_share(int rank, const reads_t* rd) {
	if (!getenv("VDJX_REPORT_SHARE")) return;
	struct rusage ru;
	getrusage(RUSAGE_SELF, &ru);
	fprintf(stderr, "share\trank\t%d\trecords\t%zu\tof\t%llu\tpairs\t%u\thost_pool_bytes\t%zu\tmaxrss_kb\t%ld\n", rank, rd->n_primary + rd->n_secondary,
	        (unsigned long long) rd->total_records, rd->n_pairs, (rd->n_primary + rd->n_secondary) * (2 * (size_t) rd->rl + 1), ru.ru_maxrss);
}

int main(int argc, char** argv) {
	t_start = t_prev = time(NULL);
	cli c;
	if (parse(argc, argv, &c)) return 255;                  /* the reference exits with -1 */
	/* --gpus N: one process per GPU.  The parent is rank 0; ranks 1..N-1 are forked BEFORE anything touches a GPU.  Every rank reads
	 * the input and keeps ITS share of the pool (the pairs whose name hashes to it), the ranks deal the k-mer build's slices out to
	 * each other on the devices, build the graph together, and the others then serve rank 0's scorer calls (vdjx_mgpu.h) while it
	 * runs the serial traversal. */
	if (getenv("VDJX_DUMP_SHARE")) {            /* (test hook, no GPU: "rank,nranks" -> that rank's share: its records on stdout, their places on stderr) */
		int rk = 0, nr = 1;
		reads_t sh;
		if (sscanf(getenv("VDJX_DUMP_SHARE"), "%d,%d", &rk, &nr) != 2 || nr < 1 || rk < 0 || rk >= nr || load_reads(&c, &sh, rk, nr, nr > 1)) return 255;
		const size_t R = sh.n_primary + sh.n_secondary;
		fprintf(stderr, "share\t%d\t%d\trl\t%d\trecords\t%zu\ttotal\t%llu\tpairs\t%u\n", rk, nr, sh.rl, R, (unsigned long long) sh.total_records, sh.n_pairs);
		for (size_t i = 0; i < R; i++) fprintf(stderr, "rec\t%u\t%u\t%u\t%d\t%d\t%s\n", sh.scan_index[i], sh.reg_rank[i], sh.pair_id[i], sh.read_num[i], sh.is_rc[i], sh.names[sh.pair_id[i]]);
		fwrite(sh.primary, 2 * (size_t) sh.rl + 1, sh.n_primary, stdout);
		if (sh.n_secondary) fwrite(sh.secondary, 2 * (size_t) sh.rl + 1, sh.n_secondary, stdout);
		return 0;
	}
	int rank = 0;
	if (c.gpus < 1 || c.gpus > 256) { fprintf(stderr, "--gpus must be in [1,256]\n"); return 255; }
	const int use_mgpu = c.gpus > 1 || getenv("VDJX_FORCE_MGPU") != NULL;      /* (the variable: a one-rank run of the same code path) */
	/* ranks that share ONE device (the multi-rank tests on a one-GPU box; RCCL refuses two ranks on a device): bytes move through the
	 * host.  Otherwise rank r drives GPU r and the bytes move by RCCL over xGMI. */
	const int one_device = getenv("VDJX_MGPU_ONE_DEVICE") != NULL;
	const char* transport = one_device ? "host" : (getenv("VDJX_MGPU_TRANSPORT") ? getenv("VDJX_MGPU_TRANSPORT") : "rccl");
	int* fds = NULL;
	if (c.gpus > 1) {
		fds = (int*) malloc((size_t) c.gpus * (size_t) c.gpus * sizeof(int));
		if (vdjx_comm_sockets(c.gpus, !strcmp(transport, "host"), fds)) { fprintf(stderr, "%s\n", vdjx_comm_last_error()); return 255; }
		struct sigaction sa;
		memset(&sa, 0, sizeof sa);
		sa.sa_handler = on_sigchld;
		sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
		sigaction(SIGCHLD, &sa, NULL);
		atexit(kill_kids);
	}
	const pid_t parent = getpid();
	sigset_t chld, before;
	sigemptyset(&chld);
	sigaddset(&chld, SIGCHLD);
	sigprocmask(SIG_BLOCK, &chld, &before);       /* (no rank's end is handled before every rank is on the list) */
	for (int r = 1; r < c.gpus; r++) {
		fflush(stdout); fflush(stderr);
		const pid_t pid = fork();
		if (pid < 0) { perror("fork"); return 255; }
		if (pid == 0) {
			rank = r; g_rank = r; g_nkids = 0;
			signal(SIGCHLD, SIG_DFL);
			sigprocmask(SIG_SETMASK, &before, NULL);
			(void) prctl(PR_SET_PDEATHSIG, SIGKILL);
			if (getppid() != parent) _exit(1);             /* (the parent was gone before the request took effect) */
			break;
		}
		g_kids[r] = pid;
		g_nkids = r;
	}
	if (fds) vdjx_comm_sockets_keep(c.gpus, rank, fds);
	if (rank == 0) sigprocmask(SIG_SETMASK, &before, NULL);
	if (rank == 0) status("START");
	/* the GPU context starts on a thread of its own while this one reads the input: bringing up the HIP runtime takes a few hundred
	 * milliseconds, about what parsing a few hundred thousand pairs takes.  (After the forks: no rank touches a GPU before it is a
	 * process of its own, and nothing is ever exec'ed from here.) */
	const int device = one_device ? 0 : rank;              /* rank r drives GPU r */
	init_job ij;
	memset(&ij, 0, sizeof ij);
	ij.device = device;
	pthread_t init_th;
	const int init_threaded = getenv("VDJX_INIT_SERIAL") == NULL && pthread_create(&init_th, NULL, init_thread, &ij) == 0;
	reads_t rd;
	/* an input error between here and the join below must not run exit() and its handlers while the HIP runtime is still coming up on the
	 * other thread: wait for that thread, then leave with the documented code */
#define BAIL(code) do { if (init_threaded) pthread_join(init_th, NULL); return (code); } while (0)
	if (load_reads(&c, &rd, rank, c.gpus, use_mgpu)) BAIL(255);
	c.hp.read_length = rd.rl;
	c.hp.threads = c.threads;
	if (rank == 0) fprintf(stderr, "read length:\t%d\n", rd.rl);

	uint32_t *vc = NULL, *jc = NULL;
	size_t nv = 0, nj = 0;
	if (load_codes(c.v_anchors, c.anchor_mismatches, &vc, &nv) || load_codes(c.j_anchors, c.anchor_mismatches, &jc, &nj)) BAIL(255);
	qsort(vc, nv, 4, cmp_u32);
	qsort(jc, nj, 4, cmp_u32);
	char** vlines = NULL;
	size_t nvl = 0;
	if (load_vregion(c.source_sim_file, &vlines, &nvl)) BAIL(255);
#undef BAIL

	vdjx_ctx* gx = NULL;
	if (getenv("VDJX_TIMES")) status("(inputs read)");
	if (init_threaded) pthread_join(init_th, NULL);
	else init_thread(&ij);
	if (ij.rc) { fprintf(stderr, "vdjx_init: %s\n", ij.err); return 1; }
	gx = ij.gx;
	if (getenv("VDJX_TIMES")) status("(vdjx_init done)");
	vdjx_mgpu* mg = NULL;
	if (use_mgpu) {
		unsigned char id[VDJX_COMM_ID_BYTES];
		memset(id, 0, sizeof id);
		/* the communicator's bootstrap id: made by rank 0 (RCCL only; zeros otherwise), handed to every rank over its control socket --
		 * whatever the transport, so that the hand-over itself runs in the one-device tests too */
		const int* my_fds = fds ? fds + (size_t) rank * (size_t) c.gpus : NULL;
		if (rank == 0) {
			if (!strcmp(transport, "rccl") && vdjx_comm_unique_id(id)) { fprintf(stderr, "%s\n", vdjx_comm_last_error()); return 1; }
			for (int r = 1; r < c.gpus; r++)
				for (size_t at = 0; at < sizeof id;) {
					const ssize_t k = write(my_fds[r], id + at, sizeof id - at);
					if (k <= 0) { if (k < 0 && errno == EINTR) continue; perror("write"); return 1; }
					at += (size_t) k;
				}
		} else {
			for (size_t at = 0; at < sizeof id;) {
				const ssize_t k = read(my_fds[0], id + at, sizeof id - at);
				if (k <= 0) { if (k < 0 && errno == EINTR) continue; fprintf(stderr, "rank %d: no communicator id from rank 0\n", rank); return 1; }
				at += (size_t) k;
			}
		}
		vdjx_comm* cm = NULL;
		if (vdjx_comm_init(transport, rank, c.gpus, device, my_fds, id, &cm)) { fprintf(stderr, "rank %d: %s\n", rank, vdjx_comm_last_error()); return 1; }
		if (vdjx_mgpu_init(cm, device, &mg)) { fprintf(stderr, "rank %d: %s\n", rank, vdjx_mgpu_last_error()); return 1; }
	}
	VX(vdjx_anchor_sets_load(gx, vc, nv, jc, nj));
	VX(vdjx_vregion_load(gx, (const char* const*) vlines, nvl, c.hp.vregion_kmer_size));
	status("POST_VJF_INIT");

	vdjx_pool* px = NULL;
	if (!use_mgpu) {
		VX(vdjx_pool_load(gx, rd.primary, rd.n_primary, rd.secondary, rd.n_secondary, rd.rl, &px));
		if (getenv("VDJX_TIMES")) status("(pool loaded)");
		/* the read index (add_read_info, quick_map3.c:126-149) is begun on the library's index stream and built BESIDE the k-mer build below:
		 * both only read the packed pool, and nothing needs the index before the traversal's first window (A2:841).  It is ended after
		 * the build (VDJX_INDEX_SERIAL=1: built here, waited for, as until round 5) */
		if (getenv("VDJX_INDEX_SERIAL")) VX(vdjx_read_index_build(gx, px, rd.pair_id, rd.read_num, rd.is_rc, rd.reg_rank, rd.n_pairs));
		else VX(vdjx_read_index_build_begin(gx, px, rd.pair_id, rd.read_num, rd.is_rc, rd.reg_rank, rd.n_pairs));
		if (getenv("VDJX_TIMES")) status(getenv("VDJX_INDEX_SERIAL") ? "(read index built)" : "(read index begun)");
	} else if (vdjx_mgpu_load(mg, gx, rd.primary, rd.n_primary, rd.rl, rd.scan_index, rd.pair_id, rd.read_num, rd.is_rc, rd.reg_rank, rd.n_pairs, rd.total_records)) {
		fprintf(stderr, "%s\n", vdjx_mgpu_last_error());
		return 1;
	}
	if (vdjx_stat(gx, "pool_other_bases"))
		fprintf(stderr, "warning: %llu bases other than ACGTN in the reads%s are treated as N (the reference would carry them inside k-mers)\n",
		        (unsigned long long) vdjx_stat(gx, "pool_other_bases"), use_mgpu ? " of this rank" : "");
	{	/* the read names by pair id, for the SAM records formatted on the device */
		uint64_t* noff = (uint64_t*) calloc((size_t) rd.n_pairs + 1, 8);
		for (uint32_t i = 0; i < rd.n_pairs; i++) noff[i + 1] = noff[i] + strlen(rd.names[i]);
		char* cat = (char*) malloc((size_t) noff[rd.n_pairs] + 1);
		for (uint32_t i = 0; i < rd.n_pairs; i++) memcpy(cat + noff[i], rd.names[i], (size_t) (noff[i + 1] - noff[i]));
		const int rcn = vdjx_sam_names_load(gx, cat, noff, rd.n_pairs);
		free(cat);
		free(noff);
		if (rcn) { fprintf(stderr, "vdjx_sam_names_load: %s\n", vdjx_last_error()); return 1; }
	}
	report_share(rank, &rd);
	if (use_mgpu) {                              /* the records live on the device now (the share's quality characters included) */
		free(rd.primary); rd.primary = NULL;
		free(rd.pair_id); free(rd.reg_rank); free(rd.scan_index); free(rd.read_num); free(rd.is_rc);
		rd.pair_id = rd.reg_rank = rd.scan_index = NULL; rd.read_num = rd.is_rc = NULL;
	}
	status("POST_READ_EXTRACT");
	if (rank == 0) fprintf(stderr, "Assembling...\n");
	vdjx_graph* gg = NULL;
	status("PRE_PRE_GRAPH1");                  /* A2:1387 */
	if (!use_mgpu) {
		VX(vdjx_kmer_build(gx, px, c.hp.k, c.hp.min_node_freq, c.hp.min_base_quality, &gg));
		if (getenv("VDJX_TIMES")) status("(k-mer build done)");
		VX(vdjx_read_index_build_end(gx));
		if (getenv("VDJX_TIMES")) status("(read index ended)");
	} else {
		if (vdjx_mgpu_kmer_build(mg, gx, c.hp.k, c.hp.min_node_freq, c.hp.min_base_quality, &gg)) { fprintf(stderr, "%s\n", vdjx_mgpu_last_error()); return 1; }
		if (rank == 0) fprintf(stderr, "k-mer table sharded over %d GPUs (%s): %llu bytes sent by rank 0 so far\n", c.gpus, transport, (unsigned long long) vdjx_mgpu_bytes_sent(mg));
		if (rank != 0) {                        /* the graph is identical on every rank; the serial traversal runs on rank 0, which calls on the others' scorers */
			vdjx_graph_free(gg);
			const int rcs = vdjx_mgpu_serve(mg, gx);
			if (rcs) fprintf(stderr, "%s\n", vdjx_mgpu_last_error());
			vdjx_mgpu_free(mg);
			vdjx_shutdown(gx);
			return rcs ? 1 : 0;
		}
	}
	const size_t n = vdjx_graph_nodes(gg);
	status("PRE_PRE_GRAPH2");                  /* A2:1389-1409: one device call made the table, the prune and the graph */
	status("POST_PRE_GRAPH1");
	fprintf(stderr, "Pre Num nodes: %zu\npre nodes after pruning: %zu\n", vdjx_graph_pre_nodes(gg), n);
	status("POST_PRUNE_PRE_GRAPH1");
	status("POST_BUILD_GRAPH1");
	fprintf(stderr, "Num nodes: %zu\n", n);
	status("POST_BUILD_GRAPH2");
	vdjh_graph hg;
	memset(&hg, 0, sizeof hg);
	hg.n = n;
	hg.k = c.hp.k;
	char* kmers = (char*) malloc(n * (size_t) c.hp.k + 1);
	uint32_t* freq = (uint32_t*) malloc((n + 1) * 4);
	uint8_t *hv = (uint8_t*) malloc(n + 1), *hj = (uint8_t*) malloc(n + 1), *td = (uint8_t*) malloc(n + 1), *fd = (uint8_t*) malloc(n + 1);
	uint32_t *ti = (uint32_t*) malloc((n + 1) * 16), *fi = (uint32_t*) malloc((n + 1) * 16);
	VX(vdjx_graph_export(gg, NULL, NULL, freq, hv, hj, td, ti, fd, fi, kmers));
	hg.kmers = kmers; hg.freq = freq; hg.has_v = hv; hg.has_j = hj; hg.to_deg = td; hg.to_ids = ti; hg.from_deg = fd; hg.from_ids = fi;

	hook_ud ud = {gx, &rd, &c.hp, mg};
	vdjh_hooks hk = {&ud, h_root_score, h_window_score, h_sam_body, vc, nv, jc, nj, h_status};
	vdjh_stats st;
	if (vdjh_assemble(&c.hp, &hg, &hk, "vdj_contigs.fa", "vdjer.dot", stdout, &st)) {
		fprintf(stderr, "%s\n", vdjh_last_error());
		return 1;
	}
	fprintf(stderr, "num root nodes: %zu\nProcessed roots: %zu\ncontig_candidates: %zu\nwindows scored: %zu valid: %zu\ncontigs: %zu\n",
	        st.n_roots, st.n_roots_accepted, st.n_contig_candidates, st.n_windows_scored, st.n_windows_valid, st.n_contigs_out);
	if (mg) {
		fprintf(stderr, "%llu bytes sent by rank 0 in all\n", (unsigned long long) vdjx_mgpu_bytes_sent(mg));
		if (vdjx_mgpu_finish(mg)) { fprintf(stderr, "%s\n", vdjx_mgpu_last_error()); return 1; }
	}
	for (int r = 1; r < c.gpus; r++) {                    /* the ranks have been released: they end, well */
		while (g_kid_state[r] == 0) {
			int st_ = 0;
			const pid_t p = waitpid(g_kids[r], &st_, 0);
			if (p == g_kids[r]) {
				g_kid_state[r] = (WIFEXITED(st_) && WEXITSTATUS(st_) == 0) ? 1 : 2;
				if (g_kid_state[r] == 2) fprintf(stderr, "rank %d ended with %s %d\n", r, WIFSIGNALED(st_) ? "signal" : "status", WIFSIGNALED(st_) ? WTERMSIG(st_) : WEXITSTATUS(st_));
			}
			else if (p < 0 && errno != EINTR) {
				/* reaped by the SIGCHLD handler -- which may be running on ANOTHER thread of this process (the GPU runtime has some,
				 * and a signal goes to any thread that does not block it): between its waitpid and its note of the outcome this
				 * thread can get here.  Its note comes within microseconds; a rank it found to have failed ends the process there. */
				for (int spin = 0; spin < 5000 && g_kid_state[r] == 0; spin++) usleep(1000);
				break;
			}
		}
		if (g_kid_state[r] != 1) { fprintf(stderr, "rank %d failed\n", r); return 1; }
	}
	status("FINIS");
	fflush(stdout);
	fflush(stderr);
	/* the outputs are written; the orderly teardown of a context that holds gigabytes (unmapping the workspaces piece by piece, the
	 * runtime's own exit handlers) took 160 ms of a 530 ms run at 400 k pairs -- the process ends instead and the driver reclaims what
	 * it held, as for any process.  VDJX_CLEAN_EXIT=1 keeps the teardown (leak checks). */
	if (!getenv("VDJX_CLEAN_EXIT")) _exit(0);
	vdjx_graph_free(gg);
	if (mg) vdjx_mgpu_free(mg);
	vdjx_pool_free(px);
	vdjx_shutdown(gx);
	return 0;
}
