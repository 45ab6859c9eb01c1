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
#include <sys/prctl.h>
#include <signal.h>
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

static time_t t_start, t_prev;
static int g_rank;                             /* --gpus N: only rank 0 writes the stage log */
static void status(const char* desc) {         /* status.c:22-32 without the /proc dumps */
	if (g_rank) return;
	time_t now = time(NULL);
	fprintf(stderr, "ELAPSED_SECS\t%s\t%ld\t%ld\n", desc, (long) (now - t_start), (long) (now - t_prev));
	t_prev = now;
}

/* ------------------------------------------------------------------------------------------ */
/* inputs                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
	int rl;
	uint8_t *primary, *secondary;
	size_t n_primary, n_secondary;
	uint32_t *pair_id, *reg_rank;       /* scan order: primary records then secondary */
	uint8_t *read_num, *is_rc;
	char** names;
	uint32_t n_pairs;
} reads_t;

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; } }

/* one extracted read before it becomes two pool records */
typedef struct { char pool; const char* name; int rn, rev; const char* seq; const char* qual; } read_in;

/* add_to_buffer (bam_read.c:206-244) for every read, in order: forward record, reverse-complement record, both registered */
static int build_reads(const read_in* in, size_t n, int rl, reads_t* r) {
	size_t np = 0, ns = 0;
	for (size_t i = 0; i < n; i++) { if (in[i].pool == 'P') np++; else ns++; }
	memset(r, 0, sizeof *r);
	r->rl = rl;
	const size_t rec = 2 * (size_t) rl + 1;
	r->n_primary = 2 * np;
	r->n_secondary = 2 * ns;
	const size_t R = r->n_primary + r->n_secondary;
	r->primary = (uint8_t*) calloc(r->n_primary * rec + 1, 1);
	r->secondary = (uint8_t*) calloc(r->n_secondary * rec + 1, 1);
	r->pair_id = (uint32_t*) calloc(R + 1, 4);
	r->reg_rank = (uint32_t*) calloc(R + 1, 4);
	r->read_num = (uint8_t*) calloc(R + 1, 1);
	r->is_rc = (uint8_t*) calloc(R + 1, 1);
	r->names = (char**) calloc(np + ns + 1, sizeof(char*));
	sph_table ids;                       /* read name -> pair id */
	sph_init(&ids, 0, 0);
	size_t ip = 0, is = 0;
	uint32_t reg = 0;
	for (size_t q = 0; q < n; q++) {
		const char *name = in[q].name, *seq = in[q].seq, *qual = in[q].qual;
		if ((int) strlen(seq) != rl || (int) strlen(qual) != rl) { fprintf(stderr, "read %s: length != %d\n", name, rl); sph_free(&ids); return -1; }
		size_t b = sph_find(&ids, name);
		uint32_t pid;
		if (b == (size_t) -1) {
			pid = r->n_pairs++;
			r->names[pid] = strdup(name);
			sph_map_put(&ids, r->names[pid], (void*) (uintptr_t) (pid + 1), NULL);
		} else {
			pid = (uint32_t) (uintptr_t) ids.b[b].val - 1;
		}
		const int isp = in[q].pool == 'P';
		uint8_t* base = isp ? r->primary + ip * rec : r->secondary + is * rec;
		const size_t g = isp ? ip : r->n_primary + is;
		base[0] = '0';
		memcpy(base + 1, seq, (size_t) rl);
		memcpy(base + 1 + rl, qual, (size_t) rl);
		base[rec] = '0';
		for (int i = 0; i < rl; i++) {
			base[rec + 1 + i] = (uint8_t) comp(seq[rl - 1 - i]);
			base[rec + 1 + rl + i] = (uint8_t) qual[rl - 1 - i];
		}
		for (int j = 0; j < 2; j++) {
			r->pair_id[g + j] = pid;
			r->read_num[g + j] = (uint8_t) in[q].rn;
			r->is_rc[g + j] = (uint8_t) (j ? !in[q].rev : (in[q].rev != 0));     /* add_read_info(..., bam_is_rev) then (!bam_is_rev) */
			r->reg_rank[g + j] = reg++;
		}
		if (isp) ip += 2; else is += 2;
	}
	sph_free(&ids);
	return 0;
}

/* --in <bam>: get_read_length + extract (bam_read.c:264-446) through bamx */
static int load_bam(const cli* c, reads_t* r) {
	if (!c->vdj_fasta[0] || !c->v_region[0] || !c->c_region[0]) { fprintf(stderr, "BAM input needs --chain/--ref-dir (or --vdjf, --vr, --cr)\n"); return -1; }
	bamx_reads br;
	if (bamx_extract(c->in, c->vdj_fasta, c->v_region, c->c_region, &br)) { fprintf(stderr, "%s\n", bamx_last_error()); return -1; }
	if (br.read_len != br.max_len) {
		fprintf(stderr, "reads of different lengths (%d and %d): the reference lays its pools out with one length\n", br.read_len, br.max_len);
		bamx_free(&br);
		return -1;
	}
	read_in* in = (read_in*) calloc(br.n + 1, sizeof(read_in));
	for (size_t i = 0; i < br.n; i++) {
		const bamx_read* x = &br.v[i];
		in[i].pool = x->pool; in[i].name = x->name; in[i].rn = x->read_num; in[i].rev = x->is_rev; in[i].seq = x->seq; in[i].qual = x->qual;
	}
	const int rc = build_reads(in, br.n, br.max_len, r);
	free(in);
	bamx_free(&br);
	return rc;
}

/* A rank other than 0 of `--gpus N` only ever uses ITS records [rank*S, (rank+1)*S) of the scan order (primary pool, then secondary;
 * two records per read): the text file is read twice -- once to count the reads of the two pools, once to keep the rank's own --
 * instead of every process holding the whole pool (40 GB each at 100 M pairs).  The slice comes back as the primary pool of `r`
 * (n_primary records, nothing else filled in); *stride = S. */
static int load_slice_text(const char* path, int rank, int nranks, reads_t* r, size_t* stride) {
	FILE* fp = fopen(path, "r");
	if (!fp) { fprintf(stderr, "cannot open %s\n", path); return -1; }
	char pool[8], name[512];
	static char seq[1024], qual[1024];
	int rn, rev, rl = -1;
	size_t np = 0, ns = 0;
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		const int l = (int) strlen(seq);
		if (l > rl) rl = l;
		if (pool[0] == 'P') np++; else ns++;
	}
	if (rl <= 0) { fclose(fp); fprintf(stderr, "Error retrieving read length from: %s\n", path); return -1; }
	const size_t Rt = 2 * (np + ns), S = (Rt + (size_t) nranks - 1) / (size_t) nranks;
	const size_t a = (size_t) rank * S < Rt ? (size_t) rank * S : Rt, b = a + S < Rt ? a + S : Rt;
	const size_t rec = 2 * (size_t) rl + 1;
	memset(r, 0, sizeof *r);
	r->rl = rl;
	r->n_primary = b - a;
	r->primary = (uint8_t*) calloc((b - a) * rec + 1, 1);
	*stride = S;
	rewind(fp);
	size_t ip = 0, is = 0;
	int bad = 0;
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		const int isp = pool[0] == 'P';
		const size_t g = isp ? ip : 2 * np + is;              /* scan index of the read's first record */
		if (isp) ip += 2; else is += 2;
		if (g + 2 <= a || g >= b) continue;
		if ((int) strlen(seq) != rl || (int) strlen(qual) != rl) { fprintf(stderr, "read %s: length != %d\n", name, rl); bad = 1; break; }
		for (int j = 0; j < 2; j++) {
			if (g + (size_t) j < a || g + (size_t) j >= b) continue;
			uint8_t* base = r->primary + (g + (size_t) j - a) * rec;
			base[0] = '0';
			for (int i = 0; i < rl; i++) {
				base[1 + i] = (uint8_t) (j ? comp(seq[rl - 1 - i]) : seq[i]);
				base[1 + rl + i] = (uint8_t) (j ? qual[rl - 1 - i] : qual[i]);
			}
		}
	}
	fclose(fp);
	return bad ? -1 : 0;
}

static int load_reads(const cli* c, reads_t* r, int rank, int nranks) {
	const char* path = c->in;
	const int isbam = bamx_is_bam(path);
	if (isbam < 0) { fprintf(stderr, "cannot open %s\n", path); return -1; }
	if (isbam) return load_bam(c, r);                        /* (every rank extracts: the BAM passes are not sliced) */
	if (rank > 0 && nranks > 1) {
		size_t S = 0;
		const int rc = load_slice_text(path, rank, nranks, r, &S);
		r->n_pairs = 0;
		r->n_secondary = S;                                   /* (smuggled to main: the slice's stride; a slice has no secondary pool) */
		r->secondary = NULL;
		return rc;
	}
	FILE* fp = fopen(path, "r");
	if (!fp) { fprintf(stderr, "cannot open %s\n", path); return -1; }
	char pool[8], name[512];
	static char seq[1024], qual[1024];
	int rn, rev;
	size_t n = 0, cap = 1024;
	int rl = -1;
	read_in* in = (read_in*) calloc(cap, sizeof(read_in));
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		const int l = (int) strlen(seq);
		if (l > rl) rl = l;                                   /* get_read_length: the maximum */
		if (n == cap) { cap *= 2; in = (read_in*) realloc(in, cap * sizeof(read_in)); }
		in[n].pool = pool[0] == 'P' ? 'P' : 'S';
		in[n].name = strdup(name); in[n].rn = rn; in[n].rev = rev; in[n].seq = strdup(seq); in[n].qual = strdup(qual);
		n++;
	}
	fclose(fp);
	if (rl <= 0) { fprintf(stderr, "Error retrieving read length from: %s\n", path); return -1; }
	const int rc = build_reads(in, n, rl, r);
	for (size_t i = 0; i < n; i++) { free((char*) in[i].name); free((char*) in[i].seq); free((char*) in[i].qual); }
	free(in);
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
typedef struct { vdjx_ctx* gx; const reads_t* r; const vdjh_params* p; } hook_ud;

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
	if (!getenv("VDJX_SAM_HOST")) {
		uint32_t* id_off = (uint32_t*) calloc(n + 1, 4);
		size_t tot = 0;
		for (size_t c = 0; c < n; c++) { tot += strlen(ids[c]); id_off[c + 1] = (uint32_t) tot; }
		char* cat = (char*) malloc(tot + 1);
		for (size_t c = 0; c < n; c++) memcpy(cat + id_off[c], ids[c], id_off[c + 1] - id_off[c]);
		const char* text = NULL;
		uint64_t nb = 0;
		int rc = vdjx_sam_text(u->gx, contigs, n, len, cat, id_off, &text, &nb);
		free(cat);
		free(id_off);
		if (rc) { fprintf(stderr, "vdjx_sam_text: %s\n", vdjx_last_error()); return rc; }
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

/* records [a, b) of the scan order (primary pool, then secondary) as one contiguous block */
static uint8_t* slice_records(const reads_t* r, size_t a, size_t b) {
	const size_t rec = 2 * (size_t) r->rl + 1;
	uint8_t* out = (uint8_t*) malloc((b > a ? b - a : 1) * rec);
	for (size_t i = a; i < b; i++)
		memcpy(out + (i - a) * rec, i < r->n_primary ? r->primary + i * rec : r->secondary + (i - r->n_primary) * rec, rec);
	return out;
}

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

int main(int argc, char** argv) {
	t_start = t_prev = time(NULL);
	cli c;
	if (parse(argc, argv, &c)) return 255;                  /* the reference exits with -1 */
	/* --gpus N: one process per GPU.  The parent is rank 0 and does everything a single-GPU run does; ranks 1..N-1 are forked
	 * BEFORE anything touches a GPU, take part in the sharded k-mer build with their slice of the pool, and leave. */
	if (getenv("VDJX_DUMP_SLICE")) {            /* (test hook, no GPU: "rank,nranks" -> that rank's records of the text input on stdout) */
		int rk = 0, nr = 1;
		reads_t sl;
		size_t S = 0;
		if (sscanf(getenv("VDJX_DUMP_SLICE"), "%d,%d", &rk, &nr) != 2 || load_slice_text(c.in, rk, nr, &sl, &S)) return 255;
		fprintf(stderr, "slice\t%d\t%d\trl\t%d\trecords\t%zu\tstride\t%zu\n", rk, nr, sl.rl, sl.n_primary, S);
		fwrite(sl.primary, 2 * (size_t) sl.rl + 1, sl.n_primary, stdout);
		return 0;
	}
	int rank = 0;
	int id_pipe[256][2];
	if (c.gpus < 1 || c.gpus > 256) { fprintf(stderr, "--gpus must be in [1,256]\n"); return 255; }
	if (c.gpus > 1) {
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
		if (pipe(id_pipe[r])) { perror("pipe"); return 255; }
		fflush(stdout); fflush(stderr);
		const pid_t pid = fork();
		if (pid < 0) { perror("fork"); return 255; }
		if (pid == 0) {
			rank = r; g_rank = r; g_nkids = 0;
			signal(SIGCHLD, SIG_DFL);
			sigprocmask(SIG_SETMASK, &before, NULL);
			(void) prctl(PR_SET_PDEATHSIG, SIGKILL);
			if (getppid() != parent) _exit(1);             /* (the parent was gone before the request took effect) */
			close(id_pipe[r][1]);
			break;
		}
		g_kids[r] = pid;
		g_nkids = r;
		close(id_pipe[r][0]);
	}
	if (rank == 0) sigprocmask(SIG_SETMASK, &before, NULL);
	if (rank == 0) status("START");
	reads_t rd;
	if (load_reads(&c, &rd, rank, c.gpus)) return 255;
	c.hp.read_length = rd.rl;
	c.hp.threads = c.threads;
	if (rank == 0) fprintf(stderr, "read length:\t%d\n", rd.rl);

	uint32_t *vc = NULL, *jc = NULL;
	size_t nv = 0, nj = 0;
	if (load_codes(c.v_anchors, c.anchor_mismatches, &vc, &nv) || load_codes(c.j_anchors, c.anchor_mismatches, &jc, &nj)) return 255;
	qsort(vc, nv, 4, cmp_u32);
	qsort(jc, nj, 4, cmp_u32);
	char** vlines = NULL;
	size_t nvl = 0;
	if (load_vregion(c.source_sim_file, &vlines, &nvl)) return 255;

	vdjx_ctx* gx = NULL;
	VX(vdjx_init(rank, &gx));                               /* rank r drives GPU r */
	vdjx_mgpu* mg = NULL;
	const int use_mgpu = c.gpus > 1 || getenv("VDJX_FORCE_MGPU") != NULL;      /* (the variable: a one-rank RCCL run of the same code path) */
	if (use_mgpu) {
		unsigned char id[VDJX_MGPU_ID_BYTES];
		if (rank == 0) {
			if (vdjx_mgpu_unique_id(id)) { fprintf(stderr, "%s\n", vdjx_mgpu_last_error()); return 1; }
			for (int r = 1; r < c.gpus; r++) { if (write(id_pipe[r][1], id, sizeof id) != (ssize_t) sizeof id) { perror("write"); return 1; } close(id_pipe[r][1]); }
		} else {
			if (read(id_pipe[rank][0], id, sizeof id) != (ssize_t) sizeof id) { fprintf(stderr, "rank %d: no RCCL id from rank 0\n", rank); return 1; }
			close(id_pipe[rank][0]);
		}
		if (vdjx_mgpu_init(rank, c.gpus, rank, id, &mg)) { fprintf(stderr, "rank %d: %s\n", rank, vdjx_mgpu_last_error()); return 1; }
	}
	VX(vdjx_anchor_sets_load(gx, vc, nv, jc, nj));
	VX(vdjx_vregion_load(gx, (const char* const*) vlines, nvl, c.hp.vregion_kmer_size));
	status("POST_VJF_INIT");

	vdjx_pool* px = NULL;
	if (rank == 0) {                            /* the read index (and with it the scorers) lives on rank 0's GPU: the whole pool */
		VX(vdjx_pool_load(gx, rd.primary, rd.n_primary, rd.secondary, rd.n_secondary, rd.rl, &px));
		if (vdjx_stat(gx, "pool_other_bases"))
			fprintf(stderr, "warning: %llu bases other than ACGTN in the reads are treated as N (the reference would carry them inside k-mers)\n",
			        (unsigned long long) vdjx_stat(gx, "pool_other_bases"));
		VX(vdjx_read_index_build(gx, px, rd.pair_id, rd.read_num, rd.is_rc, rd.reg_rank, rd.n_pairs));
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
		status("POST_READ_EXTRACT");
		fprintf(stderr, "Assembling...\n");
	}
	vdjx_graph* gg = NULL;
	if (rank == 0) status("PRE_PRE_GRAPH1");   /* A2:1387 */
	if (!use_mgpu) {
		VX(vdjx_kmer_build(gx, px, c.hp.k, c.hp.min_node_freq, c.hp.min_base_quality, &gg));
	} else {
		/* rank r holds records [r*S, (r+1)*S) of the scan order: the rank-major numbering of the sharded build IS the scan order */
		const int sliced = rank > 0 && rd.secondary == NULL && rd.n_pairs == 0;      /* load_slice_text: the pool IS the slice */
		const size_t Rt = rd.n_primary + rd.n_secondary;
		const size_t S = sliced ? rd.n_secondary : (Rt + (size_t) c.gpus - 1) / (size_t) c.gpus;
		const size_t a = (size_t) rank * S < Rt ? (size_t) rank * S : Rt, b = a + S < Rt ? a + S : Rt;
		uint8_t* mine = sliced ? rd.primary : slice_records(&rd, a, b);
		vdjx_pool* ps = NULL;
		VX(vdjx_pool_load(gx, mine, sliced ? rd.n_primary : b - a, NULL, 0, rd.rl, &ps));
		if (!sliced) free(mine);
		if (vdjx_mgpu_kmer_build(mg, gx, ps, c.hp.k, c.hp.min_node_freq, c.hp.min_base_quality, S ? S : 1, &gg)) {
			fprintf(stderr, "rank %d: %s\n", rank, vdjx_mgpu_last_error());
			return 1;
		}
		vdjx_pool_free(ps);
		if (rank == 0) fprintf(stderr, "k-mer table sharded over %d GPUs: %llu bytes sent by rank 0\n", c.gpus, (unsigned long long) vdjx_mgpu_bytes_sent(mg));
		vdjx_mgpu_free(mg);
		if (rank != 0) {                        /* the graph is identical on every rank; the serial traversal runs on rank 0 */
			vdjx_graph_free(gg);
			vdjx_shutdown(gx);
			return 0;
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

	hook_ud ud = {gx, &rd, &c.hp};
	vdjh_hooks hk = {&ud, h_root_score, h_window_score, h_sam_body, vc, nv, jc, nj, h_status};
	vdjh_stats st;
	if (vdjh_assemble(&c.hp, &hg, &hk, "vdj_contigs.fa", "vdjer.dot", stdout, &st)) {
		fprintf(stderr, "%s\n", vdjh_last_error());
		return 1;
	}
	fprintf(stderr, "num root nodes: %zu\nProcessed roots: %zu\ncontig_candidates: %zu\nwindows scored: %zu valid: %zu\ncontigs: %zu\n",
	        st.n_roots, st.n_roots_accepted, st.n_contig_candidates, st.n_windows_scored, st.n_windows_valid, st.n_contigs_out);
	for (int r = 1; r < c.gpus; r++) {                    /* (normally all ended, well, long ago: on_sigchld has their status) */
		while (g_kid_state[r] == 0) {
			int st_ = 0;
			const pid_t p = waitpid(g_kids[r], &st_, 0);
			if (p == g_kids[r]) g_kid_state[r] = (WIFEXITED(st_) && WEXITSTATUS(st_) == 0) ? 1 : 2;
			else if (p < 0 && errno != EINTR) break;          /* reaped by the handler meanwhile */
		}
		if (g_kid_state[r] != 1) { fprintf(stderr, "rank %d failed\n", r); return 1; }
	}
	status("FINIS");
	fflush(stdout);
	vdjx_graph_free(gg);
	vdjx_pool_free(px);
	vdjx_shutdown(gx);
	return 0;
}
