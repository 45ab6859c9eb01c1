/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Drives the *unmodified* reference sources where they lie under /root/reference
 * (compiled by oracle/Makefile into oracle/_ref/vdjer_ref, -O0 exactly like the
 * reference's Makefile:8 -- higher levels crash on the reference's missing-return UB).
 *
 * The reference is one monolithic program with no test seams, so this harness
 *   (1) #includes src/main/c/assembler2_vdj.c with main renamed, which exposes its
 *       file-scope functions (build_pre_graph, prune_pre_graph, build_graph2, ...),
 *   (2) supplies get_read_length()/extract() -- the two entry points of bam_read.c
 *       (htslib-dependent BAM I/O, out of scope) -- reading a plain
 *       text "reads file" instead of a BAM, and registering records with the
 *       reference's own add_read_info() in the same way add_to_buffer does
 *       (bam_read.c:206-244: as-is record, then reverse-complement record),
 *   (3) offers sub-commands that call reference functions directly and dump their
 *       results as TSV; tests/golden/ fixtures are generated from these dumps.
 *
 * Nothing here re-implements the hot path: every number printed comes out of a
 * reference function.  No reference source text is copied into this repository.
 *
 * Reads file (one read per line, registration order == BAM order in the reference):
 *     <pool:P|S> <name> <read_num:1|2> <is_rev:0|1> <SEQ> <QUAL>
 */
#include <stdint.h>
#include <assert.h>

#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
static long vdjx_roots_scored = 0;
static FILE* vdjx_root_log = NULL;                 /* VDJX_REF_ROOT_LOG=<file>: "<root k-mer>\t<verdict>" per consumed root */
static pthread_mutex_t vdjx_root_log_mutex = PTHREAD_MUTEX_INITIALIZER;
extern "C" int vdjx_hook_score_seq(const char* seq, int threshold);
/* count every root consumed by worker_thread (A2:1103) to detect the lost-root race */
#define score_seq(a, b) vdjx_hook_score_seq(a, b)
#define main vdjer_reference_main
#include "assembler2_vdj.c"
#undef main
#undef score_seq

int score_seq(const char* seq, int threshold); /* the real one, seq_score.c:158 */
extern "C" int vdjx_hook_score_seq(const char* seq, int threshold) {
	__sync_fetch_and_add(&vdjx_roots_scored, 1);
	int verdict = score_seq(seq, threshold);
	if (vdjx_root_log) {        /* which roots a run consumed and what the (racy, at --t > 1: seq_score.c:14) scorer said of each */
		pthread_mutex_lock(&vdjx_root_log_mutex);
		fwrite(seq, 1, kmer_size, vdjx_root_log);
		fprintf(vdjx_root_log, "\t%d\n", verdict);
		pthread_mutex_unlock(&vdjx_root_log_mutex);
	}
	return verdict;
}

/* quick_map3.c */
extern void quick_map_init();
extern void add_read_info(char* read_id, char* seq, char* quals, char read_num, char is_rc);
/* vj_filter.c / seq_to_kmer.c */
extern unsigned long seq_to_int(const char* seq);

/* ------------------------------------------------------------------------------------------ */
/* reads-file loader standing in for bam_read.c                                                */
/* ------------------------------------------------------------------------------------------ */

static char h_complement(char c) {
	switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; }
}

int get_read_length(char* reads_file) {
	FILE* fp = fopen(reads_file, "r");
	if (!fp) { fprintf(stderr, "harness: cannot open %s\n", reads_file); exit(-1); }
	char pool[8], name[512], seq[1024], qual[1024];
	int rn, rev, rl = -1;
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		int l = (int) strlen(seq);
		if (l > rl) rl = l;
	}
	fclose(fp);
	if (rl <= 0) { fprintf(stderr, "harness: no reads in %s\n", reads_file); exit(-1); }
	return rl;
}

void extract(char* reads_file, char* vdj_fasta, char* v_region, char* c_region,
		char*& primary_buf, char*& secondary_buf) {
	(void) vdj_fasta; (void) v_region; (void) c_region;
	quick_map_init();

	FILE* fp = fopen(reads_file, "r");
	if (!fp) { fprintf(stderr, "harness: cannot open %s\n", reads_file); exit(-1); }
	char pool[8], name[512], seq[1024], qual[1024];
	int rn, rev;
	size_t np = 0, ns = 0;
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		if (pool[0] == 'P') np++; else ns++;
	}
	rewind(fp);
	int rl = read_length;
	size_t rec = 2 * (size_t) rl + 1;
	primary_buf = (char*) calloc(np * 2 * rec + 1, 1);
	secondary_buf = (char*) calloc(ns * 2 * rec + 1, 1);
	char* pp = primary_buf;
	char* sp = secondary_buf;
	while (fscanf(fp, "%7s %511s %d %d %1023s %1023s", pool, name, &rn, &rev, seq, qual) == 6) {
		if ((int) strlen(seq) != rl || (int) strlen(qual) != rl) {
			fprintf(stderr, "harness: read %s has length != %d\n", name, rl); exit(-1);
		}
		char*& bp = (pool[0] == 'P') ? pp : sp;
		char* id = strdup(name);
		/* as-is record */
		bp[0] = '0';
		memcpy(bp + 1, seq, rl);
		memcpy(bp + 1 + rl, qual, rl);
		add_read_info(id, bp + 1, bp + 1 + rl, (char) rn, (char) (rev ? 1 : 0));
		bp += rec;
		/* reverse-complement record with reversed qualities */
		bp[0] = '0';
		for (int i = 0; i < rl; i++) {
			bp[1 + i] = h_complement(seq[rl - 1 - i]);
			bp[1 + rl + i] = qual[rl - 1 - i];
		}
		add_read_info(id, bp + 1, bp + 1 + rl, (char) rn, (char) (rev ? 0 : 1));
		bp += rec;
	}
	fclose(fp);
}

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static void print_n(FILE* f, const char* s, int n) { for (int i = 0; i < n; i++) fputc(s[i], f); }

static void at_exit_report() {
	fprintf(stderr, "HARNESS_ROOTS_SCORED\t%ld\n", vdjx_roots_scored);
	if (vdjx_root_log) fclose(vdjx_root_log);
}

typedef dense_hash_map<const char*, pre_node, my_hash, eqstr> pre_map_t;
typedef dense_hash_map<const char*, struct node*, my_hash, eqstr> node_map_t;

/* read "key=value" style trailing args into the reference's own params struct via its parser */
static void parse_ref_params(int argc, char** argv) {
	parse_params(argc, argv, &p);
	if (p.min_base_quality >= MAX_QUAL_SUM) p.min_base_quality = MAX_QUAL_SUM - 1; /* A2:1514-1516, done in main */
	VREGION_KMER_SIZE = p.vregion_kmer_size;
	CONTIG_SIZE = p.eval_stop - p.eval_start + 1;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: graph  -- k-mer table, prune, graph build, roots, condense (a-1, a-2, a-3)     */
/*   vdjer_ref graph <out_prefix> <vdjer flags...>                                             */
/* ------------------------------------------------------------------------------------------ */
static int cmd_graph(int argc, char** argv) {
	const char* out = argv[2];
	parse_ref_params(argc - 2, argv + 2);
	read_length = get_read_length(p.input_bam);
	kmer_size = p.kmer;
	score_seq_init(p.kmer, 1000, p.source_sim_file);
	vjf_init(p.v_anchors, p.j_anchors, p.anchor_mismatches, p.vj_min_win, p.vj_max_win,
			p.j_conserved, p.window_span, p.j_extension);
	char* input = NULL; char* unaligned = NULL;
	extract(p.input_bam, p.vdj_fasta, p.v_region, p.c_region, input, unaligned);

	char fn[4096];
	pre_map_t pre_nodes;
	pre_nodes.set_empty_key(NULL);
	char* deleted_key = (char*) calloc(kmer_size, sizeof(char));
	pre_nodes.set_deleted_key(deleted_key);
	build_pre_graph(input, pre_nodes);
	build_pre_graph(unaligned, pre_nodes);

	/* full pre-prune table (only when small) */
	snprintf(fn, sizeof fn, "%s.pre.tsv", out);
	FILE* f = fopen(fn, "w");
	if (pre_nodes.size() <= 200000) {
		for (pre_map_t::const_iterator it = pre_nodes.begin(); it != pre_nodes.end(); ++it) {
			print_n(f, it->first, kmer_size);
			fprintf(f, "\t%d\t%d\t", (int) it->second.frequency, (int) it->second.hasMultipleUniqueReads);
			print_n(f, it->second.contributingRead, read_length);
			fputc('\t', f);
			for (int j = 0; j < kmer_size; j++) fprintf(f, "%02x", (unsigned) it->second.qual_sums[j]);
			fputc('\n', f);
		}
	}
	fclose(f);
	size_t pre_size = pre_nodes.size();

	prune_pre_graph(pre_nodes);
	snprintf(fn, sizeof fn, "%s.survivors.tsv", out);
	f = fopen(fn, "w");
	for (pre_map_t::const_iterator it = pre_nodes.begin(); it != pre_nodes.end(); ++it) {
		print_n(f, it->first, kmer_size);
		fprintf(f, "\t%d\n", (int) it->second.frequency);
	}
	fclose(f);

	struct_pool* pool = (struct_pool*) calloc(1, sizeof(struct_pool));
	node_map_t* nodes = new node_map_t();
	nodes->set_empty_key(NULL);
	pool->nodes = (struct node*) calloc(pre_nodes.size() + 1, sizeof(struct node));
	pool->idx = 0;
	pool->size = pre_nodes.size() + 3;
	build_graph2(input, nodes, pool, 1, pre_nodes);
	build_graph2(unaligned, nodes, pool, 0, pre_nodes);

	/* nodes in creation order with their edge lists in list order */
	snprintf(fn, sizeof fn, "%s.nodes.tsv", out);
	f = fopen(fn, "w");
	for (int i = 0; i < pool->idx; i++) {
		struct node* n = &pool->nodes[i];
		fprintf(f, "%d\t", n->id);
		print_n(f, n->kmer, kmer_size);
		fprintf(f, "\t%d\t%d\t%d\t", (int) n->frequency, (int) n->has_vmer, (int) n->has_jmer);
		for (linked_node* l = n->toNodes; l; l = l->next) fprintf(f, "%d,", l->node->id);
		fputc('\t', f);
		for (linked_node* l = n->fromNodes; l; l = l->next) fprintf(f, "%d,", l->node->id);
		fputc('\n', f);
	}
	fclose(f);

	/* dense_hash_map iteration order of `nodes` */
	snprintf(fn, sizeof fn, "%s.node_order.tsv", out);
	f = fopen(fn, "w");
	for (node_map_t::const_iterator it = nodes->begin(); it != nodes->end(); ++it) fprintf(f, "%d\n", it->second->id);
	fclose(f);

	linked_node* roots = identify_root_nodes(nodes);
	snprintf(fn, sizeof fn, "%s.roots.tsv", out);
	f = fopen(fn, "w");
	for (linked_node* l = roots; l; l = l->next) {
		fprintf(f, "%d\t", l->node->id);
		print_n(f, l->node->kmer, kmer_size);
		fprintf(f, "\t%d\n", score_seq(l->node->kmer, p.min_source_homology_score));
	}
	fclose(f);

	condense_graph(nodes);
	snprintf(fn, sizeof fn, "%s.condensed.tsv", out);
	f = fopen(fn, "w");
	for (int i = 0; i < pool->idx; i++) {
		struct node* n = &pool->nodes[i];
		fprintf(f, "%d\t%d\t%d\t%d\t%d\t%s\t", n->id, (int) n->is_condensed, (int) n->is_filtered,
				(int) n->has_vmer, (int) n->has_jmer, n->is_condensed ? n->seq : "-");
		for (linked_node* l = n->toNodes; l; l = l->next) fprintf(f, "%d,", l->node->id);
		fputc('\n', f);
	}
	fclose(f);
	fprintf(stderr, "HARNESS_GRAPH\tpre=%zu\tsurvivors=%zu\tnodes=%d\n", pre_size, (size_t) pre_nodes.size(), pool->idx);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: score -- root scorer (a-7)                                                     */
/*   vdjer_ref score <v_region.fa> <k> <vk> <threshold> <kmers.txt>   -> "<kmer>\t<0|1>"        */
/* ------------------------------------------------------------------------------------------ */
static int cmd_score(int argc, char** argv) {
	if (argc < 7) return 2;
	kmer_size = atoi(argv[3]);
	VREGION_KMER_SIZE = atoi(argv[4]);
	int thr = atoi(argv[5]);
	score_seq_init(kmer_size, 1000, argv[2]);
	FILE* fp = fopen(argv[6], "r");
	char line[4096];
	while (fgets(line, sizeof line, fp)) {
		size_t l = strlen(line);
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if ((int) l < kmer_size) continue;
		printf("%s\t%d\n", line, score_seq(line, thr));
	}
	fclose(fp);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: map -- read->contig mapper + coverage validator (a-8, a-9)                     */
/*   vdjer_ref map <windows.txt> <vdjer flags...>                                              */
/*   per window: "W\t<idx>\t<valid>\t<npairs>" then "P\t<name>\t<pos1>\t<pos2>\t<insert>\t<rc1>\t<rc2>" */
/*   rows in mapped order, then "S\t<first,second>;..." sorted start list                       */
/* ------------------------------------------------------------------------------------------ */
static int cmd_map(int argc, char** argv) {
	const char* wfile = argv[2];
	parse_ref_params(argc - 2, argv + 2);
	read_length = get_read_length(p.input_bam);
	kmer_size = p.kmer;
	char* input = NULL; char* unaligned = NULL;
	extract(p.input_bam, p.vdj_fasta, p.v_region, p.c_region, input, unaligned);
	FILE* fp = fopen(wfile, "r");
	char* line = (char*) calloc(100000, 1);
	int idx = 0;
	while (fgets(line, 100000, fp)) {
		size_t l = strlen(line);
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (!l) continue;
		vector<mapped_pair> mapped;
		vector<pair<int, int> > starts;
		char id[64];
		snprintf(id, sizeof id, "w%d", idx);
		quick_map_process_contig(id, line, mapped, starts);
		char valid = p.read_filter_floor == 0 ? 1 : coverage_is_valid(read_length, (int) l, p.eval_start, p.eval_stop,
				p.filter_read_span, p.insert_len, p.insert_len, p.read_filter_floor, mapped, starts, 0, p.filter_mate_span);
		printf("W\t%d\t%d\t%zu\n", idx, (int) valid, mapped.size());
		for (size_t i = 0; i < mapped.size(); i++) {
			printf("P\t%s\t%d\t%d\t%d\t%d\t%d\n", mapped[i].r1->id, (int) mapped[i].pos1, (int) mapped[i].pos2,
					(int) mapped[i].insert, (int) mapped[i].r1->is_rc, (int) mapped[i].r2->is_rc);
		}
		printf("S\t");
		for (size_t i = 0; i < starts.size(); i++) printf("%d,%d;", starts[i].first, starts[i].second);
		printf("\n");
		idx++;
	}
	fclose(fp);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: vjf -- V/J window discovery (host glue f-1) in table iteration order           */
/*   vdjer_ref vjf <contigs.txt> <vdjer flags...>   -> "C\t<idx>" then "<window>\t<cdr3>" rows  */
/* ------------------------------------------------------------------------------------------ */
static int cmd_vjf(int argc, char** argv) {
	const char* cfile = argv[2];
	parse_ref_params(argc - 2, argv + 2);
	vjf_init(p.v_anchors, p.j_anchors, p.anchor_mismatches, p.vj_min_win, p.vj_max_win,
			p.j_conserved, p.window_span, p.j_extension);
	vjf_cdr3_block_buffer = (char*) calloc(1024L * 1000L, sizeof(char));
	FILE* fp = fopen(cfile, "r");
	char* line = (char*) calloc(100000, 1);
	int idx = 0;
	while (fgets(line, 100000, fp)) {
		size_t l = strlen(line);
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (!l) continue;
		dense_hash_map<const char*, const char*, vjf_hash, vjf_eqstr> wins;
		wins.set_empty_key(NULL);
		vjf_search(line, wins, 1);
		printf("C\t%d\t%zu\n", idx++, (size_t) wins.size());
		for (dense_hash_map<const char*, const char*, vjf_hash, vjf_eqstr>::iterator it = wins.begin(); it != wins.end(); ++it)
			printf("%s\t%s\n", it->first, it->second);
	}
	fclose(fp);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: hash -- MurmurHash64A(line, len, 97) and seq_to_int(line) known answers        */
/* ------------------------------------------------------------------------------------------ */
static int cmd_hash(int argc, char** argv) {
	FILE* fp = fopen(argv[2], "r");
	char line[8192];
	while (fgets(line, sizeof line, fp)) {
		size_t l = strlen(line);
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (!l) continue;
		unsigned long code = 0;
		int ok16 = l >= 16;
		for (size_t i = 0; ok16 && i < 16; i++) if (!strchr("ACGT", line[i])) ok16 = 0;
		if (ok16) code = seq_to_int(line);
		printf("%s\t%llu\t%lu\n", line, (unsigned long long) MurmurHash64A(line, (int) l, 97), ok16 ? code : 0UL);
	}
	fclose(fp);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sub-command: order -- dense_hash_map iteration order with contig_hash (len-bounded keys)    */
/*   vdjer_ref order <len> <ops.txt>   ops: "I <str>" insert, "E <str>" erase, "R" resize(0)    */
/* ------------------------------------------------------------------------------------------ */
static int cmd_order(int argc, char** argv) {
	CONTIG_SIZE = atoi(argv[2]);
	dense_hash_map<const char*, const char*, contig_hash, contig_eqstr> m;
	m.set_empty_key(NULL);
	m.set_deleted_key(DELETED_KEY);
	FILE* fp = fopen(argv[3], "r");
	char* line = (char*) calloc(100000, 1);
	while (fgets(line, 100000, fp)) {
		size_t l = strlen(line);
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (!l) continue;
		if (line[0] == 'I') m[strdup(line + 2)] = "x";
		else if (line[0] == 'E') m.erase(line + 2);
		else if (line[0] == 'R') m.resize(0);
		else if (line[0] == 'D') { /* dump */
			printf("D\t%zu\t%zu\n", (size_t) m.size(), (size_t) m.bucket_count());
			for (dense_hash_map<const char*, const char*, contig_hash, contig_eqstr>::iterator it = m.begin(); it != m.end(); ++it)
				printf("%s\n", it->first);
		}
	}
	fclose(fp);
	return 0;
}

/* seqd <anchors file> <start> <end>: the index generator process_kmers (seq_dist.c:49-71) -- its own main() is commented
 * out in the reference (seq_dist.c:73-98) and did exactly this call; rows go to stdout */
extern void process_kmers(char* input, unsigned long start, unsigned long end);
static int cmd_seqd(int argc, char** argv) {
	if (argc != 5) { fprintf(stderr, "usage: vdjer_ref seqd <anchors> <start> <end>\n"); return 2; }
	process_kmers(argv[2], strtoul(argv[3], NULL, 10), strtoul(argv[4], NULL, 10));
	return 0;
}

int main(int argc, char** argv) {
	if (argc < 2) {
		fprintf(stderr, "usage: vdjer_ref run|graph|score|map|vjf|hash|order|seqd ...\n");
		return 2;
	}
	if (!strcmp(argv[1], "run")) {
		atexit(at_exit_report);
		if (getenv("VDJX_REF_ROOT_LOG")) vdjx_root_log = fopen(getenv("VDJX_REF_ROOT_LOG"), "w");
		/* the reference's own main(): params -> read length -> scorer init -> extract -> assemble */
		return vdjer_reference_main(argc - 1, argv + 1);
	}
	if (!strcmp(argv[1], "graph")) return cmd_graph(argc, argv);
	if (!strcmp(argv[1], "score")) return cmd_score(argc, argv);
	if (!strcmp(argv[1], "map")) return cmd_map(argc, argv);
	if (!strcmp(argv[1], "vjf")) return cmd_vjf(argc, argv);
	if (!strcmp(argv[1], "hash")) return cmd_hash(argc, argv);
	if (!strcmp(argv[1], "order")) return cmd_order(argc, argv);
	if (!strcmp(argv[1], "seqd")) return cmd_seqd(argc, argv);
	fprintf(stderr, "unknown sub-command %s\n", argv[1]);
	return 2;
}
