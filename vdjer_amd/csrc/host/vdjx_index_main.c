/* vdjx_index -- the v_index / j_index generator as a command.
 *
 * Usage (the one the reference's commented-out main() documents, seq_dist.c:73-98):
 *     vdjx_index <anchors file> <start> <end>  > v_index
 * <anchors file>: one 16-base anchor per line (get_kmers, seq_dist.c:37-47: seq_to_int of each fgets line).
 * Output: "<code>\t<min distance>\n" for every code in [start, end] within 5 bases of an anchor (process_kmers,
 * seq_dist.c:49-71).  The distances come from libvdjx.so (vdjx_index_generate); this file only parses and prints. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "../../../include/vdjx.h"

/* seq_to_int / base_val (seq_to_kmer.c:6-46): A0 T1 C2 G3, first base most significant; anything else is fatal */
static int code16(const char* s, uint32_t* out) {
	uint32_t v = 0;
	for (int i = 0; i < 16; i++) {
		uint32_t b;
		switch (s[i]) {
		case 'A': b = 0; break;
		case 'T': b = 1; break;
		case 'C': b = 2; break;
		case 'G': b = 3; break;
		default: return -1;
		}
		v = (v << 2) | b;
	}
	*out = v;
	return 0;
}

int main(int argc, char** argv) {
	if (argc != 4) {
		fprintf(stderr, "Usage: vdjx_index <input> <start> <end>\n");
		return 255;
	}
	FILE* in = fopen(argv[1], "r");
	if (!in) { fprintf(stderr, "Could not open file: [%s]\n", argv[1]); return 255; }
	size_t n = 0, cap = 1024;
	uint32_t* anchors = (uint32_t*) malloc(cap * sizeof(uint32_t));
	char line[1024];
	while (fgets(line, sizeof(line), in)) {
		if (n == cap) anchors = (uint32_t*) realloc(anchors, (cap *= 2) * sizeof(uint32_t));
		if (code16(line, &anchors[n])) { fprintf(stderr, "Error converting base in: %s", line); return 255; }   /* seq_to_kmer.c:22-24 */
		n++;
	}
	fclose(in);
	const unsigned long long start = strtoull(argv[2], NULL, 10), end = strtoull(argv[3], NULL, 10);
	vdjx_ctx* ctx = NULL;
	if (vdjx_init(0, &ctx)) { fprintf(stderr, "vdjx_index: %s\n", vdjx_last_error()); return 1; }
	/* a slice at a time keeps the row arrays small; rows are ascending within and across slices */
	const unsigned long long SLICE = 1ull << 28;
	uint64_t cap_rows = 1u << 22;
	uint32_t* codes = (uint32_t*) malloc(cap_rows * 4);
	uint8_t* dists = (uint8_t*) malloc(cap_rows);
	int rc = 0;
	for (unsigned long long s0 = start; s0 <= end && s0 <= 0xFFFFFFFFull; s0 += SLICE) {
		unsigned long long e0 = s0 + SLICE - 1 < end ? s0 + SLICE - 1 : end;
		uint64_t rows = 0;
		rc = vdjx_index_generate(ctx, anchors, n, s0, e0, 5, cap_rows, &rows, codes, dists);
		if (!rc && rows > cap_rows) {
			cap_rows = rows;
			codes = (uint32_t*) realloc(codes, cap_rows * 4);
			dists = (uint8_t*) realloc(dists, cap_rows);
			rc = vdjx_index_generate(ctx, anchors, n, s0, e0, 5, cap_rows, &rows, codes, dists);
		}
		if (rc) { fprintf(stderr, "vdjx_index: %s\n", vdjx_last_error()); break; }
		for (uint64_t i = 0; i < rows; i++) printf("%lu\t%d\n", (unsigned long) codes[i], (int) dists[i]);
	}
	vdjx_shutdown(ctx);
	free(codes); free(dists); free(anchors);
	return rc ? 1 : 0;
}
