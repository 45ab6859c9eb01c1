/* bamx -- BAM/BGZF/BAI reading and V'DJer's read extraction in plain C over zlib.
 *
 * Replaces, for the `vdjer --in <bam>` path: get_read_length (bam_read.c:264-292) and extract (bam_read.c:294-446) together with
 * what they reach into htslib 1.2.1 for: sequential BAM reading (bgzf_read/seek/tell, bam_hdr_read, bam_read1), the BAI index and the
 * region iterator behind sam_itr_querys / sam_itr_next.  The file formats and the region query are written from the SAM/BAM
 * specification (BGZF 4.1, BAM 4.2, BAI 5.2, the binning scheme 5.3, region strings as samtools(1) describes them); what is taken from
 * the reference is BEHAVIOUR that its extraction depends on: where a finished region query leaves the file (extract's sequential pass
 * starts there, bam_read.c:346-374), and which records it returns.  The reference side of this row cannot be compiled here -- hts.c
 * includes a version.h that only the vendored Makefile generates, and the rules of this build forbid both running that Makefile and
 * writing a stand-in -- so the extraction's order rules are a restatement checked against an independent model (tests/bam_model.py);
 * the decoding and the region query are checked on BAM/BAI files written by real samtools (tests/golden/bam).
 */
#ifndef VDJX_BAMX_H
#define VDJX_BAMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one extracted read, in the order add_to_buffer is called (bam_read.c:404-428): the forward record; the caller derives the
 * reverse-complement record exactly as add_to_buffer does */
typedef struct {
	char pool;            /* 'P' primary_buf, 'S' secondary_buf */
	const char* name;     /* read name (owned by the bamx_reads) */
	int read_num;         /* 1 | 2 */
	int is_rev;           /* bam_is_rev */
	char* seq;            /* read_len characters of "=ACMGRSVTWYHKDBN" (NUL padded if the record is shorter: strncpy, :220) */
	char* qual;           /* read_len characters, Phred+33 */
	uint64_t seq_no;      /* its place among ALL reads the extraction emits (kept or not): add_read_info is called for read q as calls 2q, 2q+1 */
	uint64_t pool_no;     /* its place among the reads of its pool: records 2*pool_no, 2*pool_no+1 of primary_buf / secondary_buf */
} bamx_read;

typedef struct {
	bamx_read* v;
	size_t n;
	int read_len;         /* extract's: l_qseq of the first record its sequential pass sees (bam_read.c:353-355) */
	int max_len;          /* get_read_length: the longest l_qseq of the file (bam_read.c:264-292) */
	size_t n_primary_names, n_secondary_names;
	uint64_t n_primary_reads, n_secondary_reads;   /* reads emitted into either pool, kept by the caller's filter or not */
	void* arena;          /* private */
} bamx_reads;

const char* bamx_last_error(void);
/* 1 if the file starts with a gzip member (what sam_open takes for BAM), 0 if not, <0 if unreadable */
int bamx_is_bam(const char* path);
/* the whole extraction; v_region / c_region are samtools region strings ("chr14:105566277-106879844").  0 on success */
int bamx_extract(const char* bam_path, const char* vdj_fasta, const char* v_region, const char* c_region, bamx_reads* out);
/* the same passes, but only the reads `keep` says yes to are stored (all are counted and numbered): one rank of `vdjer --gpus N` keeps
 * the pairs it owns, 1/N of the pool, instead of every rank holding all of it.  keep == NULL keeps everything. */
int bamx_extract_filtered(const char* bam_path, const char* vdj_fasta, const char* v_region, const char* c_region,
                          int (*keep)(void* ud, const char* name), void* ud, bamx_reads* out);
void bamx_free(bamx_reads* r);
/* threads that inflate BGZF blocks ahead of the parser in extract's sequential passes (1: none, the parser inflates as the reference's
 * does; `vdjer --t N` passes N).  The records and their order do not depend on it. */
void bamx_threads(int n);

/* ---- lower level, exposed for the tests -------------------------------------------------------------------------- */
typedef struct bamx_file bamx_file;
typedef struct bamx_index bamx_index;
typedef struct {
	int32_t tid, pos, l_qseq, n_cigar, end;   /* end = pos + reference length of the CIGAR, or pos + 1 (sam.c:458-467) */
	uint16_t flag, bin;
	uint8_t mapq;
	char qname[256];
	char seq[1024];                           /* "=ACMGRSVTWYHKDBN" letters (reads longer than 1023 are an error here) */
	char qual[1024];                          /* Phred+33 */
	uint64_t voff;                            /* virtual offset of the record's first byte */
} bamx_rec;

bamx_file* bamx_open(const char* path);
void bamx_close(bamx_file* f);
int bamx_n_ref(const bamx_file* f);
const char* bamx_ref_name(const bamx_file* f, int tid);
int bamx_read1(bamx_file* f, bamx_rec* r);              /* >= 0 record read, -1 end of file, < -1 error */
uint64_t bamx_tell(const bamx_file* f);                  /* bgzf_tell */
int bamx_seek(bamx_file* f, uint64_t voff);
int bamx_set_threads(bamx_file* f, int n);               /* read-ahead from the reader's position on with n inflating threads (<= 1: off) */
bamx_index* bamx_index_load(const char* bam_path);       /* <bam_path>.bai */
void bamx_index_free(bamx_index* ix);
/* sam_itr_querys + the loop `while (sam_itr_next(...) >= 0)`: calls cb for every record the iterator returns; leaves the
 * file where the iterator left it.  Returns the number of records, or < 0 on error (unknown reference: the reference
 * dereferences a NULL iterator there) */
long bamx_query(bamx_file* f, const bamx_index* ix, const char* region, void (*cb)(const bamx_rec*, void*), void* ud);

#ifdef __cplusplus
}
#endif
#endif
