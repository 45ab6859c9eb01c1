/* bamx.c -- see bamx.h.  Plain C over zlib; no GPU code. */
#include "bamx.h"

#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>
#include <pthread.h>

static char g_err[512];
const char* bamx_last_error(void) { return g_err; }
static int fail(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return -2;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* BGZF (bgzf.c): blocks are gzip members with a 'BC' extra field holding the block size                              */
/* ------------------------------------------------------------------------------------------------------------------ */
/* Read-ahead for the SEQUENTIAL passes of extract (bamx_set_threads): BGZF blocks are independent deflate streams, so while the
 * consumer parses records the blocks behind it are already being inflated -- one thread reads raw blocks into a ring in file order,
 * the others inflate them, the consumer takes them in order.  (The reference inflates on the thread that parses, one block at a time:
 * with the assembly down to milliseconds the four passes over the BAM are what a run waits for.)  A seek empties the ring. */
#define RA_SLOTS 128
typedef struct {
	int64_t addr;                     /* file offset of the block */
	int bsize, count;                 /* compressed size; inflated size, or -1: error */
	int state;                        /* 0 free, 1 raw, 2 being inflated, 3 inflated */
	uint8_t raw[65536 + 64], out[65536];
} ra_slot;
typedef struct {
	pthread_t reader, *workers;
	int n_workers, started;
	pthread_mutex_t mu;
	pthread_cond_t cv, cv_work, cv_space;      /* cv: the parser waits for the block at the head; cv_work: the inflating threads for a raw block; cv_space: the
	                                            * reader for a free slot (one condition for all three woke every thread at every block: eight threads were slower than four) */
	ra_slot* ring;
	uint64_t head, tail, next_job;    /* consumer takes ring[head % RA_SLOTS]; the reader fills [tail]; workers inflate [next_job] */
	int eof, stop, err;               /* the reader saw the end of the file / shutdown / a malformed block */
	char errmsg[160];
	FILE* fp;
} readahead;

struct bamx_file {
	FILE* fp;
	int64_t block_address;            /* file offset of the current block */
	int block_offset, block_length;   /* position inside / size of its inflated data; 0/0 = not loaded */
	int64_t next_address;             /* file offset behind the current block (what ftello says without read-ahead) */
	readahead* ra;
	char path[4096];
	uint8_t inflated[65536];
	uint8_t raw[65536 + 64];
	int n_ref;
	char** ref_name;
	int32_t* ref_len;
	uint8_t* rec_data;                /* variable part of the record being read (per file: the reader is re-entrant) */
	size_t rec_cap;
	const uint8_t *cur_sq, *cur_ql;   /* the packed bases and the qualities of the record read last (read1_fields: decoded on request) */
};

/* sizes taken from a file are bounded before they reach an allocator: a corrupt or hostile BAM fails cleanly */
#define BAMX_MAX_REFS (1 << 24)
#define BAMX_MAX_NAME (1 << 16)
#define BAMX_MAX_RECORD (1u << 26)          /* 64 MB: far beyond any short-read record */
#define BAMX_MAX_BINS (1 << 22)
#define BAMX_MAX_CHUNKS (1 << 26)

static int rd_u16(const uint8_t* p) { return p[0] | (p[1] << 8); }
static uint32_t rd_u32(const uint8_t* p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
static uint64_t rd_u64(const uint8_t* p) { return (uint64_t) rd_u32(p) | ((uint64_t) rd_u32(p + 4) << 32); }

/* inflate one raw block (18-byte header checked by the caller): the inflated size, or -1 */
static int inflate_block(const uint8_t* h, int bsize, uint8_t* out) {
	z_stream zs;
	memset(&zs, 0, sizeof zs);
	zs.next_in = (Bytef*) (h + 18);
	zs.avail_in = (uInt) (bsize - 18 - 8);
	zs.next_out = out;
	zs.avail_out = 65536;
	if (inflateInit2(&zs, -15) != Z_OK) return -1;
	const int rc = inflate(&zs, Z_FINISH);
	inflateEnd(&zs);
	if (rc != Z_STREAM_END) return -1;
	if ((uint32_t) zs.total_out != rd_u32(h + bsize - 4)) return -1;
	return (int) zs.total_out;
}

static void* ra_reader(void* arg) {
	readahead* ra = (readahead*) arg;
	for (;;) {
		pthread_mutex_lock(&ra->mu);
		while (!ra->stop && ra->tail - ra->head >= RA_SLOTS) pthread_cond_wait(&ra->cv_space, &ra->mu);
		if (ra->stop) { pthread_mutex_unlock(&ra->mu); return NULL; }
		ra_slot* sl = &ra->ring[ra->tail % RA_SLOTS];
		pthread_mutex_unlock(&ra->mu);
		/* (the slot is free and only this thread fills slots: no lock while reading the file) */
		const int64_t addr = ftello(ra->fp);
		uint8_t* h = sl->raw;
		const size_t got = fread(h, 1, 18, ra->fp);
		int bad = 0, end = 0, bsize = 0;
		if (got == 0) end = 1;
		else if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || rd_u16(h + 10) != 6 || h[12] != 'B' || h[13] != 'C' || rd_u16(h + 14) != 2) bad = 1;
		else {
			bsize = rd_u16(h + 16) + 1;
			if (bsize < 26 || fread(h + 18, 1, (size_t) bsize - 18, ra->fp) != (size_t) bsize - 18) bad = 1;
		}
		pthread_mutex_lock(&ra->mu);
		if (bad) { ra->err = 1; snprintf(ra->errmsg, sizeof ra->errmsg, "not a BGZF block (or a truncated one) at file offset %lld", (long long) addr); }
		if (bad || end) { ra->eof = 1; pthread_cond_broadcast(&ra->cv); pthread_cond_broadcast(&ra->cv_work); pthread_mutex_unlock(&ra->mu); return NULL; }
		sl->addr = addr; sl->bsize = bsize; sl->state = 1;
		ra->tail++;
		pthread_cond_signal(&ra->cv_work);
		pthread_mutex_unlock(&ra->mu);
	}
}

static void* ra_worker(void* arg) {
	readahead* ra = (readahead*) arg;
	for (;;) {
		pthread_mutex_lock(&ra->mu);
		while (!ra->stop && !(ra->next_job < ra->tail) && !ra->eof) pthread_cond_wait(&ra->cv_work, &ra->mu);
		if (ra->stop || !(ra->next_job < ra->tail)) {                   /* shutdown, or the file is read and every block taken */
			const int done = ra->stop || ra->eof;
			pthread_mutex_unlock(&ra->mu);
			if (done) return NULL;
			continue;
		}
		const uint64_t job = ra->next_job++;
		ra_slot* sl = &ra->ring[job % RA_SLOTS];
		sl->state = 2;
		pthread_mutex_unlock(&ra->mu);
		const int count = inflate_block(sl->raw, sl->bsize, sl->out);
		pthread_mutex_lock(&ra->mu);
		sl->count = count;
		sl->state = 3;
		if (job == ra->head) pthread_cond_signal(&ra->cv);          /* (the parser waits for the block at the head only) */
		pthread_mutex_unlock(&ra->mu);
	}
}

static void ra_stop(bamx_file* f) {
	readahead* ra = f->ra;
	if (!ra) return;
	if (ra->started) {
		pthread_mutex_lock(&ra->mu);
		ra->stop = 1;
		pthread_cond_broadcast(&ra->cv);
		pthread_cond_broadcast(&ra->cv_work);
		pthread_cond_broadcast(&ra->cv_space);
		pthread_mutex_unlock(&ra->mu);
		pthread_join(ra->reader, NULL);
		for (int i = 0; i < ra->n_workers; i++) pthread_join(ra->workers[i], NULL);
	}
	if (ra->fp) fclose(ra->fp);
	pthread_mutex_destroy(&ra->mu);
	pthread_cond_destroy(&ra->cv);
	pthread_cond_destroy(&ra->cv_work);
	pthread_cond_destroy(&ra->cv_space);
	free(ra->workers);
	free(ra->ring);
	free(ra);
	f->ra = NULL;
}

/* (re)start the read-ahead at file offset `from` for a run of `n` threads: the parser is one of them, the others inflate (the thread
 * that reads the raw blocks sleeps in the file system most of the time) */
static int ra_start(bamx_file* f, int64_t from, int n) {
	ra_stop(f);
	n = n > 2 ? n - 1 : 1;
	readahead* ra = (readahead*) calloc(1, sizeof *ra);
	if (!ra) return fail("out of memory");
	ra->ring = (ra_slot*) calloc(RA_SLOTS, sizeof(ra_slot));
	ra->workers = (pthread_t*) calloc((size_t) n, sizeof(pthread_t));
	ra->fp = fopen(f->path, "rb");
	pthread_mutex_init(&ra->mu, NULL);
	pthread_cond_init(&ra->cv, NULL);
	pthread_cond_init(&ra->cv_work, NULL);
	pthread_cond_init(&ra->cv_space, NULL);
	f->ra = ra;
	if (!ra->ring || !ra->workers || !ra->fp || fseeko(ra->fp, (off_t) from, SEEK_SET) != 0) { ra_stop(f); return fail("cannot start the read-ahead on %s", f->path); }
	ra->n_workers = n;
	if (pthread_create(&ra->reader, NULL, ra_reader, ra) != 0) { ra_stop(f); return fail("pthread_create failed"); }
	ra->started = 1;
	for (int i = 0; i < n; i++)
		if (pthread_create(&ra->workers[i], NULL, ra_worker, ra) != 0) { ra->n_workers = i; ra_stop(f); return fail("pthread_create failed"); }
	return 0;
}

/* bgzf_read_block (bgzf.c): 0 = ok (block_length 0 at end of file), -1 = error */
static int read_block(bamx_file* f) {
	if (f->ra) {
		readahead* ra = f->ra;
		pthread_mutex_lock(&ra->mu);
		while (!(ra->head < ra->tail && ra->ring[ra->head % RA_SLOTS].state == 3) && !(ra->eof && ra->head == ra->tail)) pthread_cond_wait(&ra->cv, &ra->mu);
		if (ra->head == ra->tail) {                       /* the end of the file (or of what could be read of it) */
			const int err = ra->err;
			char msg[160];
			memcpy(msg, ra->errmsg, sizeof msg);
			pthread_mutex_unlock(&ra->mu);
			if (err) return fail("%s", msg), -1;
			f->block_length = 0;
			return 0;
		}
		ra_slot* sl = &ra->ring[ra->head % RA_SLOTS];
		pthread_mutex_unlock(&ra->mu);
		if (sl->count < 0) return fail("inflate failed in the block at %lld", (long long) sl->addr), -1;
		memcpy(f->inflated, sl->out, (size_t) sl->count);
		if (f->block_length != 0) f->block_offset = 0;
		f->block_address = sl->addr;
		f->block_length = sl->count;
		f->next_address = sl->addr + sl->bsize;
		pthread_mutex_lock(&ra->mu);
		sl->state = 0;
		ra->head++;
		pthread_cond_signal(&ra->cv_space);
		pthread_mutex_unlock(&ra->mu);
		return 0;
	}
	const int64_t addr = ftello(f->fp);
	uint8_t* h = f->raw;
	size_t got = fread(h, 1, 18, f->fp);
	if (got == 0) { f->block_length = 0; return 0; }
	if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || rd_u16(h + 10) != 6 || h[12] != 'B' || h[13] != 'C' || rd_u16(h + 14) != 2)
		return fail("not a BGZF block at file offset %lld", (long long) addr), -1;
	const int bsize = rd_u16(h + 16) + 1;
	if (bsize < 26) return fail("BGZF block too short at %lld", (long long) addr), -1;
	if (fread(h + 18, 1, (size_t) bsize - 18, f->fp) != (size_t) bsize - 18) return fail("truncated BGZF block at %lld", (long long) addr), -1;
	const int count = inflate_block(h, bsize, f->inflated);
	if (count < 0) return fail("inflate failed (or ISIZE mismatch) in the block at %lld", (long long) addr), -1;
	if (f->block_length != 0) f->block_offset = 0;      /* do not reset the offset if this read follows a seek */
	f->block_address = addr;
	f->block_length = count;
	f->next_address = addr + bsize;
	return 0;
}

/* bgzf_read (bgzf.c:547-574), including the move to the next block's address when a block is used up */
static long bz_read(bamx_file* f, void* data, size_t length) {
	size_t done = 0;
	uint8_t* out = (uint8_t*) data;
	while (done < length) {
		int avail = f->block_length - f->block_offset;
		if (avail <= 0) {
			if (read_block(f) != 0) return -1;
			avail = f->block_length - f->block_offset;
			if (avail <= 0) break;
		}
		const size_t take = length - done < (size_t) avail ? length - done : (size_t) avail;
		memcpy(out, f->inflated + f->block_offset, take);
		f->block_offset += (int) take;
		out += take;
		done += take;
	}
	if (f->block_offset == f->block_length) {
		f->block_address = f->next_address;             /* (= ftello(fp) of the plain reader: the next block's address) */
		f->block_offset = f->block_length = 0;
	}
	return (long) done;
}

uint64_t bamx_tell(const bamx_file* f) { return ((uint64_t) f->block_address << 16) | ((uint64_t) f->block_offset & 0xFFFF); }

int bamx_set_threads(bamx_file* f, int n) {
	if (n <= 1) { if (f->ra) { const int64_t at = f->next_address; ra_stop(f); if (fseeko(f->fp, (off_t) at, SEEK_SET) != 0) return fail("seek failed"); } return 0; }
	/* what lies behind the block the reader holds (or the position it was sent to) is what the read-ahead starts with */
	const int64_t from = f->block_length ? f->next_address : f->block_address;
	return ra_start(f, from, n);
}

int bamx_seek(bamx_file* f, uint64_t voff) {            /* bgzf_seek (bgzf.c:847-866) */
	if (f->ra) {                                       /* (the blocks read ahead are of no use at the new place) */
		const int n = f->ra->n_workers;
		f->block_length = 0;
		f->block_address = (int64_t) (voff >> 16);
		f->block_offset = (int) (voff & 0xFFFF);
		f->next_address = f->block_address;
		return ra_start(f, f->block_address, n);
	}
	if (fseeko(f->fp, (off_t) (voff >> 16), SEEK_SET) != 0) return fail("seek failed");
	f->next_address = (int64_t) (voff >> 16);
	f->block_length = 0;
	f->block_address = (int64_t) (voff >> 16);
	f->block_offset = (int) (voff & 0xFFFF);
	return 0;
}

int bamx_is_bam(const char* path) {
	FILE* fp = fopen(path, "rb");
	if (!fp) return -1;
	unsigned char m[2] = {0, 0};
	const size_t n = fread(m, 1, 2, fp);
	fclose(fp);
	return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* BAM header and records (sam.c: bam_hdr_read, bam_read1)                                                             */
/* ------------------------------------------------------------------------------------------------------------------ */
bamx_file* bamx_open(const char* path) {
	bamx_file* f = (bamx_file*) calloc(1, sizeof *f);
	if (!f) return NULL;
	f->fp = fopen(path, "rb");
	if (!f->fp) { fail("cannot open %s", path); free(f); return NULL; }
	snprintf(f->path, sizeof f->path, "%s", path);
	uint8_t b[8];
	if (bz_read(f, b, 4) != 4 || memcmp(b, "BAM\1", 4)) { fail("%s: no BAM magic", path); bamx_close(f); return NULL; }
	if (bz_read(f, b, 4) != 4) { fail("%s: truncated header", path); bamx_close(f); return NULL; }
	uint32_t l_text = rd_u32(b);
	while (l_text) {                                       /* the SAM text is not needed */
		uint8_t skip[4096];
		const size_t take = l_text < sizeof skip ? l_text : sizeof skip;
		if (bz_read(f, skip, take) != (long) take) { fail("%s: truncated header text", path); bamx_close(f); return NULL; }
		l_text -= (uint32_t) take;
	}
	if (bz_read(f, b, 4) != 4) { fail("%s: truncated header", path); bamx_close(f); return NULL; }
	{
		const uint32_t nr = rd_u32(b);
		if (nr > BAMX_MAX_REFS) { fail("%s: %u reference sequences: not a sane BAM header", path, nr); bamx_close(f); return NULL; }
		f->ref_name = (char**) calloc((size_t) nr + 1, sizeof(char*));
		f->ref_len = (int32_t*) calloc((size_t) nr + 1, 4);
		if (!f->ref_name || !f->ref_len) { fail("%s: out of memory for %u reference names", path, nr); bamx_close(f); return NULL; }
		f->n_ref = (int) nr;
	}
	for (int i = 0; i < f->n_ref; i++) {
		if (bz_read(f, b, 4) != 4) { fail("%s: truncated reference list", path); bamx_close(f); return NULL; }
		const uint32_t ln = rd_u32(b);
		if (ln > BAMX_MAX_NAME) { fail("%s: reference name of %u bytes", path, ln); bamx_close(f); return NULL; }
		f->ref_name[i] = (char*) calloc((size_t) ln + 1, 1);
		if (!f->ref_name[i]) { fail("%s: out of memory", path); bamx_close(f); return NULL; }
		if (bz_read(f, f->ref_name[i], ln) != (long) ln || bz_read(f, b, 4) != 4) { fail("%s: truncated reference list", path); bamx_close(f); return NULL; }
		f->ref_len[i] = (int32_t) rd_u32(b);
	}
	return f;
}

void bamx_close(bamx_file* f) {
	if (!f) return;
	ra_stop(f);
	if (f->fp) fclose(f->fp);
	for (int i = 0; i < f->n_ref; i++) free(f->ref_name ? f->ref_name[i] : NULL);
	free(f->ref_name);
	free(f->ref_len);
	free(f->rec_data);
	free(f);
}

int bamx_n_ref(const bamx_file* f) { return f->n_ref; }
const char* bamx_ref_name(const bamx_file* f, int tid) { return tid >= 0 && tid < f->n_ref ? f->ref_name[tid] : NULL; }

static int name2id(const bamx_file* f, const char* name) {   /* bam_name2id: exact match */
	for (int i = 0; i < f->n_ref; i++) if (!strcmp(f->ref_name[i], name)) return i;
	return -1;
}

/* a record's fixed fields, name and end; its sequence and qualities stay packed in the file's buffer until rec_seq / rec_qual ask
 * for them (extract's last pass looks at the name and the flag of every record of the file and keeps one in thousands) */
static void rec_seq(const bamx_file* f, bamx_rec* r) {
	const uint8_t* sq = f->cur_sq;
	for (int i = 0; i < r->l_qseq; i++) r->seq[i] = "=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> ((~i & 1) << 2)) & 0xF];
	r->seq[r->l_qseq] = 0;
}
static void rec_qual(const bamx_file* f, bamx_rec* r) {
	const uint8_t* ql = f->cur_ql;
	for (int i = 0; i < r->l_qseq; i++) r->qual[i] = (char) (ql[i] + 33);
	r->qual[r->l_qseq] = 0;
}
static int read1_fields(bamx_file* f, bamx_rec* r);
int bamx_read1(bamx_file* f, bamx_rec* r) {
	const int rc = read1_fields(f, r);
	if (rc >= 0) { rec_seq(f, r); rec_qual(f, r); }
	return rc;
}
static int read1_fields(bamx_file* f, bamx_rec* r) {
	uint8_t b[36];
	r->voff = bamx_tell(f);
	long got = bz_read(f, b, 4);
	if (got == 0) return -1;                                 /* normal end of file */
	if (got != 4) return got < 0 ? -3 : fail("truncated record");
	const uint32_t block_len = rd_u32(b);
	if (block_len < 32) return fail("record shorter than its fixed part");
	if (bz_read(f, b, 32) != 32) return fail("truncated record");
	r->tid = (int32_t) rd_u32(b);
	r->pos = (int32_t) rd_u32(b + 4);
	const uint32_t x = rd_u32(b + 8), y = rd_u32(b + 12);
	r->bin = (uint16_t) (x >> 16);
	r->mapq = (uint8_t) ((x >> 8) & 0xFF);
	const int l_qname = (int) (x & 0xFF);
	r->flag = (uint16_t) (y >> 16);
	r->n_cigar = (int32_t) (y & 0xFFFF);
	r->l_qseq = (int32_t) rd_u32(b + 16);
	if (block_len > BAMX_MAX_RECORD) return fail("record of %u bytes: not a sane BAM record", block_len);
	const size_t rest = block_len - 32;
	if (rest + 1 > f->rec_cap) {
		uint8_t* nd = (uint8_t*) realloc(f->rec_data, rest + 1024);
		if (!nd) return fail("out of memory for a record of %u bytes", block_len);
		f->rec_data = nd;
		f->rec_cap = rest + 1024;
	}
	uint8_t* data = f->rec_data;
	if (bz_read(f, data, rest) != (long) rest) return fail("truncated record");
	if (r->l_qseq < 0 || r->l_qseq > 1023) return fail("read of %d bases: longer than this reader takes", r->l_qseq);
	const size_t need = (size_t) l_qname + (size_t) r->n_cigar * 4 + (size_t) (r->l_qseq + 1) / 2 + (size_t) r->l_qseq;
	if (need > rest || l_qname < 1) return fail("record fields exceed the record");
	memcpy(r->qname, data, (size_t) l_qname);
	r->qname[l_qname] = 0;
	const uint8_t* cig = data + l_qname;
	int32_t rlen = 0;
	for (int i = 0; i < r->n_cigar; i++) {
		const uint32_t c = rd_u32(cig + 4 * i);
		const uint32_t op = c & 0xF;
		if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int32_t) (c >> 4);    /* M D N = X consume the reference */
	}
	r->end = r->pos + (r->n_cigar ? rlen : 1);               /* bam_readrec, sam.c:458-467 */
	f->cur_sq = cig + (size_t) r->n_cigar * 4;
	f->cur_ql = f->cur_sq + (size_t) (r->l_qseq + 1) / 2;
	r->seq[0] = r->qual[0] = 0;
	return (int) block_len;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* BAI (SAM specification 5.2: bins with their chunks, the linear index) and the region query (5.1.3, 5.3)             */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct { uint64_t u, v; } chunk_t;
typedef struct { uint32_t bin; int n; chunk_t* list; } bin_t;
typedef struct { int n_bin; bin_t* bins; int n_intv; uint64_t* ioff; } refidx_t;
struct bamx_index { int n_ref; refidx_t* ref; };
#define BAI_N_LVLS 5
#define BAI_MIN_SHIFT 14

static const bin_t* find_bin(const refidx_t* r, uint32_t bin) {
	for (int i = 0; i < r->n_bin; i++) if (r->bins[i].bin == bin) return &r->bins[i];
	return NULL;
}

bamx_index* bamx_index_load(const char* bam_path) {
	char fn[4200];
	snprintf(fn, sizeof fn, "%s.bai", bam_path);
	FILE* fp = fopen(fn, "rb");
	if (!fp) { fail("cannot open %s", fn); return NULL; }
	uint8_t b[16];
	if (fread(b, 1, 8, fp) != 8 || memcmp(b, "BAI\1", 4)) { fail("%s: no BAI magic", fn); fclose(fp); return NULL; }
	bamx_index* ix = (bamx_index*) calloc(1, sizeof *ix);
	if (!ix) { fclose(fp); fail("%s: out of memory", fn); return NULL; }
	if (rd_u32(b + 4) > BAMX_MAX_REFS) { fail("%s: %u references: not a sane index", fn, rd_u32(b + 4)); fclose(fp); free(ix); return NULL; }
	ix->n_ref = (int) rd_u32(b + 4);
	ix->ref = (refidx_t*) calloc((size_t) ix->n_ref + 1, sizeof(refidx_t));
	if (!ix->ref) { fail("%s: out of memory", fn); fclose(fp); free(ix); return NULL; }
	for (int i = 0; i < ix->n_ref; i++) {
		refidx_t* r = &ix->ref[i];
		if (fread(b, 1, 4, fp) != 4) goto bad;
		if (rd_u32(b) > BAMX_MAX_BINS) goto bad;
		r->n_bin = (int) rd_u32(b);
		r->bins = (bin_t*) calloc((size_t) r->n_bin + 1, sizeof(bin_t));
		if (!r->bins) { r->n_bin = 0; goto bad; }
		for (int j = 0; j < r->n_bin; j++) {
			if (fread(b, 1, 8, fp) != 8) goto bad;
			r->bins[j].bin = rd_u32(b);
			if (rd_u32(b + 4) > BAMX_MAX_CHUNKS) goto bad;
			r->bins[j].n = (int) rd_u32(b + 4);
			r->bins[j].list = (chunk_t*) calloc((size_t) r->bins[j].n + 1, sizeof(chunk_t));
			if (!r->bins[j].list) { r->bins[j].n = 0; goto bad; }
			for (int k = 0; k < r->bins[j].n; k++) {
				if (fread(b, 1, 16, fp) != 16) goto bad;
				r->bins[j].list[k].u = rd_u64(b);
				r->bins[j].list[k].v = rd_u64(b + 8);
			}
		}
		if (fread(b, 1, 4, fp) != 4) goto bad;
		if (rd_u32(b) > BAMX_MAX_CHUNKS) goto bad;
		r->n_intv = (int) rd_u32(b);
		r->ioff = (uint64_t*) calloc((size_t) r->n_intv + 1, 8);
		if (!r->ioff) { r->n_intv = 0; goto bad; }
		for (int j = 0; j < r->n_intv; j++) {
			if (fread(b, 1, 8, fp) != 8) goto bad;
			r->ioff[j] = rd_u64(b);
		}
		for (int j = 1; j < r->n_intv; j++) if (r->ioff[j] == 0) r->ioff[j] = r->ioff[j - 1];     /* "fill missing values" */
	}
	fclose(fp);
	return ix;
bad:
	fail("%s: truncated index", fn);
	fclose(fp);
	bamx_index_free(ix);
	return NULL;
}

void bamx_index_free(bamx_index* ix) {
	if (!ix) return;
	for (int i = 0; i < ix->n_ref; i++) {
		for (int j = 0; j < ix->ref[i].n_bin; j++) free(ix->ref[i].bins[j].list);
		free(ix->ref[i].bins);
		free(ix->ref[i].ioff);
	}
	free(ix->ref);
	free(ix);
}

/* A samtools region, "RNAME[:START[-END]]" (SAM specification / samtools(1): 1-based, END inclusive, thousands separators allowed in
 * the numbers, a missing END = up to the end of the reference).  Written from that description: the part after the LAST colon is a
 * coordinate range only if it reads as one (digits and commas, at most one '-', START <= END); otherwise the whole string is the
 * reference name (names may hold colons).  Result: 0-based half-open [*beg, *end), and the length of the name. */
static int region_number(const char* s, size_t n, long long* out) {
	long long v = 0;
	int digits = 0;
	for (size_t i = 0; i < n; i++) {
		if (s[i] == ',') continue;
		if (s[i] < '0' || s[i] > '9') return -1;
		if (v > (long long) INT_MAX) return -1;
		v = v * 10 + (s[i] - '0');
		digits++;
	}
	*out = v;
	return digits;
}
static size_t parse_region(const char* s, int* beg, int* end) {
	const size_t len = strlen(s);
	*beg = 0; *end = INT_MAX;
	const char* colon = strrchr(s, ':');
	if (!colon) return len;
	const char* range = colon + 1;
	const size_t rn = len - (size_t) (range - s);
	const char* dash = (const char*) memchr(range, '-', rn);
	long long a = 0, b = (long long) INT_MAX;
	const int da = region_number(range, dash ? (size_t) (dash - range) : rn, &a);
	if (da < 0) return len;                                      /* not a range: the colon belongs to the name */
	if (dash) {
		const int db = region_number(dash + 1, rn - (size_t) (dash + 1 - range), &b);
		if (db < 0) return len;
		if (db == 0) b = 0;                                      /* "chr:5-" reads an empty END as 0 with strtol: START > END, a name */
	}
	long long lo = a - 1;                                        /* 1-based START -> 0-based; START 0 (or none) means the first base */
	if (lo < 0) lo = 0;
	if (b > (long long) INT_MAX) b = INT_MAX;
	if (lo > b) return len;
	*beg = (int) lo;
	*end = (int) b;
	return (size_t) (colon - s);
}

static int cmp_chunk(const void* a, const void* b) {
	const chunk_t* x = (const chunk_t*) a;
	const chunk_t* y = (const chunk_t*) b;
	if (x->u != y->u) return x->u < y->u ? -1 : 1;
	return x->v < y->v ? -1 : x->v > y->v;
}

/* The bins that may hold alignments overlapping [beg, end), SAM specification section 5.3 (the UCSC binning scheme with a 16 kb
 * minimum interval and five levels: bin 0 spans 512 Mb; the levels start at bins 1, 9, 73, 585 and 4681 with intervals of 64 Mb,
 * 8 Mb, 1 Mb, 128 kb and 16 kb). */
static const int LEVEL_FIRST[6] = {0, 1, 9, 73, 585, 4681};
static const int LEVEL_SHIFT[6] = {29, 26, 23, 20, 17, 14};

long bamx_query(bamx_file* f, const bamx_index* ix, const char* region, void (*cb)(const bamx_rec*, void*), void* ud) {
	int beg, end;
	const size_t ne = parse_region(region, &beg, &end);
	char name[512];
	if (ne >= sizeof name) return fail("region name too long");
	memcpy(name, region, ne);
	name[ne] = 0;
	int tid = name2id(f, name);
	if (tid < 0) tid = name2id(f, region);
	if (tid < 0) return fail("region %s: the reference name is not in the BAM header (the reference crashes here: NULL iterator)", region);
	if (tid >= ix->n_ref) return fail("region %s: no index for that reference (NULL iterator in the reference)", region);
	const refidx_t* r = &ix->ref[tid];
	/* the linear index (section 5.1.3): entry w is the smallest file offset of an alignment that overlaps the 16 kb window w.  In a
	 * coordinate-sorted file no alignment that overlaps [beg, end) starts before the first one overlapping beg's window: chunks
	 * that END at or before that offset hold nothing for this region. */
	/* Which window's entry: the tightest bound would be beg's own window.  The reference's library takes a LOOSER one -- the entry of the
	 * first window of the nearest bin THAT EXISTS at or before beg's leaf bin (the leaf itself, else its left neighbours under the same
	 * parent, else the parent, and so on upwards) -- and which chunks are read decides where the query leaves the file, which is where
	 * extract's sequential pass starts (bam_read.c:346-374).  The same reads come back either way; the looser bound keeps the file
	 * position the reference's (round-4 advice: with the tight bound a region without overlapping reads in a sparse neighbourhood left
	 * the file untouched where the reference seeks and reads). */
	uint64_t min_off = 0;
	if (r->n_intv > 0 && (long long) beg < (1LL << 29)) {
		int lvl = BAI_N_LVLS;
		int bin = LEVEL_FIRST[lvl] + (beg >> LEVEL_SHIFT[lvl]);
		while (bin > 0 && !find_bin(r, (uint32_t) bin)) {
			const int parent = (bin - 1) >> 3, first_child = (parent << 3) + 1;
			if (bin > first_child) bin--;                      /* the neighbour to the left, same level */
			else { bin = parent; lvl--; }
		}
		if (bin > 0 || find_bin(r, 0)) {
			const long long w = (long long) (bin - LEVEL_FIRST[lvl]) << (LEVEL_SHIFT[lvl] - BAI_MIN_SHIFT);      /* first 16 kb window the bin covers */
			min_off = w < r->n_intv ? r->ioff[w] : 0;
		}
	}
	/* candidate chunks: those of every bin that can hold an overlapping alignment */
	size_t cap = 64, n_off = 0;
	chunk_t* off = (chunk_t*) malloc(cap * sizeof(chunk_t));
	if (!off) return fail("out of memory");
	if (beg < end) {
		long long last = (long long) end - 1;                                 /* last base of the region */
		if (last >= (1LL << 29)) last = (1LL << 29) - 1;                       /* the scheme addresses 512 Mb */
		for (int lvl = 0; lvl <= BAI_N_LVLS && (long long) beg < (1LL << 29); lvl++) {
			const int from = LEVEL_FIRST[lvl] + (beg >> LEVEL_SHIFT[lvl]), to = LEVEL_FIRST[lvl] + (int) (last >> LEVEL_SHIFT[lvl]);
			for (int bi = from; bi <= to; bi++) {
				const bin_t* p = find_bin(r, (uint32_t) bi);
				if (!p) continue;
				for (int j = 0; j < p->n; j++) {
					if (p->list[j].v <= min_off) continue;
					if (n_off == cap) {
						cap *= 2;
						chunk_t* q = (chunk_t*) realloc(off, cap * sizeof(chunk_t));
						if (!q) { free(off); return fail("out of memory"); }
						off = q;
					}
					off[n_off++] = p->list[j];
				}
			}
		}
	}
	long n_ret = 0;
	if (n_off) {
		/* one pass over the file: the chunks in file order, every stretch covered by any of them read once.  Two chunks are read as
		 * one when they overlap, touch, or the gap between them lies inside one compressed block (a seek there would decompress the
		 * same block again; what the gap holds fails the overlap test below like anything else that does not belong). */
		qsort(off, n_off, sizeof(chunk_t), cmp_chunk);
		size_t m = 0;
		for (size_t i = 1; i < n_off; i++) {
			if (off[i].u <= off[m].v || (off[i].u >> 16) == (off[m].v >> 16)) { if (off[i].v > off[m].v) off[m].v = off[i].v; }
			else off[++m] = off[i];
		}
		n_off = m + 1;
		static bamx_rec rec;
		int done = 0;
		for (size_t i = 0; i < n_off && !done; i++) {
			if (bamx_tell(f) != off[i].u && bamx_seek(f, off[i].u)) { free(off); return -2; }
			while (bamx_tell(f) < off[i].v) {
				const int rc = bamx_read1(f, &rec);
				if (rc < -1) { free(off); return rc; }
				if (rc < 0) { done = 1; break; }                                /* end of file */
				if (rec.tid != tid || rec.pos >= end) { done = 1; break; }      /* sorted by coordinate: nothing further on can overlap */
				if (rec.end > beg && rec.pos < end) { n_ret++; if (cb) cb(&rec, ud); }
			}
		}
	}
	free(off);
	return n_ret;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* string sets (the dense_hash_sets of extract: membership only, iteration order never matters)                        */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const char** slot; uint32_t* tag; size_t cap, n; size_t keylen; /* 0 = NUL-terminated strings */ } sset;      /* tag: the upper half of the key's hash (a probe compares it before it follows the pointer) */
static uint64_t hash_bytes(const char* s, size_t n) {
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; i++) { h ^= (unsigned char) s[i]; h *= 1099511628211ull; }
	return h;
}
static void sset_init(sset* t, size_t keylen) { t->cap = 1024; t->n = 0; t->keylen = keylen; t->slot = (const char**) calloc(t->cap, sizeof(char*)); t->tag = (uint32_t*) calloc(t->cap, 4); }
static void sset_free(sset* t) { free(t->slot); free(t->tag); t->slot = NULL; t->tag = NULL; }
static size_t sset_len(const sset* t, const char* s) { return t->keylen ? t->keylen : strlen(s); }
/* (h: hash_bytes of the key -- a name asked for in several sets is hashed once) */
static const char* sset_get_h(const sset* t, const char* s, size_t n, uint64_t h) {
	const uint32_t tg = (uint32_t) (h >> 32);
	for (size_t i = h & (t->cap - 1);; i = (i + 1) & (t->cap - 1)) {
		const char* c = t->slot[i];
		if (!c) return NULL;
		if (t->tag[i] == tg && (t->keylen ? !memcmp(c, s, n) : !strcmp(c, s))) return c;
	}
}
static const char* sset_get(const sset* t, const char* s) {
	const size_t n = sset_len(t, s);
	return sset_get_h(t, s, n, hash_bytes(s, n));
}
static void sset_put(sset* t, const char* stored) {       /* the caller made sure it is absent */
	if ((t->n + 1) * 2 > t->cap) {
		const char** old = t->slot;
		uint32_t* otag = t->tag;
		const size_t oc = t->cap;
		t->cap *= 2;
		t->slot = (const char**) calloc(t->cap, sizeof(char*));
		t->tag = (uint32_t*) calloc(t->cap, 4);
		for (size_t i = 0; i < oc; i++)
			if (old[i]) {
				const uint64_t h = hash_bytes(old[i], sset_len(t, old[i]));
				size_t j = h & (t->cap - 1);
				while (t->slot[j]) j = (j + 1) & (t->cap - 1);
				t->slot[j] = old[i];
				t->tag[j] = (uint32_t) (h >> 32);
			}
		free(old);
		free(otag);
	}
	const uint64_t h = hash_bytes(stored, sset_len(t, stored));
	size_t j = h & (t->cap - 1);
	while (t->slot[j]) j = (j + 1) & (t->cap - 1);
	t->slot[j] = stored;
	t->tag[j] = (uint32_t) (h >> 32);
	t->n++;
}

/* append-only storage for names, 15-mers, sequences */
typedef struct blk { struct blk* next; size_t used, cap; char data[]; } blk;
static char* arena_alloc(blk** head, size_t n) {
	if (!*head || (*head)->used + n > (*head)->cap) {
		const size_t cap = n > (1u << 22) ? n : (1u << 22);
		blk* b = (blk*) malloc(sizeof(blk) + cap);
		b->next = *head; b->used = 0; b->cap = cap;
		*head = b;
	}
	char* p = (*head)->data + (*head)->used;
	(*head)->used += n;
	return p;
}
static const char* arena_str(blk** head, const char* s) {
	const size_t n = strlen(s) + 1;
	char* p = arena_alloc(head, n);
	memcpy(p, s, n);
	return p;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* extract (bam_read.c:294-446)                                                                                        */
/* ------------------------------------------------------------------------------------------------------------------ */
static char complement(char c) {              /* bam_read.c:115-129 */
	switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; }
}

/* load_kmers (bam_read.c:180-204): 15-mers of every fgets chunk of ig_vdj.fa that is not a header, both strands; the chunk
 * loses its last character ("Remove newline") and the loop stops one k-mer short (i < strlen - 15) */
#define EXTRACT_KMER_SIZE 15
/* The 15-mers of ig_vdj.fa that consist of A, C, G, T only (all of them, in practice) also go into a table of their 30-bit codes: the
 * screen of extract's sequential pass asks 35 times per read whether a 15-mer of the read is one of them -- a rolling code and one probe
 * of a table that stays in cache instead of hashing 15 characters and comparing strings (the reference's dense_hash_set of char*). */
typedef struct { uint32_t* slot; size_t cap, n; } cset;          /* slot: code + 1, 0 = empty */
static int base2(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; } }
static void cset_init(cset* t) { t->cap = 1 << 12; t->n = 0; t->slot = (uint32_t*) calloc(t->cap, 4); }
static void cset_free(cset* t) { free(t->slot); t->slot = NULL; }
static inline size_t cset_home(const cset* t, uint32_t code) { return ((size_t) code * 0x9E3779B1u >> 7) & (t->cap - 1); }
static inline int cset_has(const cset* t, uint32_t code) {
	for (size_t i = cset_home(t, code);; i = (i + 1) & (t->cap - 1)) {
		const uint32_t c = t->slot[i];
		if (!c) return 0;
		if (c == code + 1) return 1;
	}
}
static void cset_put(cset* t, uint32_t code) {
	if (cset_has(t, code)) return;
	if ((t->n + 1) * 4 > t->cap) {
		uint32_t* old = t->slot;
		const size_t oc = t->cap;
		t->cap *= 2;
		t->slot = (uint32_t*) calloc(t->cap, 4);
		for (size_t i = 0; i < oc; i++)
			if (old[i]) { size_t j = cset_home(t, old[i] - 1); while (t->slot[j]) j = (j + 1) & (t->cap - 1); t->slot[j] = old[i]; }
		free(old);
	}
	size_t j = cset_home(t, code);
	while (t->slot[j]) j = (j + 1) & (t->cap - 1);
	t->slot[j] = code + 1;
	t->n++;
}
/* the code of the 15 characters at s, or -1 if one of them is not A, C, G or T */
static long code15(const char* s) {
	uint32_t c = 0;
	for (int i = 0; i < EXTRACT_KMER_SIZE; i++) { const int b = base2(s[i]); if (b < 0) return -1; c = (c << 2) | (uint32_t) b; }
	return (long) c;
}

static int load_vdj_kmers(const char* path, sset* set, cset* codes, size_t* n_other, blk** arena) {
	FILE* fp = fopen(path, "r");
	if (!fp) return fail("Could not open file: [%s]", path);     /* the reference prints this and then crashes in fgets */
	char buf[1024], rcb[1024];
	while (fgets(buf, sizeof buf, fp)) {
		if (buf[0] == '>' || strlen(buf) < EXTRACT_KMER_SIZE) continue;
		buf[strlen(buf) - 1] = 0;
		const size_t n = strlen(buf);
		for (size_t i = 0; i < n; i++) rcb[i] = complement(buf[n - 1 - i]);
		rcb[n] = 0;
		if (n < EXTRACT_KMER_SIZE) continue;                       /* (size_t underflow in the reference: nothing sensible to add) */
		for (size_t i = 0; i + EXTRACT_KMER_SIZE < n; i++) {
			const char* two[2] = {buf + i, rcb + i};
			for (int j = 0; j < 2; j++) {
				const long cd = code15(two[j]);
				if (cd >= 0) { cset_put(codes, (uint32_t) cd); continue; }
				if (!sset_get(set, two[j])) {                      /* (a 15-mer with another letter in it: kept as text) */
					char* p = arena_alloc(arena, EXTRACT_KMER_SIZE + 1);
					memcpy(p, two[j], EXTRACT_KMER_SIZE);
					p[EXTRACT_KMER_SIZE] = 0;
					sset_put(set, p);
					(*n_other)++;
				}
			}
		}
	}
	fclose(fp);
	return 0;
}

typedef struct { sset* set; blk** arena; } name_ud;
static void add_name_cb(const bamx_rec* r, void* ud) {
	name_ud* u = (name_ud*) ud;
	if (!sset_get(u->set, r->qname)) sset_put(u->set, arena_str(u->arena, r->qname));
}

typedef struct { int (*keep)(void* ud, const char* name); void* ud; } keep_fn;
static void push_read(bamx_reads* out, size_t* cap, char pool, const char* name, int read_num, const bamx_rec* r, blk** arena, const keep_fn* kf) {
	const uint64_t seq_no = out->n_primary_reads + out->n_secondary_reads;
	const uint64_t pool_no = pool == 'P' ? out->n_primary_reads++ : out->n_secondary_reads++;
	if (kf->keep && !kf->keep(kf->ud, name)) return;
	if (out->n == *cap) { *cap = *cap ? *cap * 2 : 4096; out->v = (bamx_read*) realloc(out->v, *cap * sizeof(bamx_read)); }
	bamx_read* x = &out->v[out->n++];
	const size_t L = (size_t) out->read_len;
	x->pool = pool; x->name = name; x->read_num = read_num; x->is_rev = (r->flag & 16) != 0;
	x->seq_no = seq_no; x->pool_no = pool_no;
	x->seq = arena_alloc(arena, L + 1);
	x->qual = arena_alloc(arena, L + 1);
	memset(x->seq, 0, L + 1);
	memset(x->qual, 0, L + 1);
	strncpy(x->seq, r->seq, L);                                    /* bam_read.c:220,223 */
	strncpy(x->qual, r->qual, L);
}

static int g_threads = 1;
void bamx_threads(int n) { g_threads = n < 1 ? 1 : (n > 64 ? 64 : n); }

int bamx_extract(const char* bam_path, const char* vdj_fasta, const char* v_region, const char* c_region, bamx_reads* out) {
	return bamx_extract_filtered(bam_path, vdj_fasta, v_region, c_region, NULL, NULL, out);
}

int bamx_extract_filtered(const char* bam_path, const char* vdj_fasta, const char* v_region, const char* c_region,
                          int (*keep)(void* ud, const char* name), void* ud, bamx_reads* out) {
	const keep_fn kf = {keep, ud};
	memset(out, 0, sizeof *out);
	blk* arena = NULL;
	static bamx_rec rec;
	int rc = 0;
	/* get_read_length (bam_read.c:264-292) is a pass of its own over the whole file in the reference (and it loads the index too): the
	 * longest l_qseq of the file.  extract's last pass reads every record anyway: the maximum is taken there, one decompression of the
	 * whole file less.  (What get_read_length would have stopped on -- a file or an index that cannot be opened -- still stops here.) */
	{
		bamx_file* f = bamx_open(bam_path);
		if (!f) return -2;
		bamx_index* ix0 = bamx_index_load(bam_path);
		bamx_close(f);
		if (!ix0) return -2;
		bamx_index_free(ix0);
		out->max_len = -1;
	}
	sset kmers, primary, secondary;
	cset codes;
	size_t n_other_kmers = 0;
	sset_init(&kmers, EXTRACT_KMER_SIZE);
	cset_init(&codes);
	sset_init(&primary, 0);
	sset_init(&secondary, 0);
	bamx_file* f = NULL;
	bamx_index* ix = NULL;
	if ((rc = load_vdj_kmers(vdj_fasta, &kmers, &codes, &n_other_kmers, &arena)) != 0) goto done;
	f = bamx_open(bam_path);
	if (!f) { rc = -2; goto done; }
	ix = bamx_index_load(bam_path);
	if (!ix) { rc = -2; goto done; }
	/* names of the reads overlapping the variable region, then the constant region (bam_read.c:316-342) */
	{
		name_ud u1 = {&primary, &arena}, u2 = {&secondary, &arena};
		if (bamx_query(f, ix, v_region, add_name_cb, &u1) < 0) { rc = -2; goto done; }
		fprintf(stderr, "primary_reads size1: [%d]\n", (int) primary.n);
		if (bamx_query(f, ix, c_region, add_name_cb, &u2) < 0) { rc = -2; goto done; }
		fprintf(stderr, "secondary_reads size1: [%d]\n", (int) secondary.n);
	}
	/* "Process unmapped reads" (bam_read.c:346-374): sequential from WHEREVER the iterators left the file */
	if (g_threads > 1 && bamx_set_threads(f, g_threads)) { rc = -2; goto done; }
	while ((rc = read1_fields(f, &rec)) >= 0) {
		if (out->read_len == 0) out->read_len = rec.l_qseq;
		rec_seq(f, &rec);                                           /* (the screen reads the bases; nobody reads the qualities here) */
		const size_t L = strlen(rec.seq);
		if (L < EXTRACT_KMER_SIZE) continue;                        /* (size_t underflow in the reference) */
		/* the reference's loop (bam_read.c:358-372) asks at every position: is this 15-mer a V/D/J one?  yes -> the name into the
		 * primary set (if absent); no, and the read is unmapped -> into the secondary set (if absent).  What it leaves behind depends
		 * on two facts only -- some position hit, some position missed -- so they are found with a rolling 2-bit code (a window with
		 * another letter in it is looked up as text, in the few 15-mers of ig_vdj.fa that hold one) and the sets are touched once. */
		int any_hit = 0, any_miss = 0;
		{
			const int want_miss = (rec.flag & 4) != 0;
			uint32_t code = 0;
			int run = 0;                                                /* consecutive A/C/G/T characters ending at the current one */
			const size_t last = L - EXTRACT_KMER_SIZE;                  /* positions 0 .. last-1 are looked at (i + 15 < L) */
			for (size_t j = 0; j + 1 < L && !(any_hit && (any_miss || !want_miss)); j++) {
				const int b = base2(rec.seq[j]);
				if (b < 0) run = 0; else { code = ((code << 2) | (uint32_t) b) & 0x3FFFFFFFu; run++; }
				if (j + 1 < EXTRACT_KMER_SIZE) continue;
				const size_t i = j + 1 - EXTRACT_KMER_SIZE;             /* the window [i, i + 15) ends at j */
				if (i >= last) break;
				int hit;
				if (run >= EXTRACT_KMER_SIZE) hit = cset_has(&codes, code);
				else hit = n_other_kmers ? sset_get(&kmers, rec.seq + i) != NULL : 0;
				if (hit) any_hit = 1; else any_miss = 1;
			}
		}
		if (any_hit && !sset_get(&primary, rec.qname)) sset_put(&primary, arena_str(&arena, rec.qname));
		if (any_miss && (rec.flag & 4) && !sset_get(&secondary, rec.qname)) sset_put(&secondary, arena_str(&arena, rec.qname));
	}
	if (rc < -1) goto done;
	rc = 0;
	bamx_close(f);
	f = NULL;
	out->n_primary_names = primary.n;
	out->n_secondary_names = secondary.n;
	/* second pass from the beginning (bam_read.c:399-432): first 0x40 and first 0x80 record of every kept name */
	{
		sset p1, p2, s1, s2;
		sset_init(&p1, 0); sset_init(&p2, 0); sset_init(&s1, 0); sset_init(&s2, 0);
		size_t cap = 0;
		f = bamx_open(bam_path);
		if (!f) { rc = -2; }
		else {
			bamx_index* ix2 = bamx_index_load(bam_path);          /* bam_open loads it again and fails without it */
			if (!ix2) rc = -2;
			bamx_index_free(ix2);
			if (!rc && g_threads > 1 && bamx_set_threads(f, g_threads)) rc = -2;
		}
		while (!rc && (rc = read1_fields(f, &rec)) >= 0) {
			rc = 0;
			if (rec.l_qseq > out->max_len) out->max_len = rec.l_qseq;          /* (get_read_length) */
			if (rec.flag & 0x900) continue;
			/* every record of the file asks both sets for its name: hashed once; the text of a record is made when it is kept */
			const size_t nl = strlen(rec.qname);
			const uint64_t nh = hash_bytes(rec.qname, nl);
			const char* nm;
			if ((nm = sset_get_h(&primary, rec.qname, nl, nh)) != NULL) {
				if ((rec.flag & 0x40) && !sset_get_h(&p1, nm, nl, nh)) { rec_seq(f, &rec); rec_qual(f, &rec); push_read(out, &cap, 'P', nm, 1, &rec, &arena, &kf); sset_put(&p1, nm); }
				else if ((rec.flag & 0x80) && !sset_get_h(&p2, nm, nl, nh)) { rec_seq(f, &rec); rec_qual(f, &rec); push_read(out, &cap, 'P', nm, 2, &rec, &arena, &kf); sset_put(&p2, nm); }
			} else if ((nm = sset_get_h(&secondary, rec.qname, nl, nh)) != NULL) {
				if ((rec.flag & 0x40) && !sset_get_h(&s1, nm, nl, nh)) { rec_seq(f, &rec); rec_qual(f, &rec); push_read(out, &cap, 'S', nm, 1, &rec, &arena, &kf); sset_put(&s1, nm); }
				else if ((rec.flag & 0x80) && !sset_get_h(&s2, nm, nl, nh)) { rec_seq(f, &rec); rec_qual(f, &rec); push_read(out, &cap, 'S', nm, 2, &rec, &arena, &kf); sset_put(&s2, nm); }
			}
		}
		if (rc == -1) rc = 0;
		if (!rc && out->max_len <= 0) rc = fail("Error retrieving read length from: %s", bam_path);
		fprintf(stderr, "primary_output1: [%d] primary_output2: [%d] secondary_output1: [%d] secondary_output2: [%d]\n", (int) p1.n, (int) p2.n,
		        (int) s1.n, (int) s2.n);
		sset_free(&p1); sset_free(&p2); sset_free(&s1); sset_free(&s2);
	}
done:
	if (f) bamx_close(f);
	bamx_index_free(ix);
	sset_free(&kmers); cset_free(&codes); sset_free(&primary); sset_free(&secondary);
	out->arena = arena;
	if (rc) bamx_free(out);
	return rc;
}

void bamx_free(bamx_reads* r) {
	if (!r) return;
	blk* b = (blk*) r->arena;
	while (b) { blk* n = b->next; free(b); b = n; }
	free(r->v);
	memset(r, 0, sizeof *r);
}
