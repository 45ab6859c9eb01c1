/*
 * vdjx_mgpu.c -- see vdjx_mgpu.h: the sharded hot path driven from C (what `vdjer --gpus N` runs).
 *
 * The compute is libvdjx's (include/vdjx.h: vdjx_shard_* for the k-mer build, vdjx_window_pairs / _fetch / vdjx_window_cover for the
 * window scorer, vdjx_sam_blocks / vdjx_sam_merge for the SAM records); this file only decides who sends what to whom and moves it
 * through vdjx_comm.  It is the C twin of vdjer_amd/shard.py (the test and bench driver over torch.distributed).
 */
#define _GNU_SOURCE
#include "vdjx_mgpu.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

enum { WS_SDIR, WS_RDIR, WS_SPARTS, WS_RPARTS, WS_Q, WS_RQ, WS_ANS, WS_RANS, WS_SURV, WS_SURV_ALL, WS_MINS, WS_UCNT, WS_A, WS_B, WS_C, WS_D, WS_SLOTS };
enum { CMD_QUIT = 1, CMD_WINDOWS = 2, CMD_SAM = 3, CMD_YIELD = 4 };

struct vdjx_mgpu {
	int rank, nranks, device;
	vdjx_comm* cm;
	/* exchange buffers: one per purpose, kept by the handle and only replaced when a call needs more (a process that builds again -- a
	 * second chain, the next sample -- allocates nothing) */
	void* ws[WS_SLOTS];
	size_t ws_cap[WS_SLOTS];
	/* the rank's share of the pool (ASCII records; the packed pool reads its quality characters here), the registration ranks and the scan positions of its records */
	void *d_share, *d_reg, *d_scan;
	void* d_info;            /* pair ids (4 B), read numbers and is_rc flags (1 B each) of the share's records: the begun index build reads them */
	vdjx_pool* share_pool;
	uint64_t total_records;
	int rl;
	/* VDJX_TIMES: wall milliseconds per phase on this rank (a rank waits inside a collective for the slowest: what the phases cost
	 * the job), printed by rank 0 when it releases the others */
	int timing;
	int n_laps;
	struct { const char* name; double ms; uint64_t calls; uint64_t bytes; } laps[48];
	struct timespec lap_t;
	uint64_t lap_bytes;
};

static void lap_start(vdjx_mgpu* m) {
	if (!m->timing) return;
	clock_gettime(CLOCK_MONOTONIC, &m->lap_t);
	m->lap_bytes = vdjx_comm_bytes_sent(m->cm);
}
static void lap(vdjx_mgpu* m, const char* name) {
	if (!m->timing) return;
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	const double ms = (t.tv_sec - m->lap_t.tv_sec) * 1e3 + (t.tv_nsec - m->lap_t.tv_nsec) / 1e6;
	const uint64_t by = vdjx_comm_bytes_sent(m->cm);
	int i = 0;
	while (i < m->n_laps && strcmp(m->laps[i].name, name)) i++;
	if (i == m->n_laps && m->n_laps < 48) { m->laps[i].name = name; m->laps[i].ms = 0; m->laps[i].calls = 0; m->laps[i].bytes = 0; m->n_laps++; }
	if (i < m->n_laps) { m->laps[i].ms += ms; m->laps[i].calls++; m->laps[i].bytes += by - m->lap_bytes; }
	m->lap_t = t;
	m->lap_bytes = by;
}
void vdjx_mgpu_print_times(const vdjx_mgpu* m, FILE* f) {
	if (!m || !m->timing) return;
	for (int i = 0; i < m->n_laps; i++)
		fprintf(f, "VDJX_MGPU_PHASE\trank\t%d\t%s\tms\t%.2f\tcalls\t%llu\tbytes_sent\t%llu\n", m->rank, m->laps[i].name, m->laps[i].ms,
		        (unsigned long long) m->laps[i].calls, (unsigned long long) m->laps[i].bytes);
}

static __thread char g_err[640];
static int fail(int rc, const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return rc;
}
const char* vdjx_mgpu_last_error(void) { return g_err; }

/* buffer `slot` of at least `bytes` bytes (contents undefined) */
static int ws_get(vdjx_mgpu* m, int slot, size_t bytes, void** out) {
	if (bytes > m->ws_cap[slot]) {
		if (m->ws[slot]) (void) hipFree(m->ws[slot]);
		m->ws[slot] = NULL; m->ws_cap[slot] = 0;
		const size_t want = bytes + bytes / 8 + 256;
		const hipError_t e = hipMalloc(&m->ws[slot], want);
		if (e != hipSuccess) return fail(VDJX_EHIP, "hipMalloc of %zu bytes (exchange buffer %d): %s", want, slot, hipGetErrorString(e));
		m->ws_cap[slot] = want;
	}
	*out = m->ws[slot];
	return 0;
}

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(VDJX_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); goto done; } } while (0)
#define VX(x) do { int r_ = (x); if (r_ != 0) { rc = fail(r_, "rank %d: %s: %s", m->rank, #x, vdjx_last_error()); goto done; } } while (0)
#define CX(x) do { int r_ = (x); if (r_ != 0) { rc = fail(r_, "rank %d: %s: %s", m->rank, #x, vdjx_comm_last_error()); goto done; } } while (0)
#define WSG(slot, bytes, out) do { void* p_ = NULL; const int r_ = ws_get(m, slot, (size_t) (bytes), &p_); if (r_) { rc = r_; goto done; } *(out) = p_; } while (0)

int vdjx_mgpu_init(vdjx_comm* cm, int device, vdjx_mgpu** out) {
	*out = NULL;
	if (!cm) return fail(VDJX_EINVAL, "vdjx_mgpu_init: no communicator");
	vdjx_mgpu* m = (vdjx_mgpu*) calloc(1, sizeof *m);
	if (!m) return fail(VDJX_EINVAL, "out of memory");
	m->cm = cm; m->rank = vdjx_comm_rank(cm); m->nranks = vdjx_comm_size(cm); m->device = device;
	m->timing = getenv("VDJX_TIMES") != NULL;
	*out = m;
	return 0;
}

void vdjx_mgpu_free(vdjx_mgpu* m) {
	if (!m) return;
	(void) hipSetDevice(m->device);
	if (m->share_pool) vdjx_pool_free(m->share_pool);
	if (m->d_share) (void) hipFree(m->d_share);
	if (m->d_reg) (void) hipFree(m->d_reg);
	if (m->d_scan) (void) hipFree(m->d_scan);
	if (m->d_info) (void) hipFree(m->d_info);
	for (int i = 0; i < WS_SLOTS; i++) if (m->ws[i]) (void) hipFree(m->ws[i]);
	vdjx_comm_free(m->cm);
	free(m);
}

uint64_t vdjx_mgpu_bytes_sent(const vdjx_mgpu* m) { return m ? vdjx_comm_bytes_sent(m->cm) : 0; }

static uint64_t sum64(const uint64_t* v, int n) { uint64_t s = 0; for (int i = 0; i < n; i++) s += v[i]; return s; }

/* ------------------------------------------------------------------------------------------------------------------ */
/* the share and its read index                                                                                        */
/* ------------------------------------------------------------------------------------------------------------------ */
int vdjx_mgpu_load(vdjx_mgpu* m, vdjx_ctx* ctx, const uint8_t* records, size_t n_records, int rl, const uint32_t* scan_index, const uint32_t* pair_id,
                   const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs, uint64_t total_records) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	const size_t rec = 2 * (size_t) rl + 1;
	uint64_t* all_rows = (uint64_t*) calloc((size_t) G, 8);
	if (!all_rows) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
	if (rl < 1 || rl > VDJX_MAX_READ_LEN) { rc = fail(VDJX_ELIMIT, "read length %d outside [1,%d]", rl, VDJX_MAX_READ_LEN); goto done; }
	if (total_records >= (1ull << 32) - (uint64_t) G) { rc = fail(VDJX_ELIMIT, "%llu records in all: at most 2^32", (unsigned long long) total_records); goto done; }
	HIPC(hipSetDevice(m->device));
	if (m->share_pool) { vdjx_pool_free(m->share_pool); m->share_pool = NULL; }
	if (m->d_share) { (void) hipFree(m->d_share); m->d_share = NULL; }
	if (m->d_reg) { (void) hipFree(m->d_reg); m->d_reg = NULL; }
	if (m->d_scan) { (void) hipFree(m->d_scan); m->d_scan = NULL; }
	if (m->d_info) { (void) hipFree(m->d_info); m->d_info = NULL; }
	m->rl = rl;
	m->total_records = total_records;
	lap_start(m);
	/* the share on the device: the packed pool of the k-mer build AND of the scorers (it keeps reading the quality characters from these
	 * records).  Round 4 dealt the ASCII records out again into slices of the scan order (one all-to-all of 101-byte records: 5 GB per
	 * rank at configs[4], 2.8 times the partial aggregates the build exchanges); the build now takes the share as it is and translates
	 * a first instance through scan_index where it leaves the rank (vdjx_shard_begin_share): no record moves between ranks. */
	{
		/* the local verdict travels WITH the record count: a rank whose share is bad must not leave the collective the others wait in
		 *  -- every rank learns of it and all fail together */
		uint64_t mine[2] = {n_records, 0}, sum = 0;
		size_t bad_at = 0;
		for (size_t i = 0; i < n_records && !mine[1]; i++) {
			const uint64_t g = scan_index[i];
			if (g >= total_records || (i && scan_index[i - 1] >= g)) { mine[1] = 1; bad_at = i; }
		}
		uint64_t* all2 = (uint64_t*) calloc((size_t) G * 2, 8);
		if (!all2) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
		const int rg = vdjx_comm_allgather_host(m->cm, mine, 16, all2);
		int bad_rank = -1;
		for (int r = 0; r < G && !rg; r++) { all_rows[r] = all2[2 * r]; sum += all2[2 * r]; if (all2[2 * r + 1] && bad_rank < 0) bad_rank = r; }
		free(all2);
		if (rg) { rc = fail(rg, "rank %d: vdjx_comm_allgather_host: %s", me, vdjx_comm_last_error()); goto done; }
		if (mine[1]) { rc = fail(VDJX_EINVAL, "rank %d: scan positions of the share must ascend below %llu (record %zu: %llu)", me, (unsigned long long) total_records, bad_at, (unsigned long long) scan_index[bad_at]); goto done; }
		if (bad_rank >= 0) { rc = fail(VDJX_EINVAL, "rank %d: rank %d's share is not in scan order", me, bad_rank); goto done; }
		if (sum != total_records) { rc = fail(VDJX_EINVAL, "rank %d: the shares hold %llu records, the pool %llu (the shares do not cover the pool once)", me, (unsigned long long) sum, (unsigned long long) total_records); goto done; }
	}
	HIPC(hipMalloc(&m->d_share, n_records * rec + 16));
	HIPC(hipMalloc(&m->d_reg, (n_records + 1) * 4));
	HIPC(hipMalloc(&m->d_scan, (n_records + 1) * 4));
	HIPC(hipMalloc(&m->d_info, (n_records + 1) * 6));
	if (n_records) {
		HIPC(hipMemcpy(m->d_info, pair_id, n_records * 4, hipMemcpyHostToDevice));
		HIPC(hipMemcpy((char*) m->d_info + (n_records + 1) * 4, read_num, n_records, hipMemcpyHostToDevice));
		HIPC(hipMemcpy((char*) m->d_info + (n_records + 1) * 5, is_rc, n_records, hipMemcpyHostToDevice));
		HIPC(hipMemcpy(m->d_share, records, n_records * rec, hipMemcpyHostToDevice));
		HIPC(hipMemcpy(m->d_reg, reg_rank, n_records * 4, hipMemcpyHostToDevice));
		HIPC(hipMemcpy(m->d_scan, scan_index, n_records * 4, hipMemcpyHostToDevice));
	}
	lap(m, "load: check + upload of the share");
	VX(vdjx_pool_load_device(ctx, (const uint8_t*) m->d_share, n_records, NULL, 0, rl, &m->share_pool));
	lap(m, "load: pack");
	/* the share's read index is BEGUN here, on the library's index stream: it is built beside the sharded k-mer build (both only read the
	 * packed share) and ended by vdjx_mgpu_kmer_build -- or by the first scorer call */
	VX(vdjx_read_index_build_device_begin(ctx, m->share_pool, (const uint32_t*) m->d_info, (const uint8_t*) m->d_info + (n_records + 1) * 4,
	                                      (const uint8_t*) m->d_info + (n_records + 1) * 5, (const uint32_t*) m->d_reg, n_pairs));
	lap(m, "load: read index of the share begun");
done:
	free(all_rows);
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* the k-mer build                                                                                                     */
/* ------------------------------------------------------------------------------------------------------------------ */
static int kmer_build_any(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, uint64_t rec_stride, const uint32_t* d_scan, uint64_t total_records, vdjx_graph** out) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	vdjx_shard* sh = NULL;
	void *d_sdir = NULL, *d_rdir = NULL, *d_sparts = NULL, *d_rparts = NULL, *d_q = NULL, *d_rq = NULL, *d_ans = NULL, *d_rans = NULL;
	void *d_surv = NULL, *d_surv_all = NULL, *d_mins = NULL, *d_ucnt = NULL;
	uint32_t* h_rdir = NULL;
	uint64_t *send_counts = (uint64_t*) calloc((size_t) G, 8), *recv_counts = (uint64_t*) calloc((size_t) G, 8), *q_out = (uint64_t*) calloc((size_t) G, 8),
	         *q_in = (uint64_t*) calloc((size_t) G, 8), *eq = (uint64_t*) calloc((size_t) G, 8), *meta = (uint64_t*) calloc((size_t) G * (size_t) (G + 2), 8);
	*out = NULL;
	if (!send_counts || !recv_counts || !q_out || !q_in || !eq || !meta) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
	const size_t W0 = vdjx_shard_record_bytes(0), W1 = vdjx_shard_record_bytes(1), W2 = vdjx_shard_record_bytes(2), W3 = vdjx_shard_record_bytes(3);
	HIPC(hipSetDevice(m->device));
	/* a lone rank skips the exchanges that would hand it its own data back (VDJX_MGPU_SELF_COLLECTIVES=1 keeps them: the one-rank RCCL test) */
	const int alone = G == 1 && getenv("VDJX_MGPU_SELF_COLLECTIVES") == NULL;
	lap_start(m);
	if (d_scan) VX(vdjx_shard_begin_share(ctx, pool, k, mf, mq, me, G, d_scan, total_records, &sh));
	else VX(vdjx_shard_begin(ctx, pool, k, mf, mq, me, G, rec_stride, &sh));
	/* 0. the ranks agree on the bucket geometry: the largest number of gated instances any of them holds */
	{
		/* ... and on whether every rank's pool is made of couples (record, reverse complement): then all cut their buckets by the smaller of a
		 * k-mer and its reverse complement and move half the tuples (vdjx_shard_symmetric) */
		uint64_t mine[2] = {0, 0}, most = 0;
		int all_sym = 1;
		VX(vdjx_shard_count(sh, &mine[0]));
		mine[1] = (uint64_t) vdjx_shard_symmetric(sh);
		CX(vdjx_comm_allgather_host(m->cm, mine, 16, meta));
		for (int r = 0; r < G; r++) { if (meta[2 * r] > most) most = meta[2 * r]; if (!meta[2 * r + 1]) all_sym = 0; }
		VX(vdjx_shard_geometry2(sh, most, all_sym));
	}
	lap(m, "build: count + agree on the geometry");
	/* 1. local aggregation; 2. the bulk exchange: per-bucket directories (their sums are the receive counts), then the partials */
	uint32_t dl = 0;
	VX(vdjx_shard_local(sh, send_counts, &dl));
	const size_t ndir = (size_t) G * dl;
	WSG(WS_SDIR, ndir * 4 + 16, &d_sdir);
	WSG(WS_RDIR, ndir * 4 + 16, &d_rdir);
	WSG(WS_SPARTS, sum64(send_counts, G) * W0 + 16, &d_sparts);
	VX(vdjx_shard_local_fill(sh, d_sdir, d_sparts));
	lap(m, "build: local aggregation");
	for (int r = 0; r < G; r++) eq[r] = dl;
	/* what every rank will send this one: the counts travel as control traffic beside the directories (no device round trip in between) */
	CX(vdjx_comm_allgather_host(m->cm, send_counts, (size_t) G * 8, meta));
	for (int r = 0; r < G; r++) recv_counts[r] = meta[(size_t) r * G + me];
	/* (a lone rank has nothing to exchange: what it would send itself IS what it received -- no device copy of its own 0.5 GB of aggregates) */
	if (alone) { d_rdir = d_sdir; d_rparts = d_sparts; }
	else {
		CX(vdjx_comm_a2av(m->cm, d_sdir, eq, d_rdir, eq, 4));
		WSG(WS_RPARTS, sum64(recv_counts, G) * W0 + 16, &d_rparts);
		CX(vdjx_comm_a2av(m->cm, d_sparts, send_counts, d_rparts, recv_counts, W0));
	}
	lap(m, "build: exchange of the partial aggregates");
	/* 3. owners merge and decide; questions and answers for the few open k-mers */
	VX(vdjx_shard_merge(sh, d_rdir, d_rparts, recv_counts, q_out));
	lap(m, "build: owner merge");
	CX(vdjx_comm_allgather_host(m->cm, q_out, (size_t) G * 8, meta));
	for (int r = 0; r < G; r++) q_in[r] = meta[(size_t) r * G + me];
	WSG(WS_Q, sum64(q_out, G) * W1 + 16, &d_q);
	WSG(WS_ANS, sum64(q_in, G) * W2 + 16, &d_ans);
	VX(vdjx_shard_queries(sh, d_q));
	if (alone) d_rq = d_q;
	else {
		WSG(WS_RQ, sum64(q_in, G) * W1 + 16, &d_rq);
		CX(vdjx_comm_a2av(m->cm, d_q, q_out, d_rq, q_in, W1));
	}
	VX(vdjx_shard_reply(sh, d_rq, q_in, d_ans));
	if (alone) d_rans = d_ans;
	else {
		WSG(WS_RANS, sum64(q_out, G) * W2 + 16, &d_rans);
		CX(vdjx_comm_a2av(m->cm, d_ans, q_in, d_rans, q_out, W2));
	}
	uint64_t ns = 0, ndist = 0;
	lap(m, "build: questions and answers");
	VX(vdjx_shard_resolve(sh, d_rans, sum64(q_out, G), &ns, &ndist));
	lap(m, "build: resolve");
	/* 4. survivors everywhere, every rank's share of add_to_graph, MIN / SUM over ranks */
	{
		uint64_t mine[2] = {ns, ndist};
		CX(vdjx_comm_allgather_host(m->cm, mine, 16, meta));
	}
	uint64_t ns_total = 0, pre_total = 0;
	for (int r = 0; r < G; r++) { recv_counts[r] = meta[2 * r]; ns_total += meta[2 * r]; pre_total += meta[2 * r + 1]; }
	WSG(WS_SURV, ns * W3 + 16, &d_surv);
	VX(vdjx_shard_survivors(sh, d_surv));
	if (alone) d_surv_all = d_surv;
	else {
		WSG(WS_SURV_ALL, ns_total * W3 + 16, &d_surv_all);
		CX(vdjx_comm_allgatherv(m->cm, d_surv, d_surv_all, recv_counts, W3));
	}
	lap(m, "build: gather of the survivors");
	WSG(WS_MINS, ns_total * 5 * 8 + 16, &d_mins);          /* in-edge first sights [4n] | node first sights [n] */
	WSG(WS_UCNT, ns_total * 4 + 16, &d_ucnt);
	VX(vdjx_shard_edges(sh, d_surv_all, ns_total, d_mins, d_ucnt, (char*) d_mins + ns_total * 32));
	lap(m, "build: graph pass over the share");
	if (!alone) {
		CX(vdjx_comm_allreduce_min_u64(m->cm, d_mins, (size_t) ns_total * 5));                     /* all-ones = none stays largest */
		CX(vdjx_comm_allreduce_sum_u32(m->cm, d_ucnt, (size_t) ns_total));
	}
	lap(m, "build: reduction of first sights and counts");
	/* 5. node numbering + list order: identical on every rank */
	VX(vdjx_shard_finish(sh, d_mins, d_ucnt, (char*) d_mins + ns_total * 32, pre_total, out));
	lap(m, "build: node numbering");
done:
	if (sh) vdjx_shard_free(sh);
	free(h_rdir); free(send_counts); free(recv_counts); free(q_out); free(q_in); free(eq); free(meta);
	return rc;
}

int vdjx_mgpu_kmer_build_pool(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, uint64_t rec_stride, vdjx_graph** out) {
	return kmer_build_any(m, ctx, pool, k, mf, mq, rec_stride, NULL, 0, out);
}

/* the same over a pool the caller packed itself that is this rank's SHARE of a pool of total_records records (by pair, any dealing that keeps the
 * scan order): record i at scan position d_scan_index[i] (device array, ascending) -- what vdjx_mgpu_load + vdjx_mgpu_kmer_build do for records
 * that start on the host */
int vdjx_mgpu_kmer_build_share(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, const uint32_t* d_scan_index, uint64_t total_records, vdjx_graph** out) {
	if (!d_scan_index) return fail(VDJX_EINVAL, "vdjx_mgpu_kmer_build_share: NULL scan index");
	return kmer_build_any(m, ctx, pool, k, mf, mq, 0, d_scan_index, total_records, out);
}

/* the read length of the pools the scorers work on (vdjx_mgpu_load sets it; a caller that loads its own pools says so here) */
void vdjx_mgpu_set_read_length(vdjx_mgpu* m, int rl) { if (m) m->rl = rl; }

/* the ranks' largest value of `mine` (the record stride of vdjx_mgpu_kmer_build_pool is the largest pool of any rank) */
int vdjx_mgpu_agree_max(vdjx_mgpu* m, uint64_t mine, uint64_t* most) {
	int rc = 0;
	uint64_t* all = (uint64_t*) calloc((size_t) m->nranks, 8);
	if (!all) return fail(VDJX_EINVAL, "out of memory");
	CX(vdjx_comm_allgather_host(m->cm, &mine, 8, all));
	*most = 0;
	for (int r = 0; r < m->nranks; r++) if (all[r] > *most) *most = all[r];
done:
	free(all);
	return rc;
}

int vdjx_mgpu_kmer_build(vdjx_mgpu* m, vdjx_ctx* ctx, int k, int mf, int mq, vdjx_graph** out) {
	if (!m->share_pool) return fail(VDJX_ESTATE, "vdjx_mgpu_kmer_build: call vdjx_mgpu_load first");
	int rc = kmer_build_any(m, ctx, m->share_pool, k, mf, mq, 0, (const uint32_t*) m->d_scan, m->total_records, out);
	lap_start(m);
	const int ri = vdjx_read_index_build_end(ctx);        /* (begun by vdjx_mgpu_load) */
	lap(m, "read index of the share: waited for after the build");
	if (ri && !rc) rc = fail(ri, "rank %d: vdjx_read_index_build_end: %s", m->rank, vdjx_last_error());
	/* a rank builds once and then serves scorer calls: what only the build needed goes back to the device -- its exchange buffers here,
	 * the library's workspaces (the peak of the build: tens of GB per rank at configs[4]) through vdjx_trim */
	(void) hipSetDevice(m->device);
	for (int i = WS_SDIR; i <= WS_UCNT; i++) if (m->ws[i]) { (void) hipFree(m->ws[i]); m->ws[i] = NULL; m->ws_cap[i] = 0; }
	if (rc == 0 && vdjx_trim(ctx) != 0) return fail(VDJX_EHIP, "rank %d: vdjx_trim: %s", m->rank, vdjx_last_error());
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* the window scorer (every rank, the same windows)                                                                    */
/* ------------------------------------------------------------------------------------------------------------------ */
static int do_window_score(vdjx_mgpu* m, vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p, uint8_t* out_valid, uint32_t* out_npairs) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	const size_t nmax = (n + (size_t) G - 1) / (size_t) G, n_mine = n > (size_t) me ? (n - (size_t) me + (size_t) G - 1) / (size_t) G : 0;
	uint32_t *ent = (uint32_t*) calloc(n + 1, 4), *npairs = (uint32_t*) calloc(n + 1, 4), *all_ent = (uint32_t*) calloc((size_t) G * n + 1, 4);
	uint32_t *send_ids = (uint32_t*) calloc(n + 1, 4), *counts = (uint32_t*) calloc((size_t) G * n_mine + 1, 4);
	uint64_t *send_counts = (uint64_t*) calloc((size_t) G, 8), *recv_counts = (uint64_t*) calloc((size_t) G, 8);
	uint8_t *mine = (uint8_t*) calloc(nmax + 1, 1), *all_valid = (uint8_t*) calloc((size_t) G * nmax + 1, 1);
	void *d_send = NULL, *d_recv = NULL;
	if (!ent || !npairs || !all_ent || !send_ids || !counts || !send_counts || !recv_counts || !mine || !all_valid) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
	if (!n) goto done;
	HIPC(hipSetDevice(m->device));
	lap_start(m);
	VX(vdjx_window_pairs(ctx, windows, n, len, ent, npairs));
	lap(m, "windows: pair lists against the share");
	CX(vdjx_comm_allgather_host(m->cm, ent, n * 4, all_ent));
	{
		size_t at = 0;
		for (int o = 0; o < G; o++)
			for (size_t w = (size_t) o; w < n; w += (size_t) G) { send_ids[at++] = (uint32_t) w; send_counts[o] += ent[w]; }
	}
	for (int s = 0; s < G; s++)
		for (size_t j = 0; j < n_mine; j++) {
			const uint32_t v = all_ent[(size_t) s * n + (size_t) me + j * (size_t) G];
			counts[(size_t) s * n_mine + j] = v;
			recv_counts[s] += v;
		}
	WSG(WS_A, sum64(send_counts, G) * 8 + 16, &d_send);
	VX(vdjx_window_pairs_fetch(ctx, send_ids, n, d_send));
	if (G == 1 && !getenv("VDJX_MGPU_SELF_COLLECTIVES")) d_recv = d_send;          /* (a lone rank's lists are the union already) */
	else {
		WSG(WS_B, sum64(recv_counts, G) * 8 + 16, &d_recv);
		CX(vdjx_comm_a2av(m->cm, d_send, send_counts, d_recv, recv_counts, 8));
	}
	lap(m, "windows: exchange of the pair lists");
	if (n_mine) VX(vdjx_window_cover(ctx, n_mine, len, m->rl, p, d_recv, (size_t) G, counts, mine));
	lap(m, "windows: coverage test of the own windows");
	CX(vdjx_comm_allgather_host(m->cm, mine, nmax, all_valid));
	lap(m, "windows: gather of the verdicts");
	for (int o = 0; o < G; o++)
		for (size_t w = (size_t) o, j = 0; w < n; w += (size_t) G, j++) out_valid[w] = all_valid[(size_t) o * nmax + j];
	if (out_npairs) {                     /* mapped pairs per window over all shares (a pair lives on one rank: the counts add) */
		if (G == 1) memcpy(out_npairs, npairs, n * 4);
		else {
			CX(vdjx_comm_allgather_host(m->cm, npairs, n * 4, all_ent));
			for (size_t w = 0; w < n; w++) { uint32_t s = 0; for (int r = 0; r < G; r++) s += all_ent[(size_t) r * n + w]; out_npairs[w] = s; }
		}
		lap(m, "windows: sum of the pair counts");
	}
done:
	free(ent); free(npairs); free(all_ent); free(send_ids); free(counts); free(send_counts); free(recv_counts); free(mine); free(all_valid);
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* the SAM records (every rank, the same contigs; the text arrives on rank 0)                                          */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef int (*sam_sink)(void* ud, const char* text, uint64_t bytes);

static int do_sam(vdjx_mgpu* m, vdjx_ctx* ctx, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off, sam_sink sink, void* ud) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	uint64_t *offs = (uint64_t*) calloc(n + 2, 8), *cnt = (uint64_t*) calloc(n + 1, 8), *all_cnt = (uint64_t*) calloc((size_t) G * n + 1, 8);
	uint64_t *meta = (uint64_t*) calloc((size_t) G * 2 + 2, 8), *send_rows = (uint64_t*) calloc((size_t) G, 8), *recv_rows = (uint64_t*) calloc((size_t) G, 8);
	uint32_t* sub_off = (uint32_t*) calloc(n + 2, 4);
	if (!offs || !cnt || !all_cnt || !meta || !send_rows || !recv_rows || !sub_off) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
	if (!n) goto done;
	HIPC(hipSetDevice(m->device));
	lap_start(m);
	/* pairs per contig over all ranks: the contigs are taken in runs of at most `budget` pairs, so that the text of one run (a few
	 * hundred bytes per pair, on every rank and all of it on rank 0) stays a few GB whatever the pool */
	VX(vdjx_map_emit(ctx, contigs, n, len, offs, NULL));
	for (size_t i = 0; i < n; i++) cnt[i] = offs[i + 1] - offs[i];
	CX(vdjx_comm_allgather_host(m->cm, cnt, n * 8, all_cnt));
	for (size_t i = 0; i < n; i++) { uint64_t s = 0; for (int r = 0; r < G; r++) s += all_cnt[(size_t) r * n + i]; cnt[i] = s; }
	lap(m, "sam: counting pass");
	uint64_t budget = 8u << 20;
	if (getenv("VDJX_MGPU_SAM_PAIRS") && atoll(getenv("VDJX_MGPU_SAM_PAIRS")) > 0) budget = (uint64_t) atoll(getenv("VDJX_MGPU_SAM_PAIRS"));
	for (size_t a = 0; a < n;) {
		size_t b = a;
		uint64_t in_run = 0;
		while (b < n && b - a < (1u << 19) && (b == a || in_run + cnt[b] <= budget)) in_run += cnt[b++];
		for (size_t i = a; i <= b; i++) sub_off[i - a] = id_off[i] - id_off[a];
		uint64_t nb = 0, nbytes = 0;
		const void *dk = NULL, *dl = NULL, *dt = NULL;
		VX(vdjx_sam_blocks(ctx, contigs + a * (size_t) len, b - a, len, ids + id_off[a], sub_off, (const uint32_t*) m->d_reg, &nb, &nbytes, &dk, &dl, &dt));
		uint64_t mine[2] = {nb, nbytes};
		lap(m, "sam: records of the own pairs");
		CX(vdjx_comm_allgather_host(m->cm, mine, 16, meta));
		uint64_t NB = 0, NBY = 0;
		for (int r = 0; r < G; r++) { NB += meta[2 * r]; NBY += meta[2 * r + 1]; }
		void *d_keys = NULL, *d_lens = NULL, *d_text = NULL;
		if (me == 0) {
			WSG(WS_A, NB * 8 + 16, &d_keys);
			WSG(WS_B, NB * 4 + 16, &d_lens);
			WSG(WS_C, NBY + 16, &d_text);
		}
		memset(send_rows, 0, (size_t) G * 8);
		memset(recv_rows, 0, (size_t) G * 8);
		send_rows[0] = nb;
		if (me == 0) for (int r = 0; r < G; r++) recv_rows[r] = meta[2 * r];
		CX(vdjx_comm_a2av(m->cm, dk, send_rows, d_keys, recv_rows, 8));
		CX(vdjx_comm_a2av(m->cm, dl, send_rows, d_lens, recv_rows, 4));
		send_rows[0] = nbytes;
		if (me == 0) for (int r = 0; r < G; r++) recv_rows[r] = meta[2 * r + 1];
		CX(vdjx_comm_a2av(m->cm, dt, send_rows, d_text, recv_rows, 1));
		lap(m, "sam: blocks to rank 0");
		if (me == 0 && NB) {
			const char* text = NULL;
			uint64_t tb = 0;
			VX(vdjx_sam_merge(ctx, NB, NBY, d_keys, d_lens, d_text, &text, &tb));
			if (sink && (rc = sink(ud, text, tb)) != 0) { rc = fail(rc, "the SAM sink failed"); goto done; }
			lap(m, "sam: merge by key on rank 0");
		}
		a = b;
	}
done:
	free(offs); free(cnt); free(all_cnt); free(meta); free(send_rows); free(recv_rows); free(sub_off);
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* rank 0's calls and the other ranks' service loop                                                                    */
/* ------------------------------------------------------------------------------------------------------------------ */
int vdjx_mgpu_window_score(vdjx_mgpu* m, vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p, uint8_t* out_valid) {
	return vdjx_mgpu_window_score2(m, ctx, windows, n, len, p, out_valid, NULL);
}

int vdjx_mgpu_window_score2(vdjx_mgpu* m, vdjx_ctx* ctx, const char* windows, size_t n, int len, const vdjx_cov_params* p, uint8_t* out_valid, uint32_t* out_npairs) {
	int rc = 0;
	if (m->rank != 0) return fail(VDJX_ESTATE, "vdjx_mgpu_window_score is rank 0's call");
	if (!n) return 0;
	if (m->nranks > 1) {
		const uint64_t cmd[4] = {CMD_WINDOWS, n, (uint64_t) len, out_npairs ? 1u : 0u};
		vdjx_cov_params pc = *p;
		CX(vdjx_comm_command_send(m->cm, cmd));
		CX(vdjx_comm_bcast_host(m->cm, &pc, sizeof pc));
		CX(vdjx_comm_bcast_host(m->cm, (void*) windows, n * (size_t) len));
	}
	rc = do_window_score(m, ctx, windows, n, len, p, out_valid, out_npairs);
done:
	return rc;
}

/* rank 0: the other ranks' vdjx_mgpu_serve_step returns (they stay in the job: the next step's collective calls follow) */
int vdjx_mgpu_yield(vdjx_mgpu* m) {
	if (m->rank != 0 || m->nranks == 1) return 0;
	const uint64_t cmd[4] = {CMD_YIELD, 0, 0, 0};
	const int rc = vdjx_comm_command_send(m->cm, cmd);
	return rc ? fail(rc, "%s", vdjx_comm_last_error()) : 0;
}

typedef struct { const char* text; uint64_t bytes; char* own; uint64_t cap; } sam_acc;
static int acc_sink(void* ud, const char* text, uint64_t bytes) {
	sam_acc* a = (sam_acc*) ud;
	if (!a->own && !a->text) { a->text = text; a->bytes = bytes; return 0; }       /* one run: the context's buffer is handed on as it is */
	if (!a->own) {
		a->cap = (a->bytes + bytes) * 2 + 4096;
		a->own = (char*) malloc((size_t) a->cap);
		if (!a->own) return VDJX_EINVAL;
		memcpy(a->own, a->text, (size_t) a->bytes);
	} else if (a->bytes + bytes + 1 > a->cap) {
		a->cap = (a->bytes + bytes) * 2 + 4096;
		char* q = (char*) realloc(a->own, (size_t) a->cap);
		if (!q) return VDJX_EINVAL;
		a->own = q;
	}
	memcpy(a->own + a->bytes, text, (size_t) bytes);
	a->bytes += bytes;
	a->text = a->own;
	return 0;
}

static __thread char* g_sam_own;       /* the text of a call that took several runs (freed by the next call) */

int vdjx_mgpu_sam_body(vdjx_mgpu* m, vdjx_ctx* ctx, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off,
                       const char** out_text, uint64_t* out_bytes) {
	int rc = 0;
	*out_text = ""; *out_bytes = 0;
	if (m->rank != 0) return fail(VDJX_ESTATE, "vdjx_mgpu_sam_body is rank 0's call");
	if (!n) return 0;
	free(g_sam_own);
	g_sam_own = NULL;
	if (m->nranks > 1) {
		const uint64_t cmd[4] = {CMD_SAM, n, (uint64_t) len, id_off[n]};
		CX(vdjx_comm_command_send(m->cm, cmd));
		CX(vdjx_comm_bcast_host(m->cm, (void*) contigs, n * (size_t) len));
		CX(vdjx_comm_bcast_host(m->cm, (void*) id_off, (n + 1) * 4));
		CX(vdjx_comm_bcast_host(m->cm, (void*) ids, id_off[n]));
	}
	sam_acc acc = {NULL, 0, NULL, 0};
	rc = do_sam(m, ctx, contigs, n, len, ids, id_off, acc_sink, &acc);
	if (rc) { free(acc.own); return rc; }
	g_sam_own = acc.own;
	if (acc.text) { *out_text = acc.text; *out_bytes = acc.bytes; }
done:
	return rc;
}

int vdjx_mgpu_finish(vdjx_mgpu* m) {
	if (m->rank == 0 && m->nranks == 1) vdjx_mgpu_print_times(m, stderr);
	if (m->rank != 0 || m->nranks == 1) return 0;
	const uint64_t cmd[4] = {CMD_QUIT, 0, 0, 0};
	vdjx_mgpu_print_times(m, stderr);
	const int rc = vdjx_comm_command_send(m->cm, cmd);
	return rc ? fail(rc, "%s", vdjx_comm_last_error()) : 0;
}

/* out_valid / out_npairs (cap entries, may be NULL): the verdicts and pair counts of the LAST window call served -- every rank computes
 * them all (do_window_score), so a caller that goes on with the same step on every rank (bench.py: the pair emission of the accepted
 * windows against the own share) has them without another exchange.  *released = 1: rank 0 quit the job; 0: it yielded. */
static int serve_impl(vdjx_mgpu* m, vdjx_ctx* ctx, int until_yield, uint8_t* out_valid, uint32_t* out_npairs, size_t cap, size_t* n_out, int* released) {
	int rc = 0;
	char *a = NULL, *b = NULL;
	uint32_t *off = NULL, *np = NULL;
	uint8_t* valid = NULL;
	if (n_out) *n_out = 0;
	if (released) *released = 0;
	if (m->rank == 0) return fail(VDJX_ESTATE, "rank 0 does not serve");
	for (;;) {
		uint64_t cmd[4];
		const int w = vdjx_comm_command_wait(m->cm, cmd);
		if (w) { rc = fail(w < 0 ? w : -5, "rank %d: rank 0 is gone", m->rank); goto done; }
		free(a); free(b); free(off); free(valid); free(np);
		a = b = NULL; off = NULL; valid = NULL; np = NULL;
		if (cmd[0] == CMD_QUIT) { if (released) *released = 1; break; }
		if (cmd[0] == CMD_YIELD) { if (until_yield) break; continue; }
		const size_t n = (size_t) cmd[1];
		const int len = (int) cmd[2];
		if (cmd[0] == CMD_WINDOWS) {
			vdjx_cov_params pc;
			a = (char*) malloc(n * (size_t) len + 1);
			valid = (uint8_t*) malloc(n + 1);
			if (!a || !valid) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
			if (cmd[3]) { np = (uint32_t*) malloc((n + 1) * 4); if (!np) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; } }
			CX(vdjx_comm_bcast_host(m->cm, &pc, sizeof pc));
			CX(vdjx_comm_bcast_host(m->cm, a, n * (size_t) len));
			if ((rc = do_window_score(m, ctx, a, n, len, &pc, valid, np))) goto done;
			if (n_out) {
				*n_out = n;
				if (out_valid && n <= cap) memcpy(out_valid, valid, n);
				if (out_npairs && np && n <= cap) memcpy(out_npairs, np, n * 4);
			}
		} else if (cmd[0] == CMD_SAM) {
			a = (char*) malloc(n * (size_t) len + 1);
			off = (uint32_t*) malloc((n + 1) * 4);
			b = (char*) malloc((size_t) cmd[3] + 1);
			if (!a || !off || !b) { rc = fail(VDJX_EINVAL, "out of memory"); goto done; }
			CX(vdjx_comm_bcast_host(m->cm, a, n * (size_t) len));
			CX(vdjx_comm_bcast_host(m->cm, off, (n + 1) * 4));
			CX(vdjx_comm_bcast_host(m->cm, b, (size_t) cmd[3]));
			if ((rc = do_sam(m, ctx, a, n, len, b, off, NULL, NULL))) goto done;
		} else { rc = fail(VDJX_EINVAL, "rank %d: unknown command %llu", m->rank, (unsigned long long) cmd[0]); goto done; }
	}
done:
	free(a); free(b); free(off); free(valid); free(np);
	return rc;
}

int vdjx_mgpu_serve(vdjx_mgpu* m, vdjx_ctx* ctx) { return serve_impl(m, ctx, 0, NULL, NULL, 0, NULL, NULL); }

int vdjx_mgpu_serve_step(vdjx_mgpu* m, vdjx_ctx* ctx, uint8_t* out_valid, uint32_t* out_npairs, size_t cap, size_t* n_out, int* released) {
	return serve_impl(m, ctx, 1, out_valid, out_npairs, cap, n_out, released);
}
