/*
 * vdjx_mgpu.c -- the sharded k-mer build driven from C over RCCL (one process per GPU, xGMI): what `vdjer --gpus N` runs.
 *
 * The reference has no counterpart (its only parallelism is pthreads over roots, A2:1287-1348).  This is the C twin of
 * vdjer_amd/shard.py: the compute is the vdjx_shard_* phases of libvdjx (include/vdjx.h), this file only moves their bytes
 * between ranks.  Every exchange is ONE grouped set of ncclSend/ncclRecv (each peer on its own xGMI link):
 *   directories (their sums ARE the receive counts) -> partial aggregates -> questions -> answers -> survivors (all-gather-v)
 * followed by two all-reduces of the per-survivor arrays (MIN over unsigned 64-bit first sights, SUM over the counts).
 */
#include "vdjx_mgpu.h"
#include "vdjx_a2a_plan.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

enum { WS_SDIR, WS_RDIR, WS_SPARTS, WS_RPARTS, WS_META, WS_Q, WS_RQ, WS_ANS, WS_RANS, WS_SURV, WS_SURV_ALL, WS_MINS, WS_UCNT, WS_SLOTS };
struct vdjx_mgpu {
	int rank, nranks, device;
	ncclComm_t comm;
	hipStream_t stream;
	uint64_t bytes_sent;
	/* the build's exchange buffers: one per purpose, kept by the handle and only replaced when a build needs more (a process that
	 * builds again -- a second chain, the next sample -- allocates nothing) */
	void* ws[WS_SLOTS];
	size_t ws_cap[WS_SLOTS];
};

static __thread char g_err[512];
static void set_err(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}
const char* vdjx_mgpu_last_error(void) { return g_err; }

/* buffer `slot` of at least `bytes` bytes (contents undefined) */
static int ws_get(vdjx_mgpu* m, int slot, size_t bytes, void** out) {
	if (bytes > m->ws_cap[slot]) {
		if (m->ws[slot]) (void) hipFree(m->ws[slot]);
		m->ws[slot] = NULL; m->ws_cap[slot] = 0;
		const size_t want = bytes + bytes / 8 + 256;
		const hipError_t e = hipMalloc(&m->ws[slot], want);
		if (e != hipSuccess) { set_err("hipMalloc of %zu bytes (exchange buffer %d): %s", want, slot, hipGetErrorString(e)); return VDJX_EHIP; }
		m->ws_cap[slot] = want;
	}
	*out = m->ws[slot];
	return 0;
}

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); rc = -2; goto done; } } while (0)
#define NCCLC(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { set_err("%s: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); rc = -5; goto done; } } while (0)
#define VX(x) do { int r_ = (x); if (r_ != 0) { set_err("%s: %s", #x, vdjx_last_error()); rc = r_; goto done; } } while (0)
#define WSG(slot, bytes, out) do { void* p_ = NULL; const int r_ = ws_get(m, slot, (size_t) (bytes), &p_); if (r_) { rc = r_; goto done; } *(out) = p_; } while (0)

int vdjx_mgpu_unique_id(void* out128) {
	ncclUniqueId id;
	if (sizeof id > VDJX_MGPU_ID_BYTES) { set_err("ncclUniqueId is %zu bytes", sizeof id); return -1; }
	ncclResult_t r = ncclGetUniqueId(&id);
	if (r != ncclSuccess) { set_err("ncclGetUniqueId: %s", ncclGetErrorString(r)); return -5; }
	memset(out128, 0, VDJX_MGPU_ID_BYTES);
	memcpy(out128, &id, sizeof id);
	return 0;
}

int vdjx_mgpu_init(int rank, int nranks, int device, const void* unique_id, vdjx_mgpu** out) {
	int rc = 0;
	*out = NULL;
	vdjx_mgpu* m = (vdjx_mgpu*) calloc(1, sizeof *m);
	m->rank = rank; m->nranks = nranks; m->device = device;
	ncclUniqueId id;
	memcpy(&id, unique_id, sizeof id);
	HIPC(hipSetDevice(device));
	HIPC(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
	{
		/* RCCL prints a version banner on the process's stdout when the communicator comes up; stdout is the SAM stream: the banner
		 * goes to stderr */
		fflush(stdout);
		const int saved = dup(1);
		if (saved >= 0) (void) dup2(2, 1);
		const ncclResult_t r_ = ncclCommInitRank(&m->comm, nranks, id, rank);
		fflush(stdout);
		if (saved >= 0) { (void) dup2(saved, 1); close(saved); }
		if (r_ != ncclSuccess) { set_err("ncclCommInitRank: %s", ncclGetErrorString(r_)); rc = -5; goto done; }
	}
	*out = m;
	return 0;
done:
	free(m);
	return rc;
}

void vdjx_mgpu_free(vdjx_mgpu* m) {
	if (!m) return;
	(void) hipSetDevice(m->device);
	if (m->comm) (void) ncclCommDestroy(m->comm);
	if (m->stream) (void) hipStreamDestroy(m->stream);
	for (int i = 0; i < WS_SLOTS; i++) if (m->ws[i]) (void) hipFree(m->ws[i]);
	free(m);
}

uint64_t vdjx_mgpu_bytes_sent(const vdjx_mgpu* m) { return m ? m->bytes_sent : 0; }

/* all-to-all-v of byte rows: send[r] rows of `row` bytes to rank r (contiguous, rank order), recv likewise */
/* RCCL moved the second half of a 1.09 GB transfer wrongly on this stack (ROCm 7.0.2 / RCCL 2.26.6: a rank sending 34 M partials to
 * itself through all_to_all_single, bytes beyond 2^29 differ, silently): no single transfer is larger than A2A_CHUNK here, and what a
 * rank sends to itself is a device copy. */
#define A2A_CHUNK ((size_t) 128 << 20)
static int a2av(vdjx_mgpu* m, const void* d_send, const uint64_t* send_rows, void* d_recv, const uint64_t* recv_rows, size_t row) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	uint64_t self[3] = {0, 0, 0};
	const size_t nst = vdjx_a2a_plan(G, me, send_rows, recv_rows, row, A2A_CHUNK, NULL, 0, self);         /* (the offsets: vdjx_a2a_plan.c, tested on the CPU) */
	vdjx_a2a_step* st = (vdjx_a2a_step*) calloc(nst + 1, sizeof *st);
	if (!st) { rc = VDJX_EHIP; goto done; }
	(void) vdjx_a2a_plan(G, me, send_rows, recv_rows, row, A2A_CHUNK, st, nst, self);
	for (int r = 0; r < G; r++) if (r != me) m->bytes_sent += send_rows[r] * row;
	if (self[2]) HIPC(hipMemcpyAsync((char*) d_recv + self[1], (const char*) d_send + self[0], (size_t) self[2], hipMemcpyDeviceToDevice, m->stream));
	for (size_t i = 0; i < nst;) {
		const uint32_t round = st[i].round;
		NCCLC(ncclGroupStart());
		for (; i < nst && st[i].round == round; i++) {
			if (st[i].send_len) NCCLC(ncclSend((const char*) d_send + st[i].send_off, (size_t) st[i].send_len, ncclChar, st[i].peer, m->comm, m->stream));
			if (st[i].recv_len) NCCLC(ncclRecv((char*) d_recv + st[i].recv_off, (size_t) st[i].recv_len, ncclChar, st[i].peer, m->comm, m->stream));
		}
		NCCLC(ncclGroupEnd());
	}
	HIPC(hipStreamSynchronize(m->stream));
done:
	free(st);
	return rc;
}

static uint64_t sum64(const uint64_t* v, int n) { uint64_t s = 0; for (int i = 0; i < n; i++) s += v[i]; return s; }

int vdjx_mgpu_kmer_build(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, uint64_t rec_stride, vdjx_graph** out) {
	int rc = 0;
	const int G = m->nranks, me = m->rank;
	vdjx_shard* sh = NULL;
	void *d_sdir = NULL, *d_rdir = NULL, *d_sparts = NULL, *d_rparts = NULL, *d_q = NULL, *d_rq = NULL, *d_ans = NULL, *d_rans = NULL;
	void *d_meta = NULL, *d_surv = NULL, *d_surv_all = NULL, *d_mins = NULL, *d_ucnt = NULL;
	uint32_t* h_rdir = NULL;
	uint64_t *send_counts = (uint64_t*) calloc((size_t) G, 8), *recv_counts = (uint64_t*) calloc((size_t) G, 8), *q_out = (uint64_t*) calloc((size_t) G, 8),
	         *q_in = (uint64_t*) calloc((size_t) G, 8), *eq = (uint64_t*) calloc((size_t) G, 8), *meta = (uint64_t*) calloc((size_t) G * (size_t) (G + 2), 8);
	*out = NULL;
	const size_t W0 = vdjx_shard_record_bytes(0), W1 = vdjx_shard_record_bytes(1), W2 = vdjx_shard_record_bytes(2), W3 = vdjx_shard_record_bytes(3);
	HIPC(hipSetDevice(m->device));
	VX(vdjx_shard_begin(ctx, pool, k, mf, mq, me, G, rec_stride, &sh));
	/* 1. local aggregation; 2. the bulk exchange: per-bucket directories (their sums are the receive counts), then the partials */
	uint32_t dl = 0;
	VX(vdjx_shard_local(sh, send_counts, &dl));
	const size_t ndir = (size_t) G * dl;
	WSG(WS_SDIR, ndir * 4 + 16, &d_sdir);
	WSG(WS_RDIR, ndir * 4 + 16, &d_rdir);
	WSG(WS_SPARTS, sum64(send_counts, G) * W0 + 16, &d_sparts);
	VX(vdjx_shard_local_fill(sh, d_sdir, d_sparts));
	for (int r = 0; r < G; r++) eq[r] = dl;
	if ((rc = a2av(m, d_sdir, eq, d_rdir, eq, 4))) goto done;
	h_rdir = (uint32_t*) malloc(ndir * 4 + 4);
	HIPC(hipMemcpy(h_rdir, d_rdir, ndir * 4, hipMemcpyDeviceToHost));
	for (int r = 0; r < G; r++) {
		uint64_t s = 0;
		for (uint32_t i = 0; i < dl; i++) s += h_rdir[(size_t) r * dl + i];
		recv_counts[r] = s;
	}
	WSG(WS_RPARTS, sum64(recv_counts, G) * W0 + 16, &d_rparts);
	if ((rc = a2av(m, d_sparts, send_counts, d_rparts, recv_counts, W0))) goto done;
	/* 3. owners merge and decide; questions and answers for the few open k-mers */
	VX(vdjx_shard_merge(sh, d_rdir, d_rparts, recv_counts, q_out));
	/* everybody learns everybody's question counts, survivor count and distinct count in one small all-gather */
	WSG(WS_META, (size_t) G * (size_t) (G + 2) * 8 + 16, &d_meta);
	{
		uint64_t* mine = meta + (size_t) me * (G + 2);
		memcpy(mine, q_out, (size_t) G * 8);
		mine[G] = mine[G + 1] = 0;
		HIPC(hipMemcpy((char*) d_meta + (size_t) me * (G + 2) * 8, mine, (size_t) (G + 2) * 8, hipMemcpyHostToDevice));
		NCCLC(ncclAllGather((char*) d_meta + (size_t) me * (G + 2) * 8, d_meta, (size_t) (G + 2), ncclUint64, m->comm, m->stream));
		HIPC(hipStreamSynchronize(m->stream));
		HIPC(hipMemcpy(meta, d_meta, (size_t) G * (G + 2) * 8, hipMemcpyDeviceToHost));
		for (int r = 0; r < G; r++) q_in[r] = meta[(size_t) r * (G + 2) + me];
	}
	WSG(WS_Q, sum64(q_out, G) * W1 + 16, &d_q);
	WSG(WS_RQ, sum64(q_in, G) * W1 + 16, &d_rq);
	WSG(WS_ANS, sum64(q_in, G) * W2 + 16, &d_ans);
	WSG(WS_RANS, sum64(q_out, G) * W2 + 16, &d_rans);
	VX(vdjx_shard_queries(sh, d_q));
	if ((rc = a2av(m, d_q, q_out, d_rq, q_in, W1))) goto done;
	VX(vdjx_shard_reply(sh, d_rq, q_in, d_ans));
	if ((rc = a2av(m, d_ans, q_in, d_rans, q_out, W2))) goto done;
	uint64_t ns = 0, ndist = 0;
	VX(vdjx_shard_resolve(sh, d_rans, sum64(q_out, G), &ns, &ndist));
	/* 4. survivors everywhere (all-gather-v as sends and receives), every rank's share of add_to_graph, MIN / SUM over ranks */
	{
		uint64_t mine[2] = {ns, ndist};
		HIPC(hipMemcpy((char*) d_meta + (size_t) me * 16, mine, 16, hipMemcpyHostToDevice));
		NCCLC(ncclAllGather((char*) d_meta + (size_t) me * 16, d_meta, 2, ncclUint64, m->comm, m->stream));
		HIPC(hipStreamSynchronize(m->stream));
		HIPC(hipMemcpy(meta, d_meta, (size_t) G * 16, hipMemcpyDeviceToHost));
	}
	uint64_t ns_total = 0, pre_total = 0;
	for (int r = 0; r < G; r++) { recv_counts[r] = meta[2 * r]; ns_total += meta[2 * r]; pre_total += meta[2 * r + 1]; send_counts[r] = ns; }
	WSG(WS_SURV, ns * W3 + 16, &d_surv);
	WSG(WS_SURV_ALL, ns_total * W3 + 16, &d_surv_all);
	VX(vdjx_shard_survivors(sh, d_surv));
	{
		/* every rank's survivors to every rank, in pieces of at most A2A_CHUNK (see a2av); the own ones by a device copy */
		size_t ro = 0, rounds = 0;
		for (int r = 0; r < G; r++) {
			const size_t a = ((size_t) recv_counts[r] * W3 + A2A_CHUNK - 1) / A2A_CHUNK;
			if (r != me && a > rounds) rounds = a;
		}
		{
			const size_t a = ((size_t) ns * W3 + A2A_CHUNK - 1) / A2A_CHUNK;
			if (G > 1 && a > rounds) rounds = a;
		}
		for (int r = 0; r < me; r++) ro += recv_counts[r] * W3;
		if (ns) HIPC(hipMemcpyAsync((char*) d_surv_all + ro, d_surv, ns * W3, hipMemcpyDeviceToDevice, m->stream));
		for (size_t rd = 0; rd < rounds; rd++) {
			const size_t a = rd * A2A_CHUNK;
			ro = 0;
			NCCLC(ncclGroupStart());
			for (int r = 0; r < G; r++) {
				const size_t sb = (size_t) ns * W3, rb = (size_t) recv_counts[r] * W3;
				if (r != me) {
					if (a < sb) NCCLC(ncclSend((const char*) d_surv + a, sb - a < A2A_CHUNK ? sb - a : A2A_CHUNK, ncclChar, r, m->comm, m->stream));
					if (a < rb) NCCLC(ncclRecv((char*) d_surv_all + ro + a, rb - a < A2A_CHUNK ? rb - a : A2A_CHUNK, ncclChar, r, m->comm, m->stream));
				}
				ro += rb;
			}
			NCCLC(ncclGroupEnd());
		}
		for (int r = 0; r < G; r++) if (r != me) m->bytes_sent += ns * W3;
		HIPC(hipStreamSynchronize(m->stream));
	}
	WSG(WS_MINS, ns_total * 5 * 8 + 16, &d_mins);          /* in-edge first sights [4n] | node first sights [n] */
	WSG(WS_UCNT, ns_total * 4 + 16, &d_ucnt);
	VX(vdjx_shard_edges(sh, d_surv_all, ns_total, d_mins, d_ucnt, (char*) d_mins + ns_total * 32));
	if (ns_total) {
		for (size_t a = 0; a < ns_total * 5; a += A2A_CHUNK / 8) {                                       /* all-ones = none stays largest */
			const size_t n = ns_total * 5 - a < A2A_CHUNK / 8 ? ns_total * 5 - a : A2A_CHUNK / 8;
			NCCLC(ncclAllReduce((char*) d_mins + a * 8, (char*) d_mins + a * 8, n, ncclUint64, ncclMin, m->comm, m->stream));
		}
		for (size_t a = 0; a < ns_total; a += A2A_CHUNK / 4) {
			const size_t n = ns_total - a < A2A_CHUNK / 4 ? ns_total - a : A2A_CHUNK / 4;
			NCCLC(ncclAllReduce((char*) d_ucnt + a * 4, (char*) d_ucnt + a * 4, n, ncclUint32, ncclSum, m->comm, m->stream));
		}
		HIPC(hipStreamSynchronize(m->stream));
		m->bytes_sent += (uint64_t) (G > 1) * ns_total * 44;
	}
	/* 5. node numbering + list order: identical on every rank */
	VX(vdjx_shard_finish(sh, d_mins, d_ucnt, (char*) d_mins + ns_total * 32, pre_total, out));
done:
	if (sh) vdjx_shard_free(sh);
	free(h_rdir); free(send_counts); free(recv_counts); free(q_out); free(q_in); free(eq); free(meta);
	return rc;
}
