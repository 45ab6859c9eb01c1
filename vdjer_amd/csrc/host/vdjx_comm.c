/* vdjx_comm.c -- see vdjx_comm.h.  Plain C over the HIP runtime API, RCCL and UNIX sockets. */
#define _GNU_SOURCE
#include "vdjx_comm.h"
#include "vdjx_a2a_plan.h"

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

struct vdjx_comm {
	int rank, G, device, is_rccl;
	int* fds;                       /* [G]: descriptor to rank j, -1 where there is none */
	ncclComm_t comm;
	hipStream_t stream;
	uint64_t bytes_sent;
	double timeout_s;
	char *hs, *hr;                  /* host staging of the "host" transport (grow-only) */
	size_t hs_cap, hr_cap;
	void* d_stage;                  /* device staging of large broadcasts */
	size_t d_cap;
};

static __thread char g_err[512];
static int fail(int rc, const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return rc;
}
const char* vdjx_comm_last_error(void) { return g_err; }

#define E_HIP (-2)
#define E_COMM (-5)
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(E_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NCCLC(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(E_COMM, "%s: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); } while (0)
/* RCCL moved the second half of a 1.09 GB transfer wrongly on this stack (ROCm 7.0.2 / RCCL 2.26.6: a rank sending 34 M partials to
 * itself through all_to_all_single, bytes beyond 2^29 differ, silently; tests/test_gpu_platform.py keeps the reproducer): no single
 * transfer is larger than A2A_CHUNK here, and what a rank sends to itself is a device copy. */
#define A2A_CHUNK ((size_t) 128 << 20)

static double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* sockets                                                                                                             */
/* ------------------------------------------------------------------------------------------------------------------ */
int vdjx_comm_sockets(int G, int mesh, int* fds) {
	for (int i = 0; i < G * G; i++) fds[i] = -1;
	if (mesh && G > VDJX_COMM_MAX_MESH) return fail(-1, "the host transport takes up to %d ranks", VDJX_COMM_MAX_MESH);
	for (int i = 0; i < G; i++)
		for (int j = i + 1; j < G; j++) {
			if (!mesh && i != 0) continue;
			int sv[2];
			if (socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, sv)) return fail(-1, "socketpair: %s", strerror(errno));
			const int big = 4 << 20;
			for (int s = 0; s < 2; s++) {
				(void) setsockopt(sv[s], SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
				(void) setsockopt(sv[s], SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
			}
			fds[i * G + j] = sv[0];
			fds[j * G + i] = sv[1];
		}
	return 0;
}

/* Ranks that are processes ALREADY (one per GPU, started by a launcher: torch.distributed.run under bench.py) meet in a directory all of
 * them know: rank i listens on <dir>/r<i>.sock, every higher rank that needs the pair connects and says who it is.  Same table row as
 * vdjx_comm_sockets would have given the rank. */
static int connect_to(const char* path, double deadline_s) {
	const double t_end = now_s() + deadline_s;
	for (;;) {
		const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
		if (fd < 0) return fail(-1, "socket: %s", strerror(errno));
		struct sockaddr_un a;
		memset(&a, 0, sizeof a);
		a.sun_family = AF_UNIX;
		snprintf(a.sun_path, sizeof a.sun_path, "%s", path);
		if (connect(fd, (struct sockaddr*) &a, sizeof a) == 0) return fd;
		close(fd);
		if (now_s() > t_end) return fail(E_COMM, "no rank listening on %s after %.0f s", path, deadline_s);
		struct timespec ts = {0, 2000000};
		nanosleep(&ts, NULL);
	}
}

int vdjx_comm_rendezvous(const char* dir, int me, int G, int mesh, int* fds_row) {
	for (int j = 0; j < G; j++) fds_row[j] = -1;
	if (G <= 1) return 0;
	if (mesh && G > VDJX_COMM_MAX_MESH) return fail(-1, "the host transport takes up to %d ranks", VDJX_COMM_MAX_MESH);
	double deadline = 120;
	if (getenv("VDJX_MGPU_TIMEOUT_S") && atof(getenv("VDJX_MGPU_TIMEOUT_S")) > 0) deadline = atof(getenv("VDJX_MGPU_TIMEOUT_S"));
	(void) mkdir(dir, 0700);
	char path[108];
	int lfd = -1, expect = 0;
	for (int j = me + 1; j < G; j++) if (mesh || me == 0) expect++;          /* the higher ranks that will call */
	if (expect) {
		if (snprintf(path, sizeof path, "%s/r%d.sock", dir, me) >= (int) sizeof path) return fail(-1, "rendezvous directory name too long");
		(void) unlink(path);
		lfd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
		struct sockaddr_un a;
		memset(&a, 0, sizeof a);
		a.sun_family = AF_UNIX;
		snprintf(a.sun_path, sizeof a.sun_path, "%s", path);
		if (lfd < 0 || bind(lfd, (struct sockaddr*) &a, sizeof a) || listen(lfd, G)) { const int e = errno; if (lfd >= 0) close(lfd); return fail(-1, "listen on %s: %s", path, strerror(e)); }
	}
	int rc = 0;
	for (int i = 0; i < me && !rc; i++) {                                       /* call the lower ranks */
		if (!mesh && i != 0) continue;
		if (snprintf(path, sizeof path, "%s/r%d.sock", dir, i) >= (int) sizeof path) { rc = fail(-1, "rendezvous directory name too long"); break; }
		const int fd = connect_to(path, deadline);
		if (fd < 0) { rc = fd; break; }
		const uint32_t who = (uint32_t) me;
		if (write(fd, &who, 4) != 4) { close(fd); rc = fail(E_COMM, "rank %d: hello to rank %d failed", me, i); break; }
		fds_row[i] = fd;
	}
	for (int n = 0; n < expect && !rc; n++) {
		struct pollfd pf = {lfd, POLLIN, 0};
		const int pr = poll(&pf, 1, (int) (deadline * 1000));
		if (pr <= 0) { rc = fail(E_COMM, "rank %d: %d of %d higher ranks did not call within %.0f s", me, expect - n, expect, deadline); break; }
		const int fd = accept4(lfd, NULL, NULL, SOCK_CLOEXEC);
		uint32_t who = 0;
		if (fd < 0 || read(fd, &who, 4) != 4 || who <= (uint32_t) me || who >= (uint32_t) G || fds_row[who] >= 0) { if (fd >= 0) close(fd); rc = fail(E_COMM, "rank %d: bad hello", me); break; }
		fds_row[who] = fd;
	}
	if (lfd >= 0) { close(lfd); snprintf(path, sizeof path, "%s/r%d.sock", dir, me); (void) unlink(path); }
	const int big = 4 << 20;
	for (int j = 0; j < G; j++) if (fds_row[j] >= 0) { (void) setsockopt(fds_row[j], SOL_SOCKET, SO_SNDBUF, &big, sizeof big); (void) setsockopt(fds_row[j], SOL_SOCKET, SO_RCVBUF, &big, sizeof big); }
	if (rc) for (int j = 0; j < G; j++) if (fds_row[j] >= 0) { close(fds_row[j]); fds_row[j] = -1; }
	return rc;
}

void vdjx_comm_sockets_keep(int G, int me, int* fds) {
	for (int i = 0; i < G; i++)
		for (int j = 0; j < G; j++)
			if (i != me && fds[i * G + j] >= 0) { close(fds[i * G + j]); fds[i * G + j] = -1; }
}

/* moves up to two streams at once (what goes out on fd_w, what comes in on fd_r; either may be absent) until both are done.
 * deadline_s < 0: no deadline.  Returns 0, 1 (the peer closed its end), or < 0. */
static int duplex(int fd_w, const char* wbuf, size_t wlen, int fd_r, char* rbuf, size_t rlen, double deadline_s) {
	const double t_end = deadline_s < 0 ? 0 : now_s() + deadline_s;
	while (wlen || rlen) {
		struct pollfd p[2];
		int n = 0, iw = -1, ir = -1;
		if (wlen) { p[n].fd = fd_w; p[n].events = POLLOUT; p[n].revents = 0; iw = n++; }
		if (rlen) {
			if (wlen && fd_r == fd_w) { p[iw].events |= POLLIN; ir = iw; }
			else { p[n].fd = fd_r; p[n].events = POLLIN; p[n].revents = 0; ir = n++; }
		}
		int to = -1;
		if (deadline_s >= 0) {
			const double left = t_end - now_s();
			if (left <= 0) return fail(E_COMM, "a peer did not answer within %.0f s (VDJX_MGPU_TIMEOUT_S)", deadline_s);
			to = left > 1000 ? 1000000 : (int) (left * 1000) + 1;
		}
		const int pr = poll(p, (nfds_t) n, to);
		if (pr < 0) { if (errno == EINTR) continue; return fail(E_COMM, "poll: %s", strerror(errno)); }
		if (pr == 0) continue;
		if (rlen && (p[ir].revents & (POLLIN | POLLHUP | POLLERR))) {
			const ssize_t k = recv(fd_r, rbuf, rlen, MSG_DONTWAIT);
			if (k == 0) return 1;
			if (k < 0) { if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return fail(E_COMM, "recv: %s", strerror(errno)); }
			else { rbuf += k; rlen -= (size_t) k; }
		}
		if (wlen && (p[iw].revents & (POLLOUT | POLLHUP | POLLERR))) {
			const ssize_t k = send(fd_w, wbuf, wlen, MSG_DONTWAIT | MSG_NOSIGNAL);
			if (k < 0) {
				if (errno == EPIPE || errno == ECONNRESET) return 1;
				if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return fail(E_COMM, "send: %s", strerror(errno));
			} else { wbuf += k; wlen -= (size_t) k; }
		}
	}
	return 0;
}
static int sock_write(vdjx_comm* c, int peer, const void* buf, size_t len) {
	const int rc = duplex(c->fds[peer], (const char*) buf, len, -1, NULL, 0, c->timeout_s);
	return rc == 1 ? fail(E_COMM, "rank %d is gone", peer) : rc;
}
static int sock_read(vdjx_comm* c, int peer, void* buf, size_t len, double deadline) {
	const int rc = duplex(-1, NULL, 0, c->fds[peer], (char*) buf, len, deadline);
	return rc == 1 ? fail(1, "rank %d is gone", peer) : rc;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* life cycle                                                                                                          */
/* ------------------------------------------------------------------------------------------------------------------ */
int vdjx_comm_unique_id(void* out128) {
	ncclUniqueId id;
	if (sizeof id > VDJX_COMM_ID_BYTES) return fail(-1, "ncclUniqueId is %zu bytes", sizeof id);
	const ncclResult_t r = ncclGetUniqueId(&id);
	if (r != ncclSuccess) return fail(E_COMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
	memset(out128, 0, VDJX_COMM_ID_BYTES);
	memcpy(out128, &id, sizeof id);
	return 0;
}

int vdjx_comm_init(const char* transport, int rank, int G, int device, const int* fds_row, const void* rccl_id, vdjx_comm** out) {
	*out = NULL;
	const int is_rccl = !strcmp(transport, "rccl");
	if (!is_rccl && strcmp(transport, "host")) return fail(-1, "transport must be rccl or host, not %s", transport);
	if (G < 1 || rank < 0 || rank >= G) return fail(-1, "bad rank %d of %d", rank, G);
	if (!is_rccl && G > VDJX_COMM_MAX_MESH) return fail(-1, "the host transport takes up to %d ranks", VDJX_COMM_MAX_MESH);
	for (int j = 0; j < G; j++) {
		const int need = j != rank && (!is_rccl || j == 0 || rank == 0);
		if (need && (!fds_row || fds_row[j] < 0)) return fail(-1, "rank %d has no socket to rank %d", rank, j);
	}
	vdjx_comm* c = (vdjx_comm*) calloc(1, sizeof *c);
	if (!c) return fail(-1, "out of memory");
	c->rank = rank; c->G = G; c->device = device; c->is_rccl = is_rccl;
	c->fds = (int*) malloc((size_t) G * sizeof(int));
	for (int j = 0; j < G; j++) c->fds[j] = fds_row ? fds_row[j] : -1;
	c->timeout_s = 600;
	if (getenv("VDJX_MGPU_TIMEOUT_S") && atof(getenv("VDJX_MGPU_TIMEOUT_S")) > 0) c->timeout_s = atof(getenv("VDJX_MGPU_TIMEOUT_S"));
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { fail(E_HIP, "rank %d, device %d: %s", rank, device, hipGetErrorString(e)); free(c->fds); free(c); return E_HIP; }
	if (is_rccl) {
		ncclUniqueId id;
		if (!rccl_id) { free(c->fds); free(c); return fail(-1, "no RCCL id"); }
		memcpy(&id, rccl_id, sizeof id);
		/* RCCL prints a version banner on the process's stdout when the communicator comes up; stdout is the SAM stream: the banner
		 * goes to stderr */
		fflush(stdout);
		const int saved = dup(1);
		if (saved >= 0) (void) dup2(2, 1);
		const ncclResult_t r_ = ncclCommInitRank(&c->comm, G, id, rank);
		fflush(stdout);
		if (saved >= 0) { (void) dup2(saved, 1); close(saved); }
		if (r_ != ncclSuccess) { fail(E_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r_)); (void) hipStreamDestroy(c->stream); free(c->fds); free(c); return E_COMM; }
	}
	*out = c;
	return 0;
}

void vdjx_comm_free(vdjx_comm* c) {
	if (!c) return;
	(void) hipSetDevice(c->device);
	if (c->comm) (void) ncclCommDestroy(c->comm);
	if (c->stream) (void) hipStreamDestroy(c->stream);
	if (c->d_stage) (void) hipFree(c->d_stage);
	free(c->hs); free(c->hr); free(c->fds);
	free(c);
}

int vdjx_comm_rank(const vdjx_comm* c) { return c->rank; }
int vdjx_comm_size(const vdjx_comm* c) { return c->G; }
const char* vdjx_comm_transport(const vdjx_comm* c) { return c->is_rccl ? "rccl" : "host"; }
uint64_t vdjx_comm_bytes_sent(const vdjx_comm* c) { return c ? c->bytes_sent : 0; }

/* the stream's work is done, or the deadline has passed: a collective whose peer hangs is aborted, not waited for */
static int wait_stream(vdjx_comm* c) {
	const double t_end = now_s() + c->timeout_s;
	unsigned spins = 0;
	for (;;) {
		const hipError_t e = hipStreamQuery(c->stream);
		if (e == hipSuccess) break;
		if (e != hipErrorNotReady) return fail(E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
		if (++spins > 2000) {            /* (the first milliseconds are polled hot: exchanges of a sharded build take less) */
			struct timespec ts = {0, 50000};
			nanosleep(&ts, NULL);
			if (now_s() > t_end) {
				if (c->comm) { (void) ncclCommAbort(c->comm); c->comm = NULL; }
				return fail(E_COMM, "rank %d: a collective did not finish within %.0f s (VDJX_MGPU_TIMEOUT_S): aborted", c->rank, c->timeout_s);
			}
		}
	}
	if (c->comm) {
		ncclResult_t as = ncclSuccess;
		if (ncclCommGetAsyncError(c->comm, &as) == ncclSuccess && as != ncclSuccess) return fail(E_COMM, "RCCL: %s", ncclGetErrorString(as));
	}
	return 0;
}

static int stage_host(char** buf, size_t* cap, size_t need) {
	if (need > *cap) {
		free(*buf);
		*cap = need + need / 4 + 4096;
		*buf = (char*) malloc(*cap);
		if (!*buf) { *cap = 0; return fail(-1, "out of host memory (%zu bytes of exchange staging)", need); }
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* bulk                                                                                                                */
/* ------------------------------------------------------------------------------------------------------------------ */


/* the "host" transport's exchange of host buffers: in step t a rank sends to me+t and receives from me-t */
static int host_a2a(vdjx_comm* c, const char* hs, const uint64_t* sb, const uint64_t* so, char* hr, const uint64_t* rb, const uint64_t* ro) {
	const int G = c->G, me = c->rank;
	if (sb[me]) memcpy(hr + ro[me], hs + so[me], (size_t) (sb[me] < rb[me] ? sb[me] : rb[me]));
	for (int t = 1; t < G; t++) {
		const int to = (me + t) % G, from = (me - t + G) % G;
		const int rc = duplex(c->fds[to], hs + so[to], (size_t) sb[to], c->fds[from], hr + ro[from], (size_t) rb[from], c->timeout_s);
		if (rc) return rc == 1 ? fail(E_COMM, "rank %d: a peer is gone", me) : rc;
		c->bytes_sent += sb[to];
	}
	return 0;
}

int vdjx_comm_a2av(vdjx_comm* c, const void* d_send, const uint64_t* send_rows, void* d_recv, const uint64_t* recv_rows, size_t row) {
	const int G = c->G, me = c->rank;
	HIPC(hipSetDevice(c->device));
	if (!c->is_rccl) {
		uint64_t sb[VDJX_COMM_MAX_MESH], so[VDJX_COMM_MAX_MESH + 1], rb[VDJX_COMM_MAX_MESH], ro[VDJX_COMM_MAX_MESH + 1];
		so[0] = ro[0] = 0;
		for (int r = 0; r < G; r++) { sb[r] = send_rows[r] * row; rb[r] = recv_rows[r] * row; so[r + 1] = so[r] + sb[r]; ro[r + 1] = ro[r] + rb[r]; }
		int rc;
		if ((rc = stage_host(&c->hs, &c->hs_cap, (size_t) so[G] + 1)) || (rc = stage_host(&c->hr, &c->hr_cap, (size_t) ro[G] + 1))) return rc;
		if (so[G]) HIPC(hipMemcpy(c->hs, d_send, (size_t) so[G], hipMemcpyDeviceToHost));
		if ((rc = host_a2a(c, c->hs, sb, so, c->hr, rb, ro))) return rc;
		if (ro[G]) HIPC(hipMemcpy(d_recv, c->hr, (size_t) ro[G], hipMemcpyHostToDevice));
		return 0;
	}
	uint64_t self[3] = {0, 0, 0};
	const size_t nst = vdjx_a2a_plan(G, me, send_rows, recv_rows, row, A2A_CHUNK, NULL, 0, self);         /* (the offsets: vdjx_a2a_plan.c, tested on the CPU) */
	vdjx_a2a_step* st = (vdjx_a2a_step*) calloc(nst + 1, sizeof *st);
	if (!st) return fail(-1, "out of memory");
	(void) vdjx_a2a_plan(G, me, send_rows, recv_rows, row, A2A_CHUNK, st, nst, self);
	for (int r = 0; r < G; r++) if (r != me) c->bytes_sent += send_rows[r] * row;
	int rc = 0;
	hipError_t he = hipSuccess;
	ncclResult_t ne = ncclSuccess;
	if (self[2]) he = hipMemcpyAsync((char*) d_recv + self[1], (const char*) d_send + self[0], (size_t) self[2], hipMemcpyDeviceToDevice, c->stream);
	for (size_t i = 0; i < nst && he == hipSuccess && ne == ncclSuccess;) {
		const uint32_t round = st[i].round;
		ne = ncclGroupStart();
		for (; i < nst && st[i].round == round && ne == ncclSuccess; i++) {
			if (st[i].send_len) ne = ncclSend((const char*) d_send + st[i].send_off, (size_t) st[i].send_len, ncclChar, st[i].peer, c->comm, c->stream);
			if (st[i].recv_len && ne == ncclSuccess) ne = ncclRecv((char*) d_recv + st[i].recv_off, (size_t) st[i].recv_len, ncclChar, st[i].peer, c->comm, c->stream);
		}
		const ncclResult_t ge = ncclGroupEnd();
		if (ne == ncclSuccess) ne = ge;
	}
	free(st);
	if (he != hipSuccess) return fail(E_HIP, "a2av: %s", hipGetErrorString(he));
	if (ne != ncclSuccess) return fail(E_COMM, "a2av: %s", ncclGetErrorString(ne));
	rc = wait_stream(c);
	return rc;
}

int vdjx_comm_allgatherv(vdjx_comm* c, const void* d_send, void* d_recv, const uint64_t* rows, size_t row) {
	const int G = c->G, me = c->rank;
	uint64_t* sr = (uint64_t*) malloc((size_t) G * 8);
	if (!sr) return fail(-1, "out of memory");
	HIPC(hipSetDevice(c->device));
	int rc = 0;
	if (!c->is_rccl) {
		/* the same rows to every peer: the send side of the exchange points every peer at the one block */
		uint64_t sb[VDJX_COMM_MAX_MESH], so[VDJX_COMM_MAX_MESH + 1], rb[VDJX_COMM_MAX_MESH], ro[VDJX_COMM_MAX_MESH + 1];
		ro[0] = 0;
		for (int r = 0; r < G; r++) { sb[r] = rows[me] * row; so[r] = 0; rb[r] = rows[r] * row; ro[r + 1] = ro[r] + rb[r]; }
		if (!(rc = stage_host(&c->hs, &c->hs_cap, (size_t) sb[me] + 1)) && !(rc = stage_host(&c->hr, &c->hr_cap, (size_t) ro[G] + 1))) {
			hipError_t e = sb[me] ? hipMemcpy(c->hs, d_send, (size_t) sb[me], hipMemcpyDeviceToHost) : hipSuccess;
			if (e == hipSuccess) rc = host_a2a(c, c->hs, sb, so, c->hr, rb, ro);
			if (e == hipSuccess && !rc && ro[G]) e = hipMemcpy(d_recv, c->hr, (size_t) ro[G], hipMemcpyHostToDevice);
			if (e != hipSuccess) rc = fail(E_HIP, "allgatherv: %s", hipGetErrorString(e));
		}
		free(sr);
		return rc;
	}
	/* every rank's rows to every rank as sends and receives, in pieces of at most A2A_CHUNK (see above); the own ones by a device copy */
	const size_t mine = (size_t) rows[me] * row;
	size_t rounds = 0, ro = 0;
	for (int r = 0; r < G; r++) {
		const size_t a = ((size_t) rows[r] * row + A2A_CHUNK - 1) / A2A_CHUNK;
		if (G > 1 && a > rounds) rounds = a;
	}
	for (int r = 0; r < me; r++) ro += (size_t) rows[r] * row;
	hipError_t he = mine ? hipMemcpyAsync((char*) d_recv + ro, d_send, mine, hipMemcpyDeviceToDevice, c->stream) : hipSuccess;
	ncclResult_t ne = ncclSuccess;
	for (size_t rd = 0; rd < rounds && he == hipSuccess && ne == ncclSuccess; rd++) {
		const size_t a = rd * A2A_CHUNK;
		ro = 0;
		ne = ncclGroupStart();
		for (int r = 0; r < G && ne == ncclSuccess; r++) {
			const size_t rb = (size_t) rows[r] * row;
			if (r != me) {
				if (a < mine) ne = ncclSend((const char*) d_send + a, mine - a < A2A_CHUNK ? mine - a : A2A_CHUNK, ncclChar, r, c->comm, c->stream);
				if (a < rb && ne == ncclSuccess) ne = ncclRecv((char*) d_recv + ro + a, rb - a < A2A_CHUNK ? rb - a : A2A_CHUNK, ncclChar, r, c->comm, c->stream);
			}
			ro += rb;
		}
		const ncclResult_t ge = ncclGroupEnd();
		if (ne == ncclSuccess) ne = ge;
	}
	free(sr);
	for (int r = 0; r < G; r++) if (r != me) c->bytes_sent += mine;
	if (he != hipSuccess) return fail(E_HIP, "allgatherv: %s", hipGetErrorString(he));
	if (ne != ncclSuccess) return fail(E_COMM, "allgatherv: %s", ncclGetErrorString(ne));
	return wait_stream(c);
}

/* the "host" transport's reductions: every rank's array to rank 0, which folds them and sends the result back */
static int host_allreduce(vdjx_comm* c, void* d_buf, size_t n, int is_min64) {
	const size_t bytes = n * (is_min64 ? 8 : 4);
	int rc;
	if (!bytes) return 0;
	if ((rc = stage_host(&c->hs, &c->hs_cap, bytes)) || (rc = stage_host(&c->hr, &c->hr_cap, bytes))) return rc;
	HIPC(hipMemcpy(c->hs, d_buf, bytes, hipMemcpyDeviceToHost));
	if (c->rank == 0) {
		for (int r = 1; r < c->G; r++) {
			if ((rc = sock_read(c, r, c->hr, bytes, c->timeout_s))) return rc;
			if (is_min64) { uint64_t *a = (uint64_t*) c->hs, *b = (uint64_t*) c->hr; for (size_t i = 0; i < n; i++) if (b[i] < a[i]) a[i] = b[i]; }
			else { uint32_t *a = (uint32_t*) c->hs, *b = (uint32_t*) c->hr; for (size_t i = 0; i < n; i++) a[i] += b[i]; }
		}
		for (int r = 1; r < c->G; r++) if ((rc = sock_write(c, r, c->hs, bytes))) return rc;
	} else {
		if ((rc = sock_write(c, 0, c->hs, bytes))) return rc;
		c->bytes_sent += bytes;
		if ((rc = sock_read(c, 0, c->hs, bytes, c->timeout_s))) return rc;
	}
	HIPC(hipMemcpy(d_buf, c->hs, bytes, hipMemcpyHostToDevice));
	return 0;
}

int vdjx_comm_allreduce_min_u64(vdjx_comm* c, void* d_buf, size_t n) {
	HIPC(hipSetDevice(c->device));
	if (c->G == 1 || !n) return 0;
	if (!c->is_rccl) return host_allreduce(c, d_buf, n, 1);
	for (size_t a = 0; a < n; a += A2A_CHUNK / 8) {
		const size_t m = n - a < A2A_CHUNK / 8 ? n - a : A2A_CHUNK / 8;
		NCCLC(ncclAllReduce((char*) d_buf + a * 8, (char*) d_buf + a * 8, m, ncclUint64, ncclMin, c->comm, c->stream));
	}
	c->bytes_sent += n * 8;
	return wait_stream(c);
}

int vdjx_comm_allreduce_sum_u32(vdjx_comm* c, void* d_buf, size_t n) {
	HIPC(hipSetDevice(c->device));
	if (c->G == 1 || !n) return 0;
	if (!c->is_rccl) return host_allreduce(c, d_buf, n, 0);
	for (size_t a = 0; a < n; a += A2A_CHUNK / 4) {
		const size_t m = n - a < A2A_CHUNK / 4 ? n - a : A2A_CHUNK / 4;
		NCCLC(ncclAllReduce((char*) d_buf + a * 4, (char*) d_buf + a * 4, m, ncclUint32, ncclSum, c->comm, c->stream));
	}
	c->bytes_sent += n * 4;
	return wait_stream(c);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* control                                                                                                             */
/* ------------------------------------------------------------------------------------------------------------------ */
int vdjx_comm_allgather_host(vdjx_comm* c, const void* mine, size_t bytes, void* all) {
	int rc;
	memcpy((char*) all + (size_t) c->rank * bytes, mine, bytes);
	if (c->G == 1 || !bytes) return 0;
	if (c->rank == 0) {
		for (int r = 1; r < c->G; r++) if ((rc = sock_read(c, r, (char*) all + (size_t) r * bytes, bytes, c->timeout_s))) return rc;
		for (int r = 1; r < c->G; r++) if ((rc = sock_write(c, r, all, bytes * (size_t) c->G))) return rc;
		return 0;
	}
	if ((rc = sock_write(c, 0, mine, bytes))) return rc;
	return sock_read(c, 0, all, bytes * (size_t) c->G, c->timeout_s);
}

int vdjx_comm_bcast_host(vdjx_comm* c, void* buf, size_t bytes) {
	int rc;
	if (c->G == 1 || !bytes) return 0;
	if (c->is_rccl && bytes >= ((size_t) 1 << 20)) {
		HIPC(hipSetDevice(c->device));
		if (bytes > c->d_cap) {
			if (c->d_stage) (void) hipFree(c->d_stage);
			c->d_stage = NULL; c->d_cap = 0;
			HIPC(hipMalloc(&c->d_stage, bytes + bytes / 4));
			c->d_cap = bytes + bytes / 4;
		}
		if (c->rank == 0) HIPC(hipMemcpyAsync(c->d_stage, buf, bytes, hipMemcpyHostToDevice, c->stream));
		for (size_t a = 0; a < bytes; a += A2A_CHUNK) {
			const size_t m = bytes - a < A2A_CHUNK ? bytes - a : A2A_CHUNK;
			NCCLC(ncclBroadcast((char*) c->d_stage + a, (char*) c->d_stage + a, m, ncclChar, 0, c->comm, c->stream));
		}
		if (c->rank != 0) HIPC(hipMemcpyAsync(buf, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
		else c->bytes_sent += bytes;
		return wait_stream(c);
	}
	if (c->rank == 0) {
		for (int r = 1; r < c->G; r++) if ((rc = sock_write(c, r, buf, bytes))) return rc;
		return 0;
	}
	return sock_read(c, 0, buf, bytes, c->timeout_s);
}

int vdjx_comm_command_send(vdjx_comm* c, const uint64_t cmd[4]) {
	int rc;
	if (c->rank != 0) return fail(-1, "only rank 0 sends commands");
	for (int r = 1; r < c->G; r++) if ((rc = sock_write(c, r, cmd, 32))) return rc;
	return 0;
}

int vdjx_comm_command_wait(vdjx_comm* c, uint64_t cmd[4]) {
	if (c->rank == 0) return fail(-1, "rank 0 does not wait for commands");
	return sock_read(c, 0, cmd, 32, -1.0);
}

int vdjx_comm_status(vdjx_comm* c, int mine, int* worst) {
	int rc, w = mine;
	if (c->rank == 0) {
		for (int r = 1; r < c->G; r++) {
			int s = 0;
			if ((rc = sock_read(c, r, &s, sizeof s, c->timeout_s))) return rc;
			if (s && !w) w = s;
		}
		if (worst) *worst = w;
		return 0;
	}
	if (worst) *worst = mine;
	return sock_write(c, 0, &mine, sizeof mine);
}
