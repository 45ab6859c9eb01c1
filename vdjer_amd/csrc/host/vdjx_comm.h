/*
 * vdjx_comm.h -- what the ranks of `vdjer --gpus N` say to each other (one process per GPU).  No counterpart in the reference
 * (its only parallelism is pthreads over roots, A2:1287-1348).
 *
 * Two kinds of traffic:
 *   bulk      device buffers: all-to-all-v of byte rows, all-gather-v, all-reduce (MIN over u64, SUM over u32), broadcast.
 *             Transport "rccl": grouped ncclSend/ncclRecv and the RCCL collectives over xGMI (each peer on its own link).
 *             Transport "host": the same calls staged through host memory over a full mesh of UNIX socket pairs -- for ranks that
 *             share ONE device (RCCL refuses two ranks on a device), i.e. the multi-rank tests on the one-GPU box; never a default.
 *   control   small host-side messages between rank 0 and the others (commands, counts): the socket pair every rank shares with
 *             rank 0, whatever the bulk transport.  A rank that waits for its next command sleeps in read(), not inside a collective.
 * The sockets are made by the caller BEFORE it forks the ranks (vdjx_comm_sockets), so no rank ever needs a rendezvous address.
 * Every wait has a deadline (VDJX_MGPU_TIMEOUT_S, default 600 s; the wait for the next COMMAND has none): a peer that hangs -- not
 * dies -- ends the run with an error instead of blocking it forever (ncclCommAbort on the RCCL side).
 */
#ifndef VDJX_COMM_H
#define VDJX_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDJX_COMM_ID_BYTES 128
#define VDJX_COMM_MAX_MESH 16          /* ranks of the "host" transport (a full mesh of socket pairs) */

typedef struct vdjx_comm vdjx_comm;

/* Before the fork: fds[i*nranks + j] = the descriptor rank i talks to rank j with (-1 where there is none).  mesh = 0: only the pairs
 * with rank 0 (control; enough for the "rccl" transport); mesh = 1: every pair.  0 on success. */
int vdjx_comm_sockets(int nranks, int mesh, int* fds);
/* Ranks that are separate processes already (started one per GPU by a launcher: bench.py under torch.distributed.run): they meet in a
 * directory they all know -- rank i listens on <dir>/r<i>.sock, the higher ranks call it -- and get the same row the table above would
 * have given them (fds_row[j] = the descriptor to rank j; mesh as above).  Collective over the ranks; 0 on success. */
int vdjx_comm_rendezvous(const char* dir, int me, int nranks, int mesh, int* fds_row);
/* After the fork, in rank `me`: closes every descriptor that is not this rank's. */
void vdjx_comm_sockets_keep(int nranks, int me, int* fds);

/* transport: "rccl" | "host".  fds_row = this rank's row of the table above (nranks entries; kept, not closed, by the communicator).
 * rccl_id (VDJX_COMM_ID_BYTES): made by rank 0 with vdjx_comm_unique_id and handed to the others by the caller.  Collective. */
int vdjx_comm_unique_id(void* out128);
int vdjx_comm_init(const char* transport, int rank, int nranks, int device, const int* fds_row, const void* rccl_id, vdjx_comm** out);
void vdjx_comm_free(vdjx_comm* c);
int vdjx_comm_rank(const vdjx_comm* c);
int vdjx_comm_size(const vdjx_comm* c);
const char* vdjx_comm_transport(const vdjx_comm* c);
uint64_t vdjx_comm_bytes_sent(const vdjx_comm* c);          /* bulk bytes this rank sent to OTHER ranks */
const char* vdjx_comm_last_error(void);

/* ---- bulk (device pointers) ---- */
/* send_rows[r] rows of `row` bytes for rank r, contiguous in rank order; recv likewise.  Returns when the data has arrived. */
int vdjx_comm_a2av(vdjx_comm* c, const void* d_send, const uint64_t* send_rows, void* d_recv, const uint64_t* recv_rows, size_t row);
/* every rank's `rows[me]` rows to every rank: d_recv = rank 0's rows, rank 1's rows, ... */
int vdjx_comm_allgatherv(vdjx_comm* c, const void* d_send, void* d_recv, const uint64_t* rows, size_t row);
int vdjx_comm_allreduce_min_u64(vdjx_comm* c, void* d_buf, size_t n);       /* in place; unsigned order (all-ones stays largest) */
int vdjx_comm_allreduce_sum_u32(vdjx_comm* c, void* d_buf, size_t n);
/* ---- control (host pointers) ---- */
/* `bytes` from every rank, rank order, to every rank */
int vdjx_comm_allgather_host(vdjx_comm* c, const void* mine, size_t bytes, void* all);
/* rank 0's buffer to everybody (large payloads travel by the bulk transport) */
int vdjx_comm_bcast_host(vdjx_comm* c, void* buf, size_t bytes);
/* a command word from rank 0: rank 0 sends, the others block (without deadline) until it arrives; 0 on success, 1 = rank 0 is gone */
int vdjx_comm_command_send(vdjx_comm* c, const uint64_t cmd[4]);
int vdjx_comm_command_wait(vdjx_comm* c, uint64_t cmd[4]);
/* everybody reports a status to rank 0; rank 0 gets the worst (largest magnitude non-zero) in *worst */
int vdjx_comm_status(vdjx_comm* c, int mine, int* worst);

#ifdef __cplusplus
}
#endif
#endif
