/*
 * vdjx_mgpu.h -- `vdjer --gpus N`: the sharded k-mer build of include/vdjx.h (vdjx_shard_*) driven from C, the bytes moved by RCCL
 * over xGMI (one process per GPU).  No counterpart in the reference (A2:1287-1348 is its only parallelism).
 */
#ifndef VDJX_MGPU_H
#define VDJX_MGPU_H

#include <stdint.h>

#include "../../../include/vdjx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VDJX_MGPU_ID_BYTES 128

typedef struct vdjx_mgpu vdjx_mgpu;

/* rank 0: the RCCL bootstrap id every rank needs (sent to the other processes by the caller, e.g. through a pipe) */
int vdjx_mgpu_unique_id(void* out128);
/* every rank: joins the communicator on `device` (collective) */
int vdjx_mgpu_init(int rank, int nranks, int device, const void* unique_id, vdjx_mgpu** out);
void vdjx_mgpu_free(vdjx_mgpu* m);
/* collective: every rank passes its slice of the pool (records [rank*rec_stride, ...) of the scan order); the same graph on every rank */
int vdjx_mgpu_kmer_build(vdjx_mgpu* m, vdjx_ctx* ctx, const vdjx_pool* pool, int k, int mf, int mq, uint64_t rec_stride, vdjx_graph** out);
uint64_t vdjx_mgpu_bytes_sent(const vdjx_mgpu* m);
const char* vdjx_mgpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
