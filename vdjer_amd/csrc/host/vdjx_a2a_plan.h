/*
 * vdjx_a2a_plan.h -- the arithmetic of the multi-GPU exchanges (vdjx_mgpu.c), apart from RCCL so that it can be tested on a CPU.
 * An all-to-all-v of byte rows is cut into rounds: in round t every pair of ranks moves bytes [t*chunk, (t+1)*chunk) of what it has
 * for each other (one grouped set of sends and receives per round); what a rank keeps for itself is one device copy.
 */
#ifndef VDJX_A2A_PLAN_H
#define VDJX_A2A_PLAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	uint32_t round;          /* steps of one round form one group */
	int32_t peer;
	uint64_t send_off, send_len;     /* bytes of the send buffer that go to `peer` in this round (len 0: nothing) */
	uint64_t recv_off, recv_len;     /* where `peer`'s bytes of this round land in the receive buffer */
} vdjx_a2a_step;

/* send_rows[r] / recv_rows[r] rows of `row` bytes for / from rank r, laid out rank after rank in both buffers.
 * Writes up to `cap` steps (round-major, peers ascending inside a round, `me` left out) and returns how many there are.
 * self[3] = {send offset, receive offset, bytes} of what the rank keeps. */
size_t vdjx_a2a_plan(int nranks, int me, const uint64_t* send_rows, const uint64_t* recv_rows, uint64_t row, uint64_t chunk,
                     vdjx_a2a_step* out, size_t cap, uint64_t self[3]);

#ifdef __cplusplus
}
#endif
#endif
