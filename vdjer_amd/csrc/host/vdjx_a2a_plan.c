/* vdjx_a2a_plan.c -- see vdjx_a2a_plan.h */
#include "vdjx_a2a_plan.h"

size_t vdjx_a2a_plan(int G, int me, const uint64_t* send_rows, const uint64_t* recv_rows, uint64_t row, uint64_t chunk,
                     vdjx_a2a_step* out, size_t cap, uint64_t self[3]) {
	uint64_t so = 0, ro = 0, rounds = 0;
	if (chunk == 0) chunk = 1;
	for (int r = 0; r < G; r++) {
		const uint64_t sb = send_rows[r] * row, rb = recv_rows[r] * row;
		if (r == me) { if (self) { self[0] = so; self[1] = ro; self[2] = sb; } }
		else {
			const uint64_t a = (sb + chunk - 1) / chunk, b = (rb + chunk - 1) / chunk;
			if (a > rounds) rounds = a;
			if (b > rounds) rounds = b;
		}
		so += sb; ro += rb;
	}
	size_t n = 0;
	for (uint64_t t = 0; t < rounds; t++) {
		so = 0; ro = 0;
		for (int r = 0; r < G; r++) {
			const uint64_t sb = send_rows[r] * row, rb = recv_rows[r] * row, a = t * chunk;
			if (r != me && (a < sb || a < rb)) {
				if (n < cap) {
					vdjx_a2a_step* s = &out[n];
					s->round = (uint32_t) t; s->peer = r;
					s->send_off = so + a; s->send_len = a < sb ? (sb - a < chunk ? sb - a : chunk) : 0;
					s->recv_off = ro + a; s->recv_len = a < rb ? (rb - a < chunk ? rb - a : chunk) : 0;
				}
				n++;
			}
			so += sb; ro += rb;
		}
	}
	return n;
}
