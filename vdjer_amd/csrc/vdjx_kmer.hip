// vdjx_kmer.hip -- k-mer table, prune and graph build on gfx950.
//
// Replaces build_pre_graph x2 + prune_pre_graph + build_graph2 x2 (A2:1388-1408).  The reference
// upserts every k-mer instance into one string-keyed hash table (72-byte random RMW per instance,
// A2:322-367).  Here the instances are radix-partitioned by a hash prefix into buckets sized for LDS,
// and every per-k-mer reduction (count, first instance, distinct-read flag, quality sums, ungated
// recount) is done bucket-locally in LDS.  All of it is integer/byte work bounded by HBM traffic; no
// MFMA applies.
//
//   K2a k_kmer_hist        per-workgroup LDS histogram of bucket sizes over a slice of the pool
//   K2b k_hist_colscan /   exclusive offsets per (workgroup, bucket)  -> deterministic placement,
//       k_bucket_scan      no global atomics
//   K2c k_kmer_scatter     LDS cursors, tuples {key_lo, key_hi, inst|gated} written bucket-contiguous
//   K3a k_bucket_aggregate LDS open-addressing table per bucket: gated count + first instance per
//                          distinct k-mer; keys with count >= max(mf,2) become candidates, their
//                          tuples are compacted in place (noise singletons die here)
//   K3b k_bucket_finalize  direct-indexed LDS arrays per bucket: distinct-read flag, quality sums
//                          (only for low-count keys, see TLOW), ungated recount -> survivors
//   K5  k_surv_table / k_graph_edges / k_node_flags   survivor lookup table, ordered edges, V/J flags
#include "vdjx_common.h"

#include <algorithm>
#include <numeric>
#include <string.h>

#define HIST_THREADS 1024
#define K3_THREADS 1024
#define K3_SLOTS 4096u              // LDS table slots per sub-pass
#define K3_SUB_TUPLES 3072u         // tuples per sub-pass the table is sized for
#define K3B_THREADS 512
#define K3B_CH 2048u                // candidates per chunk
#define K3B_A 256u                  // quality-sum rows per round
#define K3B_KW 25u                  // u32 words per row (2 x u16 sums each), k <= 50
#define NONE32 0xFFFFFFFFu
#define INST_MASK 0x7FFFFFFFu

// ----------------------------------------------------------------------------------------------
// per-record iteration shared by the histogram, scatter and edge passes
// ----------------------------------------------------------------------------------------------
struct RecView {
	u64 bhi, blo, nm, lq;
};

__device__ inline RecView load_rec(const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                   const u64* __restrict__ lowq, size_t r) {
	RecView v;
	const ulonglong2 b = ((const ulonglong2*) bases)[r];
	v.bhi = b.x; v.blo = b.y;
	v.nm = nmask[r];
	v.lq = lowq ? lowq[r] : 0ull;
	return v;
}

// ----------------------------------------------------------------------------------------------
// K2a
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HIST_THREADS) void k_kmer_hist(const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                            size_t R, int rl, int k, u32 nb_bits, size_t rpb,
                                                            u32* __restrict__ block_hist) {
	extern __shared__ u32 hist[];
	const u32 NB = 1u << nb_bits;
	for (u32 i = threadIdx.x; i < NB; i += HIST_THREADS) hist[i] = 0;
	__syncthreads();
	const size_t r0 = (size_t) blockIdx.x * rpb;
	const size_t r1 = r0 + rpb < R ? r0 + rpb : R;
	const int P = rl - k + 1;
	const u64 km = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
	for (size_t r = r0 + threadIdx.x; r < r1; r += HIST_THREADS) {
		RecView v = load_rec(bases, nmask, nullptr, r);
		for (int o = 0; o < P; o++) {
			if ((v.nm >> o) & km) continue;                     // k-mer holds an 'N' (A2:246)
			u64 khi, klo;
			vdjx_kmer_at(v.bhi, v.blo, rl, k, o, khi, klo);
			u64 h = vdjx_mix(klo, khi);
			atomicAdd(&hist[(u32) (h >> (64 - nb_bits))], 1u);
		}
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < NB; i += HIST_THREADS) block_hist[(size_t) blockIdx.x * NB + i] = hist[i];
}

// K2b: per bucket, exclusive running sum over workgroups (column scan; coalesced across buckets)
__global__ void k_hist_colscan(u32* __restrict__ block_hist, u32 nblk, u32 NB, u32* __restrict__ bucket_cnt) {
	u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= NB) return;
	u32 run = 0;
	for (u32 i = 0; i < nblk; i++) {
		u32 v = block_hist[(size_t) i * NB + b];
		block_hist[(size_t) i * NB + b] = run;
		run += v;
	}
	bucket_cnt[b] = run;
}

// exclusive scan of bucket_cnt[NB] -> bucket_start[NB+1], one 1024-thread workgroup
__global__ __launch_bounds__(1024) void k_bucket_scan(const u32* __restrict__ bucket_cnt, u32 NB, u32* __restrict__ bucket_start) {
	__shared__ u32 part[1024];
	const u32 per = (NB + 1023) / 1024;
	const u32 lo = threadIdx.x * per;
	const u32 hi = lo + per < NB ? lo + per : NB;
	u32 s = 0;
	for (u32 i = lo; i < hi; i++) s += bucket_cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for (u32 i = lo; i < hi; i++) {
		bucket_start[i] = run;
		run += bucket_cnt[i];
	}
	if (threadIdx.x == 1023) bucket_start[NB] = part[1023];
}

// ----------------------------------------------------------------------------------------------
// K2c
// ----------------------------------------------------------------------------------------------
template <typename THI>
__global__ __launch_bounds__(HIST_THREADS) void k_kmer_scatter(const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                               const u64* __restrict__ lowq, size_t R, int rl, int k, u32 nb_bits,
                                                               size_t rpb, const u32* __restrict__ block_hist,
                                                               const u32* __restrict__ bucket_start, u64* __restrict__ t_lo,
                                                               THI* __restrict__ t_hi, u32* __restrict__ t_inst) {
	extern __shared__ u32 cursor[];
	const u32 NB = 1u << nb_bits;
	for (u32 i = threadIdx.x; i < NB; i += HIST_THREADS) cursor[i] = bucket_start[i] + block_hist[(size_t) blockIdx.x * NB + i];
	__syncthreads();
	const size_t r0 = (size_t) blockIdx.x * rpb;
	const size_t r1 = r0 + rpb < R ? r0 + rpb : R;
	const int P = rl - k + 1;
	const u64 km = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
	for (size_t r = r0 + threadIdx.x; r < r1; r += HIST_THREADS) {
		RecView v = load_rec(bases, nmask, lowq, r);
		for (int o = 0; o < P; o++) {
			if ((v.nm >> o) & km) continue;
			u64 khi, klo;
			vdjx_kmer_at(v.bhi, v.blo, rl, k, o, khi, klo);
			u64 h = vdjx_mix(klo, khi);
			u32 pos = atomicAdd(&cursor[(u32) (h >> (64 - nb_bits))], 1u);
			u32 gated = ((v.lq >> o) & km) ? 0u : 0x80000000u;   // all k Phred >= 20 (A2:252)
			t_lo[pos] = klo;
			t_hi[pos] = (THI) khi;
			t_inst[pos] = gated | (u32) (r * (size_t) P + (size_t) o);
		}
	}
}

// ----------------------------------------------------------------------------------------------
// K3a: LDS hash aggregation per bucket
// ----------------------------------------------------------------------------------------------
template <typename THI>
__device__ inline int lds_insert(u64* s_klo, THI* s_khi, u64 lo, THI hi, u32 h) {
	const THI EMPTY = (THI) ~(THI) 0, LOCKED = (THI) (EMPTY - 1);
	u32 slot = h & (K3_SLOTS - 1);
	u32 probes = 0;
	while (probes < K3_SLOTS) {
		THI cur = *(volatile THI*) &s_khi[slot];
		if (cur == EMPTY) {
			THI old = atomicCAS(&s_khi[slot], EMPTY, LOCKED);
			if (old == EMPTY) {
				*(volatile u64*) &s_klo[slot] = lo;
				__threadfence_block();
				*(volatile THI*) &s_khi[slot] = hi;             // publish
				return (int) slot;
			}
			cur = old;
		}
		if (cur == LOCKED) continue;                            // another lane is writing this slot: look again
		if (cur == hi && *(volatile u64*) &s_klo[slot] == lo) return (int) slot;
		slot = (slot + 1) & (K3_SLOTS - 1);
		probes++;
	}
	return -1;
}

template <typename THI>
__device__ inline int lds_lookup(const u64* s_klo, const THI* s_khi, u64 lo, THI hi, u32 h) {
	const THI EMPTY = (THI) ~(THI) 0;
	u32 slot = h & (K3_SLOTS - 1);
	for (u32 probes = 0; probes < K3_SLOTS; probes++) {
		THI cur = s_khi[slot];
		if (cur == EMPTY) return -1;
		if (cur == hi && s_klo[slot] == lo) return (int) slot;
		slot = (slot + 1) & (K3_SLOTS - 1);
	}
	return -1;
}

template <typename THI>
__global__ __launch_bounds__(K3_THREADS) void k_bucket_aggregate(const u64* __restrict__ t_lo, const THI* __restrict__ t_hi,
                                                                 const u32* __restrict__ t_inst, const u32* __restrict__ bucket_start,
                                                                 u32 cmin, u64* __restrict__ c_lo, THI* __restrict__ c_hi,
                                                                 u32* __restrict__ c_cnt, u32* __restrict__ c_first,
                                                                 u32* __restrict__ ct_lcid, u32* __restrict__ ct_inst,
                                                                 u32* __restrict__ bucket_ncand, u32* __restrict__ bucket_nct,
                                                                 u64* __restrict__ g_distinct, u32* __restrict__ g_err) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	u64* s_klo = (u64*) smem;
	THI* s_khi = (THI*) (s_klo + K3_SLOTS);
	u32* s_cnt = (u32*) (s_khi + K3_SLOTS);
	u32* s_first = s_cnt + K3_SLOTS;
	u32* s_cidx = s_first + K3_SLOTS;
	__shared__ u32 s_ncand, s_nct, s_over, s_ndist;
	const THI EMPTY = (THI) ~(THI) 0;
	const u32 b = blockIdx.x;
	const u32 base = bucket_start[b];
	const u32 n = bucket_start[b + 1] - base;
	const u32 tid = threadIdx.x;
	if (n == 0) {
		if (tid == 0) { bucket_ncand[b] = 0; bucket_nct[b] = 0; }
		return;
	}
	u32 S = 1;
	while ((u64) S * K3_SUB_TUPLES < n) S <<= 1;
	for (;;) {
		if (tid == 0) { s_ncand = 0; s_nct = 0; s_over = 0; s_ndist = 0; }
		for (u32 s = 0; s < S; s++) {
			for (u32 i = tid; i < K3_SLOTS; i += K3_THREADS) { s_khi[i] = EMPTY; s_cnt[i] = 0; s_first[i] = NONE32; }
			__syncthreads();
			// sweep 1: gated instances only (include_kmer, A2:240-259) -> count + first (A2:332-347)
			for (u32 t = tid; t < n; t += K3_THREADS) {
				const u32 iw = t_inst[base + t];
				if (!(iw >> 31)) continue;
				const u64 lo = t_lo[base + t];
				const THI hi = t_hi[base + t];
				const u64 h = vdjx_mix(lo, (u64) hi);
				if ((u32) ((h >> 12) & (S - 1)) != s) continue;
				int slot = lds_insert<THI>(s_klo, s_khi, lo, hi, (u32) h);
				if (slot < 0) { s_over = 1; continue; }
				atomicAdd(&s_cnt[slot], 1u);
				atomicMin(&s_first[slot], iw & INST_MASK);
			}
			__syncthreads();
			if (s_over) break;
			// candidates: count >= max(mf, 2) (a k-mer seen once can never have two distinct reads, A2:349-352,476)
			for (u32 i = tid; i < K3_SLOTS; i += K3_THREADS) {
				u32 cid = NONE32;
				if (s_khi[i] != EMPTY) {
					atomicAdd(&s_ndist, 1u);
					if (s_cnt[i] >= cmin) {
						cid = atomicAdd(&s_ncand, 1u);
						c_lo[base + cid] = s_klo[i];
						c_hi[base + cid] = s_khi[i];
						c_cnt[base + cid] = s_cnt[i];
						c_first[base + cid] = s_first[i];
					}
				}
				s_cidx[i] = cid;
			}
			__syncthreads();
			// sweep 2: compact every instance (gated or not) of a candidate k-mer
			for (u32 t = tid; t < n; t += K3_THREADS) {
				const u64 lo = t_lo[base + t];
				const THI hi = t_hi[base + t];
				const u64 h = vdjx_mix(lo, (u64) hi);
				bool is_c = false;
				u32 cid = NONE32;
				if ((u32) ((h >> 12) & (S - 1)) == s) {
					int slot = lds_lookup<THI>(s_klo, s_khi, lo, hi, (u32) h);
					if (slot >= 0) { cid = s_cidx[slot]; is_c = cid != NONE32; }
				}
				u32 p = vdjx_wave_inc(&s_nct, is_c);
				if (is_c) {
					ct_lcid[base + p] = cid;
					ct_inst[base + p] = t_inst[base + t];
				}
			}
			__syncthreads();
		}
		if (!s_over) break;
		S <<= 1;
		if (S > (1u << 20)) { if (tid == 0) atomicAdd(g_err, 1u); break; }
		__syncthreads();
	}
	if (tid == 0) {
		bucket_ncand[b] = s_over ? 0 : s_ncand;
		bucket_nct[b] = s_over ? 0 : s_nct;
		atomicAdd(g_distinct, (u64) s_ndist);
	}
}

// ----------------------------------------------------------------------------------------------
// K3b: finalize candidates of a bucket
// ----------------------------------------------------------------------------------------------
struct SurvOut {
	u64* lo; u64* hi; u32* gcnt; u32* gfirst; u32* ucnt; u32* ufirst; u32* n; u32 cap;
};

template <typename THI>
__global__ __launch_bounds__(K3B_THREADS) void k_bucket_finalize(const u32* __restrict__ bucket_start, const u32* __restrict__ bucket_ncand,
                                                                 const u32* __restrict__ bucket_nct, const u64* __restrict__ c_lo,
                                                                 const THI* __restrict__ c_hi, const u32* __restrict__ c_cnt,
                                                                 const u32* __restrict__ c_first, const u32* __restrict__ ct_lcid,
                                                                 const u32* __restrict__ ct_inst, const u64* __restrict__ bases,
                                                                 const u64* __restrict__ nmask,
                                                                 const uint8_t* __restrict__ quals, int qstride, int k, int P,
                                                                 u32 mf, u32 mqq, u32 tlow, SurvOut so) {
	__shared__ u32 l_cnt[K3B_CH], l_first[K3B_CH], l_ucnt[K3B_CH], l_ufirst[K3B_CH], l_lowid[K3B_CH];
	__shared__ uint8_t l_multi[K3B_CH], l_qok[K3B_CH];
	__shared__ u32 acc[K3B_A * K3B_KW];
	__shared__ u32 s_nlow;
	const u32 b = blockIdx.x;
	const u32 nc = bucket_ncand[b];
	if (nc == 0) return;
	const u32 base = bucket_start[b];
	const u32 nt = bucket_nct[b];
	const u32 tid = threadIdx.x;
	const u32 KW = (u32) (k + 1) / 2;
	for (u32 c0 = 0; c0 < nc; c0 += K3B_CH) {
		const u32 m = nc - c0 < K3B_CH ? nc - c0 : K3B_CH;
		if (tid == 0) s_nlow = 0;
		__syncthreads();
		for (u32 i = tid; i < m; i += K3B_THREADS) {
			const u32 cnt = c_cnt[base + c0 + i];
			l_cnt[i] = cnt;
			l_first[i] = c_first[base + c0 + i];
			l_ucnt[i] = 0;
			l_ufirst[i] = NONE32;
			l_multi[i] = 0;
			l_qok[i] = 0;
			l_lowid[i] = cnt < tlow ? atomicAdd(&s_nlow, 1u) : NONE32;
		}
		__syncthreads();
		// sweep A: ungated recount + first sight (add_to_graph, A2:280-309) and the distinct-read flag (A2:349-352)
		for (u32 t = tid; t < nt; t += K3B_THREADS) {
			u32 lc = ct_lcid[base + t];
			if (lc < c0 || lc >= c0 + m) continue;
			lc -= c0;
			const u32 iw = ct_inst[base + t];
			const u32 inst = iw & INST_MASK;
			atomicAdd(&l_ucnt[lc], 1u);
			atomicMin(&l_ufirst[lc], inst);
			if ((iw >> 31) && !*(volatile uint8_t*) &l_multi[lc]) {
				const u32 rec = inst / (u32) P;
				const u32 frec = l_first[lc] / (u32) P;
				if (rec != frec) {
					const ulonglong2 x = ((const ulonglong2*) bases)[rec];
					const ulonglong2 y = ((const ulonglong2*) bases)[frec];
					// compare_read (A2:142-144) on the rl-base sequences; an 'N' is coded 0 in `bases`, so the N masks
					// take part in the comparison
					if (x.x != y.x || x.y != y.y || nmask[rec] != nmask[frec]) l_multi[lc] = 1;
				}
			}
		}
		__syncthreads();
		// quality sums, only for keys whose count cannot pass on its own (see TLOW in vdjx_kmer_build)
		const u32 nlow = s_nlow;
		for (u32 l0 = 0; l0 < nlow; l0 += K3B_A) {
			for (u32 i = tid; i < K3B_A * K3B_KW; i += K3B_THREADS) acc[i] = 0;
			__syncthreads();
			for (u32 t = tid; t < nt; t += K3B_THREADS) {
				u32 lc = ct_lcid[base + t];
				if (lc < c0 || lc >= c0 + m) continue;
				lc -= c0;
				const u32 iw = ct_inst[base + t];
				if (!(iw >> 31)) continue;
				const u32 lid = l_lowid[lc];
				if (lid < l0 || lid >= l0 + K3B_A) continue;
				const u32 inst = iw & INST_MASK;
				const u32 rec = inst / (u32) P;
				const u32 off = inst - rec * (u32) P;
				// first instance: the RECORD's first k qualities (A2:337-339); others: the k-mer's own (A2:354-361)
				const uint8_t* q = quals + (size_t) rec * (size_t) qstride + (inst == l_first[lc] ? 0u : off);
				u32* row = acc + (lid - l0) * K3B_KW;
				for (int j = 0; j < k; j++) {
					const u32 v = (uint8_t) (q[j] - 33);
					atomicAdd(&row[j >> 1], v << (16 * (j & 1)));
				}
			}
			__syncthreads();
			for (u32 i = tid; i < m; i += K3B_THREADS) {
				const u32 lid = l_lowid[i];
				if (lid < l0 || lid >= l0 + K3B_A) continue;
				const u32* row = acc + (lid - l0) * K3B_KW;
				uint8_t ok = 1;
				for (u32 w = 0; w < KW; w++) {
					const u32 v = row[w];
					if ((v & 0xFFFFu) < mqq) ok = 0;
					if (2 * w + 1 < (u32) k && (v >> 16) < mqq) ok = 0;
				}
				l_qok[i] = ok;
			}
			__syncthreads();
		}
		// prune_pre_graph (A2:467-484)
		for (u32 i = tid; i < m; i += K3B_THREADS) {
			const u32 cnt = l_cnt[i] > 32765u ? 32765u : l_cnt[i];            // A2:345-347
			const bool keep = cnt >= mf && l_multi[i] && (l_cnt[i] >= tlow || l_qok[i]);
			if (keep) {
				const u32 pos = atomicAdd(so.n, 1u);
				if (pos < so.cap) {
					so.lo[pos] = c_lo[base + c0 + i];
					so.hi[pos] = (u64) c_hi[base + c0 + i];
					so.gcnt[pos] = cnt;
					so.gfirst[pos] = l_first[i];
					so.ucnt[pos] = l_ucnt[i] > 32765u ? 32765u : l_ucnt[i]; // A2:261-265
					so.ufirst[pos] = l_ufirst[i];
				}
			}
		}
		__syncthreads();
	}
}

// ----------------------------------------------------------------------------------------------
// K5: survivor lookup table, edges, V/J flags
// ----------------------------------------------------------------------------------------------
__global__ void k_surv_table(const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, u32 n, u32* __restrict__ table, u32 mask) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	u32 slot = (u32) (vdjx_mix(s_lo[i], s_hi[i]) >> 20) & mask;
	while (atomicCAS(&table[slot], 0u, i + 1) != 0u) slot = (slot + 1) & mask;
}

__device__ inline int surv_lookup(const u32* __restrict__ table, u32 mask, const u64* __restrict__ s_lo,
                                  const u64* __restrict__ s_hi, u64 lo, u64 hi) {
	u32 slot = (u32) (vdjx_mix(lo, hi) >> 20) & mask;
	for (;;) {
		u32 v = table[slot];
		if (!v) return -1;
		if (s_lo[v - 1] == lo && s_hi[v - 1] == hi) return (int) (v - 1);
		slot = (slot + 1) & mask;
	}
}

// add_to_graph's edge bookkeeping (A2:311-318, link_nodes A2:223-237): an edge prev->curr exists when two
// adjacent offsets of one record both survive; list order is by first sight, so keep the minimum instance.
__global__ void k_graph_edges(const u64* __restrict__ bases, const u64* __restrict__ nmask, size_t R, int rl, int k,
                              const u32* __restrict__ table, u32 mask, const u64* __restrict__ s_lo, const u64* __restrict__ s_hi,
                              u32* __restrict__ edge_first, u32* __restrict__ edge_to) {
	size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= R) return;
	const int P = rl - k + 1;
	const u64 km = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
	RecView v = load_rec(bases, nmask, nullptr, r);
	int prev = -1;
	for (int o = 0; o < P; o++) {
		if ((v.nm >> o) & km) { prev = -1; continue; }
		u64 khi, klo;
		vdjx_kmer_at(v.bhi, v.blo, rl, k, o, khi, klo);
		int s = surv_lookup(table, mask, s_lo, s_hi, klo, khi);
		if (s >= 0 && prev >= 0) {
			const u32 e = (u32) prev * 4u + (u32) (klo & 3ull);
			atomicMin(&edge_first[e], (u32) (r * (size_t) P + (size_t) o));
			edge_to[e] = (u32) s;
		}
		prev = s;
	}
}

// A2:288-303: has_vmer/has_jmer from the code of the node's first 16 bases
__global__ void k_node_flags(const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, u32 n, int k,
                             const u32* __restrict__ vbits, const u32* __restrict__ jbits,
                             uint8_t* __restrict__ has_v, uint8_t* __restrict__ has_j) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (k <= 16) { has_v[i] = 1; has_j[i] = 1; return; }
	u128 key = ((u128) s_hi[i] << 64) | s_lo[i];
	u32 code = (u32) (key >> (2 * (k - 16)));
	has_v[i] = code ? (vbits[code >> 5] >> (code & 31)) & 1u : 0;
	has_j[i] = code ? (jbits[code >> 5] >> (code & 31)) & 1u : 0;
}


// ----------------------------------------------------------------------------------------------
// K6: node numbering and list building on the device.
// Node ids are creation order (new_node, A2:188-204) = rank of the node's first ungated instance.
// First instances are distinct integers below R*P, so the rank comes from a bitmap over the instance
// space and a popcount prefix (no sort).  toNodes/fromNodes are prepend-on-first-sight lists
// (link_nodes, A2:223-237) of at most 4 entries: a 4-element sort by first sight, newest first.
// ----------------------------------------------------------------------------------------------
__global__ void k_mark_first(const u32* __restrict__ ufirst, u32 n, u32* __restrict__ bits) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) atomicOr(&bits[ufirst[i] >> 5], 1u << (ufirst[i] & 31));
}

#define POPC_WORDS 4096u
__global__ __launch_bounds__(256) void k_popc_blocks(const u32* __restrict__ bits, u32 nwords, u32* __restrict__ word_pre,
                                                     u32* __restrict__ block_sum) {
	__shared__ u32 part[256];
	const u32 w0 = blockIdx.x * POPC_WORDS + threadIdx.x * 16;
	u32 loc[16];
	u32 s = 0;
	for (int i = 0; i < 16; i++) {
		const u32 w = w0 + i;
		loc[i] = s;
		s += w < nwords ? __popc(bits[w]) : 0;
	}
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 256; d <<= 1) {
		u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	const u32 excl = part[threadIdx.x] - s;
	for (int i = 0; i < 16; i++) if (w0 + i < nwords) word_pre[w0 + i] = excl + loc[i];
	if (threadIdx.x == 255) block_sum[blockIdx.x] = part[255];
}

struct NodeOut {
	u64* first_inst; u32* gcnt; u32* freq; uint8_t* hv; uint8_t* hj; u64* klo; u64* khi; char* kmers;
	uint8_t* to_deg; u32* to_ids; uint8_t* from_deg; u32* from_ids;
};

__global__ void k_node_rank(const u32* __restrict__ ufirst, u32 n, const u32* __restrict__ bits, const u32* __restrict__ word_pre,
                            const u32* __restrict__ block_pre, u32* __restrict__ rank) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u32 f = ufirst[i], w = f >> 5;
	rank[i] = block_pre[w / POPC_WORDS] + word_pre[w] + __popc(bits[w] & ((1u << (f & 31)) - 1u));
}

// in-edges: (u -> v) lands in slot (v, first base of u); a k-mer has at most 4 predecessors
__global__ void k_in_edges(const u32* __restrict__ edge_first, const u32* __restrict__ edge_to, const u64* __restrict__ s_lo,
                           const u64* __restrict__ s_hi, u32 n, int k, u32* __restrict__ in_first, u32* __restrict__ in_from) {
	u32 e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * 4u) return;
	const u32 ef = edge_first[e];
	if (ef == NONE32) return;
	const u32 u = e >> 2, v = edge_to[e];
	const u128 key = ((u128) s_hi[u] << 64) | s_lo[u];
	const u32 a = (u32) (key >> (2 * (k - 1))) & 3u;
	in_first[v * 4 + a] = ef;
	in_from[v * 4 + a] = u;
}

__device__ inline void sort4_desc(u32 (&f)[4], u32 (&id)[4]) {
	// entries with f == NONE32 are absent: give them the smallest key
#define CSWAP(a, b) { const bool sw = key[a] < key[b]; if (sw) { u64 t = key[a]; key[a] = key[b]; key[b] = t; } }
	u64 key[4];
	for (int i = 0; i < 4; i++) key[i] = f[i] == NONE32 ? 0ull : (((u64) f[i] + 1) << 32) | id[i];
	CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
	for (int i = 0; i < 4; i++) { f[i] = key[i] ? (u32) (key[i] >> 32) - 1 : NONE32; id[i] = (u32) key[i]; }
#undef CSWAP
}

__global__ void k_node_emit(const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, const u32* __restrict__ s_gcnt,
                            const u32* __restrict__ s_ucnt, const u32* __restrict__ s_ufirst, const uint8_t* __restrict__ hv,
                            const uint8_t* __restrict__ hj, const u32* __restrict__ rank, const u32* __restrict__ edge_first,
                            const u32* __restrict__ edge_to, const u32* __restrict__ in_first, const u32* __restrict__ in_from,
                            u32 n, int k, int P, NodeOut o) {
	u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	const u32 r = rank[s];
	const u32 inst = s_ufirst[s];
	o.first_inst[r] = (u64) (inst / (u32) P) * 64 + (inst % (u32) P);
	o.gcnt[r] = s_gcnt[s];
	o.freq[r] = s_ucnt[s];
	o.hv[r] = hv[s];
	o.hj[r] = hj[s];
	o.klo[r] = s_lo[s];
	o.khi[r] = s_hi[s];
	const u128 key = ((u128) s_hi[s] << 64) | s_lo[s];
	for (int j = 0; j < k; j++) {
		const u32 b = (u32) (key >> (2 * (k - 1 - j))) & 3u;
		o.kmers[(size_t) r * k + j] = b == 0 ? 'A' : (b == 1 ? 'T' : (b == 2 ? 'C' : 'G'));
	}
	u32 f[4], id[4];
	for (int e = 0; e < 4; e++) { f[e] = edge_first[s * 4 + e]; id[e] = f[e] != NONE32 ? rank[edge_to[s * 4 + e]] + 1 : 0; }
	sort4_desc(f, id);
	u32 deg = 0;
	for (int e = 0; e < 4; e++) { o.to_ids[(size_t) r * 4 + e] = f[e] != NONE32 ? id[e] : 0; deg += f[e] != NONE32; }
	o.to_deg[r] = (uint8_t) deg;
	for (int e = 0; e < 4; e++) { f[e] = in_first[s * 4 + e]; id[e] = f[e] != NONE32 ? rank[in_from[s * 4 + e]] + 1 : 0; }
	sort4_desc(f, id);
	deg = 0;
	for (int e = 0; e < 4; e++) { o.from_ids[(size_t) r * 4 + e] = f[e] != NONE32 ? id[e] : 0; deg += f[e] != NONE32; }
	o.from_deg[r] = (uint8_t) deg;
}

// ----------------------------------------------------------------------------------------------
// host driver
// ----------------------------------------------------------------------------------------------
namespace {

template <typename THI>
int kmer_build_impl(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, vdjx_graph* g) {
	const int rl = pool->rl;
	const int P = rl - k + 1;
	const size_t R = pool->n_records;
	const size_t NI = R * (size_t) P;
	hipStream_t st = c->stream;
	vdjx_work db(c);

	// ---- partition geometry: ~2048 instances per bucket, at most 2^15 buckets (128 KB LDS histogram)
	u32 nb_bits = 8;
	while (nb_bits < 15 && ((size_t) 2048 << nb_bits) < NI) nb_bits++;
	const u32 NB = 1u << nb_bits;
	u32 nblk = (u32) std::min<size_t>(512, (R + 4095) / 4096);
	if (nblk == 0) nblk = 1;
	const size_t rpb = (R + nblk - 1) / nblk;

	u32 *block_hist, *bucket_cnt, *bucket_start, *bucket_ncand, *bucket_nct, *g_err, *n_surv;
	u64* g_distinct;
	HIP_TRY(db.alloc(&block_hist, (size_t) nblk * NB));
	HIP_TRY(db.alloc(&bucket_cnt, NB));
	HIP_TRY(db.alloc(&bucket_start, NB + 1));
	HIP_TRY(db.alloc(&bucket_ncand, NB));
	HIP_TRY(db.alloc(&bucket_nct, NB));
	HIP_TRY(db.alloc(&g_err, 1));
	HIP_TRY(db.alloc(&n_surv, 1));
	HIP_TRY(db.alloc(&g_distinct, 1));
	HIP_TRY(hipMemsetAsync(g_err, 0, 4, st));
	HIP_TRY(hipMemsetAsync(n_surv, 0, 4, st));
	HIP_TRY(hipMemsetAsync(g_distinct, 0, 8, st));

	const size_t lds_hist = (size_t) NB * 4;
	HIP_TRY(hipFuncSetAttribute((const void*) k_kmer_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_hist));
	HIP_TRY(hipFuncSetAttribute((const void*) k_kmer_scatter<THI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_hist));
	{
		vdjx_prof_scope ps(c, "k_kmer_hist");
		hipLaunchKernelGGL(k_kmer_hist, dim3(nblk), dim3(HIST_THREADS), lds_hist, st, pool->d_bases, pool->d_nmask, R, rl, k, nb_bits, rpb, block_hist);
	}
	{
		vdjx_prof_scope ps(c, "k_hist_scan");
		hipLaunchKernelGGL(k_hist_colscan, dim3((NB + 255) / 256), dim3(256), 0, st, block_hist, nblk, NB, bucket_cnt);
		hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, bucket_cnt, NB, bucket_start);
	}
	u32 N = 0;
	HIP_TRY(hipMemcpyAsync(&N, bucket_start + NB, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());

	// ---- tuples + in-place candidate arrays (bucket b owns [bucket_start[b], bucket_start[b+1]) of each)
	u64 *t_lo, *c_lo;
	THI *t_hi, *c_hi;
	u32 *t_inst, *c_cnt, *c_first, *ct_lcid, *ct_inst;
	HIP_TRY(db.alloc(&t_lo, N));
	HIP_TRY(db.alloc(&t_hi, N));
	HIP_TRY(db.alloc(&t_inst, N));
	HIP_TRY(db.alloc(&c_lo, N));
	HIP_TRY(db.alloc(&c_hi, N));
	HIP_TRY(db.alloc(&c_cnt, N));
	HIP_TRY(db.alloc(&c_first, N));
	HIP_TRY(db.alloc(&ct_lcid, N));
	HIP_TRY(db.alloc(&ct_inst, N));
	{
		vdjx_prof_scope ps(c, "k_kmer_scatter");
		hipLaunchKernelGGL(k_kmer_scatter<THI>, dim3(nblk), dim3(HIST_THREADS), lds_hist, st, pool->d_bases, pool->d_nmask, pool->d_lowq,
		                   R, rl, k, nb_bits, rpb, block_hist, bucket_start, t_lo, t_hi, t_inst);
	}

	// ---- prune thresholds.  mq is clamped as A2:1514-1516; a sum >= 214 reads as 255 (A2:356-360), so the test
	// "S_j >= mq" is "true sum >= min(mq, 214)".  Every gated instance other than the first adds >= 20
	// (MIN_BASE_QUALITY) to every S_j, so a key with count >= TLOW = 1 + ceil(mqq/20) passes the quality test
	// whatever its qualities are: only keys with count < TLOW need their sums computed.
	if (mq >= 255) mq = 254;
	const u32 mqq = (u32) (mq < 0 ? 0 : (mq > 214 ? 214 : mq));
	const u32 tlow = 1 + (mqq + 19) / 20;
	const u32 cmin = (u32) std::max(mf, 2);
	const u32 mfu = (u32) std::max(mf, 0);

	const size_t lds_agg = (size_t) K3_SLOTS * (8 + sizeof(THI) + 12);
	HIP_TRY(hipFuncSetAttribute((const void*) k_bucket_aggregate<THI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_agg));
	{
		vdjx_prof_scope ps(c, "k_bucket_aggregate");
		hipLaunchKernelGGL(k_bucket_aggregate<THI>, dim3(NB), dim3(K3_THREADS), lds_agg, st, t_lo, t_hi, t_inst, bucket_start, cmin,
		                   c_lo, c_hi, c_cnt, c_first, ct_lcid, ct_inst, bucket_ncand, bucket_nct, g_distinct, g_err);
	}

	// survivors: capacity grows on demand (rerun of the cheap finalize pass)
	u32 cap = (u32) std::min<size_t>((size_t) N / 2 + 1024, (size_t) 1 << 22);
	u32 ns = 0;
	u64 *s_lo = nullptr, *s_hi = nullptr;
	u32 *s_gcnt = nullptr, *s_gfirst = nullptr, *s_ucnt = nullptr, *s_ufirst = nullptr;
	for (int attempt = 0; attempt < 2; attempt++) {
		HIP_TRY(db.alloc(&s_lo, cap)); HIP_TRY(db.alloc(&s_hi, cap));
		HIP_TRY(db.alloc(&s_gcnt, cap)); HIP_TRY(db.alloc(&s_gfirst, cap));
		HIP_TRY(db.alloc(&s_ucnt, cap)); HIP_TRY(db.alloc(&s_ufirst, cap));
		HIP_TRY(hipMemsetAsync(n_surv, 0, 4, st));
		SurvOut so{s_lo, s_hi, s_gcnt, s_gfirst, s_ucnt, s_ufirst, n_surv, cap};
		{
			vdjx_prof_scope ps(c, "k_bucket_finalize");
			hipLaunchKernelGGL(k_bucket_finalize<THI>, dim3(NB), dim3(K3B_THREADS), 0, st, bucket_start, bucket_ncand, bucket_nct,
			                   c_lo, c_hi, c_cnt, c_first, ct_lcid, ct_inst, pool->d_bases, pool->d_nmask, pool->d_quals, pool->qstride,
			                   k, P, mfu, mqq, tlow, so);
		}
		HIP_TRY(hipMemcpyAsync(&ns, n_surv, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		if (ns <= cap) break;
		cap = ns;
		if (attempt == 1) { vdjx_set_error("survivor capacity logic failed"); return VDJX_EHIP; }
	}
	u32 err = 0;
	u64 ndist = 0;
	HIP_TRY(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(&ndist, g_distinct, 8, hipMemcpyDeviceToHost));
	if (err) { vdjx_set_error("k_bucket_aggregate: %u buckets could not be split to fit LDS", err); return VDJX_EHIP; }
	g->pre_nodes = (size_t) ndist;
	g->n = ns;
	g->k = k;
	if (ns == 0) return VDJX_OK;

	// ---- graph pass: survivor table, edges, flags
	u32 tmask = 1023;
	while ((size_t) tmask + 1 < (size_t) ns * 2) tmask = tmask * 2 + 1;
	u32 *table, *edge_first, *edge_to;
	uint8_t *d_hv, *d_hj;
	HIP_TRY(db.alloc(&table, (size_t) tmask + 1));
	HIP_TRY(db.alloc(&edge_first, (size_t) ns * 4));
	HIP_TRY(db.alloc(&edge_to, (size_t) ns * 4));
	HIP_TRY(db.alloc(&d_hv, ns));
	HIP_TRY(db.alloc(&d_hj, ns));
	HIP_TRY(hipMemsetAsync(table, 0, ((size_t) tmask + 1) * 4, st));
	HIP_TRY(hipMemsetAsync(edge_first, 0xFF, (size_t) ns * 16, st));
	HIP_TRY(hipMemsetAsync(edge_to, 0xFF, (size_t) ns * 16, st));
	{
		vdjx_prof_scope ps(c, "k_surv_table");
		hipLaunchKernelGGL(k_surv_table, dim3((ns + 255) / 256), dim3(256), 0, st, s_lo, s_hi, ns, table, tmask);
	}
	{
		vdjx_prof_scope ps(c, "k_graph_edges");
		hipLaunchKernelGGL(k_graph_edges, dim3((unsigned) ((R + 255) / 256)), dim3(256), 0, st, pool->d_bases, pool->d_nmask, R, rl, k,
		                   table, tmask, s_lo, s_hi, edge_first, edge_to);
	}
	{
		vdjx_prof_scope ps(c, "k_node_flags");
		hipLaunchKernelGGL(k_node_flags, dim3((ns + 255) / 256), dim3(256), 0, st, s_lo, s_hi, ns, k, c->d_vbits, c->d_jbits, d_hv, d_hj);
	}

	// ---- node ids, ordered lists, k-mer text: all on the device (K6), the host only copies
	const u32 nwords = (u32) ((NI + 31) / 32);
	const u32 npb = (nwords + POPC_WORDS - 1) / POPC_WORDS;
	u32 *bits, *word_pre, *block_sum, *block_pre, *rank, *in_first, *in_from;
	HIP_TRY(db.alloc(&bits, nwords));
	HIP_TRY(db.alloc(&word_pre, nwords));
	HIP_TRY(db.alloc(&block_sum, npb));
	HIP_TRY(db.alloc(&block_pre, npb + 1));
	HIP_TRY(db.alloc(&rank, ns));
	HIP_TRY(db.alloc(&in_first, (size_t) ns * 4));
	HIP_TRY(db.alloc(&in_from, (size_t) ns * 4));
	NodeOut no;
	HIP_TRY(db.alloc(&no.first_inst, ns)); HIP_TRY(db.alloc(&no.gcnt, ns)); HIP_TRY(db.alloc(&no.freq, ns));
	HIP_TRY(db.alloc(&no.hv, ns)); HIP_TRY(db.alloc(&no.hj, ns)); HIP_TRY(db.alloc(&no.klo, ns)); HIP_TRY(db.alloc(&no.khi, ns));
	HIP_TRY(db.alloc(&no.kmers, (size_t) ns * k));
	HIP_TRY(db.alloc(&no.to_deg, ns)); HIP_TRY(db.alloc(&no.to_ids, (size_t) ns * 4));
	HIP_TRY(db.alloc(&no.from_deg, ns)); HIP_TRY(db.alloc(&no.from_ids, (size_t) ns * 4));
	HIP_TRY(hipMemsetAsync(bits, 0, (size_t) nwords * 4, st));
	HIP_TRY(hipMemsetAsync(in_first, 0xFF, (size_t) ns * 16, st));
	{
		vdjx_prof_scope ps(c, "k_node_order");
		hipLaunchKernelGGL(k_mark_first, dim3((ns + 255) / 256), dim3(256), 0, st, s_ufirst, ns, bits);
		hipLaunchKernelGGL(k_popc_blocks, dim3(npb), dim3(256), 0, st, bits, nwords, word_pre, block_sum);
		hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, block_sum, npb, block_pre);
		hipLaunchKernelGGL(k_node_rank, dim3((ns + 255) / 256), dim3(256), 0, st, s_ufirst, ns, bits, word_pre, block_pre, rank);
		hipLaunchKernelGGL(k_in_edges, dim3((ns * 4 + 255) / 256), dim3(256), 0, st, edge_first, edge_to, s_lo, s_hi, ns, k, in_first, in_from);
		hipLaunchKernelGGL(k_node_emit, dim3((ns + 255) / 256), dim3(256), 0, st, s_lo, s_hi, s_gcnt, s_ucnt, s_ufirst, d_hv, d_hj, rank,
		                   edge_first, edge_to, in_first, in_from, ns, k, P, no);
	}
	g->first_inst.resize(ns); g->gated_count.resize(ns); g->freq.resize(ns);
	g->has_v.resize(ns); g->has_j.resize(ns);
	g->to_deg.resize(ns); g->from_deg.resize(ns);
	g->to_ids.resize((size_t) ns * 4); g->from_ids.resize((size_t) ns * 4);
	g->key_lo.resize(ns); g->key_hi.resize(ns);
	g->kmers.resize((size_t) ns * k);
	HIP_TRY(hipMemcpyAsync(g->first_inst.data(), no.first_inst, (size_t) ns * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->gated_count.data(), no.gcnt, (size_t) ns * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->freq.data(), no.freq, (size_t) ns * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->has_v.data(), no.hv, ns, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->has_j.data(), no.hj, ns, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->key_lo.data(), no.klo, (size_t) ns * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->key_hi.data(), no.khi, (size_t) ns * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->kmers.data(), no.kmers, (size_t) ns * k, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->to_deg.data(), no.to_deg, ns, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->to_ids.data(), no.to_ids, (size_t) ns * 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->from_deg.data(), no.from_deg, ns, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(g->from_ids.data(), no.from_ids, (size_t) ns * 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

}  // namespace

extern "C" int vdjx_kmer_build(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, vdjx_graph** out) {
	if (!c || !pool || !out) { vdjx_set_error("vdjx_kmer_build: NULL argument"); return VDJX_EINVAL; }
	*out = nullptr;
	if (pool->ctx != c) { vdjx_set_error("vdjx_kmer_build: pool belongs to another context"); return VDJX_EINVAL; }
	if (k < 1 || k > VDJX_MAX_KMER || k > pool->rl) { vdjx_set_error("k=%d outside [1,min(%d,rl=%d)]", k, VDJX_MAX_KMER, pool->rl); return VDJX_ELIMIT; }
	const size_t NI = pool->n_records * (size_t) (pool->rl - k + 1);
	if (NI >= (1ull << 31)) { vdjx_set_error("records*offsets = %zu >= 2^31: shard the pool over more GPUs", NI); return VDJX_ELIMIT; }
	if (k > 16 && !c->anchors_loaded) { vdjx_set_error("vdjx_kmer_build: call vdjx_anchor_sets_load first (k > 16)"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_graph* g = new vdjx_graph();
	int rc = (2 * k - 64 <= 30) ? kmer_build_impl<u32>(c, pool, k, mf, mq, g) : kmer_build_impl<u64>(c, pool, k, mf, mq, g);
	if (rc != VDJX_OK) { delete g; return rc; }
	*out = g;
	return VDJX_OK;
}

extern "C" size_t vdjx_graph_nodes(const vdjx_graph* g) { return g ? g->n : 0; }
extern "C" size_t vdjx_graph_pre_nodes(const vdjx_graph* g) { return g ? g->pre_nodes : 0; }

extern "C" int vdjx_graph_export(const vdjx_graph* g, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq,
                                 uint8_t* has_v, uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids,
                                 uint8_t* from_deg, uint32_t* from_ids, char* kmers) {
	if (!g) { vdjx_set_error("vdjx_graph_export: NULL graph"); return VDJX_EINVAL; }
	const size_t n = g->n;
	if (first_inst) memcpy(first_inst, g->first_inst.data(), n * 8);
	if (gated_count) memcpy(gated_count, g->gated_count.data(), n * 4);
	if (freq) memcpy(freq, g->freq.data(), n * 4);
	if (has_v) memcpy(has_v, g->has_v.data(), n);
	if (has_j) memcpy(has_j, g->has_j.data(), n);
	if (to_deg) memcpy(to_deg, g->to_deg.data(), n);
	if (from_deg) memcpy(from_deg, g->from_deg.data(), n);
	if (to_ids) memcpy(to_ids, g->to_ids.data(), n * 16);
	if (from_ids) memcpy(from_ids, g->from_ids.data(), n * 16);
	if (kmers) memcpy(kmers, g->kmers.data(), n * (size_t) g->k);
	return VDJX_OK;
}

extern "C" void vdjx_graph_free(vdjx_graph* g) { delete g; }
