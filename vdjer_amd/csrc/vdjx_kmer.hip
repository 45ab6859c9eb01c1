// vdjx_kmer.hip -- k-mer table, prune and graph build on gfx950.
//
// Replaces build_pre_graph x2 + prune_pre_graph + build_graph2 x2 (A2:1388-1408).  The reference
// upserts every k-mer instance into one string-keyed hash table (72-byte random RMW per instance,
// A2:322-367).  Here the instances are radix-partitioned by a hash prefix into buckets sized for LDS,
// and every per-k-mer reduction (count, first instance, distinct-read flag, quality sums, ungated
// recount) is done bucket-locally in LDS.  All of it is integer/byte work bounded by HBM traffic; no
// MFMA applies.
//
//   K2a k_kmer_hist        per-workgroup LDS histogram of bucket sizes over a slice of the pool, added to the global counts
//   K2b k_bucket_scan      exclusive bucket starts
//   K2c k_part_records /   LDS-staged counting sort per round: tuples {key_lo, key_hi, inst|gated} written as
//       k_part_tuples      coalesced bucket runs (256 coarse buckets, then 128 fine ones inside each; large pools:
//                          2^(T-10) coarse, a counting pass, 1024 fine); positions inside a bucket from cursor bumps
//   K3a k_bucket_aggregate LDS open-addressing table per bucket: gated count + first instance per
//                          distinct k-mer; keys with count >= max(mf,2) become candidates, their
//                          tuples are compacted in place (noise singletons die here)
//   K3b k_bucket_finalize  direct-indexed LDS arrays per bucket: distinct-read flag, quality sums
//                          (only for low-count keys, see TLOW), ungated recount -> survivors
//   K5  k_surv_table / k_graph_edges / k_node_flags   survivor lookup table, ordered edges, V/J flags
#include "vdjx_common.h"

#include <algorithm>
#include <numeric>
#include <type_traits>
#include <string.h>
#include <stdlib.h>

#define HIST_THREADS 512
#define K3_THREADS 512
#define K3_SLOTS 1024u              // LDS table slots per sub-pass of k_bucket_aggregate (28 KB with u32 key_hi): ~100 distinct gated k-mers per bucket
#define LOCAL_SLOTS 2048u           // k_bucket_local keeps every distinct k-mer of the bucket, gated or not (~600)
#define K3_UNR 8
#define K3_SUB_TUPLES g_sub_tuples  // first split only for very large buckets: hot k-mers make buckets long, not wide (an overflow splits further)
__device__ u32 g_sub_tuples = 262144u;
#define K3B_THREADS 256
#define K3B_CH 1024u                // candidates per chunk
#define K3B_A 128u                  // quality-sum rows per round
#define K3B_KW 25u                  // u32 words per row (2 x u16 sums each), k <= 50
#define K3B_Q 1536u                 // remembered low-count instances per chunk (more: the quality rounds rescan the bucket)
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

// exclusive scan of bucket_cnt[NB] -> bucket_start[NB+1], one 1024-thread workgroup.  Tiles of 8,192 counters go through LDS: read and
// written with consecutive lanes on consecutive addresses, summed eight per thread from a padded layout (index i at i + i/32: the 32
// lanes of a half-wave on 32 banks).  (A thread reading its own 32 consecutive counters from global memory made 2^15 counters a chain
// of 32 dependent line fills: 52 us at 10 M pairs, between the histogram and the partition.)
#define SCAN_TILE 8192u
__global__ __launch_bounds__(1024) void k_bucket_scan(const u32* __restrict__ bucket_cnt, u32 NB, u32* __restrict__ bucket_start) {
	__shared__ u32 buf[SCAN_TILE + SCAN_TILE / 32];
	__shared__ u32 wsum[16];
	__shared__ u32 s_carry;
	const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
	if (t == 0) s_carry = 0;
	__syncthreads();
	for (u32 base = 0; base < NB; base += SCAN_TILE) {
#pragma unroll
		for (u32 j = 0; j < 8; j++) {
			const u32 i = j * 1024u + t;
			buf[i + (i >> 5)] = base + i < NB ? bucket_cnt[base + i] : 0u;
		}
		__syncthreads();
		u32 v[8], sum = 0;
#pragma unroll
		for (u32 e = 0; e < 8; e++) { const u32 i = 8u * t + e; v[e] = buf[i + (i >> 5)]; sum += v[e]; }
		const u32 incl = (u32) vdjx_wave_scan_add((int) sum);
		if (lane == 63) wsum[wv] = incl;
		__syncthreads();
		u32 run = s_carry + incl - sum;
		for (u32 w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
		for (u32 e = 0; e < 8; e++) { const u32 i = 8u * t + e; buf[i + (i >> 5)] = run; run += v[e]; }
		__syncthreads();
#pragma unroll
		for (u32 j = 0; j < 8; j++) {
			const u32 i = j * 1024u + t;
			if (base + i < NB) bucket_start[base + i] = buf[i + (i >> 5)];
		}
		if (t == 1023) s_carry = run;                         // (thread 1023's running sum is the tile's end)
		__syncthreads();
	}
	if (t == 0) bucket_start[NB] = s_carry;
}

// ----------------------------------------------------------------------------------------------
// K2c: LDS-staged partition (software write combining).
// A direct scatter into 2^15 buckets writes 4-16 B at a time to tens of millions of open write fronts;
// rocprofv3 WRITE_SIZE showed 5.7x the algorithmic bytes reaching HBM (profiles/r01b_traffic.json).
// Instead a workgroup stages PART_ROUND tuples per round in LDS, counting-sorts them by bucket there and
// writes every bucket's run with consecutive lanes on consecutive addresses.  <= 1024 buckets per pass keep
// the runs long; 2^15 buckets are reached in two passes (256 coarse x 128 fine), the second pass working
// one coarse bucket (a few MB, cache resident) at a time.  Placement inside a bucket is arbitrary
// (per-round global cursor bump): every per-k-mer reduction downstream is order-free.
// ----------------------------------------------------------------------------------------------
#define PART_THREADS 1024
#define PART_LDS_BYTES 131072
#define PART_MAXB 1024

// exclusive scan of cnt[0..n) (n <= 1024) into base[0..n], base[n] = total; all PART_THREADS threads call it.
// Two counts per thread, DPP prefix sums inside the waves, one wave for the wave totals: three barriers instead of twenty.
__device__ inline void part_scan(const u32* cnt, u32* base, u32* tmp, u32 n) {
	const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const u32 a = 2 * t < n ? cnt[2 * t] : 0, b = 2 * t + 1 < n ? cnt[2 * t + 1] : 0;
	const u32 incl = (u32) vdjx_wave_scan_add((int) (a + b));
	if (lane == 63) tmp[wv] = incl;
	__syncthreads();
	if (wv == 0) {
		const u32 w = lane < PART_THREADS / 64 ? tmp[lane] : 0;
		const u32 wi = (u32) vdjx_wave_scan_add((int) w);
		if (lane < PART_THREADS / 64) tmp[lane] = wi - w;          // exclusive offset of every wave
		if (lane == 63) tmp[PART_THREADS / 64] = wi;               // grand total
	}
	__syncthreads();
	const u32 excl = tmp[wv] + incl - (a + b);
	if (2 * t < n) base[2 * t] = excl;
	if (2 * t + 1 < n) base[2 * t + 1] = excl + a;
	if (t == 0) base[n] = tmp[PART_THREADS / 64];
	__syncthreads();
}

// out[i] = src[i*step]
__global__ void k_pick_u32(const u32* __restrict__ src, u32 step, u32 n, u32* __restrict__ out) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = src[(size_t) i * step];
}

// cursors of a partition pass: cur[i] = bucket_start[i << sh]
__global__ void k_init_cursors(const u32* __restrict__ bucket_start, u32 n, u32 sh, u32* __restrict__ cur) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cur[i] = bucket_start[(size_t) i << sh];
}


// ----------------------------------------------------------------------------------------------
// K3a: LDS hash aggregation per bucket
// ----------------------------------------------------------------------------------------------
template <typename THI, u32 SLOTS = K3_SLOTS>
__device__ inline int lds_insert(u64* s_klo, THI* s_khi, u64 lo, THI hi, u32 h) {
	const THI EMPTY = (THI) ~(THI) 0, LOCKED = (THI) (EMPTY - 1);
	u32 slot = h & (SLOTS - 1);
	u32 probes = 0;
	while (probes < SLOTS) {
		// the claimant writes klo, then PUBLISHES khi with release order; a reader that sees a published khi (acquire) therefore
		// sees that klo -- relaxed accesses to two addresses may be reordered by the compiler (on LDS the orders cost no instruction)
		THI cur = vdjx_peek_acquire(&s_khi[slot]);
		if (cur == EMPTY) {
			THI old = atomicCAS(&s_khi[slot], EMPTY, LOCKED);
			if (old == EMPTY) {
				vdjx_poke(&s_klo[slot], lo);
				vdjx_poke_release(&s_khi[slot], hi);            // publish
				return (int) slot;
			}
			cur = old;
			if (cur != LOCKED) cur = vdjx_peek_acquire(&s_khi[slot]);      // (the CAS's own load carries no order)
		}
		if (cur == LOCKED) continue;                            // another lane is writing this slot: look again
		if (cur == hi && vdjx_peek(&s_klo[slot]) == lo) return (int) slot;
		slot = (slot + 1) & (SLOTS - 1);
		probes++;
	}
	return -1;
}

template <typename THI, u32 SLOTS = K3_SLOTS>
__device__ inline int lds_lookup(const u64* s_klo, const THI* s_khi, u64 lo, THI hi, u32 h) {
	const THI EMPTY = (THI) ~(THI) 0;
	u32 slot = h & (SLOTS - 1);
	for (u32 probes = 0; probes < SLOTS; probes++) {
		THI cur = s_khi[slot];
		if (cur == EMPTY) return -1;
		if (cur == hi && s_klo[slot] == lo) return (int) slot;
		slot = (slot + 1) & (SLOTS - 1);
	}
	return -1;
}

// ==============================================================================================
// Round-2 build: only the GATED instances travel (Phase A), the recount of add_to_graph comes from the edge pass (Phase B).
//
// Whether a k-mer survives (A2:467-484) depends on its gated instances only (add_to_table is never called for the others,
// A2:240-259, 398-403): about one instance in six of a Phred-noisy pool (0.95^35).  The ungated ones matter for exactly two
// numbers of a SURVIVING k-mer -- add_to_graph's frequency and first sight (A2:261-309) -- and those are counted where the
// edge pass finds the survivors anyway.  So the partition moves 16-byte {key, instance} records of the gated instances only
// (AoS: one global_load_dwordx4 per lane), and one kernel per bucket does table + prune (the former aggregate + finalize pair,
// without their candidate arrays in global memory).
// Instance ids are 38 bits: record << 6 | offset (the layout of vdjx_graph_export's first_inst), records < 2^32.
// ==============================================================================================
#define INST_BITS 38
#define INST_MASK64 ((1ull << INST_BITS) - 1ull)
#define NONE64 0xFFFFFFFFFFFFFFFFull

struct Tup16 {                  // k <= 45: the key's high part (2k-64 <= 26 bits) shares a word with the instance id
	u64 lo, x;
	typedef u32 hi_t;
	__device__ inline u64 hi() const { return x >> INST_BITS; }
	__device__ inline u64 inst() const { return x & INST_MASK64; }
	__device__ static inline Tup16 make(u64 lo, u64 hi, u64 inst) { Tup16 t; t.lo = lo; t.x = (hi << INST_BITS) | inst; return t; }
	__device__ static inline Tup16 load(const Tup16* p) { const ulonglong2 v = *(const ulonglong2*) p; Tup16 t; t.lo = v.x; t.x = v.y; return t; }
	__device__ static inline void store(Tup16* p, const Tup16& t) { *(ulonglong2*) p = make_ulonglong2(t.lo, t.x); }
};
struct Tup24 {                  // k 46..50
	u64 lo, h, i;
	typedef u64 hi_t;
	__device__ inline u64 hi() const { return h; }
	__device__ inline u64 inst() const { return i; }
	__device__ static inline Tup24 make(u64 lo, u64 hi, u64 inst) { Tup24 t; t.lo = lo; t.h = hi; t.i = inst; return t; }
	__device__ static inline Tup24 load(const Tup24* p) { return *p; }
	__device__ static inline void store(Tup24* p, const Tup24& t) { *p = t; }
};

// the offsets whose k bases are all clean: bit o is set iff none of the bits o .. o+k-1 of `bad` is (o < P) -- OR of k shifts in
// log2(k) steps instead of P window tests
__device__ inline u64 vdjx_clean_offsets(u64 bad, int k, int P) {
	u64 inv = bad;
	int cur = 1;
	while (cur * 2 <= k) { inv |= inv >> cur; cur *= 2; }
	inv |= inv >> (k - cur);
	return ~inv & (P >= 64 ? ~0ull : (1ull << P) - 1ull);
}

// One instance in six is gated, in runs (a read is clean or it is not): a loop over the offsets of a lane's record spends most of
// its trips masked off in most lanes, and the hash of a k-mer costs ~50 instructions.  The kernels below therefore list the gated
// (lane, offset) pairs of a wave's 64 records densely in LDS (16 offsets at a time: a prefix sum over the lanes' counts, a few
// cheap trips to write the entries) and hash them with all lanes busy: ~3 dense trips per 64 records instead of 16 sparse ones.
// The offset of a listed instance differs from lane to lane: the k-mer is cut out with vdjx_kmer_at_lane (vdjx_common.h: the plain
// 128-bit shift by a per-lane amount gave wrong k-mers on gfx950 when the amounts of a wave lay on both sides of 64).
#define GL_WAVE_BYTES 3072u         // per wave: 64 x 16-byte packed bases, 1024 x 2-byte entries (lane << 6 | offset)
#define GL_WAVE_BYTES_SYM GL_WAVE_BYTES

// long reads (W words per read, vdjx_pool): a lane's record in LDS is W + 2 words (two zero words behind it: the k-mer extraction
// reads three words from the k-mer's first), entries are lane << 8 | offset
#define GL_ROW_LONG (VDJX_LONG_W + 2)
#define GL_WAVE_BYTES_LONG (64u * GL_ROW_LONG * 8u + 2048u)
// the gating kernels' view of a short record: bases and the gate mask (vdjx_pool::d_lowq = not-ACGT | Phred < 20): 24 bytes, not 32
struct GateView { u64 bhi, blo, bad; };
__device__ inline GateView load_gate(const u64* __restrict__ bases, const u64* __restrict__ gate, size_t r) {
	GateView v;
	const ulonglong2 b = ((const ulonglong2*) bases)[r];
	v.bhi = b.x; v.blo = b.y;
	v.bad = gate[r];
	return v;
}
__device__ inline vdjx_mask3 load_gate3(const u64* __restrict__ gate, size_t r) {
	vdjx_mask3 b;
	b.w0 = gate[r * VDJX_LONG_M]; b.w1 = gate[r * VDJX_LONG_M + 1]; b.w2 = gate[r * VDJX_LONG_M + 2];
	return b;
}
__device__ inline vdjx_mask3 load_bad3(const u64* __restrict__ nmask, const u64* __restrict__ lowq, size_t r) {
	vdjx_mask3 b;
	b.w0 = nmask[r * VDJX_LONG_M] | (lowq ? lowq[r * VDJX_LONG_M] : 0ull);
	b.w1 = nmask[r * VDJX_LONG_M + 1] | (lowq ? lowq[r * VDJX_LONG_M + 1] : 0ull);
	b.w2 = nmask[r * VDJX_LONG_M + 2] | (lowq ? lowq[r * VDJX_LONG_M + 2] : 0ull);
	return b;
}
__device__ inline void stage_row_long(u64* row, const u64* __restrict__ bases, size_t r) {
#pragma unroll
	for (int w = 0; w < VDJX_LONG_W; w++) row[w] = bases[r * VDJX_LONG_W + w];
	row[VDJX_LONG_W] = 0; row[VDJX_LONG_W + 1] = 0;
}

// K2a': bucket sizes over the gated instances (include_kmer, A2:240-259: no 'N', every Phred >= 20), listed densely per wave
// the canonical side of a couple's instance (vdjx_pool::sym): k-mer X at offset o of record 2i is rc(X) at offset rl-k-o of record
// 2i+1; the smaller of the two is the key (k odd: they differ)
__device__ inline bool sym_canonical(const ulonglong2 f, int rl, int k, int o, u64& khi, u64& klo) {
	u64 xh, xl, yh, yl;
	vdjx_kmer_at_lane(f.x, f.y, rl, k, o, xh, xl);
	vdjx_kmer_rc(xh, xl, k, yh, yl);             // (what the couple's second record holds at offset rl - k - o: computed, not fetched -- 16 bytes per couple less to read)
	const bool flip = yh < xh || (yh == xh && yl < xl);
	khi = flip ? yh : xh;
	klo = flip ? yl : xl;
	return flip;
}

// SYM (short reads, k odd, vdjx_pool::sym): a thread takes a COUPLE of records (2q, 2q+1; R counts couples) and lists the gated
// offsets of the first; every listed instance stands for itself and its mirror in the second record, and is counted -- and later
// moved -- once, under the smaller of the two k-mers
template <bool LONG, bool SYM = false>
__global__ __launch_bounds__(HIST_THREADS) void k_gated_hist(const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                             const u64* __restrict__ lowq, size_t R, int rl, int k, u32 nb_bits,
                                                             size_t rpb, u32* __restrict__ bucket_cnt, u32 dbg_in) {
#ifdef VDJX_ABLATE                  // profiles/histdbg.py: 1 = loads and gate masks only, 2 = + the listing, 3 = + k-mers and hashes, 4 = all but the flush
	const u32 dbg = dbg_in;
#else
	const u32 dbg = 0u;
#endif
	u32 sink = 0;
	extern __shared__ __attribute__((aligned(16))) u32 hist[];       // [NB], then GL_WAVE_BYTES per wave
	const u32 NB = 1u << nb_bits;
	for (u32 i = threadIdx.x; i < NB; i += HIST_THREADS) hist[i] = 0;
	__syncthreads();
	static_assert(!(LONG && SYM), "SYM is the short-read form");
	uint8_t* wv = (uint8_t*) (hist + NB) + (threadIdx.x >> 6) * (LONG ? GL_WAVE_BYTES_LONG : SYM ? GL_WAVE_BYTES_SYM : GL_WAVE_BYTES);
	ulonglong2* wb = (ulonglong2*) wv;
	u64* wrow = (u64*) wv;
	uint16_t* wl = (uint16_t*) (wv + (LONG ? 64u * GL_ROW_LONG * 8u : 1024u));
	const u32 lane = threadIdx.x & 63u;
	const size_t r0 = (size_t) blockIdx.x * rpb;
	const size_t r1 = r0 + rpb < R ? r0 + rpb : R;
	const int P = rl - k + 1;
	for (size_t rb = r0; rb < r1; rb += HIST_THREADS) {
		const size_t r = rb + threadIdx.x;
		u64 G = 0;
		vdjx_mask3 bad{~0ull, ~0ull, ~0ull};
		if (r < r1) {
			if (LONG) { bad = load_gate3(lowq, r); stage_row_long(wrow + lane * GL_ROW_LONG, bases, r); }
			else {
				const GateView v = load_gate(bases, lowq, SYM ? 2 * r : r);
				G = vdjx_clean_offsets(v.bad, k, P);
				wb[lane] = make_ulonglong2(v.bhi, v.blo);
			}
		}
		constexpr int LSTEP = 16;                                           // offsets listed per trip (the list's room: GL_WAVE_BYTES)
		for (int ob = 0; ob < P; ob += LSTEP) {
			const u32 g = LONG ? vdjx_clean16(bad, ob, k, P) : (u32) (G >> ob) & 0xFFFFu;
			const u32 c = (u32) __popc(g);
			const u32 incl = (u32) vdjx_wave_scan_add((int) c);
			const u32 total = (u32) __builtin_amdgcn_readlane((int) incl, 63);
			if (dbg == 1) { sink += incl; continue; }
			if (!total) continue;                                     // (wave-uniform)
			u32 at = incl - c;
			for (u32 gg = g; gg; gg &= gg - 1) wl[at++] = (uint16_t) ((lane << (LONG ? 8 : 6)) | (u32) (ob + __builtin_ctz(gg)));
			vdjx_wave_lds_fence();                                    // the list and the records are read by OTHER lanes of the wave
			if (dbg == 2) { sink += wl[lane]; vdjx_wave_lds_fence(); continue; }
			for (u32 i = lane; i < total; i += 64) {
				const u32 e = wl[i];
				u64 khi, klo;
				if (LONG) vdjx_kmer_at_words(wrow + (e >> 8) * GL_ROW_LONG, k, (int) (e & 255u), khi, klo);
				else if (SYM) (void) sym_canonical(wb[e >> 6], rl, k, (int) (e & 63u), khi, klo);
				else {
					const ulonglong2 bb = wb[e >> 6];
					vdjx_kmer_at_lane(bb.x, bb.y, rl, k, (int) (e & 63u), khi, klo);
				}
				if (dbg == 3) { sink += (u32) (vdjx_bucket_mix(klo, khi) >> (64 - nb_bits)); continue; }
				atomicAdd(&hist[(u32) (vdjx_bucket_mix(klo, khi) >> (64 - nb_bits))], 1u);
			}
			vdjx_wave_lds_fence();                                    // ... before the next 16 offsets overwrite the list
		}
	}
	__syncthreads();
	if (dbg) { if (sink == 0x12345u) bucket_cnt[0] = sink; return; }
	for (u32 i = threadIdx.x; i < NB; i += HIST_THREADS) if (hist[i]) atomicAdd(&bucket_cnt[i], hist[i]);
}

// K2c' pass 1: records -> gated tuples, LDS-staged counting sort into `nbk` coarse buckets (see K2c above).
// A round takes `rr0` records (the host sizes it from the gated fraction the histogram measured, so that the stage fills: with one
// instance in six gated, rounds sized for the worst case spend their time in barriers); a round whose tuples do not fit is
// simply retried with half the records.  The gated instances of a round are listed once (dense, see above) as descriptors
// {record in the round (16 bits), offset (6), bucket (10)}; the placement pass reads the descriptors, not the records' offsets.
#define PARTR_STAGE_BYTES 98304u
template <typename TUP, bool LONG, bool SYM = false>
__global__ __launch_bounds__(PART_THREADS) void k_part_records_g(const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                                 const u64* __restrict__ lowq, size_t R, u64 rec_base, int rl, int k,
                                                                 u32 shift, u32 nbk, size_t rpb, u32 rr0, u32* __restrict__ gcur,
                                                                 TUP* __restrict__ out) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	TUP* stage = (TUP*) smem;                                         // (the waves' instance lists live here while the round is counted)
	u32* desc = (u32*) (smem + PARTR_STAGE_BYTES);
	__shared__ u32 cnt[PART_MAXB], base[PART_MAXB + 1], cur[PART_MAXB], gbase[PART_MAXB], tmp[PART_THREADS];
	__shared__ u32 s_n;
	constexpr u32 ROUND = PARTR_STAGE_BYTES / sizeof(TUP);
	constexpr int OB = LONG ? 8 : 6;                                  // offset bits (descriptor: record in the round << (OB + 10) | offset << 10 | bucket)
	const int P = rl - k + 1;
	const u32 rr_min = ROUND / (u32) P;                              // always fits
	u32 rr = rr0 > rr_min ? rr0 : rr_min;
	const u32 mask = nbk - 1;
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	static_assert(!(LONG && SYM), "SYM is the short-read form");
	uint8_t* wv = smem + wave * (LONG ? GL_WAVE_BYTES_LONG : SYM ? GL_WAVE_BYTES_SYM : GL_WAVE_BYTES);
	ulonglong2* wb = (ulonglong2*) wv;
	u64* wrow = (u64*) wv;
	uint16_t* wl = (uint16_t*) (wv + (LONG ? 64u * GL_ROW_LONG * 8u : 1024u));
	const size_t r0 = (size_t) blockIdx.x * rpb;            // (SYM: R, rpb, rr and the descriptors count COUPLES of records)
	const size_t r1 = r0 + rpb < R ? r0 + rpb : R;
	size_t rs = r0;
	while (rs < r1) {
		const size_t re = rs + rr < r1 ? rs + rr : r1;
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) cnt[i] = 0;
		if (threadIdx.x == 0) s_n = 0;
		__syncthreads();
		for (size_t rb = rs; rb < re; rb += PART_THREADS) {
			const size_t r = rb + threadIdx.x;
			u64 G = 0;
			vdjx_mask3 bad{~0ull, ~0ull, ~0ull};
			if (r < re) {
				if (LONG) { bad = load_gate3(lowq, r); stage_row_long(wrow + lane * GL_ROW_LONG, bases, r); }
				else {
					const GateView v = load_gate(bases, lowq, SYM ? 2 * r : r);
					G = vdjx_clean_offsets(v.bad, k, P);
					wb[lane] = make_ulonglong2(v.bhi, v.blo);
				}
			}
			const u32 loc0 = (u32) (rb - rs) + wave * 64u;                // this wave's first record in the round
			constexpr int LSTEP = 16;
			for (int ob = 0; ob < P; ob += LSTEP) {
				const u32 g = LONG ? vdjx_clean16(bad, ob, k, P) : (u32) (G >> ob) & 0xFFFFu;
				const u32 c = (u32) __popc(g);
				const u32 incl = (u32) vdjx_wave_scan_add((int) c);
				const u32 total = (u32) __builtin_amdgcn_readlane((int) incl, 63);
				if (!total) continue;                                     // (wave-uniform)
				u32 at = incl - c;
				for (u32 gg = g; gg; gg &= gg - 1) wl[at++] = (uint16_t) ((lane << OB) | (u32) (ob + __builtin_ctz(gg)));
				vdjx_wave_lds_fence();                                    // (see k_gated_hist)
				u32 dbase = 0;
				if (lane == 0) dbase = atomicAdd(&s_n, total);
				dbase = (u32) __builtin_amdgcn_readlane((int) dbase, 0);
				for (u32 i = lane; i < total; i += 64) {
					const u32 e = wl[i];
					u64 khi, klo;
					if (LONG) vdjx_kmer_at_words(wrow + (e >> 8) * GL_ROW_LONG, k, (int) (e & 255u), khi, klo);
					else if (SYM) (void) sym_canonical(wb[e >> 6], rl, k, (int) (e & 63u), khi, klo);
					else {
						const ulonglong2 bb = wb[e >> 6];
						vdjx_kmer_at_lane(bb.x, bb.y, rl, k, (int) (e & 63u), khi, klo);
					}
					const u32 b = (u32) (vdjx_bucket_mix(klo, khi) >> shift) & mask;
					atomicAdd(&cnt[b], 1u);
					if (dbase + i < ROUND) desc[dbase + i] = ((loc0 + (e >> OB)) << (OB + 10)) | ((e & ((1u << OB) - 1u)) << 10) | b;
				}
				vdjx_wave_lds_fence();
			}
		}
		__syncthreads();
		if (s_n > ROUND) {                                            // (uniform) too many gated instances for the stage
			rr = rr / 2 > rr_min ? rr / 2 : rr_min;
			__syncthreads();
			continue;
		}
		part_scan(cnt, base, tmp, nbk);
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) {
			cur[i] = base[i];
			gbase[i] = cnt[i] ? atomicAdd(&gcur[i], cnt[i]) : 0u;
		}
		__syncthreads();
		const u32 n = base[nbk];
		for (u32 i = threadIdx.x; i < n; i += PART_THREADS) {
			const u32 d = desc[i];
			const size_t r = rs + (d >> (OB + 10));
			const u32 o = (d >> 10) & ((1u << OB) - 1u);
			u64 khi, klo;
			u64 inst = ((rec_base + (u64) r) << OB) | (u64) o;
			if (LONG) vdjx_kmer_at_words(bases + r * VDJX_LONG_W, k, (int) o, khi, klo);      // (two words past the last record are readable: pool_alloc)
			else if (SYM) {
				// the tuple of the canonical side: this instance, or its mirror in the couple's second record
				const bool flip = sym_canonical(((const ulonglong2*) bases)[2 * r], rl, k, (int) o, khi, klo);
				inst = flip ? ((rec_base + 2 * (u64) r + 1) << OB) | (u64) (rl - k - (int) o) : ((rec_base + 2 * (u64) r) << OB) | (u64) o;
			} else {
				const ulonglong2 bb = ((const ulonglong2*) bases)[r];
				vdjx_kmer_at_lane(bb.x, bb.y, rl, k, (int) o, khi, klo);
			}
			TUP::store(&stage[atomicAdd(&cur[d & 1023u], 1u)], TUP::make(klo, khi, inst));
		}
		__syncthreads();
		for (u32 i = threadIdx.x; i < n; i += PART_THREADS) {
			const TUP t = TUP::load(&stage[i]);
			const u32 b = (u32) (vdjx_bucket_mix(t.lo, t.hi()) >> shift) & mask;
			TUP::store(&out[gbase[b] + (i - base[b])], t);
		}
		__syncthreads();
		rs = re;
	}
}

// pass 2 (see k_part_tuples)
template <typename TUP>
__global__ __launch_bounds__(PART_THREADS) void k_part_tuples_g(const TUP* __restrict__ in, const u32* __restrict__ seg_start,
                                                                u32 seg_shift, u32 slices, u32 shift, u32 sub_bits,
                                                                u32* __restrict__ gcur, TUP* __restrict__ out) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	TUP* stage = (TUP*) smem;
	__shared__ u32 cnt[PART_MAXB], base[PART_MAXB + 1], cur[PART_MAXB], gbase[PART_MAXB], tmp[PART_THREADS];
	constexpr u32 PER = PART_LDS_BYTES / sizeof(TUP) / PART_THREADS;
	constexpr u32 ROUND = PER * PART_THREADS;
	const u32 seg = blockIdx.x / slices, sl = blockIdx.x % slices;
	const u32 nbk = 1u << sub_bits, mask = nbk - 1;
	const size_t s0 = seg_start[(size_t) seg << seg_shift], s1 = seg_start[((size_t) seg + 1) << seg_shift];
	const size_t per = (s1 - s0 + slices - 1) / slices;
	const size_t t0 = s0 + (size_t) sl * per;
	const size_t t1 = t0 + per < s1 ? t0 + per : s1;
	u32* gc = gcur + ((size_t) seg << sub_bits);
	for (size_t ts = t0; ts < t1; ts += ROUND) {
		const size_t te = ts + ROUND < t1 ? ts + ROUND : t1;
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) cnt[i] = 0;
		__syncthreads();
		TUP r_t[PER];
		u32 r_b[PER];
#pragma unroll
		for (u32 j = 0; j < PER; j++) {
			const size_t t = ts + (size_t) j * PART_THREADS + threadIdx.x;
			r_b[j] = NONE32;
			if (t < te) {
				r_t[j] = TUP::load(&in[t]);
				r_b[j] = (u32) (vdjx_bucket_mix(r_t[j].lo, r_t[j].hi()) >> shift) & mask;
			}
		}
#pragma unroll
		for (u32 j = 0; j < PER; j++) if (r_b[j] != NONE32) atomicAdd(&cnt[r_b[j]], 1u);
		__syncthreads();
		part_scan(cnt, base, tmp, nbk);
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) {
			cur[i] = base[i];
			gbase[i] = cnt[i] ? atomicAdd(&gc[i], cnt[i]) : 0u;
		}
		__syncthreads();
#pragma unroll
		for (u32 j = 0; j < PER; j++) if (r_b[j] != NONE32) stage[atomicAdd(&cur[r_b[j]], 1u)] = r_t[j];
		__syncthreads();
		const u32 n = base[nbk];
		for (u32 i = threadIdx.x; i < n; i += PART_THREADS) {
			const TUP x = stage[i];
			const u32 b = (u32) (vdjx_bucket_mix(x.lo, x.hi()) >> shift) & mask;
			TUP::store(&out[gbase[b] + (i - base[b])], x);
		}
		__syncthreads();
	}
}

template <typename TUP>
__global__ __launch_bounds__(512) void k_seg_hist_g(const TUP* __restrict__ in, const u32* __restrict__ seg_start, u32 seg_shift, u32 slices,
                                                    u32 shift, u32 sub_bits, u32* __restrict__ fine_cnt) {
	__shared__ u32 h[PART_MAXB];
	const u32 nbk = 1u << sub_bits;
	for (u32 i = threadIdx.x; i < nbk; i += 512) h[i] = 0;
	__syncthreads();
	const u32 seg = blockIdx.x / slices, sl = blockIdx.x % slices;
	const size_t s0 = seg_start[(size_t) seg << seg_shift], s1 = seg_start[((size_t) seg + 1) << seg_shift];
	const size_t per = (s1 - s0 + slices - 1) / slices;
	const size_t t0 = s0 + (size_t) sl * per;
	const size_t t1 = t0 + per < s1 ? t0 + per : s1;
	for (size_t t = t0 + threadIdx.x; t < t1; t += 512) {
		const TUP x = TUP::load(&in[t]);
		atomicAdd(&h[(u32) (vdjx_bucket_mix(x.lo, x.hi()) >> shift) & (nbk - 1)], 1u);
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < nbk; i += 512) if (h[i]) atomicAdd(&fine_cnt[((size_t) seg << sub_bits) | i], h[i]);
}

// ----------------------------------------------------------------------------------------------
// K3': table + prune of one bucket in one kernel (add_to_table A2:322-367, prune_pre_graph A2:467-484).
//   sweep 1: LDS table over the bucket's (gated) tuples: count + first instance per distinct k-mer.  The distinct-read flag
//            (compare_read, A2:142-144, 349-352: "some instance's read differs from the first instance's") is the same as "two of
//            its reads differ", so ANY earlier instance proves it: a k-mer that sits at two different offsets of two records
//            and lacks the period of that distance cannot come from equal reads -- tested against whatever instance the table
//            holds at that moment, without a fetch.  A tuple that finds its k-mer proven and already TLOW instances deep is
//            SETTLED: nothing about it can matter any more (the count only grows; qualities are summed for counts below
//            TLOW only).  In a deep clone that is nearly every tuple.  The others are listed (tuple index, slot) in LDS.
//   candidates: count >= max(mf, 2)
//   sweep 2, over the listed tuples only, all lanes busy: the exact rule for candidates still open (this instance's record
//            against the first instance's, N masks included: two 16-byte gathers) and the instances of the low-count candidates
//   quality sums: one WAVE per low-count candidate, one lane per position of the k-mer, a byte per instance and lane
//   then the survivors are appended
// (Round 2 kept the tuples in registers across both sweeps and visited every one twice; buckets of more than 4,096 tuples -- half
//  of all tuples of a Zipf repertoire sit in them -- read, hashed and looked theirs up again.  profiles/README.md, round 3.)
// ----------------------------------------------------------------------------------------------
#ifndef RD_THREADS                  // measured at 10 M pairs (threads / tuples per lane in flight / waves per SIMD): 512/8/4 1.90 ms, 1024/4/8 1.77,
#define RD_THREADS 768              // 1024/2/8 1.72, 768/4/6 1.63: two workgroups per CU (LDS), six waves per SIMD at 80 registers
#define RD_UNR 4
#define RD_WAVES 6
#endif
#define RD_SLOTS 2048u               // ~1,000 distinct gated k-mers per bucket (most are read once): half full
#define RD_SLOT_BITS 11
#define RD_PQ 4096u                 // listed tuples (more: sweep 2 rescans the bucket)
#define RD_LI 512u                  // remembered instances of low-count candidates per round
#define RD_LROWS 128u               // low-count candidates per round
#define RD_MAXLOW 11u               // instances of a low-count candidate: fewer than TLOW <= 1 + (214 + 19) / 20
#define ST_CAND 1u
#define ST_MULTI 2u
#define ST_QOK 4u
#define ST_QOK_R 8u                 // SYM: the quality test of the reverse-complement k-mer
#define ST_ID_SHIFT 8               // s_state: flags below, low-count row / survivor position above
#define ST_NOID 0xFFFFFFu

struct SurvOutG { u64* lo; u64* hi; u32* gcnt; u64* gfirst; u32* n; u32 cap; u32* n_real; };
// SYM: the survivor set is handed on CLOSED under reverse complement -- a k-mer whose own quality test failed while its reverse
// complement's passed (the first instance's quirk, A2:337-339, is not mirror-symmetric) goes along as a SHADOW (this bit of its gated
// count): phase B walks the couples' first records only and needs both sides to mirror a run; shadows get no node, and edges that
// touch one are dropped where the nodes are written (k_node_emit2) -- what the reference, which never saw them survive, has.
#define GC_SHADOW 0x80000000u
#define GC_MASK 0x7FFFFFFFu

// compare_read (A2:142-144): the two records' sequences, not-ACGT masks included
__device__ inline bool reads_equal(const u64* __restrict__ bases, const u64* __restrict__ nmask, int rl, u64 a, u64 b) {
	if (rl <= VDJX_SHORT_READ_LEN) {
		const ulonglong2 x = ((const ulonglong2*) bases)[a];
		const ulonglong2 y = ((const ulonglong2*) bases)[b];
		return x.x == y.x && x.y == y.y && nmask[a] == nmask[b];
	}
	u64 d = 0;
#pragma unroll
	for (int w = 0; w < VDJX_LONG_W; w++) d |= bases[a * VDJX_LONG_W + w] ^ bases[b * VDJX_LONG_W + w];
#pragma unroll
	for (int w = 0; w < VDJX_LONG_M; w++) d |= nmask[a * VDJX_LONG_M + w] ^ nmask[b * VDJX_LONG_M + w];
	return d == 0;
}

// the table's own hash: the bucket has used up the leading bits of vdjx_mix, and inside a bucket 11 bits of ANY decent mix of the key
// do (three 32-bit multiplications instead of the nine of vdjx_mix: the multiplier runs at a quarter of the ALU rate)
__device__ inline u32 rd_hash(u64 lo, u64 hi) {
	u32 x = (u32) lo ^ ((u32) (lo >> 32) * 0x9E3779B1u) ^ (((u32) hi ^ (u32) (hi >> 32) * 0x27D4EB2Fu) * 0x85EBCA6Bu);
	x ^= x >> 15; x *= 0xC2B2AE35u; x ^= x >> 13;
	return x;
}

// cnt[idx] += 1, mn[idx] = min(mn[idx], val) for the lanes with `active`, and how many instances the slot had counted before this
// lane's (lanes of a wave that share the first active lane's slot are folded into one add and one min: vdjx_lds_count_min64)
__device__ inline u32 rd_count_min(u32* cnt, u64* mn, u32 idx, u64 val, bool active) {
	u32 c0 = 0;
	const u64 act = __ballot(active);
	if (act) {
		const int leader = __ffsll((long long) act) - 1;
		const u32 lidx = (u32) __builtin_amdgcn_readlane((int) idx, leader);
		const bool same = active && idx == lidx;
		const u64 m = __ballot(same);
		if (__popcll(m) >= 8) {
			const u32 hi_min = vdjx_wave_min(same ? (u32) (val >> 32) : 0xFFFFFFFFu);
			const u32 lo_min = vdjx_wave_min(same && (u32) (val >> 32) == hi_min ? (u32) val : 0xFFFFFFFFu);
			u32 old = 0;
			if (__lane_id() == leader) {
				old = atomicAdd(&cnt[lidx], (u32) __popcll(m));
				atomicMin((unsigned long long*) &mn[lidx], ((unsigned long long) hi_min << 32) | lo_min);
			}
			old = (u32) __builtin_amdgcn_readlane((int) old, leader);
			if (same) c0 = old + (u32) __popcll(m & ((1ull << __lane_id()) - 1ull));
			active = active && !same;
		}
	}
	if (active) { c0 = atomicAdd(&cnt[idx], 1u); atomicMin((unsigned long long*) &mn[idx], (unsigned long long) val); }
	return c0;
}

// the buckets in descending order of their size class (bit length of the tuple count): the reduce kernel's workgroups differ in
// length by a factor of sixty, and the longest should not be among the last to start.  One workgroup: counts per class, the
// classes' places, then every bucket to a place of its class (any order inside a class).
__global__ __launch_bounds__(1024) void k_bucket_order(const u32* __restrict__ bucket_start, u32 NB, u32* __restrict__ order) {
	__shared__ u32 cls_cnt[33], cls_at[33];
	if (threadIdx.x < 33) cls_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (u32 b = threadIdx.x; b < NB; b += 1024) {
		const u32 n = bucket_start[b + 1] - bucket_start[b];
		atomicAdd(&cls_cnt[n ? 32u - (u32) __builtin_clz(n) : 0u], 1u);
	}
	__syncthreads();
	if (threadIdx.x == 0) { u32 at = 0; for (int cl = 32; cl >= 0; cl--) { cls_at[cl] = at; at += cls_cnt[cl]; } }
	__syncthreads();
	for (u32 b = threadIdx.x; b < NB; b += 1024) {
		const u32 n = bucket_start[b + 1] - bucket_start[b];
		order[atomicAdd(&cls_at[n ? 32u - (u32) __builtin_clz(n) : 0u], 1u)] = b;
	}
}

// SYM (vdjx_pool::sym, k odd): the tuples are the canonical half of the instances -- a k-mer C and, implied, its reverse complement R
// with the mirrored instances (vdjx_inst_mirror).  One table entry stands for both: same count, same distinct-read verdict (two reads
// differ iff their reverse complements do); the quality sums differ only through the first instance's quirk (A2:337-339), so the
// low-count candidates are summed once per side; up to two survivors leave per entry.  (The first instance handed on for R is A
// gated instance of R, the first one only for low counts: nothing reads it after the prune.)
template <typename TUP, bool SYM = false>
__global__ __launch_bounds__(RD_THREADS, sizeof(TUP) == 16 ? RD_WAVES : RD_WAVES / 2) void k_gated_reduce(const TUP* __restrict__ tup, const u32* __restrict__ bucket_start,
                                                             const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                             vdjx_qrows quals, int rl, int ob, int k, u64 rec_base,
                                                             u32 mf, u32 cmin, u32 mqq, u32 tlow, SurvOutG so,
                                                             u64* __restrict__ g_distinct, u32* __restrict__ g_err, u32 dbg_in,
                                                             const u32* __restrict__ order) {
	// the kernel's phase-by-phase ablation (profiles/reducedbg.py) exists only in the -DVDJX_ABLATE build (make -C vdjer_amd/csrc ablate);
	// the shipped kernel keeps one switch, 9 = take the rescan path (the tests' way into a fallback real pools rarely reach)
#ifdef VDJX_ABLATE
	const u32 dbg = dbg_in;
#else
	const u32 dbg = dbg_in == 9u ? 9u : 0u;
#endif
	typedef typename TUP::hi_t THI;
	__shared__ u64 s_klo[RD_SLOTS];
	__shared__ THI s_khi[RD_SLOTS];
	__shared__ u64 s_first[RD_SLOTS];
	__shared__ u32 s_cnt[RD_SLOTS], s_state[RD_SLOTS];
	__shared__ u32 pq[RD_PQ];                                   // tuple index << RD_SLOT_BITS | slot
	__shared__ u64 low_inst[RD_LI];
	__shared__ u32 low_n[RD_LROWS];
	__shared__ unsigned short low_slot[RD_LROWS];
	__shared__ u32 s_over, s_ndist, s_nlow, s_npq, s_nsurv, s_base, s_nreal;
	const THI EMPTY = (THI) ~(THI) 0;
	const u32 b = order ? order[blockIdx.x] : blockIdx.x;
	const u32 base = bucket_start[b];
	const u32 n = bucket_start[b + 1] - base;
	const u32 tid = threadIdx.x;
	if (n == 0) return;
	const TUP* T = tup + base;
	const u32 om = (1u << ob) - 1u;
	const u32 per_low = tlow > 1 ? (tlow - 1 < RD_MAXLOW ? tlow - 1 : RD_MAXLOW) : 1;       // a low-count candidate has fewer than tlow instances
	const u32 lrows = RD_LI / per_low < RD_LROWS ? RD_LI / per_low : RD_LROWS;
	u32 S = 1;
	while ((u64) S * K3_SUB_TUPLES < n) S <<= 1;
	u32 ndist_total = 0;
	// A bucket whose distinct k-mers do not fit the table is done in S hash-selected sub-passes.  Survivors leave per sub-pass,
	// so a split must be known to fit BEFORE its first real sub-pass: S = 1 is simply tried (its overflow shows before anything
	// is emitted); any S > 1 is verified first by a keys-only dry run of all its sub-passes.
	bool verify = S > 1;
	for (;;) {
		if (tid == 0) { s_over = 0; s_ndist = 0; }
		__syncthreads();
		for (u32 s = 0; s < S; s++) {
			for (u32 i = tid; i < RD_SLOTS; i += RD_THREADS) { s_khi[i] = EMPTY; s_cnt[i] = 0; s_first[i] = NONE64; s_state[i] = 0; }
			if (tid == 0) { s_nlow = 0; s_npq = (n >> (32 - RD_SLOT_BITS) || dbg == 9) ? RD_PQ + 1 : 0; s_nsurv = 0; s_nreal = 0; }     // (an index that does not fit an entry: rescan; VDJX_RD_DBG=9: the tests' way into the rescan)
			__syncthreads();
			// ---- sweep 1
			for (u32 t0 = 0; t0 < n; t0 += RD_UNR * RD_THREADS) {
				TUP r_t[RD_UNR];
#pragma unroll
				for (int j = 0; j < RD_UNR; j++) {
					const u32 t = t0 + j * RD_THREADS + tid;
					if (t < n) r_t[j] = TUP::load(&T[t]);
				}
#pragma unroll
				for (int j = 0; j < RD_UNR; j++) {
					// (no lane leaves the body early: the count / first update folds the lanes of a wave that hit one slot -- a deep
					// clone's k-mer holds most tuples of its bucket, and same-address LDS atomics go one at a time)
					const u32 t = t0 + j * RD_THREADS + tid;
					int slot = -1;
					if (t < n) {
						const u32 h = rd_hash(r_t[j].lo, r_t[j].hi());
						if (!(S > 1 && ((h >> 12) & (S - 1)) != s)) {
							slot = lds_insert<THI, RD_SLOTS>(s_klo, s_khi, r_t[j].lo, (THI) r_t[j].hi(), h);
							if (slot < 0) s_over = 1;
						}
					}
					const bool live = slot >= 0 && !verify;
					const u64 inst = r_t[j].inst();
					const u32 c0 = rd_count_min(s_cnt, s_first, (u32) (live ? slot : 0), inst, live);
					bool list = false;
					if (live) {
						u32 st = vdjx_peek(&s_state[slot]);
						if (!(st & ST_MULTI)) {
							const u64 f = vdjx_peek(&s_first[slot]);                 // (some instance of this k-mer, at least this wave's earliest)
							if (f != NONE64 && (f >> ob) != (inst >> ob)) {
								const u32 o1 = (u32) inst & om, o0 = (u32) f & om;
								const u32 d = o1 > o0 ? o1 - o0 : o0 - o1;
								if (d && d < (u32) k && !vdjx_kmer_has_period(r_t[j].hi(), r_t[j].lo, k, d)) { atomicOr(&s_state[slot], ST_MULTI); st |= ST_MULTI; }
							}
						}
						list = !((st & ST_MULTI) && c0 + 1 >= tlow);
					}
					const u32 qi = vdjx_wave_inc(&s_npq, list);
					if (list && qi < RD_PQ) pq[qi] = (t << RD_SLOT_BITS) | (u32) slot;
				}
			}
			__syncthreads();
			if (s_over || verify) { if (s_over) break; continue; }
			if (dbg == 1) return;
			// ---- candidates (a k-mer seen once can never have two distinct reads, A2:349-352, 476)
			for (u32 i = tid; i < RD_SLOTS; i += RD_THREADS) {
				if (s_khi[i] == EMPTY) continue;
				atomicAdd(&s_ndist, SYM ? 2u : 1u);
				const u32 c = s_cnt[i];
				if (c >= cmin) s_state[i] = (s_state[i] & ST_MULTI) | ST_CAND | ((c < tlow ? atomicAdd(&s_nlow, 1u) : ST_NOID) << ST_ID_SHIFT);
			}
			__syncthreads();
			// ---- sweep 2 over the listed tuples (or, if the list ran over, over the bucket): per round of low-count candidates their
			// instances; in the first round the exact distinct-read rule for the candidates still open.  Then the quality sums of the
			// round: first instance = the RECORD's first k qualities (A2:337-339, the load-bearing bug), the others their own
			// (A2:354-361); a sum >= 214 reads as 255 (A2:356-360): the test is sum >= min(mq, 214)
			const u32 nlow = s_nlow;
			const bool listed = s_npq <= RD_PQ;
			const u32 nscan = listed ? s_npq : n;
			for (u32 l0 = 0; l0 == 0 || l0 < nlow; l0 += lrows) {
				for (u32 i = tid; i < RD_LROWS; i += RD_THREADS) low_n[i] = 0;
				__syncthreads();
				for (u32 e = tid; e < nscan; e += RD_THREADS) {
					u32 slot, t;
					TUP x;
					if (listed) {
						slot = pq[e] & (RD_SLOTS - 1); t = pq[e] >> RD_SLOT_BITS;
						if (!(vdjx_peek(&s_state[slot]) & ST_CAND)) continue;
						x = TUP::load(&T[t]);
					} else {
						x = TUP::load(&T[e]);
						const u32 h = rd_hash(x.lo, x.hi());
						if (S > 1 && ((h >> 12) & (S - 1)) != s) continue;
						const int sl = lds_lookup<THI, RD_SLOTS>(s_klo, s_khi, x.lo, (THI) x.hi(), h);
						if (sl < 0) continue;
						slot = (u32) sl;
					}
					const u32 st = vdjx_peek(&s_state[slot]);
					if (!(st & ST_CAND)) continue;
					const u64 inst = x.inst(), fi = s_first[slot];
					const u32 lid = st >> ST_ID_SHIFT;
					if (lid != ST_NOID && lid >= l0 && lid < l0 + lrows) {
						const u32 pos = atomicAdd(&low_n[lid - l0], 1u);
						if (pos < per_low) low_inst[(lid - l0) * per_low + pos] = inst;
						low_slot[lid - l0] = (unsigned short) slot;
					}
					if (l0 == 0 && !(st & ST_MULTI) && (inst >> ob) != (fi >> ob)) {
						const u32 o1 = (u32) inst & om, o0 = (u32) fi & om;
						const u32 d = o1 > o0 ? o1 - o0 : o0 - o1;
						if ((d && d < (u32) k && !vdjx_kmer_has_period(x.hi(), x.lo, k, d))
						    || !reads_equal(bases, nmask, rl, (inst >> ob) - rec_base, (fi >> ob) - rec_base)) atomicOr(&s_state[slot], ST_MULTI);
					}
				}
				__syncthreads();
				if (dbg == 4) return;
				const u32 rows = nlow - l0 < lrows ? nlow - l0 : lrows;
				const u32 lane = tid & 63u;
				for (u32 r = tid >> 6; r < (nlow ? rows : 0u); r += RD_THREADS / 64) {
					const u32 slot = low_slot[r], cnt = low_n[r];
					const u64 fi = s_first[slot];
					// (every instance's byte asked for before the first is added: the rows lie anywhere in the pool)
					u32 qv[RD_MAXLOW];
#pragma unroll
					for (u32 i = 0; i < RD_MAXLOW; i++) {
						qv[i] = 33u;
						if (i < cnt && i < per_low) {
							const u64 inst = low_inst[r * per_low + i];
							const uint8_t* q = quals.row((inst >> ob) - rec_base) + (inst == fi ? 0u : (u32) inst & om);
							if (lane < (u32) k) qv[i] = q[lane];
						}
					}
					u32 sum = 0;
#pragma unroll
					for (u32 i = 0; i < RD_MAXLOW; i++) sum += (u32) (uint8_t) (qv[i] - 33u);
					const bool lowq = lane < (u32) k && sum < mqq;
					if (!__ballot(lowq) && lane == 0) atomicOr(&s_state[slot], ST_QOK);
					if (SYM) {
						// the reverse-complement k-mer: the mirrored instances, the first of THEM with its record's first k qualities
						u64 fr = NONE64;
#pragma unroll
						for (u32 i = 0; i < RD_MAXLOW; i++)
							if (i < cnt && i < per_low) { const u64 m = vdjx_inst_mirror(low_inst[r * per_low + i], ob, rl, k); fr = m < fr ? m : fr; }
#pragma unroll
						for (u32 i = 0; i < RD_MAXLOW; i++) {
							qv[i] = 33u;
							if (i < cnt && i < per_low) {
								const u64 inst = vdjx_inst_mirror(low_inst[r * per_low + i], ob, rl, k);
								const uint8_t* q = quals.row((inst >> ob) - rec_base) + (inst == fr ? 0u : (u32) inst & om);
								if (lane < (u32) k) qv[i] = q[lane];
							}
						}
						u32 sum_r = 0;
#pragma unroll
						for (u32 i = 0; i < RD_MAXLOW; i++) sum_r += (u32) (uint8_t) (qv[i] - 33u);
						const bool lowq_r = lane < (u32) k && sum_r < mqq;
						if (!__ballot(lowq_r) && lane == 0) atomicOr(&s_state[slot], ST_QOK_R);
					}
				}
				__syncthreads();
			}
			if (dbg == 5) return;
			// ---- prune_pre_graph (A2:467-484); the global survivor counter is bumped once per workgroup and sub-pass
			for (u32 i = tid; i < RD_SLOTS; i += RD_THREADS) {
				const u32 st = s_state[i];
				bool keep = false, keep_r = false;
				if (st & ST_CAND) {
					const u32 craw = s_cnt[i];
					const u32 c = craw > 32765u ? 32765u : craw;                  // A2:345-347
					keep = c >= mf && (st & ST_MULTI) && (craw >= tlow || (st & ST_QOK));
					if (SYM) keep_r = c >= mf && (st & ST_MULTI) && (craw >= tlow || (st & ST_QOK_R));
				}
				// (reused: position among this sub-pass's survivors; SYM: both sides leave, << 2 | which of them are real)
				if (SYM) {
					s_state[i] = (keep || keep_r) ? (atomicAdd(&s_nsurv, 2u) << 2) | (u32) keep | ((u32) keep_r << 1) : NONE32;
					if (keep || keep_r) atomicAdd(&s_nreal, (u32) keep + (u32) keep_r);
				} else s_state[i] = keep ? atomicAdd(&s_nsurv, 1u) : NONE32;
			}
			__syncthreads();
			if (tid == 0) { s_base = s_nsurv ? atomicAdd(so.n, s_nsurv) : 0; if (SYM && s_nreal) atomicAdd(so.n_real, s_nreal); }
			__syncthreads();
			for (u32 i = tid; i < RD_SLOTS; i += RD_THREADS) {
				if (s_state[i] == NONE32) continue;
				u32 pos = s_base + (SYM ? s_state[i] >> 2 : s_state[i]);
				const u32 cgo = s_cnt[i] > 32765u ? 32765u : s_cnt[i];
				if (pos < so.cap) {
					so.lo[pos] = s_klo[i];
					so.hi[pos] = (u64) s_khi[i];
					so.gcnt[pos] = cgo | (SYM && !(s_state[i] & 1u) ? GC_SHADOW : 0u);
					so.gfirst[pos] = s_first[i];
				}
				if (SYM) {
					pos += 1u;
					if (pos < so.cap) {
						u64 rh, rlo;
						vdjx_kmer_rc((u64) s_khi[i], s_klo[i], k, rh, rlo);
						so.lo[pos] = rlo;
						so.hi[pos] = rh;
						so.gcnt[pos] = cgo | (!(s_state[i] & 2u) ? GC_SHADOW : 0u);
						so.gfirst[pos] = vdjx_inst_mirror(s_first[i], ob, rl, k);      // A gated instance of R, not its first: after the prune gfirst is read by
						                                                                // ONE place, the VDJX_SYNC_DEBUG print of an unseen survivor (recount_status_check) -- a place to look, not a result
					}
				}
			}
			__syncthreads();
		}
		__syncthreads();
		if (s_over) {
			if (S >= (1u << 20)) { if (tid == 0) atomicAdd(g_err, 1u); break; }
			S <<= 1;
			verify = true;
			__syncthreads();
			continue;
		}
		if (verify) { verify = false; __syncthreads(); continue; }       // the split fits: now for real
		ndist_total = s_ndist;
		break;
	}
	if (tid == 0) atomicAdd(&g_distinct[(b & 63u) * 16u], (u64) ndist_total);
}

// ----------------------------------------------------------------------------------------------
// Sharded build: every rank aggregates ITS instances per distinct k-mer (LDS, per bucket), the
// owner of the hash prefix merges the partial aggregates of all ranks, and only k-mers whose verdict needs
// per-read data (distinct-read flag still open, count below TLOW) cost a second, tiny question/answer round.
//   partial  = {key, local gated count + local distinct-read flag, local first gated instance}   (A.1 restated per rank);
//              only k-mers with a gated instance travel: five in six instances of a Phred-noisy pool are ungated
//   merged   : counts add (each capped at 32765 = MAX_FREQUENCY, A2:345-347), firsts take the minimum,
//   flag     = OR of the local flags, or: the ranks' first records differ (asked in round 2),
//   S_j      = first record's qualities (owner of the global first) + everybody else's own (round 2).
//   The ungated recount of add_to_graph (A2:261-309) needs every instance of a SURVIVING k-mer, wherever it lives:
//   each rank keeps {count, first} over all its instances per k-mer and looks the survivors up once they are known;
//   the caller reduces those two small arrays over ranks (SUM, MIN) together with the edge arrays.
// ----------------------------------------------------------------------------------------------
struct Partial { u64 lo, hi; u32 cg, h; u64 fg; };     // 32 bytes, what travels: gated count (bit 31 = local distinct-read flag), the key's table hash (rd_hash: the owner's merge
                                                       // looks every partial up three times and need not hash it again), first gated instance
#define PART_FLAG 0x80000000u
#define CNT_CAP 32765u
// SYM (pools made of couples, k odd; see k_gated_reduce): ONE aggregate per pair {k-mer, its reverse complement}, under the smaller of
// the two -- the canonical k-mer the tuples carry -- in the same 32 bytes: count and flag are the same on both sides (every gated
// instance has its mirror image in the record next door), the first gated instances are not, and both travel:
//   lo | hi (2k - 64 <= 36 bits) + count (15 bits) and flag in bits 48..63 | first gated instance (38 bits) + 26 bits of the table
//   hash | the reverse complement's first gated instance
struct PartialS { u64 lo, hic, fgh, fgr; };
static_assert(sizeof(PartialS) == sizeof(Partial), "vdjx_shard_record_bytes(0) is one number");
#define PS_H_BITS 26
#define PS_MAX_SPLIT (1u << (PS_H_BITS - 12))        // the merge picks its sub-pass by hash bits 12 and up
__device__ inline u64 ps_hi(const PartialS& p) { return p.hic & 0xFFFFFFFFFFFFull; }
__device__ inline u32 ps_cg(const PartialS& p) { const u32 c = (u32) (p.hic >> 48); return (c & 0x7FFFu) | ((c & 0x8000u) ? PART_FLAG : 0u); }
__device__ inline u64 ps_fg(const PartialS& p) { return p.fgh & INST_MASK64; }
__device__ inline u32 ps_h(const PartialS& p) { return (u32) (p.fgh >> INST_BITS); }
#define NEED_SEQ 1u
#define NEED_Q 2u
#define PID_MASK 0x3FFFFFFFu
#define REPLY_KQ 52                 // quality bytes per row (k <= 50)
#define REPLY_BYTES 408             // pid | need, first instance (u64), the record's bases and N mask (8 words: 2 + 1 or 5 + 3), 3 quality rows;
                                    // SYM: from byte 240 the reverse complement's first instance and ITS 3 quality rows

// this rank's partial aggregates of one bucket of its gated tuples: count, first instance, "saw two different reads" (any two:
// the owner ORs the ranks' flags and compares their first records); k-mers whose count is below TLOW also list their instances.
// The sweeps are k_gated_reduce's: settled tuples (k-mer proven, TLOW instances deep) are done after the first, the others are
// listed and visited by a second, dense one (a list that runs over: rescan).  A low-count k-mer's fill counter sits in the upper
// half of its count word (the count is below TLOW <= 12).
#define LG_THREADS 768
#define LG_UNR 4
#define LG_FLAG 0x80000000u
#define LG_NOLIST 0x7FFFFFFFu
// SYM (see k_gated_reduce): the tuples are the canonical half; a table entry leaves as ONE aggregate for the k-mer and its reverse
// complement together (PartialS: the reverse complement's first instance is the smallest mirrored one -- exact where it can matter,
// i.e. below TLOW instances, where the instances are listed).  All ranks cut the buckets by the canonical k-mer.
template <typename TUP, bool SYM = false>
__global__ __launch_bounds__(LG_THREADS, sizeof(TUP) == 16 ? 6 : 3) void k_gated_local(const TUP* __restrict__ tup, const u32* __restrict__ bucket_start,
                                                            const u64* __restrict__ bases, const u64* __restrict__ nmask, u64 rec_base, int k, int rl, int ob,
                                                            u32 tlow, Partial* __restrict__ sparse_g, u32* __restrict__ sparse_ref,
                                                            u32* __restrict__ nd_g, u64* __restrict__ low_inst, u32* __restrict__ g_err,
                                                            const u32* __restrict__ order) {
	typedef typename TUP::hi_t THI;
	const size_t obase = (size_t) bucket_start[order ? order[blockIdx.x] : blockIdx.x];
	__shared__ u64 s_klo[LOCAL_SLOTS];
	__shared__ THI s_khi[LOCAL_SLOTS];
	__shared__ u64 s_mg[LOCAL_SLOTS];
	__shared__ u32 s_cg[LOCAL_SLOTS], s_st[LOCAL_SLOTS];        // s_st: LG_FLAG | offset of the k-mer's instance list in the bucket (LG_NOLIST: none)
	__shared__ u32 pq[RD_PQ];
	__shared__ u32 s_ng, s_nlow, s_over, s_npq;
	static_assert(LOCAL_SLOTS == (1u << RD_SLOT_BITS), "list entries carry the slot in RD_SLOT_BITS bits");
	const THI EMPTY = (THI) ~(THI) 0;
	const u32 b = order ? order[blockIdx.x] : blockIdx.x;      // (largest buckets first: k_bucket_order)
	const u32 base = bucket_start[b];
	const u32 n = bucket_start[b + 1] - base;
	const u32 tid = threadIdx.x;
	const u32 om = (1u << ob) - 1u;
	if (n == 0) { if (tid == 0) nd_g[b] = 0; return; }
	const TUP* T = tup + base;
	u32 S = 1;
	while ((u64) S * K3_SUB_TUPLES < n) S <<= 1;
	for (;;) {
		if (tid == 0) { s_ng = 0; s_nlow = 0; s_over = 0; }
		for (u32 s = 0; s < S; s++) {
			for (u32 i = tid; i < LOCAL_SLOTS; i += LG_THREADS) { s_khi[i] = EMPTY; s_cg[i] = 0; s_mg[i] = NONE64; s_st[i] = 0; }
			if (tid == 0) s_npq = n >> (32 - RD_SLOT_BITS) ? RD_PQ + 1 : 0;
			__syncthreads();
			// ---- sweep 1 (k_gated_reduce)
			for (u32 t0 = 0; t0 < n; t0 += LG_UNR * LG_THREADS) {
				TUP r_t[LG_UNR];
#pragma unroll
				for (int j = 0; j < LG_UNR; j++) {
					const u32 t = t0 + j * LG_THREADS + tid;
					if (t < n) r_t[j] = TUP::load(&T[t]);
				}
#pragma unroll
				for (int j = 0; j < LG_UNR; j++) {
					const u32 t = t0 + j * LG_THREADS + tid;
					int slot = -1;
					if (t < n) {
						const u32 h = rd_hash(r_t[j].lo, r_t[j].hi());
						if (!(S > 1 && ((h >> 12) & (S - 1)) != s)) {
							slot = lds_insert<THI, LOCAL_SLOTS>(s_klo, s_khi, r_t[j].lo, (THI) r_t[j].hi(), h);
							if (slot < 0) s_over = 1;
						}
					}
					const bool live = slot >= 0;
					const u64 inst = r_t[j].inst();
					const u32 c0 = rd_count_min(s_cg, s_mg, (u32) (live ? slot : 0), inst, live);
					bool list = false;
					if (live) {
						u32 st = vdjx_peek(&s_st[slot]);
						if (!(st & LG_FLAG)) {
							const u64 f = vdjx_peek(&s_mg[slot]);
							if (f != NONE64 && (f >> ob) != (inst >> ob)) {
								const u32 o1 = (u32) inst & om, o0 = (u32) f & om;
								const u32 d = o1 > o0 ? o1 - o0 : o0 - o1;
								if (d && d < (u32) k && !vdjx_kmer_has_period(r_t[j].hi(), r_t[j].lo, k, d)) { atomicOr(&s_st[slot], LG_FLAG); st |= LG_FLAG; }
							}
						}
						list = !((st & LG_FLAG) && c0 + 1 >= tlow);
					}
					const u32 qi = vdjx_wave_inc(&s_npq, list);
					if (list && qi < RD_PQ) pq[qi] = (t << RD_SLOT_BITS) | (u32) slot;
				}
			}
			__syncthreads();
			if (s_over) break;
			// keys whose count alone cannot pass the quality test (count < TLOW) list their instances: the owner may ask
			for (u32 i = tid; i < LOCAL_SLOTS; i += LG_THREADS) {
				const u32 cg = s_cg[i];
				s_st[i] = (s_st[i] & LG_FLAG) | ((cg && cg < tlow) ? atomicAdd(&s_nlow, cg) : LG_NOLIST);
			}
			__syncthreads();
			// ---- sweep 2 over the listed tuples (or the bucket): the lists, and the exact distinct-read rule for keys still open
			const bool listed = s_npq <= RD_PQ;
			const u32 nscan = listed ? s_npq : n;
			for (u32 e = tid; e < nscan; e += LG_THREADS) {
				u32 slot;
				TUP x;
				if (listed) {
					slot = pq[e] & (LOCAL_SLOTS - 1);
					const u32 st0 = vdjx_peek(&s_st[slot]);
					if ((st0 & LG_NOLIST) == LG_NOLIST && (st0 & LG_FLAG)) continue;       // (neither a list nor an open flag)
					x = TUP::load(&T[pq[e] >> RD_SLOT_BITS]);
				} else {
					x = TUP::load(&T[e]);
					const u32 h = rd_hash(x.lo, x.hi());
					if (S > 1 && ((h >> 12) & (S - 1)) != s) continue;
					const int sl = lds_lookup<THI, LOCAL_SLOTS>(s_klo, s_khi, x.lo, (THI) x.hi(), h);
					if (sl < 0) continue;
					slot = (u32) sl;
				}
				const u32 st = vdjx_peek(&s_st[slot]);
				const u32 loff = st & LG_NOLIST;
				const u64 inst = x.inst();
				u32 cg = vdjx_peek(&s_cg[slot]);
				if (loff != LG_NOLIST) {
					cg &= 0xFFFFu;
					low_inst[base + loff + (atomicAdd(&s_cg[slot], 0x10000u) >> 16)] = inst;
				}
				if (cg < 2 || (st & LG_FLAG)) continue;
				const u64 fi = s_mg[slot];
				const u64 rec = (inst >> ob) - rec_base, frec = (fi >> ob) - rec_base;
				if (rec != frec) {
					const u32 o1 = (u32) inst & om, o0 = (u32) fi & om;
					const u32 d = o1 > o0 ? o1 - o0 : o0 - o1;
					if (d && d < (u32) k && !vdjx_kmer_has_period(x.hi(), x.lo, k, d)) atomicOr(&s_st[slot], LG_FLAG);
					else if (!reads_equal(bases, nmask, rl, rec, frec)) atomicOr(&s_st[slot], LG_FLAG);
				}
			}
			__syncthreads();
			for (u32 i = tid; i < LOCAL_SLOTS; i += LG_THREADS) {
				if (s_khi[i] == EMPTY) continue;
				const u32 st = s_st[i];
				const u32 loff = st & LG_NOLIST;
				const u32 cg = loff != LG_NOLIST ? s_cg[i] & 0xFFFFu : s_cg[i];
				Partial p;
				p.lo = s_klo[i]; p.hi = (u64) s_khi[i];
				p.fg = s_mg[i];
				p.cg = (cg > CNT_CAP ? CNT_CAP : cg) | ((st & LG_FLAG) ? PART_FLAG : 0u);
				p.h = rd_hash(p.lo, p.hi);
				const u32 gi = atomicAdd(&s_ng, 1u);
				sparse_ref[obase + gi] = loff != LG_NOLIST ? base + loff : NONE32;
				if (SYM) {
					// the reverse complement's first gated instance: the smallest mirrored one, exactly where the owner may ask about it
					u64 fr = vdjx_inst_mirror(p.fg, ob, rl, k);
					if (loff != LG_NOLIST)
						for (u32 j = 0; j < cg; j++) { const u64 m = vdjx_inst_mirror(low_inst[base + loff + j], ob, rl, k); fr = m < fr ? m : fr; }
					PartialS q;
					q.lo = p.lo;
					q.hic = p.hi | ((u64) ((p.cg & 0x7FFFu) | ((p.cg & PART_FLAG) ? 0x8000u : 0u)) << 48);
					q.fgh = p.fg | ((u64) (p.h & ((1u << PS_H_BITS) - 1u)) << INST_BITS);
					q.fgr = fr;
					((PartialS*) sparse_g)[obase + gi] = q;
				} else sparse_g[obase + gi] = p;
			}
			__syncthreads();
		}
		if (!s_over) break;
		// (nothing leaves the bucket before all its sub-passes are done: the counters restart with the finer split)
		S <<= 1;
		if (S > (1u << 20)) { if (tid == 0) atomicAdd(g_err, 1u); break; }
		__syncthreads();
	}
	if (tid == 0) nd_g[b] = s_over ? 0 : s_ng;
}

__global__ __launch_bounds__(256) void k_compact_partials(const Partial* __restrict__ sparse, const u32* __restrict__ sparse_ref,
                                                          const u32* __restrict__ bucket_start, const u32* __restrict__ nd,
                                                          const u32* __restrict__ dstart, Partial* __restrict__ dense, u32* __restrict__ dense_ref,
                                                          const u32* __restrict__ scan, int ob, bool sym) {
	const u32 b = blockIdx.x;
	const u64* src = (const u64*) (sparse + (size_t) bucket_start[b]);
	u64* dst = (u64*) (dense + dstart[b]);
	const u32 m = nd[b];
	const u64 om = (1ull << ob) - 1ull;
	for (u32 i = threadIdx.x; i < m * 4; i += 256) {
		u64 v = src[i];
		if (scan) {          // share mode: the first instances leave as scan positions (PartialS: both of them, the first below its hash bits)
			if ((i & 3u) == 3u) v = ((u64) scan[v >> ob] << ob) | (v & om);
			else if (sym && (i & 3u) == 2u) { const u64 f = v & INST_MASK64; v = (v & ~INST_MASK64) | ((u64) scan[f >> ob] << ob) | (f & om); }
		}
		dst[i] = v;
	}
	for (u32 i = threadIdx.x; i < m; i += 256) dense_ref[dstart[b] + i] = sparse_ref[(size_t) bucket_start[b] + i];
}

// seg_cnt[s*NBo + b] (what arrived) -> seg_off (absolute offsets into the receive buffer); src_base[s] = start of source s.
// One workgroup per source: exclusive scan of its NBo counts.
__global__ __launch_bounds__(1024) void k_seg_offsets(const u32* __restrict__ seg_cnt, const u32* __restrict__ src_base, u32 G, u32 NBo,
                                                      u32* __restrict__ seg_off) {
	__shared__ u32 part[1024];
	const u32 s = blockIdx.x;
	const u32* cnt = seg_cnt + (size_t) s * NBo;
	u32* off = seg_off + (size_t) s * (NBo + 1);
	const u32 per = (NBo + 1023) / 1024;
	const u32 lo = threadIdx.x * per;
	const u32 hi = lo + per < NBo ? lo + per : NBo;
	u32 sum = 0;
	for (u32 i = lo; i < hi; i++) sum += cnt[i];
	part[threadIdx.x] = sum;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		const u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u32 run = src_base[s] + (threadIdx.x ? part[threadIdx.x - 1] : 0);
	for (u32 i = lo; i < hi; i++) { off[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) off[NBo] = src_base[s] + part[1023];
}

struct PendOut { u64* lo; u64* hi; u32* cg; u64* mg; u32* need; u32* n; u32 cap; u64* mgr; };      // (mgr: SYM, the reverse complement's first gated instance)

#define MERGE_THREADS 512
#define MERGE_SLOTS 2048u             // a bucket holds ~1,000 distinct gated k-mers of all ranks together at 3,000 tuples per bucket
// SYM: the aggregates are PartialS, one per pair of mirrored k-mers under the canonical one: one table entry decides both (count and
// flag are the same on both sides); what is decided here leaves as two survivors side by side, what is still open as one question
// s_st, one word per table entry with two lives.  While the aggregates arrive: sources that sent the k-mer (at most 256) | MS_FLAG per
// source that saw two different reads.  Once decided: NONE32 nothing to do | place among this sub-pass's survivors (top two bits
// clear) | question number | what is asked << 30 (NEED_SEQ, NEED_Q: never zero)
#define MS_FLAG 0x10000u
__device__ inline bool ms_open(u32 st) { return st != NONE32 && (st >> 30) != 0u; }
template <typename THI, bool SYM = false>
__global__ __launch_bounds__(MERGE_THREADS) void k_bucket_merge(const Partial* __restrict__ recv, const u32* __restrict__ seg_off,
                                                                u32 MG, u32 G, u32 NBo,
                                                                const u32* __restrict__ src_base, u32 s_mult, u32 cmin, u32 tlow,
                                                                SurvOutG so, PendOut po,
                                                                uint2* __restrict__ queries, u32* __restrict__ g_nq,
                                                                u64* __restrict__ g_distinct, u32* __restrict__ g_err, int k) {
	__shared__ u64 s_klo[MERGE_SLOTS];
	__shared__ THI s_khi[MERGE_SLOTS];
	__shared__ u64 s_mg[MERGE_SLOTS];
	__shared__ u64 s_mgr[SYM ? MERGE_SLOTS : 1];
	__shared__ u32 s_cg[MERGE_SLOTS], s_st[MERGE_SLOTS];
	__shared__ u32 s_over, s_ndist, s_total, s_ns, s_np, s_sbase, s_pbase;
	__shared__ u32 s_pend[MERGE_SLOTS / 32];               // bit h & (MERGE_SLOTS - 1): a k-mer with that home slot has a question open
	const THI EMPTY = (THI) ~(THI) 0;
	const u32 b = blockIdx.x;
	const u32 tid = threadIdx.x;
	// one workgroup merges MG consecutive buckets (their partials are contiguous in every source's list): enough work per table
	if (tid == 0) {
		u32 tot = 0;
		for (u32 s = 0; s < G; s++) tot += seg_off[(size_t) s * (NBo + 1) + (b + 1) * MG] - seg_off[(size_t) s * (NBo + 1) + b * MG];
		s_total = tot;
	}
	__syncthreads();
	const u32 total = s_total;
	if (total == 0) return;
	u32 S = 1;
	while ((u64) S * (MERGE_SLOTS * 3 / 4) < total) S <<= 1;      // (a sub-pass's k-mers cannot outnumber its partials: the table never fills)
	S *= s_mult;
	if (SYM && S > PS_MAX_SPLIT) { if (tid == 0) atomicAdd(g_err, 1u); return; }        // (a PartialS carries PS_H_BITS bits of the hash)
	// an aggregate's key and hash, whichever way it is packed
	auto key_of = [&](u32 at, u64& lo, u64& hi, u32& h) {
		if (SYM) { const PartialS q = ((const PartialS*) recv)[at]; lo = q.lo; hi = ps_hi(q); h = ps_h(q); }
		else { const Partial q = recv[at]; lo = q.lo; hi = q.hi; h = q.h; }
	};
	{
		if (tid == 0) { s_over = 0; s_ndist = 0; }
		for (u32 sp = 0; sp < S; sp++) {
			for (u32 i = tid; i < MERGE_SLOTS; i += MERGE_THREADS) { s_khi[i] = EMPTY; s_cg[i] = 0; s_mg[i] = NONE64; s_st[i] = 0; if (SYM) s_mgr[i] = NONE64; }
			__syncthreads();
			for (u32 s = 0; s < G; s++) {
				const u32 off = seg_off[(size_t) s * (NBo + 1) + b * MG], cnt = seg_off[(size_t) s * (NBo + 1) + (b + 1) * MG] - off;
				for (u32 i = tid; i < cnt; i += MERGE_THREADS) {
					u64 lo, hi, fg, fgr = 0;
					u32 h, cg;
					if (SYM) { const PartialS q = ((const PartialS*) recv)[off + i]; lo = q.lo; hi = ps_hi(q); h = ps_h(q); cg = ps_cg(q); fg = ps_fg(q); fgr = q.fgr; }
					else { const Partial q = recv[off + i]; lo = q.lo; hi = q.hi; h = q.h; cg = q.cg; fg = q.fg; }
					if (((h >> 12) & (S - 1)) != sp) continue;
					const int slot = lds_insert<THI, MERGE_SLOTS>(s_klo, s_khi, lo, (THI) hi, h);
					if (slot < 0) { s_over = 1; continue; }
					atomicAdd(&s_cg[slot], cg & ~PART_FLAG);
					atomicMin((unsigned long long*) &s_mg[slot], (unsigned long long) fg);
					if (SYM) atomicMin((unsigned long long*) &s_mgr[slot], (unsigned long long) fgr);
					atomicAdd(&s_st[slot], 1u | ((cg & PART_FLAG) ? MS_FLAG : 0u));          // (the flag: added once per source, at most 256 times -- it may carry into the bits above, all of which read as "set")
				}
			}
			__syncthreads();
			if (s_over) break;
			// decide; the global output counters are bumped ONCE per workgroup (a few addresses shared by every workgroup
			// serialise in L2: per-k-mer or per-wave bumps cost more than the merge itself)
			if (tid == 0) { s_ns = 0; s_np = 0; }
			for (u32 i = tid; i < MERGE_SLOTS / 32; i += MERGE_THREADS) s_pend[i] = 0;
			__syncthreads();
			for (u32 i0 = 0; i0 < MERGE_SLOTS; i0 += MERGE_THREADS) {
				const u32 i = i0 + tid;
				u32 need = 0, cg = 0;
				bool live = s_khi[i] != EMPTY;
				if (live) {
					cg = s_cg[i];
					atomicAdd(&s_ndist, SYM ? 2u : 1u);
					if (cg < cmin) live = false;                 // count >= max(mf, 2): A2:349-352,476
				}
				const u32 st0 = s_st[i];
				if (live && !(st0 >> 16)) {
					if ((st0 & 0xFFFFu) < 2) live = false;       // one rank holds every gated instance and saw one read only
					else need |= NEED_SEQ;
				}
				if (live && cg < tlow) need |= NEED_Q;
				if (live && !need) s_st[i] = atomicAdd(&s_ns, SYM ? 2u : 1u);
				else s_st[i] = NONE32;
				if (live && need) {
					s_st[i] = atomicAdd(&s_np, 1u) | (need << 30);
					const u32 hm = rd_hash(s_klo[i], (u64) s_khi[i]) & (MERGE_SLOTS - 1);
					atomicOr(&s_pend[hm >> 5], 1u << (hm & 31));
				}
			}
			__syncthreads();
			if (tid == 0) {
				s_sbase = s_ns ? atomicAdd(so.n, s_ns) : 0;
				s_pbase = s_np ? atomicAdd(po.n, s_np) : 0;
			}
			__syncthreads();
			for (u32 i0 = 0; i0 < MERGE_SLOTS; i0 += MERGE_THREADS) {
				const u32 i = i0 + tid;
				const u32 st = s_st[i];
				if (st != NONE32 && !(st >> 30)) {
					const u32 pos = s_sbase + st;
					if (pos < so.cap) {
						const u32 cg = s_cg[i];
						so.lo[pos] = s_klo[i]; so.hi[pos] = (u64) s_khi[i];
						so.gcnt[pos] = cg > CNT_CAP ? CNT_CAP : cg; so.gfirst[pos] = s_mg[i];
					}
					if (SYM && pos + 1 < so.cap) {
						const u32 cg = s_cg[i];
						u64 rh, rlo;
						vdjx_kmer_rc((u64) s_khi[i], s_klo[i], k, rh, rlo);
						so.lo[pos + 1] = rlo; so.hi[pos + 1] = rh;
						so.gcnt[pos + 1] = cg > CNT_CAP ? CNT_CAP : cg; so.gfirst[pos + 1] = s_mgr[i];
					}
				} else if (st != NONE32) {
					const u32 need = st >> 30;
					const u32 pid = s_pbase + (st & PID_MASK);
					if (pid < po.cap) {
						po.lo[pid] = s_klo[i]; po.hi[pid] = (u64) s_khi[i];
						po.cg[pid] = s_cg[i]; po.mg[pid] = s_mg[i];
						if (SYM) po.mgr[pid] = s_mgr[i];
						po.need[pid] = need;
					}
					s_st[i] = pid | (need << 30);
				}
			}
			__syncthreads();
			// questions to every rank that holds gated instances of an open k-mer
			for (u32 s = 0; s < G; s++) {
				const u32 off = seg_off[(size_t) s * (NBo + 1) + b * MG], cnt = seg_off[(size_t) s * (NBo + 1) + (b + 1) * MG] - off;
				// count this source's questions, reserve their places with ONE bump of the source's counter, then write them
				if (tid == 0) s_ns = 0;
				__syncthreads();
				u32 mine = 0;
				for (u32 i = tid; i < cnt; i += MERGE_THREADS) {
					u64 lo, hi;
					u32 h;
					key_of(off + i, lo, hi, h);
					if (((h >> 12) & (S - 1)) != sp) continue;
					if (!((s_pend[(h & (MERGE_SLOTS - 1)) >> 5] >> (h & 31)) & 1u)) continue;       // (few k-mers have a question open: most partials stop here)
					const int slot = lds_lookup<THI, MERGE_SLOTS>(s_klo, s_khi, lo, (THI) hi, h);
					if (slot >= 0 && ms_open(s_st[slot])) mine++;
				}
				if (mine) atomicAdd(&s_ns, mine);
				__syncthreads();
				if (tid == 0) { s_sbase = s_ns ? atomicAdd(&g_nq[s], s_ns) : 0; s_np = 0; }
				__syncthreads();
				if (s_ns) {
					for (u32 i = tid; i < cnt; i += MERGE_THREADS) {
						u64 lo, hi;
						u32 h;
						key_of(off + i, lo, hi, h);
						if (((h >> 12) & (S - 1)) != sp) continue;
						if (!((s_pend[(h & (MERGE_SLOTS - 1)) >> 5] >> (h & 31)) & 1u)) continue;
						const int slot = lds_lookup<THI, MERGE_SLOTS>(s_klo, s_khi, lo, (THI) hi, h);
						if (slot < 0) continue;
						const u32 pid = s_st[slot];
						if (ms_open(pid)) queries[src_base[s] + s_sbase + atomicAdd(&s_np, 1u)] = make_uint2(off + i - src_base[s], pid);
					}
				}
				__syncthreads();
			}
			__syncthreads();
		}
		// a sub-pass that does not fit the table: earlier sub-passes have already appended their results, so the HOST
		// resets the outputs and relaunches everything with a larger split (s_mult); rare by construction
		if (s_over && tid == 0) atomicAdd(g_err, 1u);
	}
	if (tid == 0) atomicAdd(g_distinct, (u64) s_ndist);
}

// answers of the rank that holds the instances: one wave per question
//   [0,4) pid | need<<30   [8,16) this rank's first gated instance   [16,80) its record's bases, then its N mask (2 + 1 words, or 5 + 3
//   for reads of more than 64 bases; the rest zero)
//   [80,132) QS: sum of the own qualities of this rank's OTHER gated instances (saturating at 255: >= 214 reads as 255 anyway)
//   [132,184) the first instance's own qualities   [184,236) its RECORD's first k qualities (A2:337-339)
//   SYM: [240,248) the first gated instance of the reverse complement, [248,404) the same three rows for it (over the mirrored instances)
#define REPLY_Q0 80
#define REPLY_R0 240
#define REPLY_RQ0 248
template <bool SYM>
__global__ __launch_bounds__(64) void k_shard_reply(const uint2* __restrict__ queries, u32 nq, const u32* __restrict__ owner_off, u32 G,
                                                    const Partial* __restrict__ dense, const u32* __restrict__ dense_ref,
                                                    const u32* __restrict__ dstart, u32 NBo, const u64* __restrict__ low_inst,
                                                    const u64* __restrict__ bases, const u64* __restrict__ nmask,
                                                    vdjx_qrows quals, u64 rec_base, const u32* __restrict__ scan, u32 n_local, int k, int rl, int ob,
                                                    uint8_t* __restrict__ replies) {
	const u32 qi = blockIdx.x;                                // (a wave per question and workgroup; four waves to a workgroup: 0.25 -> 0.35 ms)
	if (qi >= nq) return;
	const u32 lane = threadIdx.x;
	u32 o = 0;
	while (o + 1 < G && owner_off[o + 1] <= qi) o++;
	const uint2 q = queries[qi];
	const u32 di = dstart[(size_t) o * NBo] + q.x;
	u64 finst, finst_r = 0;
	u32 p_cg;
	if (SYM) { const PartialS p = ((const PartialS*) dense)[di]; finst = ps_fg(p); finst_r = p.fgr; p_cg = ps_cg(p); }
	else { const Partial p = dense[di]; finst = p.fg; p_cg = p.cg; }
	const u32 need = q.y >> 30;
	const u32 om = (1u << ob) - 1u;
	// share mode: the aggregate carries a scan position; the local record is where the share holds it (the share ascends)
	auto local_rec = [&](u64 inst) -> u64 {
		if (!scan) return (inst >> ob) - rec_base;
		const u32 g = (u32) (inst >> ob);
		u32 lo = 0, hi = n_local;
		while (lo + 1 < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (scan[mid] <= g) lo = mid; else hi = mid; }
		return lo;
	};
	const u64 frec = local_rec(finst);
	const u32 foff = (u32) finst & om;
	const u64 finst_local = (frec << ob) | foff;
	const u64 frec_r = SYM ? local_rec(finst_r) : 0;
	const u32 foff_r = (u32) finst_r & om;
	const u64 finst_r_local = (frec_r << ob) | foff_r;
	uint8_t* out = replies + (size_t) qi * REPLY_BYTES;
	if (lane == 0) {
		((u32*) out)[0] = q.y;
		((u32*) out)[1] = 0;
		((u64*) out)[1] = finst;
		if (SYM) ((u64*) out)[REPLY_R0 / 8] = finst_r;
	}
	if (!SYM && lane < (REPLY_BYTES - REPLY_R0) / 4) ((u32*) (out + REPLY_R0))[lane] = 0;      // (the reverse complement's part of an answer is not used: zeros, not stale device bytes, on the wire)
	if (lane < 8) {
		const int W = rl <= VDJX_SHORT_READ_LEN ? 2 : VDJX_LONG_W, M = rl <= VDJX_SHORT_READ_LEN ? 1 : VDJX_LONG_M;
		((u64*) out)[2 + lane] = (int) lane < W ? bases[frec * W + lane] : ((int) lane < W + M ? nmask[frec * M + (lane - W)] : 0ull);
	}
	if ((int) lane >= k) return;
	u32 acc = 0, acc_r = 0;
	const u32 ref = dense_ref[di];
	if ((need & NEED_Q) && ref != NONE32) {          // a question about the sums only comes for a count below TLOW: the list exists
		const u32 cg = p_cg & ~PART_FLAG;
		for (u32 i = 0; i < cg; i++) {
			const u64 inst = low_inst[ref + i];
			if (inst != (scan ? finst_local : finst)) acc += (u32) (uint8_t) (quals.row((inst >> ob) - rec_base)[((u32) inst & om) + lane] - 33);
			if (SYM) {
				const u64 m = vdjx_inst_mirror(inst, ob, rl, k);
				if (m != (scan ? finst_r_local : finst_r)) acc_r += (u32) (uint8_t) (quals.row((m >> ob) - rec_base)[((u32) m & om) + lane] - 33);
			}
		}
	}
	const uint8_t* fr = quals.row(frec);
	out[REPLY_Q0 + lane] = (uint8_t) (acc > 255u ? 255u : acc);
	out[REPLY_Q0 + REPLY_KQ + lane] = (uint8_t) (fr[foff + lane] - 33);
	out[REPLY_Q0 + 2 * REPLY_KQ + lane] = (uint8_t) (fr[lane] - 33);
	if (SYM) {
		const uint8_t* frr = quals.row(frec_r);
		out[REPLY_RQ0 + lane] = (uint8_t) (acc_r > 255u ? 255u : acc_r);
		out[REPLY_RQ0 + REPLY_KQ + lane] = (uint8_t) (frr[foff_r + lane] - 33);
		out[REPLY_RQ0 + 2 * REPLY_KQ + lane] = (uint8_t) (frr[lane] - 33);
	}
}

// owner: which answer comes from the rank of the global first instance
// (SYM: and which from the rank of the reverse complement's -- p_r0[np + pid])
__global__ void k_resolve_first(const uint8_t* __restrict__ replies, u32 nr, const u64* __restrict__ p_mg, u32* __restrict__ p_r0,
                                const u64* __restrict__ p_mgr, u32 np) {
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nr) return;
	const uint8_t* me = replies + (size_t) r * REPLY_BYTES;
	const u32 pid = ((const u32*) me)[0] & PID_MASK;
	if (((const u64*) me)[1] == p_mg[pid]) p_r0[pid] = r;
	if (p_mgr && ((const u64*) me)[REPLY_R0 / 8] == p_mgr[pid]) p_r0[np + pid] = r;
}

// one wave per answer: lane j adds the answer's j-th quality sum (64 consecutive words per wave instead of k scattered atomics of
// one thread)
// SYM (np != 0): the sums of the reverse complement in the second half of p_S (np * 64 words on)
__global__ __launch_bounds__(256) void k_resolve_add(const uint8_t* __restrict__ replies, u32 nr, const u32* __restrict__ p_r0, int k, u32* __restrict__ p_fl,
                                                     u32* __restrict__ p_S, u32 np) {
	const u32 r = blockIdx.x * 4u + (threadIdx.x >> 6);
	const u32 lane = threadIdx.x & 63u;
	if (r >= nr) return;
	const uint8_t* me = replies + (size_t) r * REPLY_BYTES;
	const u32 w0 = ((const u32*) me)[0];
	const u32 pid = w0 & PID_MASK, need = w0 >> 30;
	const u32 r0 = p_r0[pid];
	if (r0 == NONE32) return;                   // cannot happen: the owner of the minimum always answers
	if ((need & NEED_SEQ) && r != r0 && lane == 0) {
		const u64* a = (const u64*) me;
		const u64* b = (const u64*) (replies + (size_t) r0 * REPLY_BYTES);
		u64 d = 0;
#pragma unroll
		for (int w = 2; w < 10; w++) d |= a[w] ^ b[w];
		if (d) p_fl[pid] = 1;
	}
	if ((need & NEED_Q) && (int) lane < k) {
		const uint8_t* first = me + REPLY_Q0 + (r == r0 ? 2 * REPLY_KQ : REPLY_KQ);
		atomicAdd(&p_S[(size_t) pid * 64 + lane], (u32) me[REPLY_Q0 + lane] + (u32) first[lane]);
		if (np) {
			const uint8_t* first_r = me + REPLY_RQ0 + (r == p_r0[np + pid] ? 2 * REPLY_KQ : REPLY_KQ);
			atomicAdd(&p_S[((size_t) np + pid) * 64 + lane], (u32) me[REPLY_RQ0 + lane] + (u32) first_r[lane]);
		}
	}
}

// SYM: an open pair {k-mer, reverse complement} has one verdict on its reads and one per side on its qualities (the first instance's
// quirk, A2:337-339, is not mirror-symmetric); if either side stays, both leave, the other as a SHADOW (GC_SHADOW, see k_gated_reduce)
template <bool SYM>
__global__ void k_resolve_keep(PendOut po, u32 np, const u32* __restrict__ p_fl, const u32* __restrict__ p_S, int k, u32 mf, u32 mqq, u32 tlow,
                               SurvOutG so) {
	const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = p < np;
	const u32 need = in ? po.need[p] : 0u;
	bool keep = in && ((need & NEED_SEQ) ? p_fl[p] != 0 : true);
	const u32 cg = in ? po.cg[p] : 0u;
	const u32 cgc = cg > CNT_CAP ? CNT_CAP : cg;
	keep = keep && cgc >= mf;
	bool keep_r = keep;
	if (keep && cg < tlow) {
		for (int j = 0; j < k; j++) if (p_S[(size_t) p * 64 + j] < mqq) { keep = false; break; }
		if (SYM) for (int j = 0; j < k; j++) if (p_S[((size_t) np + p) * 64 + j] < mqq) { keep_r = false; break; }
	}
	if (!SYM) {
		const u32 pos = vdjx_wave_inc(so.n, keep);
		if (keep && pos < so.cap) {
			so.lo[pos] = po.lo[p]; so.hi[pos] = po.hi[p];
			so.gcnt[pos] = cgc; so.gfirst[pos] = po.mg[p];
		}
		return;
	}
	const bool any = keep || keep_r;
	const u32 pos = vdjx_wave_inc(so.n, any, 2u);
	if (any && pos + 1 < so.cap) {
		so.lo[pos] = po.lo[p]; so.hi[pos] = po.hi[p];
		so.gcnt[pos] = cgc | (keep ? 0u : GC_SHADOW); so.gfirst[pos] = po.mg[p];
		u64 rh, rlo;
		vdjx_kmer_rc(po.hi[p], po.lo[p], k, rh, rlo);
		so.lo[pos + 1] = rlo; so.hi[pos + 1] = rh;
		so.gcnt[pos + 1] = cgc | (keep_r ? 0u : GC_SHADOW); so.gfirst[pos + 1] = po.mgr[p];
	}
}

// share mode of the sharded build: local first sights (record << ob | offset, all-ones = none) -> the same with the record's scan position
__global__ void k_first_to_scan(u64* __restrict__ a, size_t na, u64* __restrict__ b, size_t nb, const u32* __restrict__ scan, int ob) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= na + nb) return;
	u64* p = i < na ? a + i : b + (i - na);
	const u64 v = *p;
	if (v != NONE64) *p = ((u64) scan[v >> ob] << ob) | (v & ((1ull << ob) - 1ull));
}

// ----------------------------------------------------------------------------------------------
// V/J flags of the nodes
// ----------------------------------------------------------------------------------------------
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


// ==============================================================================================
// Phase B: add_to_graph's bookkeeping (A2:261-320) for the survivors, bucket-local like everything else.
//
// Nine survivors in ten have exactly one surviving successor, and a read follows such a chain for all of its offsets.  The
// survivors are therefore renumbered in CHAIN ORDER first (list ranking over the single-successor links): along a chain the
// successor of survivor p is p + 1, and what the walk over the records needs to know about 16 consecutive survivors (is p + 1
// the successor, with which last base) fits one 64-bit word.  A record then costs one hash lookup where a run of survivors
// starts and one word per 16 survivors it runs through -- not a link load per offset -- and a whole run inside a block of 16
// becomes ONE 8-byte item {first survivor, length, instance id of its first k-mer, first base of the surviving predecessor
// k-mer if the offset before the run survived too}, not one item per instance.
// Items are partitioned by survivor range (LDS-staged, like the tuples) and one workgroup per range counts them in LDS:
// node frequency, first sight (= creation order), and the first sight of every in-edge (v, first base of u) -- the edge
// u -> v under the name of its head.  Blocks of 16 are spread over the ranges by a multiplicative permutation: a deep clone's
// chain would otherwise put all of its instances into one or two workgroups.
//   item = (length - 1) << (38 + pb) | survivor (pb bits) << 38 | has_prev << 37 | pred base << 35 | local record (29 bits) << 6 | offset
//   pb = 26 - lb, lb = min(4, 26 - ceil(log2(survivors))): runs are cut at 2^lb instances when the survivors need the bits
// ==============================================================================================
#define IT_INST_BITS 35
#define IT_INST_MASK ((1ull << IT_INST_BITS) - 1ull)
#define IT_SURV_SHIFT 38
int vdjx_sort_pairs_raw(void* tmp, size_t* tmp_bytes, hipStream_t st, u64* k_in, u64* k_out, u32* v_in, u32* v_out, u32 n, unsigned end_bit);     // vdjx_rindex.hip
#define IT_HOLE 0xFFFFFFFFFFFFFFFFull
#define WALK_THREADS 256

// the survivor table: open addressing over 16-byte slots {key_lo, key_hi | (survivor index + 1) << 36} (k <= 50: the key's high
// part has at most 36 bits), all zero = empty -- a lookup is ONE random line fill; skey[] (keys by index) serves the per-survivor
// kernels.  bloom: one bit per survivor hash, 16 bits of filter per survivor: L2-resident (2 MB at 10 M pairs)
#define SLOT_HI_BITS 36
#define SLOT_HI_MASK ((1ull << SLOT_HI_BITS) - 1ull)
struct SurvTable { const ulonglong2* slots; u32 mask; const ulonglong2* skey; const u32* bloom; u32 bloom_mask; };

// item layout and the block permutation (host: stage_recount)
struct ItemFmt {
	u32 pmask;       // survivor field
	u32 len_shift;   // 38 + pb
	u32 maxlen;      // 2^lb
	u32 mul, inv;    // scattered block = (block * mul) & bmask, block = (scattered * inv) & bmask
	u32 bmask;       // 2^t - 1 >= number of blocks - 1
	u32 ns;          // survivors
};
__device__ inline u32 it_surv(const ItemFmt& f, u64 x) { return (u32) (x >> IT_SURV_SHIFT) & f.pmask; }
__device__ inline u32 it_len(const ItemFmt& f, u64 x) { return f.maxlen > 1 ? (u32) (x >> f.len_shift) + 1u : 1u; }
__device__ inline u32 it_scat(const ItemFmt& f, u32 p) { return ((((p >> 4) * f.mul) & f.bmask) << 4) | (p & 15u); }    // position in the scattered index space

__global__ void k_surv_table2(const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, u32 n, ulonglong2* __restrict__ slots, u32 mask,
                              ulonglong2* __restrict__ skey, u32* __restrict__ bloom, u32 bloom_mask) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 lo = s_lo[i], hi = s_hi[i];
	skey[i] = make_ulonglong2(lo, hi);
	const u64 h = vdjx_mix(lo, hi);
	{
		const u32 bit = (u32) (h >> 40) & bloom_mask;
		atomicOr(&bloom[bit >> 5], 1u << (bit & 31));
	}
	u32 slot = (u32) (h >> 20) & mask;
	const unsigned long long entry = hi | ((unsigned long long) (i + 1) << SLOT_HI_BITS);
	while (atomicCAS((unsigned long long*) &slots[slot].y, 0ull, entry) != 0ull) slot = (slot + 1) & mask;
	slots[slot].x = lo;                                     // (read by later launches only)
}

// lookup of a k-mer that most likely does NOT survive (the start of a run in the walk): the filter answers "no" for 15 in 16 of
// those without leaving the L2 (the table, 32-64 MB at 10 M pairs, is a random line fill over the fabric)
__device__ inline int surv_lookup2f(const SurvTable& t, u64 lo, u64 hi);

__device__ inline int surv_lookup2(const SurvTable& t, u64 lo, u64 hi) {
	u32 slot = (u32) (vdjx_mix(lo, hi) >> 20) & t.mask;
	for (;;) {
		const ulonglong2 v = t.slots[slot];
		if (!v.y) return -1;
		if (v.x == lo && (v.y & SLOT_HI_MASK) == hi) return (int) (v.y >> SLOT_HI_BITS) - 1;
		slot = (slot + 1) & t.mask;
	}
}

__device__ inline int surv_lookup2f(const SurvTable& t, u64 lo, u64 hi) {
	const u64 h = vdjx_mix(lo, hi);
	const u32 bit = (u32) (h >> 40) & t.bloom_mask;
	if (!((t.bloom[bit >> 5] >> (bit & 31)) & 1u)) return -1;
	return surv_lookup2(t, lo, hi);
}

// ---- chain order ------------------------------------------------------------------------------
// succ[u*4+b] = survivor index of (key_u << 2 | b) mod 4^k, or NONE32 (arrival numbering).  The chains are a heavy-path decomposition
// of the graph: u -> v is a chain link iff v is u's HEAVIEST successor (gated count; in a deep clone every k-mer also has surviving
// error branches, the clone's own path is the heavy one) AND u is v's heaviest predecessor.  Ties: the successor with the larger last
// base, the predecessor with the larger COMPLEMENT of its first base -- the rule is then its own mirror image (the successors of a
// k-mer are the reverse complements of its reverse complement's predecessors, last base <-> complement of the first), so over a
// survivor set that is closed under reverse complement the chains come in mirrored pairs: rc(u -> v) = rc(v) -> rc(u) is a link too,
// which is what lets the walk derive a couple's second record from its first (k_walk_items SYM).
// pred[v] = max over the predecessors u of (count << 32 | complement of u's first base << 30 | u); bestsucc[u].
__global__ void k_succ_links2(SurvTable t, u32 n, int k, const u32* __restrict__ gcnt, u32* __restrict__ succ, unsigned long long* __restrict__ pred,
                              u32* __restrict__ bestsucc) {
	const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
	if (u >= n) return;
	const ulonglong2 kk = t.skey[u];
	const u128 base = (((u128) kk.y << 64) | kk.x) << 2;
	const u128 km = k < 64 ? (((u128) 1) << (2 * k)) - 1 : ~(u128) 0;
	const int sh = 2 * (k - 1);
	const u32 fb = (u32) (sh < 64 ? kk.x >> sh : kk.y >> (sh - 64)) & 3u;
	const unsigned long long mine = ((unsigned long long) (gcnt[u] & GC_MASK) << 32) | ((unsigned long long) (fb ^ 1u) << 30) | (unsigned long long) u;
	u32 best = NONE32, best_g = 0;
	for (u32 b = 0; b < 4; b++) {
		const u128 key = (base | (u128) b) & km;
		const int s = surv_lookup2(t, (u64) key, (u64) (key >> 64));
		succ[u * 4 + b] = s >= 0 ? (u32) s : NONE32;
		if (s >= 0 && (u32) s != u) {
			const u32 g = gcnt[s] & GC_MASK;
			if (best == NONE32 || g >= best_g) { best = (u32) s; best_g = g; }
			atomicMax(&pred[s], mine);
		}
	}
	bestsucc[u] = best;
}

// list ranking by pointer jumping, in place: pd[v] = ancestor << 32 | distance to it (ancestor NONE32: v is a head).  Every entry
// is a true statement at all times, so concurrent updates are harmless.  A node is resolved when its ancestor is a head.  With
// CHAIN_JUMPS jumps per node a launch multiplies the resolved distance by at least CHAIN_JUMPS + 1 (all other nodes standing still)
// and normally by 2^CHAIN_JUMPS; launches after the one that found nothing to do return at once.
#define CHAIN_JUMPS 31                  // (measured at 1.05 M survivors, the whole chain order: 3 jumps per launch 0.82 ms, 7 0.59, 15 0.52, 31 / 63 / 255 0.48)
__global__ void k_chain_init(const unsigned long long* __restrict__ pred, const u32* __restrict__ bestsucc, u32 n, u64* __restrict__ pd) {
	const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	const unsigned long long x = pred[v];            // 0: no predecessor
	const u32 u = (u32) x & 0x3FFFFFFFu;             // the heaviest one
	pd[v] = x != 0ull && bestsucc[u] == v ? ((u64) u << 32) | 1ull : ((u64) NONE32 << 32);
}

// SYM: partner[p] = the survivor that is p's reverse complement (chain order).  The set is closed (GC_SHADOW) and the chains are
// mirrored pairs: along a link p -> p + 1 the partners run backwards, partner[p + 1] = partner[p] - 1.  Anything else is counted
// in bad[0] (the build then walks every record: kmer_build_impl2).
__global__ void k_partner(SurvTable t, u32 n, int k, u32* __restrict__ partner) {
	const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n) return;
	const ulonglong2 kk = t.skey[p];
	u64 rh, rlo;
	vdjx_kmer_rc(kk.y, kk.x, k, rh, rlo);
	const int q = surv_lookup2(t, rlo, rh);
	partner[p] = q >= 0 ? (u32) q : NONE32;
}
__global__ void k_partner_check(const u32* __restrict__ partner, const u64* __restrict__ linw, u32 n, u32* __restrict__ bad) {
	const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n) return;
	bool wrong = partner[p] == NONE32;
	const bool linked = (linw[p >> 4] >> (15u - (p & 15u))) & 1ull;
	if (!wrong && linked && (p + 1 >= n || partner[p + 1] != partner[p] - 1u)) wrong = true;
	const u64 m = __ballot(wrong);
	if (m && __lane_id() == 0) atomicAdd(bad, (u32) __popcll(m));
}

__global__ void k_chain_jump(u64* __restrict__ pd, u32 n, const u32* __restrict__ open_prev, u32* __restrict__ open_now) {
	if (open_prev && *open_prev == 0) return;          // (every node resolved by an earlier launch)
	const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
	bool open = false;
	if (v < n) {
		u64 x = __atomic_load_n(&pd[v], __ATOMIC_RELAXED);
		for (int it = 0; it < CHAIN_JUMPS; it++) {
			const u32 p = (u32) (x >> 32);
			if (p == NONE32) break;
			const u64 y = __atomic_load_n(&pd[p], __ATOMIC_RELAXED);
			const u32 pp = (u32) (y >> 32);
			if (pp == NONE32) break;                      // p is the head
			x = ((u64) pp << 32) | (u64) ((u32) x + (u32) y);
			__atomic_store_n(&pd[v], x, __ATOMIC_RELAXED);
			open = true;
		}
	}
	if (__ballot(open) && __lane_id() == 0) atomicAdd(open_now, 1u);
}

// chain lengths at the heads.  Nodes still unresolved after all launches sit on pure cycles (tandem repeats): nothing resolved
// depends on them, each becomes a chain of its own.
__device__ inline void chain_of(const u64* pd, u32 v, u32& head, u32& dist) {
	const u64 x = pd[v];
	const u32 p = (u32) (x >> 32);
	head = v; dist = 0;
	if (p == NONE32) return;
	if ((u32) (pd[p] >> 32) != NONE32) return;             // unresolved
	head = p; dist = (u32) x;
}
__global__ void k_chain_len(const u64* __restrict__ pd, u32 n, u32* __restrict__ len) {
	const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	u32 h, d;
	chain_of(pd, v, h, d);
	atomicAdd(&len[h], 1u);
}

// exclusive scan of a[0..n) in three launches: block sums (SCAN_BLOCK per workgroup), their scan (k_bucket_scan), the blocks again
#define SCAN_BLOCK 4096u
__global__ __launch_bounds__(256) void k_scan_sums(const u32* __restrict__ a, u32 n, u32* __restrict__ sums) {
	__shared__ u32 part[4];
	const u32 b0 = blockIdx.x * SCAN_BLOCK;
	u32 s = 0;
	for (u32 i = threadIdx.x; i < SCAN_BLOCK; i += 256) s += b0 + i < n ? a[b0 + i] : 0u;
	s = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_scan_apply(const u32* __restrict__ a, u32 n, const u32* __restrict__ sum_start, u32* __restrict__ out) {
	__shared__ u32 part[5];
	const u32 b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16;
	u32 loc[16];
	u32 s = 0;
	for (int i = 0; i < 16; i++) { loc[i] = s; s += b0 + i < n ? a[b0 + i] : 0u; }
	const u32 incl = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 base = sum_start[blockIdx.x] + incl - s;
	for (u32 w = 0; w < (threadIdx.x >> 6); w++) base += part[w];
	for (int i = 0; i < 16; i++) if (b0 + i < n) out[b0 + i] = base + loc[i];
}

// the new index of every survivor, and the arrays in the new order
__global__ void k_chain_place(const u64* __restrict__ pd, const u32* __restrict__ off, u32 n, u32* __restrict__ newidx) {
	const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	u32 h, d;
	chain_of(pd, v, h, d);
	newidx[v] = off[h] + d;
}
__global__ void k_chain_permute(const u32* __restrict__ newidx, u32 n, const u64* __restrict__ lo, const u64* __restrict__ hi, const u32* __restrict__ gcnt,
                                const u64* __restrict__ gfirst, const u32* __restrict__ succ, u64* __restrict__ lo2, u64* __restrict__ hi2,
                                u32* __restrict__ gcnt2, u64* __restrict__ gfirst2, ulonglong2* __restrict__ skey2, u32* __restrict__ succ2) {
	const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	const u32 q = newidx[v];
	const u64 l = lo[v], h = hi[v];
	lo2[q] = l; hi2[q] = h; gcnt2[q] = gcnt[v]; gfirst2[q] = gfirst[v];
	skey2[q] = make_ulonglong2(l, h);
	const uint4 s = *(const uint4*) &succ[(size_t) v * 4];
	uint4 r;
	r.x = s.x == NONE32 ? NONE32 : newidx[s.x];
	r.y = s.y == NONE32 ? NONE32 : newidx[s.y];
	r.z = s.z == NONE32 ? NONE32 : newidx[s.z];
	r.w = s.w == NONE32 ? NONE32 : newidx[s.w];
	*(uint4*) &succ2[(size_t) q * 4] = r;
}
__global__ void k_table_remap(ulonglong2* __restrict__ slots, u32 n_slots, const u32* __restrict__ newidx) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_slots) return;
	const u64 y = slots[i].y;
	if (!y) return;
	slots[i].y = (y & SLOT_HI_MASK) | ((u64) (newidx[(u32) (y >> SLOT_HI_BITS) - 1] + 1u) << SLOT_HI_BITS);
}

// per block of 16 survivors (new numbering), survivor j of the block: linw bit 15-j: p + 1 is a successor; bit 31-j: there are
// other successors (look into succ[]); bits 32 + 2*(15-j): the last base of p + 1 -- first survivor most significant, like the
// bases of a packed read -- and 2 bits per survivor in fbw, the first base of its k-mer (the name of the in-edge p -> p + 1 at its head)
// partner (SYM walk, else null): p + 1 counts as "the successor along the chain" only if the step is a mirror image too, partner[p + 1]
// == partner[p] - 1 -- true of every chain link (k_succ_links2), but p + 1 may also be the head of the NEXT chain that happens to be a
// successor of p: the walk may run on across such a seam, its mirror image may not.
__global__ void k_chain_words(const u32* __restrict__ succ, const ulonglong2* __restrict__ skey, u32 n, int k, u64* __restrict__ linw, u32* __restrict__ fbw,
                              const u32* __restrict__ partner) {
	const u32 p = blockIdx.x * blockDim.x + threadIdx.x;      // (whole blocks of 16 lanes: nobody returns before the shuffles)
	const u32 j = p & 15u;
	u64 w = 0;
	u32 fb = 0;
	if (p < n) {
		const uint4 s = *(const uint4*) &succ[(size_t) p * 4];
		const u32 sv[4] = {s.x, s.y, s.z, s.w};
		const bool mirrored = !partner || (p + 1 < n && partner[p] != NONE32 && partner[p + 1] == partner[p] - 1u);
		for (u32 b = 0; b < 4; b++) if (sv[b] != NONE32) {
			if (sv[b] == p + 1 && mirrored) w |= (1ull << (15u - j)) | ((u64) b << (32u + 2u * (15u - j)));
			else w |= 1ull << (31u - j);
		}
		const ulonglong2 kk = skey[p];
		const int sh = 2 * (k - 1);
		fb = (u32) (sh < 64 ? kk.x >> sh : kk.y >> (sh - 64)) & 3u;
	}
	u32 fw = fb << (2 * j);
	for (int d = 1; d < 16; d <<= 1) {
		w |= ((u64) (u32) __shfl_xor((int) (w >> 32), d) << 32) | (u32) __shfl_xor((int) (u32) w, d);
		fw |= (u32) __shfl_xor((int) fw, d);
	}
	if (j == 0 && p < n) { linw[p >> 4] = w; fbw[p >> 4] = fw; }
}

// ---- the walk ---------------------------------------------------------------------------------
// One lane per record, in rounds.  A round of a lane starts at an offset whose k-mer is a known survivor s -- from a table lookup
// (the first valid offset of the record; after a gap, the next offset the k-mer filter lets through: the filter bits of all
// remaining offsets are fetched at once, as independent loads) or from the successor list of the node the previous round ended on
// -- and follows the chain s, s + 1, ... as far as the read agrees with it: the chain's 16 next bases and link bits (two words of
// linw) against the read's 16 next bases and valid offsets, one XOR and three count-leading-zeros, no loop over the offsets.  The
// run goes out as one item per block of 16 it touches.  Reads of a clone take one round; a sequencing error, a branch or an N adds
// one.  All lanes of a wave do their lookups of a round together, so the wave waits for memory a few times per record, not at
// every offset.
// Every wave appends its items to blocks of `blk_items` slots it reserves from the global cursor (unused slots of a block are
// filled with IT_HOLE); item counts per survivor range go to range_cnt.
// the two record formats behind one face (vdjx_pool): what the walk asks of a record, and the masks over its offsets
struct WalkRecS {                       // reads of up to 64 bases: two registers
	u64 bhi, blo; int rl;
	__device__ inline void kmer_u(int k, int o, u64& hi, u64& lo) const { vdjx_kmer_at(bhi, blo, rl, k, o, hi, lo); }          // o uniform
	__device__ inline void kmer_l(int k, int o, u64& hi, u64& lo) const { vdjx_kmer_at_lane(bhi, blo, rl, k, o, hi, lo); }     // o per lane
	__device__ inline u32 bases16(int bi) const {                // bases bi .. bi+15, first most significant, zeros past the read
		const int sh = 2 * (rl - 16 - bi);
		const u32 rw_dn = vdjx_bits_at_lane(bhi, blo, (u32) (sh > 0 ? sh : 0));
		const u32 up = (u32) (sh < 0 ? -sh : 0);
		return (rw_dn << (up & 31u)) & (0u - (u32) (up < 32u));
	}
	__device__ inline u32 base(int i) const { return vdjx_base_at_lane(bhi, blo, rl, i); }
};
struct WalkRecL {                       // longer reads: a row of words in LDS
	const u64* row;
	__device__ inline void kmer_u(int k, int o, u64& hi, u64& lo) const { vdjx_kmer_at_words(row, k, o, hi, lo); }
	__device__ inline void kmer_l(int k, int o, u64& hi, u64& lo) const { vdjx_kmer_at_words(row, k, o, hi, lo); }
	__device__ inline u32 bases16(int bi) const { return vdjx_bases16_words(row, bi); }
	__device__ inline u32 base(int i) const { return vdjx_base_words(row, i); }
};
__device__ inline bool wm_test(u64 m, int i) { return (m >> i) & 1ull; }
__device__ inline bool wm_test(const vdjx_mask3& m, int i) { return m.test(i); }
__device__ inline void wm_set(u64& m, int i) { m |= 1ull << i; }
__device__ inline void wm_set(vdjx_mask3& m, int i) { m.set(i); }
__device__ inline int wm_next(u64 m, int from, int none) { const u64 x = from < 64 ? m & (~0ull << from) : 0ull; return x ? __builtin_ctzll(x) : none; }
__device__ inline int wm_next(const vdjx_mask3& m, int from, int none) { return m.next(from, none); }
__device__ inline u32 wm_slice32(u64 m, int from) { return from < 64 ? (u32) (m >> from) : 0u; }
__device__ inline u32 wm_slice32(const vdjx_mask3& m, int from) { return (u32) m.slice(from); }

// SYM (vdjx_pool::sym, a survivor set closed under reverse complement, mirrored chains: k_succ_links2): only the FIRST record of every
// couple is walked (R counts couples); a run of survivors pp .. pp + L - 1 at the offsets po .. of record 2q is, read backwards, the run
// partner[pp] - (L - 1) .. partner[pp] at the offsets rl - k - (po + L - 1) .. of record 2q + 1, whose predecessor there is the mirror of
// the node that FOLLOWS the run here.  Both runs leave as items; the table lookups -- what this kernel waits for -- are made once.
template <bool LONG, bool SYM = false>
__global__ __launch_bounds__(WALK_THREADS) void k_walk_items(const u64* __restrict__ bases, const u64* __restrict__ nmask, size_t R, int rl, int ob_bits, int k,
                                                             SurvTable t, const u32* __restrict__ succ, const u64* __restrict__ linw, ItemFmt f,
                                                             u32 range_shift, u32 n_ranges,
                                                             u64* __restrict__ raw, u64 raw_cap, u32 blk_items,
                                                             unsigned long long* __restrict__ g_cursor,
                                                             u32* __restrict__ range_cnt, u32* __restrict__ g_err, u32 dbg_in,
                                                             const u32* __restrict__ partner) {
	static_assert(!(LONG && SYM), "SYM is the short-read form");
	// the ablation switches of profiles/walkdbg.sh (filter off, lookups faked, outputs dropped, per-wave statistics) exist only in the
	// -DVDJX_ABLATE build: the shipped walk has none of those branches
#ifdef VDJX_ABLATE
	const u32 dbg = dbg_in;
#else
	(void) dbg_in;
	constexpr u32 dbg = 0u;
#endif
	extern __shared__ u32 hist[];                 // [n_ranges] (+ LONG: one row of GL_ROW_LONG words per thread)
	typedef typename std::conditional<LONG, vdjx_mask3, u64>::type MaskT;
	typedef typename std::conditional<LONG, WalkRecL, WalkRecS>::type RecT;
	for (u32 i = threadIdx.x; i < n_ranges; i += WALK_THREADS) hist[i] = 0;
	__syncthreads();
	u64* myrow = (u64*) (hist + ((n_ranges + 1u) & ~1u)) + threadIdx.x * GL_ROW_LONG;
	const int P = rl - k + 1;
	const size_t per = ((R + gridDim.x - 1) / gridDim.x + WALK_THREADS - 1) / WALK_THREADS * WALK_THREADS;
	const size_t r0 = (size_t) blockIdx.x * per;
	const size_t r1 = r0 + per < R ? r0 + per : R;
	const int lane = __lane_id();
	u64 blk = 0;                                  // this wave's block (wave-uniform), valid once have_blk
	u32 fill = blk_items;
#ifdef WALK_STATS                       // (build with -DWALK_STATS, run with VDJX_WALK_DBG=32: what a wave's trips are spent on; the counters cost scalar registers)
	u32 st_rows = 0, st_rounds = 0, st_filter = 0, st_lookup = 0, st_items = 0, st_chain = 0;
#define WST(x) x
#else
#define WST(x)
#endif
	bool have_blk = false, dead = false;
	// Records that leave their chain (an error, a branch, noise: four in ten) need the k-mer filter: 16 k-mers cut out and hashed, half
	// of this kernel's instructions -- issued for the whole wave when ONE lane asks.  A lane of a fresh row therefore only goes as
	// far as its first lookup and the chains carry it; where it would ask the filter, the record is set aside (record, offset) in
	// the wave's queue, and the queued records are walked 64 at a time, every lane with the same need.  (VDJX_WALK_DBG=32 with
	// -DWALK_STATS counted, per row of 64 records: 4.6 rounds, the filter in 1.3 of them; profiles/README.md, round 3.)
	__shared__ u64 dq[WALK_THREADS / 64][128];
	__shared__ ulonglong2 dq_b[LONG ? 1 : WALK_THREADS / 64][LONG ? 1 : 128];      // (short reads: the record travels with its queue entry --
	__shared__ u64 dq_v[LONG ? 1 : WALK_THREADS / 64][LONG ? 1 : 128];             //  read again from the pool it is two random line fills per lane)
	const u32 wq = threadIdx.x >> 6;
	u32 qn = 0;                                   // (wave-uniform)
	auto walk_row = [&](auto fresh_tag, const size_t r, const bool live, const int o_from, const ulonglong2 qb, const u64 qv) {
		constexpr bool FRESH = decltype(fresh_tag)::value;
		RecT v;
		MaskT V{};                                    // the offsets whose k bases are all valid (bit o of nm = base o is N or masked)
		if constexpr (LONG) {
			v.row = myrow;
			if (live) {
				stage_row_long(myrow, bases, r);
				const vdjx_mask3 nm = load_bad3(nmask, nullptr, r);
				for (int o16 = 0; o16 < P; o16 += 16) {
					const u64 g = (u64) vdjx_clean16(nm, o16, k, P) << (o16 & 63);
					if ((o16 >> 6) == 0) V.w0 |= g; else if ((o16 >> 6) == 1) V.w1 |= g; else V.w2 |= g;
				}
			} else
				for (int w = 0; w < GL_ROW_LONG; w++) myrow[w] = 0;
		} else {
			v.bhi = v.blo = 0; v.rl = rl;
			if (!FRESH) { if (live) { v.bhi = qb.x; v.blo = qb.y; V = qv; } }
			else if (live) {
				const RecView rv = load_rec(bases, nmask, nullptr, r);
				v.bhi = rv.bhi; v.blo = rv.blo;
				u64 inv = rv.nm;
				int cur = 1;
				while (cur * 2 <= k) { inv |= inv >> cur; cur *= 2; }
				inv |= inv >> (k - cur);
				V = ~inv & (P >= 64 ? ~0ull : (1ull << P) - 1ull);
			}
		}
		int o = wm_next(V, o_from, P);                          // the offset this lane works on; P: done
		int s = -1;                                              // its survivor, if known
		u32 in = 0;                                              // has_prev << 2 | first base of the predecessor k-mer
		bool first = FRESH, have_c = false;
		MaskT C{};                                               // valid offsets the k-mer filter lets through (once have_c)
		WST(st_rows++;)
		while (__ballot(o < P)) {
			WST(st_rounds++;)
			// ---- lookups ----
			bool need = o < P && s < 0;
			if (FRESH) {
				const bool later = need && !first;
				const u64 lm = __ballot(later);
				if (lm) {
					if (later) {
						const u32 qi = qn + (u32) __popcll(lm & ((1ull << lane) - 1ull));
						dq[wq][qi] = ((u64) r << 8) | (u64) (u32) o;
						if constexpr (!LONG) { dq_b[wq][qi] = make_ulonglong2(v.bhi, v.blo); dq_v[wq][qi] = V; }
						o = P; need = false;
					}
					qn += (u32) __popcll(lm);
				}
			}
			WST(if (__ballot(need)) st_lookup++;)
			if (!FRESH && __ballot(need && !first && !have_c)) {
				WST(st_filter++;)      // (wave-uniform) filter bits of all remaining offsets, 4 loads in flight
				const bool mine = need && !first && !have_c;
				for (int ob = 0; ob < P; ob += 4) {
					u32 wv[4], bt[4];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const int oo = ob + j;
						wv[j] = 0; bt[j] = 0;
						if (mine && oo >= o && oo < P && wm_test(V, oo)) {
							u64 khi, klo;
							v.kmer_u(k, oo, khi, klo);
							const u32 bit = (u32) (vdjx_mix(klo, khi) >> 40) & t.bloom_mask;
							bt[j] = bit & 31u;
							wv[j] = (dbg & 8u) ? 0xFFFFFFFFu : t.bloom[bit >> 5];
						}
					}
#pragma unroll
					for (int j = 0; j < 4; j++) if ((wv[j] >> bt[j]) & 1u) wm_set(C, ob + j);
				}
				if (mine) have_c = true;
			}
			if (need) {
				bool filtered = false;
				if (!first) {
					o = wm_next(C, o, P);
					filtered = true;
				}
				first = false;
				if (o < P) {
					u64 khi, klo;
					v.kmer_l(k, o, khi, klo);
					if (dbg & 8u) s = (klo & 3u) ? (int) ((u32) vdjx_mix(klo, khi) % f.ns) : -1;
					else s = (filtered || (dbg & 16u)) ? surv_lookup2(t, klo, khi) : surv_lookup2f(t, klo, khi);
					in = 0;
					if (s < 0) o++;
				}
			}
			// ---- the chain from s ----
			const bool act = o < P && s >= 0;
			WST(if (__ballot(act)) st_chain++;)
			u32 rem = 0, pp = 0, po = 0;
			if (act) {
				const u32 a = (u32) s & 15u;
				const u64 w0 = linw[(u32) s >> 4], w1 = linw[((u32) s >> 4) + 1];
				// windows over the 16 nodes s .. s+15, node s+j at bit 15-j (links, others) / bits 2*(15-j) (bases)
				const u32 lw = (((((u32) w0 & 0xFFFFu) << 16) | ((u32) w1 & 0xFFFFu)) << a) >> 16;
				const u32 ow = ((((u32) w0 & 0xFFFF0000u) | ((u32) w1 >> 16)) << a) >> 16;
				const u32 bw = (u32) (((((w0 >> 32) << 32) | (w1 >> 32)) << (2 * a)) >> 32);
				const u32 rw = v.bases16(o + k);                   // the read's bases o+k .. o+k+15 in the same layout
				const u32 x = bw ^ rw;
				const u32 nl = ~lw & 0xFFFFu;
				const u32 m_base = x ? (u32) __builtin_clz(x) >> 1 : 16u;
				const u32 m_lin = nl ? (u32) __builtin_clz(nl) - 16u : 16u;
				const u32 nv = ~wm_slice32(V, o + 1);                // (17 offsets matter; bits at and above P are invalid)
				const u32 m_valid = nv ? (u32) __builtin_ctz(nv) : 32u;
				u32 steps = m_base < m_lin ? m_base : m_lin;
				steps = steps < m_valid ? steps : m_valid;
				const bool capped = steps >= 16u;                     // (reads with more than 17 offsets) the chain goes on: next round from s + 16
				if (capped) steps = 15u;
				rem = steps + 1u; pp = (u32) s; po = (u32) o;
				// where the next round starts
				const int o2 = o + (int) rem;
				int s2 = -1;
				if (o2 < P && wm_test(V, o2)) {
					if (capped) s2 = s + 16;
					else if ((ow >> (15u - steps)) & 1u) {                     // the last node has successors off the chain
						const u32 bb = v.base(o2 + k - 1);
						const u32 nx = succ[((u32) s + steps) * 4u + bb];
						s2 = nx == NONE32 ? -1 : (int) nx;
					}
				}
				o = s2 >= 0 || o2 >= P ? o2 : o2 + 1;                  // nothing survives at o2: the search goes on behind it
				s = s2;
			}
			// ---- the run, one item per block of 16 (and per 2^lb nodes) ----
			const u32 pin0 = in;
			if (act) in = s >= 0 ? (4u | v.base(o - 1)) : 0u;      // (next round: first base of the k-mer at o - 1)
			// SYM: the partner of the run's first node, asked for before the run's own items are written
			const u32 run_len = rem, run_pp = pp, run_po = po;
			u32 q_last = 0;
			if (SYM && act) q_last = partner[run_pp];
			auto emit = [&](auto mir_tag, u32 rem_, u32 pp_, u32 po_, u32 pin_, const size_t rec_) {
				constexpr bool MIR = decltype(mir_tag)::value;
				while (__ballot(rem_ > 0)) {
					WST(st_items++;)
					const bool close = rem_ > 0;
					u32 len = 16u - (pp_ & 15u);
					len = len < rem_ ? len : rem_;
					len = len < f.maxlen ? len : f.maxlen;
					const u64 item = (f.maxlen > 1 ? (u64) (len - 1) << f.len_shift : 0ull) | ((u64) pp_ << IT_SURV_SHIFT) | ((u64) pin_ << 35) | ((u64) rec_ << ob_bits) | (u64) po_;
					const u64 m = __ballot(close);
					const u32 cnt = (u32) __popcll(m);
					if (fill + cnt > blk_items) {                             // (wave-uniform)
						if (have_blk) for (u32 i = fill + (u32) lane; i < blk_items; i += 64) raw[blk + i] = IT_HOLE;
						unsigned long long nb = 0;
						if (lane == 0) nb = atomicAdd(g_cursor, (unsigned long long) blk_items);
						nb = ((unsigned long long) (u32) __builtin_amdgcn_readlane((int) (nb >> 32), 0) << 32) | (u32) __builtin_amdgcn_readlane((int) nb, 0);
						if (nb + blk_items > raw_cap) { if (lane == 0 && !dead) atomicAdd(g_err, 1u); dead = true; have_blk = false; }
						else { blk = nb; have_blk = true; }
						fill = 0;
					}
					if (close && !dead) {
						if (!(dbg & 1u)) raw[blk + fill + (u32) __popcll(m & ((1ull << lane) - 1ull))] = item;
						if (!(dbg & 2u)) atomicAdd(&hist[it_scat(f, pp_) >> range_shift], 1u);
					}
					fill += cnt;
					if (close) {
						rem_ -= len; pp_ += len; po_ += len;
						// the next piece's predecessor is the node before it: here the base at po - 1; in the mirrored record the base at
						// po' - 1 there, i.e. the complement of this record's base rl - po'
						pin_ = 4u | (MIR ? (v.base(rl - (int) po_) ^ 1u) : v.base((int) po_ - 1));
					}
				}
			};
			emit(std::false_type{}, rem, pp, po, pin0, r);
			if constexpr (SYM) {
				// the same run in the couple's second record (see above): it ends where this one starts; its predecessor is the mirror of
				// this run's successor (known now: s), named by the complement of that k-mer's last base
				u32 mrem = 0, mpp = 0, mpo = 0, mpin = 0;
				if (act) {
					if (q_last == NONE32 || q_last + 1u < run_len) { if (!dead) atomicAdd(g_err + 2, 1u); }      // (cannot happen: the set is closed, the chains mirrored)
					else {
						mrem = run_len; mpp = q_last - (run_len - 1u);
						mpo = (u32) (rl - k) - (run_po + run_len - 1u);
						mpin = s >= 0 ? (4u | (v.base(rl - (int) mpo) ^ 1u)) : 0u;
					}
				}
				emit(std::true_type{}, mrem, mpp, mpo, mpin, r + 1);
			}
		}
	};
	auto walk_queued = [&](u32 take) {                // the last `take` <= 64 queued records, one per lane
		vdjx_wave_lds_fence();
		const bool live = (u32) lane < take;
		const u64 e = live ? dq[wq][qn - take + (u32) lane] : 0ull;
		ulonglong2 qb = make_ulonglong2(0ull, 0ull);
		u64 qv = 0;
		if constexpr (!LONG) if (live) { qb = dq_b[wq][qn - take + (u32) lane]; qv = dq_v[wq][qn - take + (u32) lane]; }
		vdjx_wave_lds_fence();
		qn -= take;
		walk_row(std::false_type{}, (size_t) (e >> 8), live, (int) (e & 255u), qb, qv);
	};
	for (size_t rb = r0; rb < r1; rb += WALK_THREADS) {
		const size_t r = rb + threadIdx.x;                   // (SYM: a couple; its first record is 2r)
		walk_row(std::true_type{}, SYM ? 2 * r : r, r < r1, 0, make_ulonglong2(0ull, 0ull), 0ull);
		while (qn >= 64) walk_queued(64);
	}
	if (qn) walk_queued(qn);
	if (have_blk) for (u32 i = fill + (u32) lane; i < blk_items; i += 64) raw[blk + i] = IT_HOLE;
#ifdef WALK_STATS
	if ((dbg & 32u) && lane == 0) {
		atomicAdd(&g_cursor[1], (unsigned long long) st_rows); atomicAdd(&g_cursor[2], (unsigned long long) st_rounds); atomicAdd(&g_cursor[3], (unsigned long long) st_filter);
		atomicAdd(&g_cursor[4], (unsigned long long) st_lookup); atomicAdd(&g_cursor[5], (unsigned long long) st_chain); atomicAdd(&g_cursor[6], (unsigned long long) st_items);
	}
#endif
	__syncthreads();
	for (u32 i = threadIdx.x; i < n_ranges; i += WALK_THREADS) if (hist[i]) atomicAdd(&range_cnt[i], hist[i]);
}

// items -> items grouped by survivor range (see k_part_tuples); holes are dropped.  `n_raw` is read from the device cursor.
__global__ __launch_bounds__(PART_THREADS) void k_part_items(const u64* __restrict__ in, const unsigned long long* __restrict__ n_raw, ItemFmt f,
                                                             u32 range_shift, u32 nbk, u32* __restrict__ gcur, u64* __restrict__ out) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	u64* stage = (u64*) smem;
	__shared__ u32 cnt[PART_MAXB], base[PART_MAXB + 1], cur[PART_MAXB], gbase[PART_MAXB], tmp[PART_THREADS];
	constexpr u32 PER = PART_LDS_BYTES / 8 / PART_THREADS;          // 16 items per thread per round
	constexpr u32 ROUND = PER * PART_THREADS;
	const size_t N = (size_t) *n_raw;
	const size_t per = ((N + gridDim.x - 1) / gridDim.x + ROUND - 1) / ROUND * ROUND;
	const size_t t0 = (size_t) blockIdx.x * per;
	const size_t t1 = t0 + per < N ? t0 + per : N;
	for (size_t ts = t0; ts < t1; ts += ROUND) {
		const size_t te = ts + ROUND < t1 ? ts + ROUND : t1;
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) cnt[i] = 0;
		__syncthreads();
		u64 r_it[PER];
		u32 r_b[PER];
#pragma unroll
		for (u32 j = 0; j < PER; j += 2) {                            // 16-byte loads
			const size_t t = ts + ((size_t) (j / 2) * PART_THREADS + threadIdx.x) * 2;
			r_it[j] = IT_HOLE; r_it[j + 1] = IT_HOLE;
			if (t + 1 < te) { const ulonglong2 v = *(const ulonglong2*) &in[t]; r_it[j] = v.x; r_it[j + 1] = v.y; }
			else if (t < te) r_it[j] = in[t];
		}
#pragma unroll
		for (u32 j = 0; j < PER; j++) {
			r_b[j] = NONE32;
			if (r_it[j] != IT_HOLE) { r_b[j] = it_scat(f, it_surv(f, r_it[j])) >> range_shift; atomicAdd(&cnt[r_b[j]], 1u); }
		}
		__syncthreads();
		part_scan(cnt, base, tmp, nbk);
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) {
			cur[i] = base[i];
			gbase[i] = cnt[i] ? atomicAdd(&gcur[i], cnt[i]) : 0u;
		}
		__syncthreads();
#pragma unroll
		for (u32 j = 0; j < PER; j++) if (r_b[j] != NONE32) stage[atomicAdd(&cur[r_b[j]], 1u)] = r_it[j];
		__syncthreads();
		const u32 n = base[nbk];
		for (u32 i = threadIdx.x; i < n; i += PART_THREADS) {
			const u64 x = stage[i];
			const u32 b = it_scat(f, it_surv(f, x)) >> range_shift;
			out[gbase[b] + (i - base[b])] = x;
		}
		__syncthreads();
	}
}

// second level (more than 1024 survivor ranges): segment `seg` of the level-1 output is split over `slices` workgroups and cut
// into 2^sub_bits ranges
__global__ __launch_bounds__(PART_THREADS) void k_part_items2(const u64* __restrict__ in, const u32* __restrict__ seg_start, u32 seg_shift, u32 slices, ItemFmt f,
                                                              u32 range_shift, u32 sub_bits, u32* __restrict__ gcur, u64* __restrict__ out) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	u64* stage = (u64*) smem;
	__shared__ u32 cnt[PART_MAXB], base[PART_MAXB + 1], cur[PART_MAXB], gbase[PART_MAXB], tmp[PART_THREADS];
	constexpr u32 PER = PART_LDS_BYTES / 8 / PART_THREADS;
	constexpr u32 ROUND = PER * PART_THREADS;
	const u32 seg = blockIdx.x / slices, sl = blockIdx.x % slices;
	const u32 nbk = 1u << sub_bits, mask = nbk - 1;
	const size_t s0 = seg_start[(size_t) seg << seg_shift], s1 = seg_start[((size_t) seg + 1) << seg_shift];
	const size_t per = (s1 - s0 + slices - 1) / slices;
	const size_t t0 = s0 + (size_t) sl * per;
	const size_t t1 = t0 + per < s1 ? t0 + per : s1;
	u32* gc = gcur + ((size_t) seg << sub_bits);
	for (size_t ts = t0; ts < t1; ts += ROUND) {
		const size_t te = ts + ROUND < t1 ? ts + ROUND : t1;
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) cnt[i] = 0;
		__syncthreads();
		u64 r_it[PER];
		u32 r_b[PER];
#pragma unroll
		for (u32 j = 0; j < PER; j++) {
			const size_t t = ts + (size_t) j * PART_THREADS + threadIdx.x;
			r_b[j] = NONE32;
			if (t < te) { r_it[j] = in[t]; r_b[j] = (it_scat(f, it_surv(f, r_it[j])) >> range_shift) & mask; }
		}
#pragma unroll
		for (u32 j = 0; j < PER; j++) if (r_b[j] != NONE32) atomicAdd(&cnt[r_b[j]], 1u);
		__syncthreads();
		part_scan(cnt, base, tmp, nbk);
		for (u32 i = threadIdx.x; i < nbk; i += PART_THREADS) {
			cur[i] = base[i];
			gbase[i] = cnt[i] ? atomicAdd(&gc[i], cnt[i]) : 0u;
		}
		__syncthreads();
#pragma unroll
		for (u32 j = 0; j < PER; j++) if (r_b[j] != NONE32) stage[atomicAdd(&cur[r_b[j]], 1u)] = r_it[j];
		__syncthreads();
		const u32 n = base[nbk];
		for (u32 i = threadIdx.x; i < n; i += PART_THREADS) {
			const u64 x = stage[i];
			const u32 b = (it_scat(f, it_surv(f, x)) >> range_shift) & mask;
			out[gbase[b] + (i - base[b])] = x;
		}
		__syncthreads();
	}
}

// ---- the recount ------------------------------------------------------------------------------
// One workgroup per range of SB scattered positions (SB/16 blocks of survivors).  A run [a, e] inside a block (positions in the
// block) with first instance I gives element x the instance I + (x - a).  No run is walked element by element:
//   counts    every run adds 1 to the block's counter e and takes 1 from counter a - 1 (a > 0): count(x) = sum of the counters >= x
//   minima    of v = I + 15 - a (first(x) = v - 15 + x), by the run's shape: a == 0 -> prefix slot e, e == 15 -> suffix slot a, and an
//             interior run (a read with an error, a chain end) -> TWO window slots of width w = 2^floor(log2 length), the windows
//             [a, a + w) and (e - w, e]: they cover the run, lie inside it and overlap, so every neighbouring pair (x - 1, x) of the
//             run is inside one of them -- which is what the in-edge (x - 1 -> x) needs
// and the survivors' values are put together at the end:
//   first(x) = min over the slots whose run holds x, - 15 + x;   in-edge (x, first base of x - 1) = the same over the slots holding
//   x - 1 and x (and the runs' own in-edges at their heads, `ef`)
// (Round 3: the interior runs used to be walked -- one lane's 14 trips for the whole wave: 0.77 -> see profiles/README.md.)
// Lanes of a wave that hit the same slot (deep clones) are combined before they touch LDS.
#define RC_THREADS 1024
#define RC_UNR 4
template <typename IT> struct RcNone;
template <> struct RcNone<u32> { static constexpr u32 v = 0xFFFFFFFFu; };
template <> struct RcNone<u64> { static constexpr u64 v = 0xFFFFFFFFFFFFFFFFull; };
__device__ inline void rc_min(u32* p, u32 v) { atomicMin(p, v); }
__device__ inline void rc_min(u64* p, u64 v) { atomicMin((unsigned long long*) p, (unsigned long long) v); }

// slots of block bl: counters [bl*17 + e]; minima [bl*77 + ...]: prefix by e (0..15), [+15 + a] suffix by a (1..15), [+31 + window]
// (16 counters and 76 minima per block; the odd strides keep the blocks' counter 15 -- where every suffix run counts -- on different banks)
#define RC_CS 17u
#define RC_MS 77u
// window of width 2^k starting at s (1 <= s, s + 2^k - 1 <= 14): 14 + 13 + 11 + 7 = 45 slots
__device__ inline u32 rc_win(u32 k, u32 s) { return 31u + ((0x261B0E00u >> (8u * k)) & 0xFFu) + s - 1u; }
// IT = u32 while the local instance ids fit 32 bits (fewer than 2^26 records on this GPU)
template <u32 SB, typename IT>
__global__ __launch_bounds__(RC_THREADS) void k_recount(const u64* __restrict__ items, const u32* __restrict__ range_start, ItemFmt f, u32 ns, u64 rec_base, int ob,
                                                        const u32* __restrict__ fbw, u32* __restrict__ ucnt, u64* __restrict__ ufirst, u64* __restrict__ in_first,
                                                        unsigned long long* __restrict__ n_inst) {
	__shared__ u32 c[SB / 16 * RC_CS];
	__shared__ IT mv[SB / 16 * RC_MS];
	__shared__ IT ef[SB * 4];
	__shared__ u32 fbl[SB / 16];
	__shared__ unsigned long long tot;
	constexpr IT NONE = RcNone<IT>::v;
	const u32 b = blockIdx.x;
	const u32 s0 = b * SB;
	const u32 nb = (ns + 15u) >> 4;
	for (u32 i = threadIdx.x; i < SB / 16 * RC_CS; i += RC_THREADS) c[i] = 0;
	for (u32 i = threadIdx.x; i < SB / 16 * RC_MS; i += RC_THREADS) mv[i] = NONE;
	for (u32 i = threadIdx.x; i < SB * 4; i += RC_THREADS) ef[i] = NONE;
	for (u32 i = threadIdx.x; i < SB / 16; i += RC_THREADS) {
		const u32 sb = (s0 >> 4) + i;
		const u32 blk = (sb * f.inv) & f.bmask;
		fbl[i] = sb <= f.bmask && blk < nb ? fbw[blk] : 0u;
	}
	if (threadIdx.x == 0) tot = 0;
	__syncthreads();
	const u32 i0 = range_start[b], i1 = range_start[b + 1];
	// whole waves stay together (ballots, DPP); RC_UNR items per lane are loaded before the first is used (the loop is bound by the
	// latency of the item loads, not by their bytes)
	const u32 iend = i0 + (i1 - i0 + RC_THREADS * RC_UNR - 1) / (RC_THREADS * RC_UNR) * (RC_THREADS * RC_UNR);
	for (u32 ib = i0 + threadIdx.x; ib < iend; ib += RC_THREADS * RC_UNR) {
		u64 xs[RC_UNR];
#pragma unroll
		for (int u = 0; u < RC_UNR; u++) { const u32 i = ib + (u32) u * RC_THREADS; xs[u] = i < i1 ? items[i] : IT_HOLE; }
#pragma unroll
		for (int u = 0; u < RC_UNR; u++) {
			const u64 x = xs[u];
			const bool live = x != IT_HOLE;
			const u32 p = it_surv(f, x);
			const u32 len = live ? it_len(f, x) : 1u;
			const u32 sl = live ? it_scat(f, p) - s0 : 0u;             // local position of the run's first survivor
			const u32 a = sl & 15u, e = a + len - 1u;
			const u64 inst = x & IT_INST_MASK;
			const bool shaped = live && (a == 0u || e == 15u);
			const u32 cb = (sl >> 4) * RC_CS, mb = (sl >> 4) * RC_MS;
			const u32 slot = mb + (a == 0u ? e : 15u + a);
			const IT val = (IT) (inst + 15u - a);
			// the lanes that share the first shaped lane's slot are folded into one add and one min when they are many (deep clones)
			bool mine = shaped;
			const u64 act = __ballot(shaped);
			if (act) {
				const int leader = __ffsll((long long) act) - 1;
				const u32 lslot = (u32) __builtin_amdgcn_readlane((int) slot, leader);
				const bool same = shaped && slot == lslot;
				const u64 m = __ballot(same);
				if (__popcll(m) >= 8) {
					// minimum of the (up to 35-bit) values in two steps: low words among the lanes holding the minimal high word
					const u64 v64 = inst + 15u - a;
					const u32 hi_min = vdjx_wave_min(same ? (u32) (v64 >> 32) : 0xFFFFFFFFu);
					const u32 lo_min = vdjx_wave_min(same && (u32) (v64 >> 32) == hi_min ? (u32) v64 : 0xFFFFFFFFu);
					if (__lane_id() == leader) {                          // (the lanes of one slot share a and e)
						const u32 n = (u32) __popcll(m);
						atomicAdd(&c[cb + e], n);
						if (a) atomicAdd(&c[cb + a - 1u], 0u - n);
						rc_min(&mv[lslot], (IT) (((u64) hi_min << 32) | lo_min));
					}
					mine = shaped && !same;
				}
			}
			if (live && (mine || !shaped)) {
				atomicAdd(&c[cb + e], 1u);
				if (a) atomicAdd(&c[cb + a - 1u], 0xFFFFFFFFu);
			}
			if (mine) rc_min(&mv[slot], val);
			if (live && !shaped) {                                       // an interior run: two windows of width 2^k <= its length
				const u32 k = 31u - (u32) __builtin_clz(len), w = 1u << k;
				rc_min(&mv[mb + rc_win(k, a)], val);
				if (len != w) rc_min(&mv[mb + rc_win(k, e + 1u - w)], val);
			}
			if (live && ((x >> 37) & 1ull)) {
				IT* ep = &ef[sl * 4 + (u32) ((x >> 35) & 3ull)];
				if (vdjx_peek(ep) > (IT) inst) rc_min(ep, (IT) inst);                  // first sights only ever decrease
			}
		}
	}
	__syncthreads();
	const u64 add = rec_base << ob;
	u32 my_cnt = 0;
	for (u32 i = threadIdx.x; i < SB; i += RC_THREADS) {
		const u32 sb = (s0 + i) >> 4;
		const u32 blk = (sb * f.inv) & f.bmask;
		const u32 x = i & 15u, p = (blk << 4) | x;
		if (sb > f.bmask || blk >= nb || p >= ns) continue;
		const u32 cbase = (i >> 4) * RC_CS, mbase = (i >> 4) * RC_MS;
		u32 cc = 0;
		IT mm = NONE;                                          // every run that holds x
		IT me = NONE;                                          // every run that holds x - 1 and x
		for (u32 e = x; e < 16; e++) { cc += c[cbase + e]; const IT t = mv[mbase + e]; mm = t < mm ? t : mm; if (x) me = t < me ? t : me; }
		for (u32 a = 1; a <= x; a++) { const IT t = mv[mbase + 15 + a]; mm = t < mm ? t : mm; if (a < x) me = t < me ? t : me; }
#pragma unroll
		for (u32 k = 0; k < 4; k++) {                          // windows [s, s + w): 1 <= s, s + w - 1 <= 14
			const u32 w = 1u << k;
			const u32 lo = x >= w ? x - w + 1u : 1u, hi = x < 15u - w ? x : 15u - w;
			for (u32 s = lo ? lo : 1u; s <= hi; s++) { const IT t = mv[mbase + rc_win(k, s)]; mm = t < mm ? t : mm; if (s < x) me = t < me ? t : me; }
		}
		my_cnt += cc;
		ucnt[p] = cc;
		ufirst[p] = mm == NONE ? NONE64 : (u64) mm - 15u + x + add;
		IT e4[4] = {ef[i * 4], ef[i * 4 + 1], ef[i * 4 + 2], ef[i * 4 + 3]};
		if (x && me != NONE) {
			const u32 pa = (fbl[i >> 4] >> (2 * (x - 1))) & 3u;
			const IT t = (IT) (me - 15u + x);
			if (t < e4[pa]) e4[pa] = t;
		}
		for (u32 q = 0; q < 4; q++) in_first[(size_t) p * 4 + q] = e4[q] == NONE ? NONE64 : (u64) e4[q] + add;
	}
	if (my_cnt) atomicAdd(&tot, (unsigned long long) my_cnt);
	__syncthreads();
	if (threadIdx.x == 0 && tot) atomicAdd(n_inst, tot);
}

// survivors the walk never met (whole-pool builds: there are none, every survivor has gated instances in this pool)
__global__ void k_count_unseen(const u64* __restrict__ ufirst, u32 n, u32* __restrict__ out) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool un = i < n && ufirst[i] == NONE64;
	const u64 m = __ballot(un);
	if (m && __lane_id() == 0) atomicAdd(out, (u32) __popcll(m));
}

// in-edge slots (v, first base of u) -> the predecessor u, and the out-edge slot (u, last base of v) of the same edge
__global__ void k_edges_from_in(SurvTable t, u32 n, int k, const u64* __restrict__ in_first, u32* __restrict__ in_from,
                                u64* __restrict__ edge_first, u32* __restrict__ edge_to) {
	const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * 4u) return;
	const u64 fs = in_first[e];
	if (fs == NONE64) { in_from[e] = NONE32; return; }
	const u32 v = e >> 2, a = e & 3u;
	const ulonglong2 kk = t.skey[v];
	const u128 kv = ((u128) kk.y << 64) | kk.x;
	const u128 ku = (kv >> 2) | ((u128) a << (2 * (k - 1)));
	const int u = surv_lookup2(t, (u64) ku, (u64) (ku >> 64));
	in_from[e] = u >= 0 ? (u32) u : NONE32;        // (always found: the walk saw it survive)
	if (u >= 0) {
		const u32 oe = (u32) u * 4u + ((u32) kk.x & 3u);
		edge_first[oe] = fs;
		edge_to[oe] = v;
	}
}

// ----------------------------------------------------------------------------------------------
// multi-GPU: survivor records as exchanged between owners
// ----------------------------------------------------------------------------------------------
struct SurvRec { u64 lo, hi; u32 gcnt, pad; u64 gfirst; };   // 32 bytes: what owners exchange

__global__ void k_surv_pack(const u64* __restrict__ lo, const u64* __restrict__ hi, const u32* __restrict__ gcnt,
                            const u64* __restrict__ gfirst, u32 n, SurvRec* __restrict__ out) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	SurvRec r{lo[i], hi[i], gcnt[i], 0u, gfirst[i]};
	out[i] = r;
}

// (n_real: the survivors that are not shadows, GC_SHADOW)
__global__ void k_surv_unpack(const SurvRec* __restrict__ in, u32 n, u64* __restrict__ lo, u64* __restrict__ hi, u32* __restrict__ gcnt,
                              u64* __restrict__ gfirst, u32* __restrict__ n_real) {
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in_range = i < n;
	SurvRec r{0, 0, GC_SHADOW, 0, 0};
	if (in_range) {
		r = in[i];
		lo[i] = r.lo; hi[i] = r.hi; gcnt[i] = r.gcnt; gfirst[i] = r.gfirst;
	}
	(void) vdjx_wave_inc(n_real, in_range && !(r.gcnt & GC_SHADOW));
}

// ----------------------------------------------------------------------------------------------
// K6: node numbering and list building on the device.
// Node ids are creation order (new_node, A2:188-204) = rank of the node's first ungated instance.
// First instances are distinct integers: the rank is the position in their sorted order (k_rank_keys, a radix sort, k_rank_scatter).
// toNodes/fromNodes are prepend-on-first-sight lists
// (link_nodes, A2:223-237) of at most 4 entries: a 4-element sort by first sight, newest first.
// ----------------------------------------------------------------------------------------------
struct NodeOut {
	u64* first_inst; u32* gcnt; u32* freq; uint8_t* hv; uint8_t* hj; u64* klo; u64* khi; char* kmers;
	uint8_t* to_deg; u32* to_ids; uint8_t* from_deg; u32* from_ids;
};

// node numbering by first sight = the survivors sorted by their first instance (a bitmap over the whole instance space with a
// popcount prefix did this without a sort: 80 MB cleared, marked, counted and prefixed for one million set bits, 0.24 ms at 10 M
// pairs, and growing with the pool, not with the graph)
__global__ void k_rank_keys(const u64* __restrict__ ufirst, const u32* __restrict__ gcnt, u32 n, u64* __restrict__ key, u32* __restrict__ idx) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { key[i] = (gcnt[i] & GC_SHADOW) ? NONE64 : ufirst[i]; idx[i] = i; }      // (shadows sort behind every node: they get none)
}
__global__ void k_rank_scatter(const u64* __restrict__ key, const u32* __restrict__ idx, u32 n, u32* __restrict__ rank) {
	const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const u32 i = idx[j];
	rank[i] = key[j] == NONE64 ? i : j;               // (an unseen survivor sorts last: the build fails, any number does)
}

// roots (identify_root_nodes, A2:653-676: nodes without predecessor) as an ascending index list
__global__ __launch_bounds__(256) void k_root_count(const uint8_t* __restrict__ from_deg, u32 n, u32* __restrict__ block_cnt) {
	__shared__ u32 cnt;
	if (threadIdx.x == 0) cnt = 0;
	__syncthreads();
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	const bool r = i < n && from_deg[i] == 0;
	const u64 b = __ballot(r);
	if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt, (u32) __popcll(b));
	__syncthreads();
	if (threadIdx.x == 0) block_cnt[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(256) void k_root_list(const uint8_t* __restrict__ from_deg, u32 n, const u32* __restrict__ block_start,
                                                   u32* __restrict__ roots) {
	__shared__ u32 wave_cnt[4];
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	const bool r = i < n && from_deg[i] == 0;
	const u64 b = __ballot(r);
	const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (lane == 0) wave_cnt[w] = (u32) __popcll(b);
	__syncthreads();
	u32 base = block_start[blockIdx.x];
	for (u32 j = 0; j < w; j++) base += wave_cnt[j];
	if (r) roots[base + (u32) __popcll(b & ((1ull << lane) - 1))] = i;
}

__device__ inline void sort4_desc64(u64 (&f)[4], u32 (&id)[4]) {
	// newest first sight first (prepend-on-first-sight lists, A2:223-237); absent entries (NONE64) go last
#define CSW(a, b) { const bool sw = kk[a] < kk[b]; if (sw) { u64 t = kk[a]; kk[a] = kk[b]; kk[b] = t; u32 u = id[a]; id[a] = id[b]; id[b] = u; } }
	u64 kk[4];
	for (int i = 0; i < 4; i++) kk[i] = f[i] == NONE64 ? 0ull : f[i] + 1;
	CSW(0, 1) CSW(2, 3) CSW(0, 2) CSW(1, 3) CSW(1, 2)
	for (int i = 0; i < 4; i++) f[i] = kk[i] ? kk[i] - 1 : NONE64;
#undef CSW
}

__global__ void k_node_emit2(const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, const u32* __restrict__ s_gcnt,
                             const u32* __restrict__ s_ucnt, const u64* __restrict__ s_ufirst, const uint8_t* __restrict__ hv,
                             const uint8_t* __restrict__ hj, const u32* __restrict__ rank, const u64* __restrict__ edge_first,
                             const u32* __restrict__ edge_to, const u64* __restrict__ in_first, const u32* __restrict__ in_from,
                             u32 n, int k, NodeOut o) {
	const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	if (s_gcnt[s] & GC_SHADOW) return;                       // a k-mer that went along for its reverse complement's sake only (GC_SHADOW)
	const u32 r = rank[s];
	o.first_inst[r] = s_ufirst[s];
	o.gcnt[r] = s_gcnt[s];
	o.freq[r] = s_ucnt[s] > 32765u ? 32765u : s_ucnt[s];            // A2:261-265
	o.hv[r] = hv[s];
	o.hj[r] = hj[s];
	const u128 key = ((u128) s_hi[s] << 64) | s_lo[s];
	for (int j = 0; j < k; j++) {
		const u32 b = (u32) (key >> (2 * (k - 1 - j))) & 3u;
		o.kmers[(size_t) r * k + j] = b == 0 ? 'A' : (b == 1 ? 'T' : (b == 2 ? 'C' : 'G'));
	}
	u64 f[4];
	u32 id[4];
	// (an edge to or from a shadow does not exist: the reference never saw that k-mer survive, A2:316-318)
	for (int e = 0; e < 4; e++) {
		f[e] = edge_first[(size_t) s * 4 + e];
		if (f[e] != NONE64 && (s_gcnt[edge_to[(size_t) s * 4 + e]] & GC_SHADOW)) f[e] = NONE64;
		id[e] = f[e] != NONE64 ? rank[edge_to[(size_t) s * 4 + e]] + 1 : 0;
	}
	sort4_desc64(f, id);
	u32 deg = 0;
	for (int e = 0; e < 4; e++) { o.to_ids[(size_t) r * 4 + e] = f[e] != NONE64 ? id[e] : 0; deg += f[e] != NONE64; }
	o.to_deg[r] = (uint8_t) deg;
	for (int e = 0; e < 4; e++) {
		f[e] = in_first[(size_t) s * 4 + e];
		if (f[e] != NONE64 && (in_from[(size_t) s * 4 + e] == NONE32 || (s_gcnt[in_from[(size_t) s * 4 + e]] & GC_SHADOW))) f[e] = NONE64;
		id[e] = f[e] != NONE64 ? rank[in_from[(size_t) s * 4 + e]] + 1 : 0;
	}
	sort4_desc64(f, id);
	deg = 0;
	for (int e = 0; e < 4; e++) { o.from_ids[(size_t) r * 4 + e] = f[e] != NONE64 ? id[e] : 0; deg += f[e] != NONE64; }
	o.from_deg[r] = (uint8_t) deg;
}

// ----------------------------------------------------------------------------------------------
// host driver: the build as stages, shared by the single-GPU call and the sharded (multi-GPU) phases
// ----------------------------------------------------------------------------------------------
namespace {

// allocations that outlive one API call (the sharded phases): the context's second arena
struct PersistAlloc {
	vdjx_ctx* c;
	explicit PersistAlloc(vdjx_ctx* ctx) : c(ctx) {}
	template <typename T> hipError_t alloc(T** out, size_t n) {
		*out = (T*) c->shard_arena.alloc((n ? n : 1) * sizeof(T));
		return *out ? hipSuccess : hipErrorOutOfMemory;
	}
	vdjx_arena::mark_t mark() const { return c->shard_arena.mark(); }
	void release_to(vdjx_arena::mark_t m) { c->shard_arena.release_to(m); }
};

struct PoolView { const u64* bases; const u64* nmask; vdjx_qrows quals; int rl; int ob; };

size_t tune(const char* name, size_t dflt) {      // undocumented tuning knobs for experiments (profiles/README.md)
	const char* v = getenv(name);
	return v && atol(v) > 0 ? (size_t) atol(v) : dflt;
}

// ==============================================================================================
// round-2 host driver
// ==============================================================================================
template <typename TUP> struct GTuples {
	TUP* t = nullptr;
	u32* bucket_start = nullptr;      // [NB+1]
	u32 NB = 0, N = 0;
};

struct SurvivorsG {
	u64 *lo = nullptr, *hi = nullptr, *gfirst = nullptr, *ufirst = nullptr;
	u32 *gcnt = nullptr, *ucnt = nullptr;
	u32 n = 0, n_real = 0;
	u64 ndist = 0;
};

inline u32 ceil_log2_u64(u64 x) { u32 b = 0; while ((1ull << b) < x) b++; return b; }

// VDJX_SYNC_DEBUG=1: wait for the stream after every stage and say so (which launch faulted)
static void dbg_sync(vdjx_ctx* c, const char* what) {
	static const bool on = getenv("VDJX_SYNC_DEBUG") != nullptr;
	if (!on) return;
	const hipError_t e = hipStreamSynchronize(c->stream);
	fprintf(stderr, "[vdjx] %s: %s\n", what, hipGetErrorString(e));
}

// Phase A, partition: the gated instances of `pool` as tuples grouped by the top T bits of the k-mer hash (T chosen from their
// number: ~4096 per bucket).  One host read (the tuple count) sizes everything that follows.
// the histogram of the gated instances over the hash buckets, and their number (one host wait): what the cut below starts from
struct GatedHist { u32 HB = 0, NBH = 0, N = 0; u32* hstart = nullptr; bool sym = false; };
// sym: the pool's records come in couples (record, its reverse complement: vdjx_pool::sym) and k is odd: the kernels walk the couples
// and move one tuple per pair of mirrored instances (k_gated_hist SYM)
static bool build_sym(const vdjx_pool* pool, int k) {
	static const bool on = getenv("VDJX_NO_SYM") == nullptr;
	return on && pool->sym && pool->W == 2 && (k & 1) && pool->n_records % 2 == 0;
}
template <typename A>
int stage_gated_hist(vdjx_ctx* c, A& db, const vdjx_pool* pool, int k, size_t per_bucket, u64 geometry_instances, GatedHist* gh, bool sym = false) {
	hipStream_t st = c->stream;
	const size_t R = sym ? pool->n_records / 2 : pool->n_records;           // (sym: couples)
	const int P = pool->rl - k + 1;
	const u64 NI = geometry_instances ? geometry_instances : (u64) R * (u64) P;
	static const size_t dflt = tune("VDJX_GATED_BUCKET", 3072);     // gated tuples per bucket: about a third are distinct k-mers (RD_SLOTS)
	const size_t per = per_bucket ? per_bucket : dflt;
	// histogram resolution: every bucket count the build could choose is a prefix of it (<= 2^15: 128 KB of LDS; 2^14 for long reads,
	// whose waves stage 5.6 KB of records each beside the histogram: 2^15 asked for 173 KB and every build of more than ~1 M pairs of
	// 2 x 100 bp failed -- the parity tests' pools were too small to get there; profiles/longreads.py now runs them at size)
	const u32 hb_max = pool->W > 2 ? 14u : 15u;
	u32 HB = ceil_log2_u64((NI + per - 1) / per);
	HB = std::max(8u, std::min(hb_max, HB));
	const u32 NBH = 1u << HB;
	static const u32 hist_dbg = (u32) tune("VDJX_HIST_DBG", 0);      // (ablation build only: profiles/histdbg.py)
	static const size_t hist_blocks = tune("VDJX_HIST_BLOCKS", 512);
	u32 nblk = (u32) std::min<size_t>(hist_blocks, (R + 4095) / 4096);
	if (nblk == 0) nblk = 1;
	size_t rpb = (R + nblk - 1) / nblk;
	u32 *hcnt, *hstart;
	HIP_TRY(db.alloc(&hcnt, NBH));
	HIP_TRY(db.alloc(&hstart, NBH + 1));
	HIP_TRY(hipMemsetAsync(hcnt, 0, (size_t) NBH * 4, st));
	const bool lng = pool->W > 2;
	const size_t lds_hist = (size_t) NBH * 4 + (HIST_THREADS / 64) * (lng ? GL_WAVE_BYTES_LONG : sym ? GL_WAVE_BYTES_SYM : GL_WAVE_BYTES);
	HIP_TRY(hipFuncSetAttribute((const void*) k_gated_hist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_hist));
	HIP_TRY(hipFuncSetAttribute((const void*) k_gated_hist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_hist));
	HIP_TRY(hipFuncSetAttribute((const void*) k_gated_hist<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_hist));
	{
		vdjx_prof_scope ps(c, "k_gated_hist");
		if (lng) hipLaunchKernelGGL(k_gated_hist<true>, dim3(nblk), dim3(HIST_THREADS), lds_hist, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, pool->rl, k, HB, rpb, hcnt, hist_dbg);
		else if (sym) hipLaunchKernelGGL((k_gated_hist<false, true>), dim3(nblk), dim3(HIST_THREADS), lds_hist, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, pool->rl, k, HB, rpb, hcnt, hist_dbg);
		else hipLaunchKernelGGL(k_gated_hist<false>, dim3(nblk), dim3(HIST_THREADS), lds_hist, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, pool->rl, k, HB, rpb, hcnt, hist_dbg);
	}
	dbg_sync(c, "k_gated_hist");
	hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, hcnt, NBH, hstart);
	HIP_TRY(hipMemcpyAsync(c->h_pin, hstart + NBH, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
#ifdef VDJX_ABLATE
	if (hist_dbg) return VDJX_ESTATE;             // (a cut histogram sizes nothing: the build stops here, profiles/histdbg.py reads the kernel's time)
#endif
	const u32 N = *(const u32*) c->h_pin;
	c->stats["gated_instances"] = sym ? 2ull * N : N;      // k-mer instances that pass include_kmer (A2:240-259) in this pool (sym: a tuple stands for two)
	gh->HB = HB; gh->NBH = NBH; gh->N = N; gh->hstart = hstart; gh->sym = sym;
	return VDJX_OK;
}

// the cut: the gated instances as tuples, by bucket.  Bucket bits from the actual number of gated instances; in the sharded build every
// rank must cut the same buckets, so there the geometry follows a number all ranks agree on (`geometry_instances`: the largest count
// of any rank when they have compared them, a bound from the stride when they have not)
template <typename TUP, typename A>
int stage_gated_cut(vdjx_ctx* c, A& db, const vdjx_pool* pool, u64 rec_base, int k, size_t per_bucket, u64 geometry_instances, const GatedHist& gh, GTuples<TUP>* out) {
	hipStream_t st = c->stream;
	const bool sym = gh.sym;
	const size_t R = sym ? pool->n_records / 2 : pool->n_records;           // (sym: couples, stage_gated_hist)
	const int P = pool->rl - k + 1;
	static const size_t dflt = tune("VDJX_GATED_BUCKET", 3072);
	const size_t per = per_bucket ? per_bucket : dflt;
	const u32 hb_max = pool->W > 2 ? 14u : 15u;
	const bool lng = pool->W > 2;
	const u32 HB = gh.HB, NBH = gh.NBH, N = gh.N;
	u32* hstart = gh.hstart;
	size_t rpb;
	out->N = N;
	const u64 Ng = geometry_instances ? geometry_instances : (u64) N;
	static const size_t refine = tune("VDJX_REFINE_TUPLES", 4096);
	u32 T = 8;
	while (T < HB && ((u64) per << T) < Ng) T++;
	u32 extra = 0;
	// (beyond the histogram's 2^15 buckets the tuples of a bucket are cut again, BEFORE a bucket's distinct k-mers -- a third of its
	// tuples -- come near the 2,048 slots of the reduce kernel's table: at 6,100 tuples per bucket (20 M pairs) every bucket overflowed
	// its table after probing it to the brim, 52 ms instead of 3)
	const size_t refine_at = per_bucket ? std::max<size_t>(per_bucket, refine / 2) : refine;
	if (T == HB && HB == hb_max && Ng / NBH > refine_at) {
		extra = 1;
		while (extra < 5 && (Ng >> extra) / NBH > refine_at) extra++;
	}
	const u32 Tt = T + extra;
	const u32 NBt = 1u << Tt;
	HIP_TRY(db.alloc(&out->t, (size_t) N + 1));
	constexpr u32 stage_tuples = PARTR_STAGE_BYTES / (u32) sizeof(TUP);
	constexpr u32 lds_partr = PARTR_STAGE_BYTES + stage_tuples * 4;
	HIP_TRY(hipFuncSetAttribute((const void*) k_part_records_g<TUP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_partr));
	HIP_TRY(hipFuncSetAttribute((const void*) k_part_records_g<TUP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_partr));
	HIP_TRY(hipFuncSetAttribute((const void*) k_part_records_g<TUP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_partr));
	HIP_TRY(hipFuncSetAttribute((const void*) k_part_tuples_g<TUP>, hipFuncAttributeMaxDynamicSharedMemorySize, PART_LDS_BYTES));
	// pass geometry: <= 1024 buckets in one pass; otherwise 256 coarse x the rest (large pools: 2^(Tt-10) coarse x 1024)
	u32 cbits = Tt, fbits = 0;
	if (Tt > 10) { cbits = extra ? Tt - 10 : 8; fbits = Tt - cbits; }
	const u32 NBc = 1u << cbits;
	u32 nblk2 = (u32) std::min<size_t>(1024, (R + 2047) / 2048);
	if (nblk2 == 0) nblk2 = 1;
	rpb = (R + nblk2 - 1) / nblk2;
	u32* gcur;
	HIP_TRY(db.alloc(&gcur, NBc));
	TUP* l1 = out->t;
	if (fbits) HIP_TRY(db.alloc(&l1, (size_t) N + 1));
	{
		vdjx_prof_scope ps(c, "k_part_records");
		hipLaunchKernelGGL(k_init_cursors, dim3((NBc + 255) / 256), dim3(256), 0, st, hstart, NBc, HB - cbits, gcur);
		// records per round: three quarters of the stage at the measured gated fraction, in whole sweeps of the workgroup
		const u64 NIl = (u64) R * (u64) P;
		u64 rr = N ? (u64) stage_tuples * 3 / 4 * NIl / ((u64) N * (u64) P) : 1u << 16;
		rr = std::max<u64>(PART_THREADS, std::min<u64>(rr / PART_THREADS * PART_THREADS, lng ? 1u << 14 : 1u << 16));      // (the descriptor's record field)
		if (lng) hipLaunchKernelGGL((k_part_records_g<TUP, true>), dim3(nblk2), dim3(PART_THREADS), lds_partr, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, rec_base,
		                            pool->rl, k, 64 - cbits, NBc, rpb, (u32) rr, gcur, l1);
		else if (sym) hipLaunchKernelGGL((k_part_records_g<TUP, false, true>), dim3(nblk2), dim3(PART_THREADS), lds_partr, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, rec_base,
		                                 pool->rl, k, 64 - cbits, NBc, rpb, (u32) rr, gcur, l1);
		else hipLaunchKernelGGL((k_part_records_g<TUP, false>), dim3(nblk2), dim3(PART_THREADS), lds_partr, st, pool->d_bases, pool->d_nmask, pool->d_lowq, R, rec_base,
		                        pool->rl, k, 64 - cbits, NBc, rpb, (u32) rr, gcur, l1);
	}
	dbg_sync(c, "k_part_records");
	u32* tstart;                                   // starts of the final buckets
	HIP_TRY(db.alloc(&tstart, NBt + 1));
	out->NB = NBt;
	out->bucket_start = tstart;
	static const u32 slices = (u32) tune("VDJX_PART_SLICES", 8);
	if (extra) {
		u32* fine_cnt;
		HIP_TRY(db.alloc(&fine_cnt, NBt));
		HIP_TRY(hipMemsetAsync(fine_cnt, 0, (size_t) NBt * 4, st));
		vdjx_prof_scope ps(c, "k_seg_hist");
		hipLaunchKernelGGL(k_seg_hist_g<TUP>, dim3(NBc * slices), dim3(512), 0, st, l1, hstart, HB - cbits, slices, 64 - Tt, fbits, fine_cnt);
		hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, fine_cnt, NBt, tstart);
	} else {
		hipLaunchKernelGGL(k_pick_u32, dim3((NBt + 1 + 255) / 256), dim3(256), 0, st, hstart, 1u << (HB - Tt), NBt + 1, tstart);
	}
	if (fbits) {
		u32* gcur2;
		HIP_TRY(db.alloc(&gcur2, NBt));
		vdjx_prof_scope ps(c, "k_part_tuples");
		hipLaunchKernelGGL(k_init_cursors, dim3((NBt + 255) / 256), dim3(256), 0, st, tstart, NBt, 0u, gcur2);
		hipLaunchKernelGGL(k_part_tuples_g<TUP>, dim3(NBc * slices), dim3(PART_THREADS), PART_LDS_BYTES, st, l1, hstart, HB - cbits, slices, 64 - Tt, fbits,
		                   gcur2, out->t);
	}
	return VDJX_OK;
}

template <typename TUP, typename A>
int stage_gated_partition(vdjx_ctx* c, A& db, const vdjx_pool* pool, u64 rec_base, int k, size_t per_bucket, u64 geometry_instances, GTuples<TUP>* out, bool sym = false) {
	GatedHist gh;
	int rc = stage_gated_hist(c, db, pool, k, per_bucket, geometry_instances, &gh, sym);
	if (rc) return rc;
	return stage_gated_cut<TUP>(c, db, pool, rec_base, k, per_bucket, geometry_instances, gh, out);
}

// Phase A, table + prune per bucket
template <typename TUP, typename A>
int stage_gated_reduce(vdjx_ctx* c, A& db, const GTuples<TUP>& t, const PoolView& pv, u64 rec_base, int k, int mf, int mq, SurvivorsG* sv, bool sym = false) {
	hipStream_t st = c->stream;
	{
		static const u32 sub = (u32) tune("VDJX_SUB_TUPLES", 262144);
		if (c->sub_tuples_set != sub) {            // (the module's variable: the same for every context of the process, set again by each once)
			HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sub_tuples), &sub, 4, 0, hipMemcpyHostToDevice, st));
			c->sub_tuples_set = sub;
		}
	}
	if (mq >= 255) mq = 254;                                        // A2:1514-1516
	const u32 mqq = (u32) (mq < 0 ? 0 : (mq > 214 ? 214 : mq));      // a sum >= 214 reads as 255 (A2:356-360)
	const u32 tlow = 1 + (mqq + 19) / 20;                           // see k_bucket_finalize
	const u32 cmin = (u32) std::max(mf, 2);
	const u32 mfu = (u32) std::max(mf, 0);
	const u32 cap = (sym ? t.N : t.N / 2) + 16;                     // a survivor has at least two gated instances (sym: two survivors per two tuples)
	u32 *g_err, *n_surv;
	u64* g_distinct;
	HIP_TRY(db.alloc(&sv->lo, cap)); HIP_TRY(db.alloc(&sv->hi, cap)); HIP_TRY(db.alloc(&sv->gcnt, cap)); HIP_TRY(db.alloc(&sv->gfirst, cap));
	HIP_TRY(db.alloc(&g_distinct, 64 * 16 + 2));                     // one block, cleared and read back in one piece: the counters of distinct k-mers | error word, survivors | real survivors
	g_err = (u32*) (g_distinct + 64 * 16);
	n_surv = g_err + 1;
	u32* n_real = g_err + 2;
	HIP_TRY(hipMemsetAsync(g_distinct, 0, (64 * 16 + 2) * 8, st));
	SurvOutG so{sv->lo, sv->hi, sv->gcnt, sv->gfirst, n_surv, cap, n_real};
	static const u32 rd_dbg = (u32) tune("VDJX_RD_DBG", 0);        // profiles/reducedbg.py: the kernel stops after a phase
	if (t.N) {
		vdjx_prof_scope ps(c, "k_gated_reduce");
		static const bool by_size = tune("VDJX_RD_ORDER", 1) == 1;
		u32* order = nullptr;
		if (by_size && t.NB >= 1024) {
			HIP_TRY(db.alloc(&order, t.NB));
			hipLaunchKernelGGL(k_bucket_order, dim3(1), dim3(1024), 0, st, t.bucket_start, t.NB, order);
		}
		if (sym) hipLaunchKernelGGL((k_gated_reduce<TUP, true>), dim3(t.NB), dim3(RD_THREADS), 0, st, t.t, t.bucket_start, pv.bases, pv.nmask, pv.quals, pv.rl, pv.ob, k, rec_base,
		                            mfu, cmin, mqq, tlow, so, g_distinct, g_err, rd_dbg, (const u32*) order);
		else hipLaunchKernelGGL(k_gated_reduce<TUP>, dim3(t.NB), dim3(RD_THREADS), 0, st, t.t, t.bucket_start, pv.bases, pv.nmask, pv.quals, pv.rl, pv.ob, k, rec_base,
		                        mfu, cmin, mqq, tlow, so, g_distinct, g_err, rd_dbg, (const u32*) order);
	}
	u64* spread = (u64*) c->h_pin;                    // [64 * 16], then ns, err
	u32* tail = (u32*) (spread + 64 * 16);
	static_assert((64 * 16 + 2) * 8 == VDJX_HPIN_SYNC_BYTES, "h_pin layout (vdjx_common.h)");
	HIP_TRY(hipMemcpyAsync(spread, g_distinct, VDJX_HPIN_SYNC_BYTES, hipMemcpyDeviceToHost, st));      // (one transfer: tail[] = error word, survivors, real survivors)
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	const u32 ns = tail[1], err = tail[0];
	if (err) { vdjx_set_error("k_gated_reduce: %u buckets could not be split to fit LDS", err); return VDJX_EHIP; }
	if (ns > cap) { vdjx_set_error("survivor capacity logic failed (%u > %u)", ns, cap); return VDJX_EHIP; }
	sv->ndist = 0;
	for (int i = 0; i < 64; i++) sv->ndist += spread[i * 16];
	sv->n = ns;
	sv->n_real = sym ? tail[2] : ns;             // (sym: the set is closed under reverse complement; the rest are shadows, GC_SHADOW)
	c->stats["kmer_build_shadows"] = ns - sv->n_real;
	return VDJX_OK;
}

// Phase B over the records of `pool` (local numbering; rec_base is added to the first sights): node frequency and first sight,
// in-edge first sights and, derived from them, both edge directions.  Output arrays are caller-provided:
//   ucnt u32 [ns] (raw count), ufirst u64 [ns], in_first u64 [ns*4], in_from u32 [ns*4], edge_first u64 [ns*4], edge_to u32 [ns*4]
struct RecountOut { u32* ucnt; u64* ufirst; u64* in_first; u32* in_from; u64* edge_first; u32* edge_to; };
// what stage_recount leaves on the device for a caller that waits later (one host wait less per build): read back by recount_status_read
struct RecountStatus { const u32* g_err = nullptr; const u32* n_items = nullptr; const unsigned long long* n_inst = nullptr; u32 ns = 0; };
#define VDJX_ESYMWALK (-1000)          // (internal) the mirrored walk met a survivor set or chains that are not mirror images: walk every record instead
static int recount_status_check(vdjx_ctx* c, const u32* hp, u32 ns) {          // hp: err[2], n_items, -, inst (u64), sym violations [2]
	if (hp[6] || hp[7]) { vdjx_set_error("the mirrored walk found %u + %u places where the chains are not mirror images", hp[6], hp[7]); return VDJX_ESYMWALK; }
	if (hp[0]) { vdjx_set_error("k_walk_items: item buffer too small (%u waves stopped)", hp[0]); return VDJX_EHIP; }
	if (hp[1]) { vdjx_set_error("internal error: %u of %u surviving k-mers were not met again in the records (VDJX_SYNC_DEBUG=1 names them)", hp[1], ns); return VDJX_EHIP; }
	c->stats["recount_items"] = hp[2];                                    // runs of surviving k-mer instances of this pool (8 bytes each)
	c->stats["recount_instances"] = *(const unsigned long long*) (hp + 4);          // the instances themselves
	return VDJX_OK;
}

template <typename A>
int stage_recount(vdjx_ctx* c, A& db, const vdjx_pool* pool, u64 rec_base, int k, SurvivorsG& sv, const RecountOut& ro,
                  bool derive_edges, SurvTable* table_out = nullptr, RecountStatus* defer = nullptr, bool sym_walk = false) {
	hipStream_t st = c->stream;
	const size_t R = pool->n_records;
	const int P = pool->rl - k + 1;
	const u32 ns = sv.n;
	if (ns >= (1u << 26)) { vdjx_set_error("more than 2^26 surviving k-mers: not supported by the recount items"); return VDJX_ELIMIT; }
	if (R >= (1ull << 29)) { vdjx_set_error("more than 2^29 records on one GPU: not supported by the recount items"); return VDJX_ELIMIT; }
	if ((u64) R * (u64) P >= (1ull << 32)) { vdjx_set_error("records*offsets = %llu >= 2^32 on one GPU: shard the pool over more GPUs", (unsigned long long) ((u64) R * (u64) P)); return VDJX_ELIMIT; }
	u32 tmask = 1023;
	while ((size_t) tmask + 1 < (size_t) ns * 2) tmask = tmask * 2 + 1;
	unsigned long long* pred;
	ulonglong2* table;
	u32 *succ0, *succ, *bloom, *clen, *coff, *csum, *csum_start, *newidx, *jump_open, *fbw, *gcnt2, *bestsucc, *partner = nullptr;
	u64 *pd, *lo2, *hi2, *gfirst2, *linw;
	ulonglong2 *skey0, *skey;
	u32 bloom_bits = 1u << 16;
	while (bloom_bits < (1u << 30) && (size_t) bloom_bits < (size_t) ns * 16) bloom_bits <<= 1;
	const u32 nb16 = (ns + 15u) >> 4;
	const u32 n_scan = (ns + SCAN_BLOCK - 1) / SCAN_BLOCK;
	const u32 n_jump = (ceil_log2_u64((u64) ns + 1) + 4) / 5 + 1;      // launches that resolve every chain of up to `ns` nodes (32x each at least)
	// what has to start as zeros lies in one block, cleared by one launch (six small clears cost the stream their turn-arounds)
	{
		auto up = [](size_t b) { return (b + 255) & ~(size_t) 255; };
		const size_t b_bloom = up(bloom_bits / 8), b_table = up(((size_t) tmask + 1) * 16), b_pred = up((size_t) ns * 8), b_clen = up((size_t) ns * 4),
		             b_jump = up((size_t) (n_jump + 1) * 4), b_linw = up(((size_t) nb16 + 1) * 8);     // (linw: the walk reads the word behind a run's first block)
		uint8_t* z;
		const size_t zb = b_bloom + b_table + b_pred + b_clen + b_jump + b_linw;
		HIP_TRY(db.alloc(&z, zb));
		HIP_TRY(hipMemsetAsync(z, 0, zb, st));
		table = (ulonglong2*) z; z += b_table;
		bloom = (u32*) z; z += b_bloom;
		pred = (unsigned long long*) z; z += b_pred;
		clen = (u32*) z; z += b_clen;
		jump_open = (u32*) z; z += b_jump;
		linw = (u64*) z;
	}
	HIP_TRY(db.alloc(&skey0, ns)); HIP_TRY(db.alloc(&skey, ns));
	HIP_TRY(db.alloc(&succ0, (size_t) ns * 4)); HIP_TRY(db.alloc(&succ, (size_t) ns * 4));
	HIP_TRY(db.alloc(&pd, ns)); HIP_TRY(db.alloc(&coff, ns));
	HIP_TRY(db.alloc(&csum, n_scan)); HIP_TRY(db.alloc(&csum_start, n_scan + 1)); HIP_TRY(db.alloc(&newidx, ns));
	HIP_TRY(db.alloc(&lo2, ns)); HIP_TRY(db.alloc(&hi2, ns)); HIP_TRY(db.alloc(&gcnt2, ns)); HIP_TRY(db.alloc(&gfirst2, ns));
	HIP_TRY(db.alloc(&fbw, nb16));
	HIP_TRY(db.alloc(&bestsucc, ns));
	if (sym_walk) HIP_TRY(db.alloc(&partner, ns));
	const SurvTable tb0{table, tmask, skey0, bloom, bloom_bits - 1};      // arrival numbering
	const SurvTable tb{table, tmask, skey, bloom, bloom_bits - 1};        // chain order (after k_table_remap)
	if (table_out) *table_out = tb;
	const dim3 gs((ns + 255) / 256), bs(256);
	{
		vdjx_prof_scope ps(c, "k_surv_table");
		hipLaunchKernelGGL(k_surv_table2, gs, bs, 0, st, sv.lo, sv.hi, ns, table, tmask, skey0, bloom, bloom_bits - 1);
		hipLaunchKernelGGL(k_succ_links2, gs, bs, 0, st, tb0, ns, k, sv.gcnt, succ0, pred, bestsucc);
	}
	{
		// chain order: the survivors, their table entries and their successor lists renumbered
		vdjx_prof_scope ps(c, "k_chain_order");
		hipLaunchKernelGGL(k_chain_init, gs, bs, 0, st, pred, bestsucc, ns, pd);
		for (u32 j = 0; j < n_jump; j++) hipLaunchKernelGGL(k_chain_jump, gs, bs, 0, st, pd, ns, j ? jump_open + j - 1 : (const u32*) nullptr, jump_open + j);
		hipLaunchKernelGGL(k_chain_len, gs, bs, 0, st, pd, ns, clen);
		hipLaunchKernelGGL(k_scan_sums, dim3(n_scan), dim3(256), 0, st, clen, ns, csum);
		hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, csum, n_scan, csum_start);
		hipLaunchKernelGGL(k_scan_apply, dim3(n_scan), dim3(256), 0, st, clen, ns, csum_start, coff);
		hipLaunchKernelGGL(k_chain_place, gs, bs, 0, st, pd, coff, ns, newidx);
		hipLaunchKernelGGL(k_chain_permute, gs, bs, 0, st, newidx, ns, sv.lo, sv.hi, sv.gcnt, sv.gfirst, succ0, lo2, hi2, gcnt2, gfirst2, skey, succ);
		hipLaunchKernelGGL(k_table_remap, dim3(tmask / 256 + 1), bs, 0, st, table, tmask + 1, newidx);
		if (sym_walk) hipLaunchKernelGGL(k_partner, gs, bs, 0, st, tb, ns, k, partner);          // (the table answers in chain order now)
		hipLaunchKernelGGL(k_chain_words, gs, bs, 0, st, succ, skey, ns, k, linw, fbw, (const u32*) partner);
	}
	dbg_sync(c, "k_chain_order");
	if (getenv("VDJX_SYNC_DEBUG")) {              // the new numbering is a permutation, and every key finds itself
		std::vector<u32> ni(ns), seen(ns, 0);
		(void) hipMemcpy(ni.data(), newidx, (size_t) ns * 4, hipMemcpyDeviceToHost);
		u32 dup = 0, oob = 0;
		for (u32 i = 0; i < ns; i++) { if (ni[i] >= ns) oob++; else if (seen[ni[i]]++) dup++; }
		std::vector<u64> pdh(ns);
		std::vector<unsigned long long> prh(ns);
		(void) hipMemcpy(pdh.data(), pd, (size_t) ns * 8, hipMemcpyDeviceToHost);
		(void) hipMemcpy(prh.data(), pred, (size_t) ns * 8, hipMemcpyDeviceToHost);
		u32 unresolved = 0;
		for (u32 i = 0; i < ns; i++) { const u32 p0 = (u32) (pdh[i] >> 32); if (p0 != NONE32 && (u32) (pdh[p0] >> 32) != NONE32) unresolved++; }
		fprintf(stderr, "[vdjx] chain order: %u survivors, %u duplicate new indices, %u out of range, %u unresolved (cycles), %u jump launches\n", ns, dup, oob, unresolved, n_jump);
		for (u32 i = 0, shown = 0; i < ns && shown < 6; i++) if (ni[i] < ns && seen[ni[i]] > 1) {
			fprintf(stderr, "[vdjx]   old %u -> new %u: pd = (%u, %u), pred word %llx\n", i, ni[i], (u32) (pdh[i] >> 32), (u32) pdh[i], prh[i]);
			shown++;
		}
	}
	sv.lo = lo2; sv.hi = hi2; sv.gcnt = gcnt2; sv.gfirst = gfirst2;
	// item layout and the permutation of the blocks of 16 over the ranges
	ItemFmt f;
	{
		const u32 sb = ceil_log2_u64(ns);
		static const u32 force_lb = (u32) tune("VDJX_RC_LEN_BITS", 4);      // (test knob: the shorter runs of pools with more than 2^22 survivors)
		const u32 lb = std::min(std::min(4u, 26u - sb), force_lb);
		const u32 pb = 26u - lb;
		f.pmask = (1u << pb) - 1u;
		f.len_shift = IT_SURV_SHIFT + pb;
		f.maxlen = 1u << lb;
		const u32 t = ceil_log2_u64(nb16);
		f.bmask = (1u << t) - 1u;
		f.mul = t ? ((0x9E3779B9u >> (32 - t)) | 1u) : 1u;
		u32 x = f.mul;
		for (int i = 0; i < 5; i++) x *= 2u - f.mul * x;                // inverse mod 2^32 (Newton), then mod 2^t
		f.inv = x & f.bmask;
		f.ns = ns;
	}
	const u32 nsp = (f.bmask + 1u) << 4;                               // scattered index space
	// ranges of the scattered space (one recount workgroup each, LDS arrays indexed by position - range start): the smallest range
	// size that keeps the ranges <= 1024 (one partition pass); beyond the largest size a second partition level
	static const bool force_wide = tune("VDJX_RC_WIDE", 0) != 0;                    // (test knobs: the paths of very large pools on small ones)
	static const u32 cap_shift = (u32) tune("VDJX_RC_MAX_SHIFT", 12);
	const bool narrow = R < (1ull << (32 - pool->ob)) && !force_wide;            // local instance ids (+15) fit 32 bits
	static const u32 max_ranges = (u32) std::min<size_t>(PART_MAXB, std::max<size_t>(2, tune("VDJX_RC_MAX_RANGES", PART_MAXB)));
	const u32 max_shift = std::max(8u, std::min(narrow ? 11u : 10u, cap_shift));
	u32 range_shift = 8;
	while (range_shift < max_shift && (nsp >> range_shift) > max_ranges) range_shift++;
	const u32 n_ranges = std::max(1u, nsp >> range_shift);
	u32 l2bits = 0;
	while (((n_ranges + (1u << l2bits) - 1) >> l2bits) > max_ranges) l2bits++;
	// two levels: as even a split as the limits allow -- 1,024 coarse segments cut in two each meant 8,192 workgroups with a round and a
	// half of items apiece for the second level (2.2 ms for 100 M items: reads of 100 bases at 5 M pairs, 50 bases at 20 M; 64 x 32: 0.7 ms)
	if (l2bits) l2bits = std::min(10u, std::max(l2bits, (ceil_log2_u64(n_ranges) + 1u) / 2u));
	const u32 n_coarse = (n_ranges + (1u << l2bits) - 1) >> l2bits;
	const u32 n_ranges_p = n_coarse << l2bits;                      // padded: every coarse segment has 2^l2bits ranges
	// raw item blocks
	static const size_t walk_blocks = tune("VDJX_WALK_BLOCKS", 7168);      // (seven waves per SIMD are resident: 28 blocks per CU = one resident set x 4; measured 4096 2.45 ms, 7168 2.32, 16384 2.52)
	const size_t Rw = sym_walk ? R / 2 : R;                          // what the walk's lanes take: records, or couples of them
	u32 nblk = (u32) std::min<size_t>(walk_blocks, (Rw + WALK_THREADS * 8 - 1) / (WALK_THREADS * 8));
	// (1,536 blocks are resident at once -- six waves per SIMD --: a pool that asks for a few more pays a second, nearly empty round with
	// the latency of a full one.  1 M pairs, 1,953 blocks asked: 0.41 ms; 1,500: 0.36; 1,700: 0.40.  3 M pairs: 3,000 0.77, 5,000 0.81)
	static const u32 walk_resident = (u32) tune("VDJX_WALK_RESIDENT", 1536);
	if (walk_resident && nblk > walk_resident && nblk < walk_blocks) nblk = nblk / walk_resident * walk_resident;
	// (a pool too small to fill one resident set at eight rows of records per wave takes fewer rows per wave, down to one: 100 k pairs
	// were 98 blocks -- a wave per SIMD on a third of the CUs, every lookup's latency in the open -- and 0.119 ms)
	if (walk_resident && nblk < walk_resident) nblk = (u32) std::min<size_t>(walk_resident, (Rw + WALK_THREADS - 1) / WALK_THREADS);
	if (nblk == 0) nblk = 1;
	const size_t nwaves = (size_t) nblk * (WALK_THREADS / 64);
	const size_t NI = R * (size_t) P;
	u32 blk_items = 4096;
	while (blk_items > 256 && nwaves * blk_items > NI / 2 + 65536) blk_items >>= 1;
	const u64 raw_cap = (u64) NI + (u64) NI / (blk_items / 64) + nwaves * (u64) blk_items + blk_items;
	u64 *raw, *items;
	unsigned long long *g_cursor, *n_inst;
	u32 *range_cnt, *range_start, *g_err, *gcur;
	const vdjx_arena::mark_t tmp_mark = db.mark();          // what follows (items and their bookkeeping: the large part) is dead when the recount is done
	HIP_TRY(db.alloc(&raw, (size_t) raw_cap));
	HIP_TRY(db.alloc(&g_cursor, 4 + ((size_t) n_ranges_p + 1) / 2));      // one cleared block: cursor, instance count | error words | range counts
	n_inst = g_cursor + 1;
	g_err = (u32*) (g_cursor + 2);
	range_cnt = (u32*) (g_cursor + 4);
	HIP_TRY(db.alloc(&range_start, n_ranges_p + 1));
	HIP_TRY(hipMemsetAsync(g_cursor, 0, 32 + (size_t) n_ranges_p * 4, st));
	if (sym_walk) {          // the survivors' reverse complements, and whether the chains are mirror images (g_err[3]: read with the other status words)
		vdjx_prof_scope ps(c, "k_chain_order");
		hipLaunchKernelGGL(k_partner_check, gs, bs, 0, st, partner, linw, ns, g_err + 3);
	}
	const bool lng = pool->W > 2;
	if (lng && R >= (1ull << 27)) { vdjx_set_error("more than 2^27 records of long reads on one GPU: not supported by the recount items"); return VDJX_ELIMIT; }
	const size_t lds_walk = (((size_t) n_ranges_p + 1) & ~(size_t) 1) * 4 + (lng ? (size_t) WALK_THREADS * GL_ROW_LONG * 8 : 0);
	if ((size_t) n_ranges_p * 4 > 64 * 1024) { vdjx_set_error("too many survivor ranges for the walk histogram"); return VDJX_ELIMIT; }
	HIP_TRY(hipFuncSetAttribute((const void*) k_walk_items<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_walk));
	HIP_TRY(hipFuncSetAttribute((const void*) k_walk_items<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_walk));
	HIP_TRY(hipFuncSetAttribute((const void*) k_walk_items<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_walk));
	// VDJX_WALK_DBG (profiles/walkdbg.sh): an ABLATED copy of the walk runs first into scratch outputs, timed as k_walk_dbg; the real
	// one follows untouched.  Bits: 1 no item stores, 2 no range histogram, 8 no filter / table / key loads at run starts.
#ifdef VDJX_ABLATE
	static const u32 walk_dbg = (u32) tune("VDJX_WALK_DBG", 0);
#else
	constexpr u32 walk_dbg = 0;
#endif
	if (R && walk_dbg != 0u) {
		u64* raw2;
		unsigned long long* cur2;
		u32* cnt2;
		HIP_TRY(db.alloc(&raw2, (size_t) raw_cap));
		HIP_TRY(db.alloc(&cur2, 8));
		HIP_TRY(db.alloc(&cnt2, n_ranges_p));
		HIP_TRY(hipMemsetAsync(cur2, 0, 64, st));
		HIP_TRY(hipMemsetAsync(cnt2, 0, (size_t) n_ranges_p * 4, st));
		vdjx_prof_scope ps(c, "k_walk_dbg");
		if (lng) hipLaunchKernelGGL(k_walk_items<true>, dim3(nblk), dim3(WALK_THREADS), lds_walk, st, pool->d_bases, pool->d_nmask, R, pool->rl, pool->ob, k, tb, succ, linw, f, range_shift,
		                            n_ranges_p, raw2, raw_cap, blk_items, cur2, cnt2, g_err, walk_dbg, (const u32*) nullptr);
		else hipLaunchKernelGGL(k_walk_items<false>, dim3(nblk), dim3(WALK_THREADS), lds_walk, st, pool->d_bases, pool->d_nmask, R, pool->rl, pool->ob, k, tb, succ, linw, f, range_shift,
		                        n_ranges_p, raw2, raw_cap, blk_items, cur2, cnt2, g_err, walk_dbg, (const u32*) nullptr);
		HIP_TRY(hipMemsetAsync(g_err, 0, 4, st));
		if (walk_dbg & 32u) {                      // per wave: rows of 64 records, rounds, and the rounds that ran each part
			unsigned long long hs[8];
			HIP_TRY(hipMemcpyAsync(hs, cur2, 64, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			const char* nm[] = {"", "rows", "rounds", "filter", "lookup", "chain", "item_trips"};
			for (int i = 1; i < 7; i++) c->stats[std::string("walk_dbg_") + nm[i]] = hs[i];
		}
	}
	if (R) {
		vdjx_prof_scope ps(c, "k_walk_items");
		if (lng) hipLaunchKernelGGL(k_walk_items<true>, dim3(nblk), dim3(WALK_THREADS), lds_walk, st, pool->d_bases, pool->d_nmask, R, pool->rl, pool->ob, k, tb, succ, linw, f, range_shift,
		                            n_ranges_p, raw, raw_cap, blk_items, g_cursor, range_cnt, g_err, (u32) tune("VDJX_WALK_FLAGS", 0), (const u32*) nullptr);
		else if (sym_walk) hipLaunchKernelGGL((k_walk_items<false, true>), dim3(nblk), dim3(WALK_THREADS), lds_walk, st, pool->d_bases, pool->d_nmask, R / 2, pool->rl, pool->ob, k, tb, succ, linw, f, range_shift,
		                                      n_ranges_p, raw, raw_cap, blk_items, g_cursor, range_cnt, g_err, (u32) tune("VDJX_WALK_FLAGS", 0), (const u32*) partner);
		else hipLaunchKernelGGL(k_walk_items<false>, dim3(nblk), dim3(WALK_THREADS), lds_walk, st, pool->d_bases, pool->d_nmask, R, pool->rl, pool->ob, k, tb, succ, linw, f, range_shift,
		                        n_ranges_p, raw, raw_cap, blk_items, g_cursor, range_cnt, g_err, (u32) tune("VDJX_WALK_FLAGS", 0), (const u32*) nullptr);
	}
	dbg_sync(c, "k_walk_items");
	hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, range_cnt, n_ranges_p, range_start);
	// the partitioned items: at most one per instance.  A build that waits for this stage anyway (the sharded one: `defer` is null)
	// asks how many the walk made and sizes the two partition buffers by that (a tenth of the bound: 12 GB less per rank at 12.5 M
	// pairs); the one-GPU build does not stop for it.
	size_t items_cap = NI + 1;
	if (!defer) {
		u32* hp = (u32*) c->h_pin;
		HIP_TRY(hipMemcpyAsync(hp, range_start + n_ranges_p, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		items_cap = (size_t) hp[0] + 1;
	}
	HIP_TRY(db.alloc(&items, items_cap));
	HIP_TRY(hipFuncSetAttribute((const void*) k_part_items, hipFuncAttributeMaxDynamicSharedMemorySize, PART_LDS_BYTES));
	HIP_TRY(db.alloc(&gcur, n_ranges_p));
	{
		vdjx_prof_scope ps(c, "k_part_items");
		u32 npb = (u32) std::min<size_t>(1024, (NI / 4 + 65535) / 65536);
		if (npb == 0) npb = 1;
		if (!l2bits) {
			hipLaunchKernelGGL(k_init_cursors, dim3((n_ranges_p + 255) / 256), dim3(256), 0, st, range_start, n_ranges_p, 0u, gcur);
			hipLaunchKernelGGL(k_part_items, dim3(npb), dim3(PART_THREADS), PART_LDS_BYTES, st, raw, g_cursor, f, range_shift, n_ranges_p, gcur, items);
		} else {
			// level 1 into `n_coarse` segments (written over a second buffer), level 2 inside each segment
			u64* l1;
			u32* gcur1;
			HIP_TRY(db.alloc(&l1, items_cap));
			HIP_TRY(db.alloc(&gcur1, n_coarse));
			HIP_TRY(hipFuncSetAttribute((const void*) k_part_items2, hipFuncAttributeMaxDynamicSharedMemorySize, PART_LDS_BYTES));
			hipLaunchKernelGGL(k_init_cursors, dim3((n_coarse + 255) / 256), dim3(256), 0, st, range_start, n_coarse, l2bits, gcur1);
			hipLaunchKernelGGL(k_part_items, dim3(npb), dim3(PART_THREADS), PART_LDS_BYTES, st, raw, g_cursor, f, range_shift + l2bits, n_coarse, gcur1, l1);
			hipLaunchKernelGGL(k_init_cursors, dim3((n_ranges_p + 255) / 256), dim3(256), 0, st, range_start, n_ranges_p, 0u, gcur);
			const u32 sl2 = std::max(8u, std::min(64u, 2048u / n_coarse));          // workgroups per coarse segment
			hipLaunchKernelGGL(k_part_items2, dim3(n_coarse * sl2), dim3(PART_THREADS), PART_LDS_BYTES, st, l1, range_start, l2bits, sl2, f, range_shift, l2bits, gcur, items);
		}
	}
	dbg_sync(c, "k_part_items");
	{
		vdjx_prof_scope ps(c, "k_recount");
#define RC_LAUNCH(SB, IT) hipLaunchKernelGGL((k_recount<SB, IT>), dim3(n_ranges), dim3(RC_THREADS), 0, st, items, range_start, f, ns, rec_base, pool->ob, fbw, ro.ucnt, ro.ufirst, ro.in_first, n_inst)
		if (narrow) {
			switch (range_shift) {
				case 8: RC_LAUNCH(256, u32); break;
				case 9: RC_LAUNCH(512, u32); break;
				case 10: RC_LAUNCH(1024, u32); break;
				default: RC_LAUNCH(2048, u32); break;
			}
		} else {
			switch (range_shift) {
				case 8: RC_LAUNCH(256, u64); break;
				case 9: RC_LAUNCH(512, u64); break;
				default: RC_LAUNCH(1024, u64); break;
			}
		}
#undef RC_LAUNCH
	}
	dbg_sync(c, "k_recount");
	if (derive_edges) {
		hipLaunchKernelGGL(k_count_unseen, dim3((ns + 255) / 256), dim3(256), 0, st, ro.ufirst, ns, g_err + 1);
		HIP_TRY(hipMemsetAsync(ro.edge_first, 0xFF, (size_t) ns * 32, st));
		hipLaunchKernelGGL(k_edges_from_in, dim3((ns * 4 + 255) / 256), dim3(256), 0, st, tb, ns, k, ro.in_first, ro.in_from, ro.edge_first, ro.edge_to);
	}
	if (defer && !getenv("VDJX_SYNC_DEBUG")) {         // the caller reads the status when it waits anyway
		defer->g_err = g_err; defer->n_items = range_start + n_ranges_p; defer->n_inst = n_inst; defer->ns = ns;
		return VDJX_OK;
	}
	if (defer) defer->g_err = nullptr;
	u32* hp = (u32*) c->h_pin;                        // err[2], n_items, -, inst (u64), mirrored walk's complaints [2]
	HIP_TRY(hipMemcpyAsync(hp, g_err, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hp + 2, range_start + n_ranges_p, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hp + 4, n_inst, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hp + 6, g_err + 2, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (hp[6] || hp[7]) { vdjx_set_error("the mirrored walk found %u + %u places where the chains are not mirror images", hp[6], hp[7]); return VDJX_ESYMWALK; }
	const u32 err[2] = {hp[0], hp[1]}, n_items = hp[2];
	const unsigned long long inst = *(const unsigned long long*) (hp + 4);
	if (err[0]) { vdjx_set_error("k_walk_items: item buffer too small (%u waves stopped)", err[0]); return VDJX_EHIP; }
	if (err[1]) {
		if (getenv("VDJX_SYNC_DEBUG")) {          // name the k-mers
			std::vector<u64> uf(ns), lo(ns), hi(ns), gf(ns);
			std::vector<u32> gc(ns);
			(void) hipMemcpy(uf.data(), ro.ufirst, (size_t) ns * 8, hipMemcpyDeviceToHost);
			(void) hipMemcpy(lo.data(), sv.lo, (size_t) ns * 8, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hi.data(), sv.hi, (size_t) ns * 8, hipMemcpyDeviceToHost);
			(void) hipMemcpy(gf.data(), sv.gfirst, (size_t) ns * 8, hipMemcpyDeviceToHost);
			(void) hipMemcpy(gc.data(), sv.gcnt, (size_t) ns * 4, hipMemcpyDeviceToHost);
			for (u32 i = 0, shown = 0; i < ns && shown < 8; i++) if (uf[i] == NONE64) {
				char txt[65];
				for (int j = 0; j < k; j++) { const int sh = 2 * (k - 1 - j); const u32 c2 = (u32) (sh < 64 ? lo[i] >> sh : hi[i] >> (sh - 64)) & 3u; txt[j] = "ATCG"[c2]; }
				txt[k] = 0;
				fprintf(stderr, "[vdjx] unseen survivor %u: %s gated count %u first gated instance record %llu offset %llu\n", i, txt, gc[i],
				        (unsigned long long) (gf[i] >> 6), (unsigned long long) (gf[i] & 63));
				shown++;
			}
		}
		vdjx_set_error("internal error: %u of %u surviving k-mers were not met again in the records", err[1], ns);
		return VDJX_EHIP;
	}
	c->stats["recount_items"] = n_items;            // runs of surviving k-mer instances of this pool (8 bytes each)
	c->stats["recount_instances"] = inst;           // the instances themselves
	db.release_to(tmp_mark);                        // (the stream has been waited for: nothing reads the items any more)
	return VDJX_OK;
}

// K6 for the round-2 arrays
template <typename A>
int stage_finish2(vdjx_ctx* c, A& db, const SurvivorsG& sv, const RecountOut& ro, u64 n_records_total, int k, int P, int ob, vdjx_graph* g) {
	hipStream_t st = c->stream;
	const u32 ns = sv.n;
	const u32 nr = sv.n_real <= ns ? sv.n_real : ns;          // the nodes: the survivors that are not shadows (GC_SHADOW; they sort last)
	g->n = nr;
	g->k = k;
	g->ctx = c;
	g->device = c->device;
	if (ns == 0) return VDJX_OK;
	uint8_t *d_hv, *d_hj;
	HIP_TRY(db.alloc(&d_hv, ns));
	HIP_TRY(db.alloc(&d_hj, ns));
	{
		vdjx_prof_scope ps(c, "k_node_flags");
		hipLaunchKernelGGL(k_node_flags, dim3((ns + 255) / 256), dim3(256), 0, st, sv.lo, sv.hi, ns, k, c->d_vbits, c->d_jbits, d_hv, d_hj);
	}
	u64 *rk_key, *rk_key2;
	u32 *rk_idx, *rk_idx2, *rank;
	HIP_TRY(db.alloc(&rk_key, ns));
	HIP_TRY(db.alloc(&rk_key2, ns));
	HIP_TRY(db.alloc(&rk_idx, ns));
	HIP_TRY(db.alloc(&rk_idx2, ns));
	HIP_TRY(db.alloc(&rank, ns));
	// instance ids are below n_records_total << ob = at most 2^rk_bits - 1; NONE64 (an unseen survivor: the build fails anyway), cut to
	// rk_bits bits, is 2^rk_bits - 1 and still sorts behind every id.  (One bit more "to keep NONE64 apart" made 33 bits and a fifth
	// pass of the sort at 10 M pairs.)
	unsigned rk_bits = 1;
	while (rk_bits < 64 && (1ull << rk_bits) <= ((n_records_total << ob) | 1ull)) rk_bits++;
	auto up = [](size_t b) { return (b + 255) & ~(size_t) 255; };
	const size_t need = up((size_t) ns * 8) + 5 * up((size_t) ns * 4) + 4 * up(ns) + 2 * up((size_t) ns * 16) + up((size_t) ns * k);
	{
		hipError_t e = c->blocks.acquire(need, &g->d_block, &g->block_cap);
		if (e != hipSuccess) { vdjx_set_error("graph alloc (%zu bytes): %s", need, hipGetErrorString(e)); return VDJX_EHIP; }
	}
	char* bp = g->d_block;
	auto carve = [&](size_t b) { char* r = bp; bp += up(b); return r; };
	NodeOut no;
	no.klo = no.khi = nullptr;
	no.first_inst = g->d_first_inst = (u64*) carve((size_t) ns * 8);
	no.gcnt = g->d_gcnt = (u32*) carve((size_t) ns * 4);
	no.freq = g->d_freq = (u32*) carve((size_t) ns * 4);
	no.to_ids = g->d_to_ids = (u32*) carve((size_t) ns * 16);
	no.from_ids = g->d_from_ids = (u32*) carve((size_t) ns * 16);
	no.hv = g->d_hv = (uint8_t*) carve(ns);
	no.hj = g->d_hj = (uint8_t*) carve(ns);
	no.to_deg = g->d_to_deg = (uint8_t*) carve(ns);
	no.from_deg = g->d_from_deg = (uint8_t*) carve(ns);
	no.kmers = g->d_kmers = carve((size_t) ns * k);
	g->export_bytes = (size_t) (bp - g->d_block);
	g->d_roots = (u32*) carve((size_t) ns * 4);
	{
		vdjx_prof_scope ps(c, "k_node_order");
		hipLaunchKernelGGL(k_rank_keys, dim3((ns + 255) / 256), dim3(256), 0, st, sv.ufirst, sv.gcnt, ns, rk_key, rk_idx);
		{
			size_t tb = 0;
			int rc_ = vdjx_sort_pairs_raw(nullptr, &tb, st, rk_key, rk_key2, rk_idx, rk_idx2, ns, rk_bits);
			if (rc_) return rc_;
			char* tmp;
			HIP_TRY(db.alloc(&tmp, tb + 256));
			if ((rc_ = vdjx_sort_pairs_raw(tmp, &tb, st, rk_key, rk_key2, rk_idx, rk_idx2, ns, rk_bits))) return rc_;
		}
		hipLaunchKernelGGL(k_rank_scatter, dim3((ns + 255) / 256), dim3(256), 0, st, rk_key2, rk_idx2, ns, rank);
		hipLaunchKernelGGL(k_node_emit2, dim3((ns + 255) / 256), dim3(256), 0, st, sv.lo, sv.hi, sv.gcnt, sv.ucnt, sv.ufirst, d_hv, d_hj, rank,
		                   ro.edge_first, ro.edge_to, ro.in_first, ro.in_from, ns, k, no);
	}
	u32 n_roots = 0;
	{
		const u32 nrb = (nr + 255) / 256;
		u32 *rb_cnt, *rb_start;
		HIP_TRY(db.alloc(&rb_cnt, nrb + 1));
		HIP_TRY(db.alloc(&rb_start, nrb + 1));
		vdjx_prof_scope ps(c, "k_root_list");
		if (nrb) hipLaunchKernelGGL(k_root_count, dim3(nrb), dim3(256), 0, st, no.from_deg, nr, rb_cnt);
		hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, rb_cnt, nrb, rb_start);
		if (nrb) hipLaunchKernelGGL(k_root_list, dim3(nrb), dim3(256), 0, st, no.from_deg, nr, rb_start, g->d_roots);
		HIP_TRY(hipMemcpyAsync(c->h_pin, rb_start + nrb, 4, hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	n_roots = *(const u32*) c->h_pin;
	g->n_roots = n_roots;
	vdjx_prof_collect(c, false);
	return VDJX_OK;
}

template <typename TUP>
int kmer_build_impl2(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, vdjx_graph* g) {
	const int P = pool->rl - k + 1;
	vdjx_work db(c);
	GTuples<TUP> t;
	const bool sym = build_sym(pool, k);
	c->stats["kmer_build_sym"] = sym ? 1 : 0;
	int rc = stage_gated_partition<TUP>(c, db, pool, 0, k, 0, 0, &t, sym);
	if (rc) return rc;
	PoolView pv{pool->d_bases, pool->d_nmask, vdjx_qrows{pool->d_quals, pool->d_quals2, pool->q_split, pool->qstride}, pool->rl, pool->ob};
	SurvivorsG sv;
	rc = stage_gated_reduce<TUP>(c, db, t, pv, 0, k, mf, mq, &sv, sym);
	if (rc) return rc;
	vdjx_ri_open_gate(c);          // phase A is done: a begun read-index build of this context runs beside the graph pass (vdjx_rindex.hip)
	dbg_sync(c, "gated_reduce");
	g->pre_nodes = (size_t) sv.ndist;
	RecountOut ro{};
	if (sv.n) {
		HIP_TRY(db.alloc(&sv.ucnt, sv.n)); HIP_TRY(db.alloc(&sv.ufirst, sv.n));
		HIP_TRY(db.alloc(&ro.in_first, (size_t) sv.n * 4)); HIP_TRY(db.alloc(&ro.in_from, (size_t) sv.n * 4));
		HIP_TRY(db.alloc(&ro.edge_first, (size_t) sv.n * 4)); HIP_TRY(db.alloc(&ro.edge_to, (size_t) sv.n * 4));
		ro.ucnt = sv.ucnt; ro.ufirst = sv.ufirst;
		// phase B: over the couples' first records when the survivor set came out of phase A closed under reverse complement (sym); should
		// the chains turn out not to be mirror images after all (they are, by construction: k_succ_links2) the walk is done again over
		// every record -- a survivor set with shadows is as good an input to that one
		static const bool sym_walk_on = getenv("VDJX_NO_SYM_WALK") == nullptr;
		for (int attempt = 0; attempt < 2; attempt++) {
			const bool sym_walk = sym && sym_walk_on && attempt == 0;
			c->stats["kmer_build_sym_walk"] = sym_walk ? 1 : 0;
			RecountStatus rs;
			rc = stage_recount(c, db, pool, 0, k, sv, ro, true, nullptr, &rs, sym_walk);
			if (rc == VDJX_ESYMWALK && sym_walk) { c->stats["kmer_build_sym_walk_retries"] += 1; continue; }      // (VDJX_SYNC_DEBUG: the stage read its own status)
			if (rc) return rc;
			u32* hp = (u32*) c->h_pin + 1024;              // (behind what stage_finish2 reads back)
			hp[6] = hp[7] = 0;
			if (rs.g_err) {
				// (the instance count and the four status words lie side by side in the recount's cleared block: one transfer, put in order below)
				HIP_TRY(hipMemcpyAsync(hp + 8, rs.n_inst, 24, hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipMemcpyAsync(hp + 2, rs.n_items, 4, hipMemcpyDeviceToHost, c->stream));
			}
			rc = stage_finish2(c, db, sv, ro, pool->n_records, k, P, pool->ob, g);          // (waits for the stream)
			if (rc) return rc;
			if (rs.g_err) { hp[4] = hp[8]; hp[5] = hp[9]; hp[0] = hp[10]; hp[1] = hp[11]; hp[6] = hp[12]; hp[7] = hp[13]; }
			rc = rs.g_err ? recount_status_check(c, hp, rs.ns) : VDJX_OK;
			if (rc != VDJX_ESYMWALK || !sym_walk) return rc == VDJX_ESYMWALK ? VDJX_EHIP : rc;
			c->stats["kmer_build_sym_walk_retries"] += 1;
			if (g->d_block) { c->blocks.release(g->d_block, g->block_cap); g->d_block = nullptr; g->block_cap = 0; }
		}
		return VDJX_EHIP;
	}
	return stage_finish2(c, db, sv, ro, pool->n_records, k, P, pool->ob, g);
}

}  // namespace

extern "C" int vdjx_kmer_build(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, vdjx_graph** out) {
	if (!c || !pool || !out) { vdjx_set_error("vdjx_kmer_build: NULL argument"); return VDJX_EINVAL; }
	*out = nullptr;
	if (pool->ctx != c) { vdjx_set_error("vdjx_kmer_build: pool belongs to another context"); return VDJX_EINVAL; }
	if (k < 1 || k > VDJX_MAX_KMER || k > pool->rl) { vdjx_set_error("k=%d outside [1,min(%d,rl=%d)]", k, VDJX_MAX_KMER, pool->rl); return VDJX_ELIMIT; }
	if (k > 16 && !c->anchors_loaded) { vdjx_set_error("vdjx_kmer_build: call vdjx_anchor_sets_load first (k > 16)"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	vdjx_graph* g = new vdjx_graph();
	const int rc = k <= 45 ? kmer_build_impl2<Tup16>(c, pool, k, mf, mq, g) : kmer_build_impl2<Tup24>(c, pool, k, mf, mq, g);
	if (rc != VDJX_OK) { vdjx_graph_free(g); return rc; }
	*out = g;
	return VDJX_OK;
}

// ==============================================================================================
// Sharded build: one process per GPU; the caller moves the bytes between ranks
// (torch.distributed over RCCL in vdjer_amd/shard.py, RCCL directly in the C host).  Record numbering: rank r's
// records are [r*rec_stride, r*rec_stride + R_r); k-mer ownership = bucket index / buckets per owner (any number of ranks: the
// buckets per owner are the quotient rounded up, to a multiple of 16 when it is not exact; the last owner has the fewest).
// Instance ids are global and 38 bits wide (record << 6 | offset; record << 8 | offset for reads of more than 64 bases):
// nranks * rec_stride < 2^32 (2^30) records.
// ==============================================================================================
struct vdjx_shard {
	vdjx_ctx* c = nullptr;
	const vdjx_pool* pool = nullptr;
	int k = 0, mf = 0, mq = 0, rank = 0, nranks = 1;
	u64 rec_stride = 0;
	// share mode (vdjx_shard_begin_share): the rank's records are a SHARE of the pool in ascending scan order, record i at scan
	// position scan[i]; the kernels work on local record numbers (an ascending renumbering keeps every "first") and what leaves
	// the rank -- a partial aggregate's first instance, the first sights of nodes and in-edges -- goes through scan[] on its way out
	const u32* scan = nullptr;
	u64 total_records = 0;
	u64 rec_base() const { return scan ? 0 : rec_stride * (u64) rank; }
	u64 records_in_all() const { return scan ? total_records : rec_stride * (u64) nranks; }
	// this rank's records are couples (record, its reverse complement) and k is odd: the local phase can move one tuple per pair of
	// mirrored instances (k_gated_local SYM) -- IF every rank does (the buckets are then cut by the canonical k-mer): the ranks tell each
	// other (vdjx_shard_symmetric) and the caller passes the verdict on (vdjx_shard_geometry2)
	bool sym_capable = false, sym = false;
	bool wide = false;                              // Tup24 (k > 45)
	// local phase: this rank's gated tuples by bucket, its partial aggregates (dense, bucket order)
	u32 NBf = 0, NBo = 0;
	u32 *nd = nullptr, *dstart = nullptr;          // partials per bucket, their dense offsets
	Partial* dense = nullptr;
	u32* dense_ref = nullptr;                      // per dense partial: where its instances are listed (count < TLOW), or NONE
	u64* low_inst = nullptr;
	u32 n_dense = 0;
	// owner phase
	SurvivorsG local_sv, all_sv;
	u32 sv_cap = 0;
	u32* n_surv = nullptr;
	PendOut po{};
	u32 n_pend = 0;
	uint2* queries = nullptr;
	std::vector<u32> src_base, nq;
	u32 *p_fl = nullptr, *p_S = nullptr, *p_r0 = nullptr;
	u32 mqq = 0, tlow = 0;
	SurvTable tb{};                                // survivor table of the recount (kept for the edge derivation in finish)
	int phase = 0;
	// vdjx_shard_count: the histogram of this rank's gated instances, kept for vdjx_shard_local; the number the ranks agreed on
	bool have_hist = false;
	GatedHist gh;
	u64 agreed = 0;
	// vdjx_shard_local -> vdjx_shard_local_fill: the partial aggregates by bucket before they are laid end to end
	Partial* sparse = nullptr;
	u32* sparse_ref = nullptr;
	vdjx_arena::mark_t fill_mark{0, 0};
	const u32* tuple_bucket_start = nullptr;
	u32 NBt = 0;
};

static int shard_begin_impl(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, int rank, int nranks,
                            uint64_t rec_stride, const uint32_t* d_scan, uint64_t total_records, vdjx_shard** out) {
	if (!c || !pool || !out) { vdjx_set_error("vdjx_shard_begin: NULL argument"); return VDJX_EINVAL; }
	*out = nullptr;
	if (c->live_shard) { vdjx_set_error("vdjx_shard_begin: a sharded build is already in flight on this context"); return VDJX_ESTATE; }
	if (nranks < 1 || nranks > 256 || rank < 0 || rank >= nranks) { vdjx_set_error("1 <= nranks <= 256, 0 <= rank < nranks"); return VDJX_EINVAL; }
	if (k < 1 || k > VDJX_MAX_KMER || k > pool->rl) { vdjx_set_error("k=%d outside [1,min(%d,rl=%d)]", k, VDJX_MAX_KMER, pool->rl); return VDJX_ELIMIT; }
	if (d_scan) {
		if (total_records < pool->n_records) { vdjx_set_error("total_records %llu < local records %zu", (unsigned long long) total_records, pool->n_records); return VDJX_EINVAL; }
		if (total_records >= (1ull << 32) || total_records >= (1ull << (INST_BITS - pool->ob))) { vdjx_set_error("global record count %llu: scan positions are 32 bits and instance ids record << %d | offset in %d bits", (unsigned long long) total_records, pool->ob, INST_BITS); return VDJX_ELIMIT; }
		rec_stride = (total_records + (uint64_t) nranks - 1) / (uint64_t) nranks;        // (only the geometry bound of a build without vdjx_shard_count reads it)
	} else {
		if (rec_stride < pool->n_records) { vdjx_set_error("rec_stride %llu < local records %zu", (unsigned long long) rec_stride, pool->n_records); return VDJX_EINVAL; }
		if (rec_stride * (uint64_t) nranks >= (1ull << (INST_BITS - pool->ob))) { vdjx_set_error("global record count %llu >= 2^%d (instance ids are record << %d | offset in %d bits)", (unsigned long long) (rec_stride * nranks), INST_BITS - pool->ob, pool->ob, INST_BITS); return VDJX_ELIMIT; }
	}
	if (k > 16 && !c->anchors_loaded) { vdjx_set_error("vdjx_shard_begin: call vdjx_anchor_sets_load first (k > 16)"); return VDJX_ESTATE; }
	if (pool->W > 2 && pool->n_records >= (1ull << 27)) { vdjx_set_error("more than 2^27 records of long reads on one GPU: not supported by the recount items"); return VDJX_ELIMIT; }
	vdjx_shard* s = new vdjx_shard();
	s->c = c; s->pool = pool; s->k = k; s->mf = mf; s->mq = mq; s->rank = rank; s->nranks = nranks;
	s->rec_stride = rec_stride;
	s->scan = d_scan; s->total_records = total_records;
	s->sym_capable = build_sym(pool, k) && s->rec_base() % 2 == 0;
	s->wide = k > 45;
	if (mq >= 255) mq = 254;                                        // A2:1514-1516
	s->mqq = (u32) (mq < 0 ? 0 : (mq > 214 ? 214 : mq));
	s->tlow = 1 + (s->mqq + 19) / 20;
	c->live_shard = s;
	*out = s;
	return VDJX_OK;
}

extern "C" int vdjx_shard_begin(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, int rank, int nranks,
                                uint64_t rec_stride, vdjx_shard** out) {
	return shard_begin_impl(c, pool, k, mf, mq, rank, nranks, rec_stride, nullptr, 0, out);
}

// the same build over a SHARE of the pool: record i of `pool` is record d_scan_index[i] of the scan order (A2:1388-1390) of a pool
// of total_records records; the positions ascend (a share keeps the pool's order) and no two ranks hold the same one.  The array
// lives on the device and must stay valid until vdjx_shard_free.
extern "C" int vdjx_shard_begin_share(vdjx_ctx* c, const vdjx_pool* pool, int k, int mf, int mq, int rank, int nranks,
                                      const uint32_t* d_scan_index, uint64_t total_records, vdjx_shard** out) {
	if (pool && pool->n_records && !d_scan_index) { vdjx_set_error("vdjx_shard_begin_share: NULL scan index"); return VDJX_EINVAL; }
	static const uint32_t none = 0;
	return shard_begin_impl(c, pool, k, mf, mq, rank, nranks, 0, d_scan_index ? d_scan_index : &none, total_records, out);
}

extern "C" void vdjx_shard_free(vdjx_shard* s) {
	if (!s) return;
	(void) hipSetDevice(s->c->device);
	(void) hipStreamSynchronize(s->c->stream);
	s->c->shard_arena.reset();
	s->c->live_shard = nullptr;
	delete s;
}

/* sizes of the records the caller moves between ranks: 0 partial aggregate, 1 question, 2 answer, 3 survivor */
extern "C" size_t vdjx_shard_record_bytes(int kind) {
	return kind == 0 ? sizeof(Partial) : kind == 1 ? sizeof(uint2) : kind == 2 ? REPLY_BYTES : kind == 3 ? sizeof(SurvRec) : 0;
}

// every rank cuts the SAME buckets.  Without vdjx_shard_count the geometry follows a bound all ranks know (a quarter of the stride's
// instances: the gated fraction of real pools lies between a sixth and a half), not the local count
static u64 shard_geometry_bound(const vdjx_shard* s) {
	const int P = s->pool->rl - s->k + 1;
	return std::max<u64>(1, s->rec_stride * (u64) P / 4);
}

template <typename TUP>
static int shard_local_impl(vdjx_shard* s) {
	vdjx_ctx* c = s->c;
	hipStream_t st = c->stream;
	PersistAlloc db(c);
	const u64 rec_base = s->rec_base();
	GTuples<TUP> t;
	int rc = VDJX_OK;
	if (s->have_hist && s->gh.sym != s->sym) s->have_hist = false;          // (counted before the ranks had compared their pools: once more, the other way)
	const size_t per_b = 0;
	if (!s->have_hist) rc = stage_gated_hist(c, db, s->pool, s->k, per_b, shard_geometry_bound(s) / (s->sym ? 2 : 1), &s->gh, s->sym);
	if (rc) return rc;
	s->have_hist = true;
	// (the ranks' largest count if they compared them, vdjx_shard_count + vdjx_shard_geometry: buckets of the size the one-GPU build
	// cuts -- at 10 M pairs per rank half as many as the bound asks for, and one partition level less)
	// what lives until the answers are out first; then the aggregates by bucket (until vdjx_shard_local_fill has laid them end to end);
	// then the tuples (until this function returns): the arena gives the later ones up again in that order
	const size_t cap = (size_t) s->gh.N + 1;
	const size_t nb_max = ((size_t) 1 << 20) + 16 * (size_t) s->nranks + 2;          // (stage_gated_cut: at most 2^(15+5) buckets; the directory's padding)
	Partial* sparse;
	u32 *g_err, *sparse_ref;
	HIP_TRY(db.alloc(&s->low_inst, cap));
	HIP_TRY(db.alloc(&s->dense_ref, cap));
	HIP_TRY(db.alloc(&s->nd, nb_max));
	HIP_TRY(db.alloc(&s->dstart, nb_max + 1));
	HIP_TRY(db.alloc(&g_err, 1));
	u32* keep_starts;                          // the tuples' bucket starts, kept below everything the fill gives up (the compaction reads them)
	HIP_TRY(db.alloc(&keep_starts, nb_max + 1));
	s->fill_mark = db.mark();
	HIP_TRY(db.alloc(&sparse, cap));
	HIP_TRY(db.alloc(&sparse_ref, cap));
	const vdjx_arena::mark_t tuple_mark = db.mark();
	rc = stage_gated_cut<TUP>(c, db, s->pool, rec_base, s->k, per_b, (s->agreed ? s->agreed : shard_geometry_bound(s)) / (s->sym ? 2 : 1), s->gh, &t);
	if (rc) return rc;
	if (t.NB < (u32) s->nranks) { vdjx_set_error("vdjx_shard_local: fewer buckets (%u) than ranks", t.NB); return VDJX_ELIMIT; }
	// buckets per owner; the directory every rank sends has nranks * NBo entries (the ones past the last bucket are empty)
	s->NBo = t.NB / (u32) s->nranks;
	if (s->NBo * (u32) s->nranks != t.NB) s->NBo = (s->NBo + 1 + 15) & ~15u;
	s->NBf = s->NBo * (u32) s->nranks;
	if ((size_t) s->NBf + 1 > nb_max) { vdjx_set_error("vdjx_shard_local: %u buckets", s->NBf); return VDJX_ELIMIT; }
	HIP_TRY(hipMemsetAsync(g_err, 0, 4, st));
	if (s->NBf > t.NB) HIP_TRY(hipMemsetAsync(s->nd + t.NB, 0, (size_t) (s->NBf - t.NB) * 4, st));
	{
		static const u32 sub = (u32) tune("VDJX_SUB_TUPLES", 262144);
		if (c->sub_tuples_set != sub) {            // (the module's variable: the same for every context of the process, set again by each once)
			HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sub_tuples), &sub, 4, 0, hipMemcpyHostToDevice, st));
			c->sub_tuples_set = sub;
		}
	}
	{
		vdjx_prof_scope ps(c, "k_gated_local");
		static const bool by_size = tune("VDJX_RD_ORDER", 1) == 1;
		u32* order = nullptr;
		if (by_size && t.NB >= 1024) {
			HIP_TRY(db.alloc(&order, t.NB));
			hipLaunchKernelGGL(k_bucket_order, dim3(1), dim3(1024), 0, st, t.bucket_start, t.NB, order);
		}
		if (s->sym) hipLaunchKernelGGL((k_gated_local<TUP, true>), dim3(t.NB), dim3(LG_THREADS), 0, st, t.t, t.bucket_start, s->pool->d_bases, s->pool->d_nmask, rec_base,
		                               s->k, s->pool->rl, s->pool->ob, s->tlow, sparse, sparse_ref, s->nd, s->low_inst, g_err, (const u32*) order);
		else hipLaunchKernelGGL(k_gated_local<TUP>, dim3(t.NB), dim3(LG_THREADS), 0, st, t.t, t.bucket_start, s->pool->d_bases, s->pool->d_nmask, rec_base,
		                        s->k, s->pool->rl, s->pool->ob, s->tlow, sparse, sparse_ref, s->nd, s->low_inst, g_err, (const u32*) order);
	}
	hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, s->nd, s->NBf, s->dstart);
	u32* d_pick;
	const u32 G = (u32) s->nranks;
	HIP_TRY(db.alloc(&d_pick, G + 1));
	hipLaunchKernelGGL(k_pick_u32, dim3(1), dim3(512), 0, st, s->dstart, s->NBo, G + 1, d_pick);
	std::vector<u32> pick(G + 1);
	u32 err = 0;
	HIP_TRY(hipMemcpyAsync(pick.data(), d_pick, (G + 1) * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&err, g_err, 4, hipMemcpyDeviceToHost, st));
	// (the bucket starts move below the tuples on the device, in stream order, before the one wait of this phase: round 5 brought them to
	// the host and back with two synchronous copies after it)
	HIP_TRY(hipMemcpyAsync(keep_starts, t.bucket_start, ((size_t) t.NB + 1) * 4, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (err) { vdjx_set_error("k_gated_local: %u buckets could not be split to fit LDS", err); return VDJX_EHIP; }
	s->src_base.assign(pick.begin(), pick.end());        // reused below as "what goes to owner o" until the merge overwrites it
	s->n_dense = pick[G];
	// (the aggregates are laid end to end by vdjx_shard_local_fill, straight into the caller's send buffer)
	s->sparse = sparse; s->sparse_ref = sparse_ref; s->NBt = t.NB;
	s->dense = nullptr;
	// the tuples are done with
	db.release_to(tuple_mark);
	s->tuple_bucket_start = keep_starts;
	return VDJX_OK;
}

// phase 1: this rank's partial aggregates; send_counts[o] = how many go to owner o; *dir_len = buckets per owner
extern "C" int vdjx_shard_local(vdjx_shard* s, uint64_t* send_counts, uint32_t* dir_len) {
	if (!s || !send_counts || !dir_len) { vdjx_set_error("vdjx_shard_local: NULL argument"); return VDJX_EINVAL; }
	if (s->phase != 0) { vdjx_set_error("vdjx_shard_local: already called"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	int rc = s->wide ? shard_local_impl<Tup24>(s) : shard_local_impl<Tup16>(s);
	if (rc) return rc;
	vdjx_ri_open_gate(s->c);       // the local phase A is done: a begun read-index build runs beside the exchange and the graph pass
	for (int g = 0; g < s->nranks; g++) send_counts[g] = s->src_base[g + 1] - s->src_base[g];
	*dir_len = s->NBo;
	s->phase = 1;
	return VDJX_OK;
}

// the bytes of phase 1 into caller buffers: d_dir = partials per bucket (u32 [nranks*dir_len], owner-major),
// d_partials = sum(send_counts) records of vdjx_shard_record_bytes(0), owner-contiguous
extern "C" int vdjx_shard_local_fill(vdjx_shard* s, void* d_dir, void* d_partials) {
	if (!s || s->phase < 1) { vdjx_set_error("vdjx_shard_local_fill: call vdjx_shard_local first"); return VDJX_ESTATE; }
	if (!d_dir || (s->n_dense && !d_partials)) { vdjx_set_error("vdjx_shard_local_fill: NULL buffer"); return VDJX_EINVAL; }
	if (((uintptr_t) d_partials & 15u) != 0) { vdjx_set_error("vdjx_shard_local_fill: d_partials must be 16-byte aligned"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	hipStream_t st = s->c->stream;
	HIP_TRY(hipMemcpyAsync(d_dir, s->nd, (size_t) s->NBf * 4, hipMemcpyDeviceToDevice, st));
	if (s->n_dense) {
		// the aggregates of every bucket end to end, in the caller's buffer -- which from here on IS this rank's list of them: the
		// answers to the owners' questions are looked up in it (vdjx_shard_reply)
		vdjx_prof_scope ps(s->c, "k_compact_partials");
		hipLaunchKernelGGL(k_compact_partials, dim3(s->NBt), dim3(256), 0, st, s->sparse, s->sparse_ref, s->tuple_bucket_start, s->nd, s->dstart, (Partial*) d_partials, s->dense_ref, s->scan, s->pool->ob, s->sym);
	}
	s->dense = (Partial*) d_partials;
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (s->sparse) {                                   // the aggregates by bucket have been laid end to end: their space goes back
		PersistAlloc db(s->c);
		db.release_to(s->fill_mark);
		s->sparse = nullptr; s->sparse_ref = nullptr; s->tuple_bucket_start = nullptr;
	}
	return VDJX_OK;
}

// this rank's gated instances (one pass over its records); the ranks compare these numbers (MAX) and hand the largest to
// vdjx_shard_geometry, so that all cut the buckets the one-GPU build would cut for the largest rank.  Optional.
extern "C" int vdjx_shard_count(vdjx_shard* s, uint64_t* gated_instances) {
	if (!s || !gated_instances) { vdjx_set_error("vdjx_shard_count: NULL argument"); return VDJX_EINVAL; }
	if (s->phase != 0) { vdjx_set_error("vdjx_shard_count: call it before vdjx_shard_local"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	if (!s->have_hist) {
		// (counted the way this rank would like to build: over couples if its pool is made of them; vdjx_shard_geometry2 has the last word)
		PersistAlloc db(s->c);
		const int rc = stage_gated_hist(s->c, db, s->pool, s->k, 0, shard_geometry_bound(s) / (s->sym_capable ? 2 : 1), &s->gh, s->sym_capable);
		if (rc) return rc;
		s->have_hist = true;
	}
	*gated_instances = (u64) s->gh.N * (s->gh.sym ? 2 : 1);
	return VDJX_OK;
}
// 1: this rank's pool is made of couples (record, its reverse complement) and k is odd -- its local phase could move half the tuples.
// Only if ALL ranks can (the buckets are then cut by the smaller of a k-mer and its reverse complement): the caller ANDs the ranks'
// answers and hands the result to vdjx_shard_geometry2.
extern "C" int vdjx_shard_symmetric(const vdjx_shard* s) { return s && s->sym_capable ? 1 : 0; }
extern "C" int vdjx_shard_geometry2(vdjx_shard* s, uint64_t agreed_instances, int all_symmetric) {
	if (!s) { vdjx_set_error("vdjx_shard_geometry2: NULL argument"); return VDJX_EINVAL; }
	if (all_symmetric && !s->sym_capable) { vdjx_set_error("vdjx_shard_geometry2: this rank's pool is not made of couples"); return VDJX_EINVAL; }
	const int rc = vdjx_shard_geometry(s, agreed_instances);
	if (rc) return rc;
	s->sym = all_symmetric != 0;
	s->c->stats["kmer_build_sym"] = s->sym ? 1 : 0;          // (what vdjx_kmer_build reports: the callers' byte models read it)
	return VDJX_OK;
}
extern "C" int vdjx_shard_geometry(vdjx_shard* s, uint64_t agreed_instances) {
	if (!s) { vdjx_set_error("vdjx_shard_geometry: NULL argument"); return VDJX_EINVAL; }
	if (s->phase != 0) { vdjx_set_error("vdjx_shard_geometry: call it before vdjx_shard_local"); return VDJX_ESTATE; }
	if (agreed_instances < (u64) s->gh.N * (s->gh.sym ? 2 : 1)) { vdjx_set_error("vdjx_shard_geometry: %llu is less than this rank's own %llu gated instances", (unsigned long long) agreed_instances, (unsigned long long) s->gh.N * (s->gh.sym ? 2 : 1)); return VDJX_EINVAL; }
	s->agreed = agreed_instances ? agreed_instances : 1;
	return VDJX_OK;
}

template <typename THI>
static int shard_merge_impl(vdjx_shard* s, const u32* d_recv_dir, const Partial* d_recv, const uint64_t* recv_counts, uint64_t* query_counts) {
	vdjx_ctx* c = s->c;
	hipStream_t st = c->stream;
	PersistAlloc db(c);
	const u32 G = (u32) s->nranks, NBo = s->NBo;
	s->src_base.assign(G + 1, 0);
	for (u32 g = 0; g < G; g++) s->src_base[g + 1] = s->src_base[g] + (u32) recv_counts[g];
	const u32 total = s->src_base[G];
	u32 *d_src_base, *seg_off, *g_nq, *g_err, *n_pend;
	u64* g_distinct;
	HIP_TRY(db.alloc(&d_src_base, G + 1));
	HIP_TRY(db.alloc(&seg_off, (size_t) G * (NBo + 1)));
	HIP_TRY(db.alloc(&g_nq, G));
	HIP_TRY(db.alloc(&g_err, 1));
	HIP_TRY(db.alloc(&n_pend, 1));
	HIP_TRY(db.alloc(&s->n_surv, 1));
	HIP_TRY(db.alloc(&g_distinct, 1));
	HIP_TRY(db.alloc(&s->queries, (size_t) total + 1));
	const u32 cap = (s->sym ? 2 : 1) * total + 2;            // (SYM: an aggregate can leave as two survivors)
	s->sv_cap = cap;
	SurvivorsG& v = s->local_sv;
	HIP_TRY(db.alloc(&v.lo, cap)); HIP_TRY(db.alloc(&v.hi, cap)); HIP_TRY(db.alloc(&v.gcnt, cap)); HIP_TRY(db.alloc(&v.gfirst, cap));
	PendOut& po = s->po;
	HIP_TRY(db.alloc(&po.lo, cap)); HIP_TRY(db.alloc(&po.hi, cap)); HIP_TRY(db.alloc(&po.cg, cap)); HIP_TRY(db.alloc(&po.mg, cap));
	HIP_TRY(db.alloc(&po.need, cap));
	po.mgr = nullptr;
	if (s->sym) HIP_TRY(db.alloc(&po.mgr, cap));
	po.n = n_pend; po.cap = cap;
	HIP_TRY(hipMemcpyAsync(d_src_base, s->src_base.data(), (G + 1) * 4, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_seg_offsets, dim3(G), dim3(1024), 0, st, d_recv_dir, d_src_base, G, NBo, seg_off);
	const u32 cmin = (u32) std::max(s->mf, 2);
	SurvOutG so{v.lo, v.hi, v.gcnt, v.gfirst, s->n_surv, cap, nullptr};
	s->nq.assign(G, 0);
	// buckets per merge table: about MERGE_SLOTS*3/8 partials each (the local buckets are sized for tuples, not for distinct k-mers)
	u32 MG = 1;
	while (MG < NBo && NBo % (MG * 2) == 0 && (u64) total * (MG * 2) <= (u64) NBo * (MERGE_SLOTS * 3 / 8)) MG <<= 1;
	u32 np = 0, ns = 0, err = 0;
	u64 ndist = 0;
	for (u32 s_mult = 1;; s_mult *= 4) {
		HIP_TRY(hipMemsetAsync(g_nq, 0, (size_t) G * 4, st));
		HIP_TRY(hipMemsetAsync(g_err, 0, 4, st));
		HIP_TRY(hipMemsetAsync(n_pend, 0, 4, st));
		HIP_TRY(hipMemsetAsync(s->n_surv, 0, 4, st));
		HIP_TRY(hipMemsetAsync(g_distinct, 0, 8, st));
		if (total) {
			vdjx_prof_scope ps(c, "k_bucket_merge");
			if (s->sym) hipLaunchKernelGGL((k_bucket_merge<THI, true>), dim3(NBo / MG), dim3(MERGE_THREADS), 0, st, d_recv, seg_off, MG, G, NBo, d_src_base, s_mult, cmin,
			                               s->tlow, so, po, s->queries, g_nq, g_distinct, g_err, s->k);
			else hipLaunchKernelGGL(k_bucket_merge<THI>, dim3(NBo / MG), dim3(MERGE_THREADS), 0, st, d_recv, seg_off, MG, G, NBo, d_src_base, s_mult, cmin,
			                        s->tlow, so, po, s->queries, g_nq, g_distinct, g_err, s->k);
		}
		HIP_TRY(hipMemcpyAsync(s->nq.data(), g_nq, (size_t) G * 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&np, n_pend, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&ns, s->n_surv, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&err, g_err, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&ndist, g_distinct, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		if (!err) break;
		if (s_mult >= (1u << 16)) { vdjx_set_error("k_bucket_merge: %u buckets could not be split to fit LDS", err); return VDJX_EHIP; }
	}
	s->n_pend = np;
	c->stats["shard_partials_received"] = total;
	c->stats["shard_open_kmers"] = np;
	c->stats["shard_decided_at_merge"] = ns;
	{
		u64 nqs = 0;
		for (u32 q : s->nq) nqs += q;
		c->stats["shard_questions"] = nqs;
	}
	v.n = ns;                    // decided without questions; vdjx_shard_resolve appends the rest
	v.ndist = ndist;
	for (u32 g = 0; g < G; g++) query_counts[g] = s->nq[g];
	const size_t sides = s->sym ? 2 : 1;                     // (SYM: the reverse complements' answering rank and sums behind the k-mers')
	HIP_TRY(db.alloc(&s->p_fl, (size_t) np + 1));
	HIP_TRY(db.alloc(&s->p_r0, sides * np + 1));
	HIP_TRY(db.alloc(&s->p_S, sides * np * 64 + 1));
	if (np) {
		HIP_TRY(hipMemsetAsync(s->p_fl, 0, (size_t) np * 4, st));
		HIP_TRY(hipMemsetAsync(s->p_r0, 0xFF, sides * np * 4, st));
		HIP_TRY(hipMemsetAsync(s->p_S, 0, sides * np * 256, st));
	}
	return VDJX_OK;
}

// phase 2 (owner): what every rank sent for this rank's hash prefix (source-major: d_recv_dir u32 [nranks*dir_len],
// d_recv_partials sum(recv_counts) records) -> decided k-mers + questions; query_counts[r] = questions for rank r
extern "C" int vdjx_shard_merge(vdjx_shard* s, const void* d_recv_dir, const void* d_recv_partials, const uint64_t* recv_counts,
                                uint64_t* query_counts) {
	if (!s || !recv_counts || !query_counts || !d_recv_dir) { vdjx_set_error("vdjx_shard_merge: NULL argument"); return VDJX_EINVAL; }
	if (s->phase != 1) { vdjx_set_error("vdjx_shard_merge: call vdjx_shard_local first (once)"); return VDJX_ESTATE; }
	uint64_t tot = 0;
	for (int g = 0; g < s->nranks; g++) tot += recv_counts[g];
	if (tot >= (1ull << 30)) { vdjx_set_error("vdjx_shard_merge: too many partial aggregates for one owner (%llu): use more ranks", (unsigned long long) tot); return VDJX_ELIMIT; }
	if (tot && !d_recv_partials) { vdjx_set_error("vdjx_shard_merge: NULL buffer"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	int rc = s->wide ? shard_merge_impl<u64>(s, (const u32*) d_recv_dir, (const Partial*) d_recv_partials, recv_counts, query_counts)
	                 : shard_merge_impl<u32>(s, (const u32*) d_recv_dir, (const Partial*) d_recv_partials, recv_counts, query_counts);
	if (rc) return rc;
	s->phase = 2;
	return VDJX_OK;
}

// the questions, grouped by the rank they go to (sum(query_counts) records of vdjx_shard_record_bytes(1))
extern "C" int vdjx_shard_queries(vdjx_shard* s, void* d_out) {
	if (!s || s->phase < 2) { vdjx_set_error("vdjx_shard_queries: call vdjx_shard_merge first"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	hipStream_t st = s->c->stream;
	size_t at = 0;
	for (int g = 0; g < s->nranks; g++) {
		if (!s->nq[g]) continue;
		if (!d_out) { vdjx_set_error("vdjx_shard_queries: NULL buffer"); return VDJX_EINVAL; }
		HIP_TRY(hipMemcpyAsync((uint2*) d_out + at, s->queries + s->src_base[g], (size_t) s->nq[g] * sizeof(uint2), hipMemcpyDeviceToDevice, st));
		at += s->nq[g];
	}
	HIP_TRY(hipStreamSynchronize(st));
	return VDJX_OK;
}

// phase 3 (every rank): answer the questions of the owners; counts[o] = questions from owner o (grouped in that order);
// d_replies = sum(counts) records of vdjx_shard_record_bytes(2), same order
extern "C" int vdjx_shard_reply(vdjx_shard* s, const void* d_queries, const uint64_t* counts, void* d_replies) {
	if (!s || !counts) { vdjx_set_error("vdjx_shard_reply: NULL argument"); return VDJX_EINVAL; }
	if (s->phase < 1 || (s->n_dense && !s->dense)) { vdjx_set_error("vdjx_shard_reply: call vdjx_shard_local and vdjx_shard_local_fill first"); return VDJX_ESTATE; }
	vdjx_ctx* c = s->c;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	const u32 G = (u32) s->nranks;
	std::vector<u32> off(G + 1, 0);
	for (u32 g = 0; g < G; g++) off[g + 1] = off[g] + (u32) counts[g];
	const u32 nq = off[G];
	if (!nq) return VDJX_OK;
	if (!d_queries || !d_replies) { vdjx_set_error("vdjx_shard_reply: NULL buffer"); return VDJX_EINVAL; }
	PersistAlloc db(c);
	u32* d_off;
	HIP_TRY(db.alloc(&d_off, G + 1));
	hipStream_t st = c->stream;
	HIP_TRY(hipMemcpyAsync(d_off, off.data(), (G + 1) * 4, hipMemcpyHostToDevice, st));
	const vdjx_pool* p = s->pool;
	{
		vdjx_prof_scope ps(c, "k_shard_reply");
		if (s->sym) hipLaunchKernelGGL(k_shard_reply<true>, dim3(nq), dim3(64), 0, st, (const uint2*) d_queries, nq, d_off, G, s->dense, s->dense_ref, s->dstart, s->NBo,
		                               s->low_inst, p->d_bases, p->d_nmask, vdjx_qrows{p->d_quals, p->d_quals2, p->q_split, p->qstride}, s->rec_base(), s->scan, (u32) p->n_records, s->k, p->rl, p->ob, (uint8_t*) d_replies);
		else hipLaunchKernelGGL(k_shard_reply<false>, dim3(nq), dim3(64), 0, st, (const uint2*) d_queries, nq, d_off, G, s->dense, s->dense_ref, s->dstart, s->NBo,
		                        s->low_inst, p->d_bases, p->d_nmask, vdjx_qrows{p->d_quals, p->d_quals2, p->q_split, p->qstride}, s->rec_base(), s->scan, (u32) p->n_records, s->k, p->rl, p->ob, (uint8_t*) d_replies);
	}
	HIP_TRY(hipStreamSynchronize(st));          // `off` staging dies with this frame
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

// phase 4 (owner): the answers (grouped by answering rank, in the order the questions were listed) -> this rank's survivors
extern "C" int vdjx_shard_resolve(vdjx_shard* s, const void* d_replies, uint64_t n_replies, uint64_t* n_survivors, uint64_t* n_distinct) {
	if (!s || !n_survivors || !n_distinct) { vdjx_set_error("vdjx_shard_resolve: NULL argument"); return VDJX_EINVAL; }
	if (s->phase != 2) { vdjx_set_error("vdjx_shard_resolve: call vdjx_shard_merge first (once)"); return VDJX_ESTATE; }
	uint64_t expect = 0;
	for (u32 q : s->nq) expect += q;
	if (n_replies != expect) { vdjx_set_error("vdjx_shard_resolve: %llu answers for %llu questions", (unsigned long long) n_replies, (unsigned long long) expect); return VDJX_EINVAL; }
	vdjx_ctx* c = s->c;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	SurvivorsG& v = s->local_sv;
	if (s->n_pend) {
		if (!d_replies) { vdjx_set_error("vdjx_shard_resolve: NULL buffer"); return VDJX_EINVAL; }
		const u32 nr = (u32) n_replies;
		SurvOutG so{v.lo, v.hi, v.gcnt, v.gfirst, s->n_surv, s->sv_cap, nullptr};
		vdjx_prof_scope ps(c, "k_shard_resolve");
		hipLaunchKernelGGL(k_resolve_first, dim3((nr + 255) / 256), dim3(256), 0, st, (const uint8_t*) d_replies, nr, s->po.mg, s->p_r0, (const u64*) s->po.mgr, s->n_pend);
		hipLaunchKernelGGL(k_resolve_add, dim3((nr + 3) / 4), dim3(256), 0, st, (const uint8_t*) d_replies, nr, s->p_r0, s->k, s->p_fl, s->p_S, s->sym ? s->n_pend : 0u);
		if (s->sym) hipLaunchKernelGGL(k_resolve_keep<true>, dim3((s->n_pend + 255) / 256), dim3(256), 0, st, s->po, s->n_pend, s->p_fl, s->p_S, s->k,
		                               (u32) std::max(s->mf, 0), s->mqq, s->tlow, so);
		else hipLaunchKernelGGL(k_resolve_keep<false>, dim3((s->n_pend + 255) / 256), dim3(256), 0, st, s->po, s->n_pend, s->p_fl, s->p_S, s->k,
		                        (u32) std::max(s->mf, 0), s->mqq, s->tlow, so);
	}
	u32 ns = 0;
	HIP_TRY(hipMemcpyAsync(&ns, s->n_surv, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	c->stats["shard_kept_after_answers"] = ns - v.n;
	v.n = ns;
	*n_survivors = ns;
	*n_distinct = v.ndist;
	s->phase = 3;
	return VDJX_OK;
}

// this rank's survivors as 32-byte records {key_lo, key_hi, gated count, -, gated first (u64)}
extern "C" int vdjx_shard_survivors(vdjx_shard* s, void* d_out) {
	if (!s || s->phase < 3) { vdjx_set_error("vdjx_shard_survivors: call vdjx_shard_resolve first"); return VDJX_ESTATE; }
	const SurvivorsG& v = s->local_sv;
	if (!v.n) return VDJX_OK;
	if (!d_out) { vdjx_set_error("vdjx_shard_survivors: NULL buffer"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(s->c->device));
	vdjx_clear_errors();
	hipLaunchKernelGGL(k_surv_pack, dim3((v.n + 255) / 256), dim3(256), 0, s->c->stream, v.lo, v.hi, v.gcnt, v.gfirst, v.n, (SurvRec*) d_out);
	HIP_TRY(hipStreamSynchronize(s->c->stream));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

// every rank: all survivors (rank order) -> this rank's share of add_to_graph's bookkeeping over ITS records (A2:261-320):
// d_in_first u64 [ns_total*4] (first sight of the in-edge (v, first base of u), global instance id, all-ones = none),
// d_ufirst u64 [ns_total] (first sight of the node), d_ucnt u32 [ns_total] (instances); the caller reduces over ranks:
// MIN (unsigned order) for d_in_first and d_ufirst, SUM for d_ucnt
extern "C" int vdjx_shard_edges(vdjx_shard* s, const void* d_surv_all, uint64_t ns_total, void* d_in_first, void* d_ucnt, void* d_ufirst) {
	if (!s || s->phase < 3) { vdjx_set_error("vdjx_shard_edges: call vdjx_shard_resolve first"); return VDJX_ESTATE; }
	if (ns_total >= (1ull << 26)) { vdjx_set_error("vdjx_shard_edges: too many survivors"); return VDJX_ELIMIT; }
	vdjx_ctx* c = s->c;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	PersistAlloc db(c);
	SurvivorsG& a = s->all_sv;
	a.n = (u32) ns_total;
	a.n_real = a.n;
	s->phase = 4;
	if (!ns_total) return VDJX_OK;
	if (!d_surv_all || !d_in_first || !d_ucnt || !d_ufirst) { vdjx_set_error("vdjx_shard_edges: NULL buffer"); return VDJX_EINVAL; }
	HIP_TRY(db.alloc(&a.lo, a.n)); HIP_TRY(db.alloc(&a.hi, a.n)); HIP_TRY(db.alloc(&a.gcnt, a.n)); HIP_TRY(db.alloc(&a.gfirst, a.n));
	u32* d_nreal;
	HIP_TRY(db.alloc(&d_nreal, 1));
	HIP_TRY(hipMemsetAsync(d_nreal, 0, 4, c->stream));
	hipLaunchKernelGGL(k_surv_unpack, dim3((a.n + 255) / 256), dim3(256), 0, c->stream, (const SurvRec*) d_surv_all, a.n, a.lo, a.hi, a.gcnt, a.gfirst, d_nreal);
	u32* h_nreal = (u32*) c->h_pin + 1536;                 // (stage_recount reads its own status into the first words)
	HIP_TRY(hipMemcpyAsync(h_nreal, d_nreal, 4, hipMemcpyDeviceToHost, c->stream));
	RecountOut ro{(u32*) d_ucnt, (u64*) d_ufirst, (u64*) d_in_first, nullptr, nullptr, nullptr};
	// the survivors of a build over couples came out of the merge closed under reverse complement (shadows included, k_resolve_keep):
	// the walk takes the couples' first records only, as in the one-GPU build.  Should it find chains that are not mirror images (it
	// cannot, by construction), this rank walks every record instead -- from the SAME survivor list, so that its numbering stays the
	// one the other ranks use
	static const bool sym_walk_on = getenv("VDJX_NO_SYM_WALK") == nullptr;
	const bool sym_walk = s->sym && sym_walk_on;
	c->stats["kmer_build_sym_walk"] = sym_walk ? 1 : 0;
	const SurvivorsG a_in = a;
	int rc = stage_recount(c, db, s->pool, s->rec_base(), s->k, a, ro, false, &s->tb, nullptr, sym_walk);
	if (rc == VDJX_ESYMWALK && sym_walk) {
		c->stats["kmer_build_sym_walk_retries"] += 1;
		a = a_in;
		rc = stage_recount(c, db, s->pool, s->rec_base(), s->k, a, ro, false, &s->tb, nullptr, false);
	}
	if (rc == VDJX_OK) a.n_real = *h_nreal;              // (the stage has waited for the stream)
	if (rc || !s->scan) return rc == VDJX_ESYMWALK ? VDJX_EHIP : rc;
	// share mode: the first sights are local instance ids; on their way to the reduction over ranks they become scan positions
	hipLaunchKernelGGL(k_first_to_scan, dim3((a.n * 5 + 255) / 256), dim3(256), 0, c->stream, (u64*) d_in_first, (size_t) a.n * 4, (u64*) d_ufirst, (size_t) a.n, s->scan, s->pool->ob);
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

// every rank: the reduced arrays -> the graph (identical on all ranks)
extern "C" int vdjx_shard_finish(vdjx_shard* s, const void* d_in_first, const void* d_ucnt, const void* d_ufirst,
                                 uint64_t pre_nodes_total, vdjx_graph** out) {
	if (!s || !out || s->phase < 4) { vdjx_set_error("vdjx_shard_finish: call vdjx_shard_edges first"); return VDJX_ESTATE; }
	*out = nullptr;
	vdjx_ctx* c = s->c;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	PersistAlloc db(c);
	const int P = s->pool->rl - s->k + 1;
	SurvivorsG& a = s->all_sv;
	RecountOut ro{};
	if (a.n) {
		if (!d_ucnt || !d_ufirst || !d_in_first) { vdjx_set_error("vdjx_shard_finish: NULL buffer"); return VDJX_EINVAL; }
		a.ucnt = (u32*) d_ucnt;
		a.ufirst = (u64*) d_ufirst;
		ro.ucnt = a.ucnt; ro.ufirst = a.ufirst; ro.in_first = (u64*) d_in_first;
		HIP_TRY(db.alloc(&ro.in_from, (size_t) a.n * 4));
		HIP_TRY(db.alloc(&ro.edge_first, (size_t) a.n * 4));
		HIP_TRY(db.alloc(&ro.edge_to, (size_t) a.n * 4));
		HIP_TRY(hipMemsetAsync(ro.edge_first, 0xFF, (size_t) a.n * 32, c->stream));
		hipLaunchKernelGGL(k_edges_from_in, dim3((a.n * 4 + 255) / 256), dim3(256), 0, c->stream, s->tb, a.n, s->k, ro.in_first, ro.in_from, ro.edge_first, ro.edge_to);
	}
	vdjx_graph* g = new vdjx_graph();
	g->pre_nodes = (size_t) pre_nodes_total;
	int rc = stage_finish2(c, db, a, ro, s->records_in_all(), s->k, P, s->pool->ob, g);
	if (rc) { vdjx_graph_free(g); return rc; }
	*out = g;
	return VDJX_OK;
}

extern "C" size_t vdjx_graph_nodes(const vdjx_graph* g) { return g ? g->n : 0; }
extern "C" size_t vdjx_graph_pre_nodes(const vdjx_graph* g) { return g ? g->pre_nodes : 0; }
extern "C" size_t vdjx_graph_roots(const vdjx_graph* g) { return g ? g->n_roots : 0; }

static int graph_export_on(const vdjx_graph* g, hipStream_t st, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq, uint8_t* has_v,
                           uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids, uint8_t* from_deg, uint32_t* from_ids, char* kmers) {
	const size_t n = g->n;
	// straight into the caller's arrays: DMA speed when they are pinned (vdjx_host_alloc), staged by the runtime otherwise
	if (first_inst) HIP_TRY(hipMemcpyAsync(first_inst, g->d_first_inst, n * 8, hipMemcpyDeviceToHost, st));
	if (gated_count) HIP_TRY(hipMemcpyAsync(gated_count, g->d_gcnt, n * 4, hipMemcpyDeviceToHost, st));
	if (freq) HIP_TRY(hipMemcpyAsync(freq, g->d_freq, n * 4, hipMemcpyDeviceToHost, st));
	if (has_v) HIP_TRY(hipMemcpyAsync(has_v, g->d_hv, n, hipMemcpyDeviceToHost, st));
	if (has_j) HIP_TRY(hipMemcpyAsync(has_j, g->d_hj, n, hipMemcpyDeviceToHost, st));
	if (to_deg) HIP_TRY(hipMemcpyAsync(to_deg, g->d_to_deg, n, hipMemcpyDeviceToHost, st));
	if (from_deg) HIP_TRY(hipMemcpyAsync(from_deg, g->d_from_deg, n, hipMemcpyDeviceToHost, st));
	if (to_ids) HIP_TRY(hipMemcpyAsync(to_ids, g->d_to_ids, n * 16, hipMemcpyDeviceToHost, st));
	if (from_ids) HIP_TRY(hipMemcpyAsync(from_ids, g->d_from_ids, n * 16, hipMemcpyDeviceToHost, st));
	if (kmers) HIP_TRY(hipMemcpyAsync(kmers, g->d_kmers, n * (size_t) g->k, hipMemcpyDeviceToHost, st));
	return VDJX_OK;
}

extern "C" int vdjx_graph_export(const vdjx_graph* g, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq,
                                 uint8_t* has_v, uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids,
                                 uint8_t* from_deg, uint32_t* from_ids, char* kmers) {
	if (!g) { vdjx_set_error("vdjx_graph_export: NULL graph"); return VDJX_EINVAL; }
	if (g->n == 0) return VDJX_OK;
	if (!g->d_block || !vdjx_ctx_alive(g->ctx)) { vdjx_set_error("vdjx_graph_export: the graph's context is gone"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(g->device));
	int rc = graph_export_on(g, g->ctx->stream, first_inst, gated_count, freq, has_v, has_j, to_deg, to_ids, from_deg, from_ids, kmers);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(g->ctx->stream));
	return VDJX_OK;
}

// the same copies on the context's copy stream: they run beside whatever the caller does next with the context (the root
// scorer works on the device-resident graph while its host copy is still in flight); vdjx_graph_export_end waits for them
extern "C" int vdjx_graph_export_begin(const vdjx_graph* g, uint64_t* first_inst, uint32_t* gated_count, uint32_t* freq,
                                       uint8_t* has_v, uint8_t* has_j, uint8_t* to_deg, uint32_t* to_ids,
                                       uint8_t* from_deg, uint32_t* from_ids, char* kmers) {
	if (!g) { vdjx_set_error("vdjx_graph_export_begin: NULL graph"); return VDJX_EINVAL; }
	if (g->n == 0) return VDJX_OK;
	if (!g->d_block || !vdjx_ctx_alive(g->ctx)) { vdjx_set_error("vdjx_graph_export_begin: the graph's context is gone"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(g->device));
	// the build that made `g` has synchronised the main stream before it returned: the arrays are final
	return graph_export_on(g, g->ctx->copy_stream, first_inst, gated_count, freq, has_v, has_j, to_deg, to_ids, from_deg, from_ids, kmers);
}

// The ten arrays of vdjx_graph_export lie in ONE device block: their byte offsets in it (first_inst, gated_count, freq, has_v, has_j,
// to_deg, to_ids, from_deg, from_ids, kmers -- each with vdjx_graph_nodes entries of its type, rows of 4 ids / k letters) and the bytes
// that hold them all.  vdjx_graph_export_block[_begin] copies those bytes in ONE transfer: ten transfers cost a small pool's step a
// tenth of its time in calls alone (0.17 ms of 2.3 at 1 M pairs).
extern "C" int vdjx_graph_block_layout(const vdjx_graph* g, uint64_t* offsets, uint64_t* bytes) {
	if (!g || !offsets || !bytes) { vdjx_set_error("vdjx_graph_block_layout: NULL argument"); return VDJX_EINVAL; }
	const char* b = g->d_block;
	const void* at[10] = {g->d_first_inst, g->d_gcnt, g->d_freq, g->d_hv, g->d_hj, g->d_to_deg, g->d_to_ids, g->d_from_deg, g->d_from_ids, g->d_kmers};
	for (int i = 0; i < 10; i++) offsets[i] = g->n ? (uint64_t) ((const char*) at[i] - b) : 0;
	*bytes = g->n ? g->export_bytes : 0;
	return VDJX_OK;
}
static int graph_block_on(const vdjx_graph* g, bool copy_stream, void* host_block, const char* who) {
	if (!g) { vdjx_set_error("%s: NULL graph", who); return VDJX_EINVAL; }
	if (g->n == 0) return VDJX_OK;
	if (!host_block) { vdjx_set_error("%s: NULL buffer", who); return VDJX_EINVAL; }
	if (!g->d_block || !vdjx_ctx_alive(g->ctx)) { vdjx_set_error("%s: the graph's context is gone", who); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(g->device));
	// (round 6 tried a copy kernel of its own with a small footprint -- 32 to 4,096 workgroups storing 16 bytes per lane into the page-locked
	// block -- instead of the runtime's blit: the same 1.77 ms for the 128 MB of a 10 M-pair graph, but the step went 10.9 -> 15.3-16.8 ms:
	// the kernels that ran beside it, k_pool_pack above all, slowed down three- to fourfold, which beside the runtime's copy they do not)
	HIP_TRY(hipMemcpyAsync(host_block, g->d_block, g->export_bytes, hipMemcpyDeviceToHost, copy_stream ? g->ctx->copy_stream : g->ctx->stream));
	return VDJX_OK;
}
extern "C" int vdjx_graph_export_block(const vdjx_graph* g, void* host_block) {
	const int rc = graph_block_on(g, false, host_block, "vdjx_graph_export_block");
	if (rc || !g || g->n == 0) return rc;
	HIP_TRY(hipStreamSynchronize(g->ctx->stream));
	return VDJX_OK;
}
// (on the copy stream, like vdjx_graph_export_begin; vdjx_graph_export_end waits)
extern "C" int vdjx_graph_export_block_begin(const vdjx_graph* g, void* host_block) {
	return graph_block_on(g, true, host_block, "vdjx_graph_export_block_begin");
}

extern "C" int vdjx_graph_export_end(const vdjx_graph* g) {
	if (!g) { vdjx_set_error("vdjx_graph_export_end: NULL graph"); return VDJX_EINVAL; }
	if (!vdjx_ctx_alive(g->ctx)) { vdjx_set_error("vdjx_graph_export_end: the graph's context is gone"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(g->device));
	HIP_TRY(hipStreamSynchronize(g->ctx->copy_stream));
	return VDJX_OK;
}

extern "C" void vdjx_graph_free(vdjx_graph* g) {
	if (!g) return;
	if (g->d_block) {
		(void) hipSetDevice(g->device);
		if (vdjx_ctx_alive(g->ctx)) {
			(void) hipStreamSynchronize(g->ctx->copy_stream);      // an export that was begun and never ended
			if (g->ctx->root_pending && g->ctx->root_pending_g == g && g->ctx->root_stream) (void) hipStreamSynchronize(g->ctx->root_stream);      // a root scoring begun on this graph and never ended
			g->ctx->blocks.release(g->d_block, g->block_cap);
		}
		else (void) hipFree(g->d_block);
	}
	delete g;
}

