// vdjx_anchor.hip -- V/J anchor membership.
//
// The reference keeps two dense_hash_set<unsigned long> of 16-mer codes (vj_filter.c:53-78) loaded
// from v_index/j_index rows "<code>\t<dist>" with dist <= --am (vj_filter.c:56-68).  Real indices
// hold every 16-mer within Hamming distance 5 of an anchor (seq_dist.c:49-71): 10^7-10^8 codes.
// On a 288 GB part the exact, hash-free representation is affordable: one 2^32-bit bitmap per set
// (512 MiB each), a probe is one 4-byte load.  Code 0 (poly-A) is the sets' empty key and can never
// be a member (vj_filter.c:317-318).
#include "vdjx_common.h"

__global__ void k_bitmap_set(const u32* __restrict__ codes, size_t n, u32* __restrict__ bits) {
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	u32 c = codes[i];
	if (c) atomicOr(&bits[c >> 5], 1u << (c & 31));
}

// seq_to_int over every offset of a contig + matches_vmer/jmer (vj_filter.c:221-238)
__global__ void k_anchor_probe(const char* __restrict__ contig, int len, const u32* __restrict__ vbits,
                               const u32* __restrict__ jbits, uint8_t* __restrict__ out_v, uint8_t* __restrict__ out_j,
                               u32* __restrict__ bad) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= len - 16) return;
	u32 code = 0;
	bool ok = true;
	for (int j = 0; j < 16; j++) {
		u32 b;
		switch (contig[i + j]) {
		case 'A': b = 0; break;
		case 'T': b = 1; break;
		case 'C': b = 2; break;
		case 'G': b = 3; break;
		default: b = 0; ok = false; break;
		}
		code = (code << 2) | b;
	}
	if (!ok) { atomicAdd(bad, 1u); out_v[i] = 0; out_j[i] = 0; return; }   // the reference exits (seq_to_kmer.c:22-24)
	out_v[i] = code ? (vbits[code >> 5] >> (code & 31)) & 1u : 0;
	out_j[i] = code ? (jbits[code >> 5] >> (code & 31)) & 1u : 0;
}

// the set's codes go up into a buffer the context keeps (no hipMalloc / hipFree -- the latter waits for the whole device -- per load: a
// --config4 step loads a new chain's sets with every pool), behind the clearing of its bitmap on the same stream; the caller waits once
static int load_set(vdjx_ctx* c, const u32* codes, size_t n, u32** d_bits, size_t tmp_at) {
	const size_t words = (size_t) 1 << 27;
	if (!*d_bits) HIP_TRY(hipMalloc(d_bits, words * 4));
	HIP_TRY(hipMemsetAsync(*d_bits, 0, words * 4, c->stream));
	if (n) {
		u32* d_codes = c->d_anchor_tmp + tmp_at;
		HIP_TRY(hipMemcpyAsync(d_codes, codes, n * 4, hipMemcpyHostToDevice, c->stream));
		hipLaunchKernelGGL(k_bitmap_set, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, d_codes, n, *d_bits);
	}
	return VDJX_OK;
}

extern "C" int vdjx_anchor_sets_load(vdjx_ctx* c, const uint32_t* v_codes, size_t nv, const uint32_t* j_codes, size_t nj) {
	if (!c || (nv && !v_codes) || (nj && !j_codes)) { vdjx_set_error("vdjx_anchor_sets_load: NULL argument"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	const size_t need = (nv + nj + 2) * 4;
	if (need > c->anchor_tmp_cap) {
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (c->d_anchor_tmp) (void) hipFree(c->d_anchor_tmp);
		c->d_anchor_tmp = nullptr; c->anchor_tmp_cap = 0;
		HIP_TRY(hipMalloc(&c->d_anchor_tmp, need + need / 4));
		c->anchor_tmp_cap = need + need / 4;
	}
	int rc = load_set(c, v_codes, nv, &c->d_vbits, 0);
	if (rc) return rc;
	rc = load_set(c, j_codes, nj, &c->d_jbits, nv + 1);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));            // (the codes have left the caller's arrays)
	HIP_TRY(hipGetLastError());
	c->anchors_loaded = true;
	return VDJX_OK;
}

extern "C" int vdjx_anchor_probe(vdjx_ctx* c, const char* contig, int len, uint8_t* out_v, uint8_t* out_j) {
	if (!c || !contig || !out_v || !out_j) { vdjx_set_error("vdjx_anchor_probe: NULL argument"); return VDJX_EINVAL; }
	if (!c->anchors_loaded) { vdjx_set_error("vdjx_anchor_probe: anchor sets not loaded"); return VDJX_ESTATE; }
	int n = len - 16;
	if (n <= 0) return VDJX_OK;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	char* d_c = nullptr;
	uint8_t* d_o = nullptr;
	u32* d_bad = nullptr;
	HIP_TRY(hipMalloc(&d_c, (size_t) len));
	hipError_t e = hipMalloc(&d_o, (size_t) 2 * n + 16);
	if (e != hipSuccess) { (void) hipFree(d_c); vdjx_set_error("alloc: %s", hipGetErrorString(e)); return VDJX_EHIP; }
	d_bad = (u32*) (d_o + (((size_t) 2 * n + 3) & ~(size_t) 3));
	u32 bad = 0;
	e = hipMemcpyAsync(d_c, contig, (size_t) len, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, c->stream);
	if (e == hipSuccess) {
		vdjx_prof_scope ps(c, "k_anchor_probe");
		hipLaunchKernelGGL(k_anchor_probe, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_c, len, c->d_vbits, c->d_jbits, d_o, d_o + n, d_bad);
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out_v, d_o, (size_t) n, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(out_j, d_o + n, (size_t) n, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	(void) hipFree(d_c);
	(void) hipFree(d_o);
	if (e != hipSuccess) { vdjx_set_error("anchor probe: %s", hipGetErrorString(e)); return VDJX_EHIP; }
	if (bad) { vdjx_set_error("vdjx_anchor_probe: contig holds %u non-ACGT 16-mers (the reference exits here, seq_to_kmer.c:22-24)", bad); return VDJX_EINVAL; }
	return VDJX_OK;
}

// ==============================================================================================
// f-3: the v_index / j_index generator (seq_dist.c) and its shortcut
// ==============================================================================================
// process_kmers (seq_dist.c:49-71) walks every code of a range and takes the minimum base distance (edit_dist,
// seq_dist.c:11-27: differing 2-bit groups) over all anchors: range x anchors distance evaluations, embarrassingly parallel.
// One thread takes four consecutive codes, the anchors are read from LDS (every lane the same address).
#define IDX_THREADS 256
#define IDX_PER 4
#define IDX_BLOCK (IDX_THREADS * IDX_PER)
#define IDX_LDS_ANCHORS 4096

__device__ inline u32 base_dist(u32 a, u32 b) {
	const u32 x = a ^ b;
	return (u32) __popc((x | (x >> 1)) & 0x55555555u);
}

__global__ __launch_bounds__(IDX_THREADS) void k_index_dist(const u32* __restrict__ anchors, u32 n_anchors, u64 start, u64 count, u32 max_dist,
                                                            u32* __restrict__ dist4, u32* __restrict__ block_cnt) {
	__shared__ u32 s_a[IDX_LDS_ANCHORS];
	__shared__ u32 s_cnt;
	const u64 first = (u64) blockIdx.x * IDX_BLOCK + (u64) threadIdx.x * IDX_PER;
	u32 best[IDX_PER];
#pragma unroll
	for (int j = 0; j < IDX_PER; j++) best[j] = 17;                   // SEQ_LEN + 1 (seq_dist.c:57)
	if (threadIdx.x == 0) s_cnt = 0;
	for (u32 a0 = 0; a0 < n_anchors; a0 += IDX_LDS_ANCHORS) {
		const u32 m = n_anchors - a0 < IDX_LDS_ANCHORS ? n_anchors - a0 : IDX_LDS_ANCHORS;
		__syncthreads();
		for (u32 i = threadIdx.x; i < m; i += IDX_THREADS) s_a[i] = anchors[a0 + i];
		__syncthreads();
		const u32 c0 = (u32) (start + first);
		for (u32 i = 0; i < m; i++) {
			const u32 a = s_a[i];
#pragma unroll
			for (int j = 0; j < IDX_PER; j++) {
				const u32 d = base_dist(c0 + (u32) j, a);
				best[j] = d < best[j] ? d : best[j];
			}
		}
	}
	u32 packed = 0, mine = 0;
#pragma unroll
	for (int j = 0; j < IDX_PER; j++) {
		const bool in = first + j < count;
		const u32 d = in ? best[j] : 255u;
		packed |= (d & 0xFFu) << (8 * j);
		mine += in && d <= max_dist;
	}
	if (first < count) dist4[first / IDX_PER] = packed;
	if (mine) atomicAdd(&s_cnt, mine);
	__syncthreads();
	if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(1024) void k_index_scan(const u32* __restrict__ cnt, u32 n, u32* __restrict__ start_out) {
	__shared__ u32 part[1024];
	const u32 per = (n + 1023) / 1024;
	const u32 lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
	u32 s = 0;
	for (u32 i = lo; i < hi; i++) s += cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		const u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for (u32 i = lo; i < hi; i++) { start_out[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) start_out[n] = part[1023];
}

// rows in ascending code order (the order process_kmers prints them)
__global__ __launch_bounds__(IDX_THREADS) void k_index_emit(const u32* __restrict__ dist4, u64 start, u64 count, u32 max_dist,
                                                            const u32* __restrict__ block_start, u32* __restrict__ codes,
                                                            uint8_t* __restrict__ dists) {
	__shared__ u32 s_pre[IDX_THREADS];
	const u64 first = (u64) blockIdx.x * IDX_BLOCK + (u64) threadIdx.x * IDX_PER;
	const u32 packed = first < count ? dist4[first / IDX_PER] : 0xFFFFFFFFu;
	u32 mine = 0;
#pragma unroll
	for (int j = 0; j < IDX_PER; j++) mine += ((packed >> (8 * j)) & 0xFFu) <= max_dist;
	s_pre[threadIdx.x] = mine;
	__syncthreads();
	for (u32 d = 1; d < IDX_THREADS; d <<= 1) {
		const u32 v = threadIdx.x >= d ? s_pre[threadIdx.x - d] : 0;
		__syncthreads();
		s_pre[threadIdx.x] += v;
		__syncthreads();
	}
	u32 at = block_start[blockIdx.x] + s_pre[threadIdx.x] - mine;
#pragma unroll
	for (int j = 0; j < IDX_PER; j++) {
		const u32 d = (packed >> (8 * j)) & 0xFFu;
		if (d <= max_dist) { codes[at] = (u32) (start + first + j); dists[at] = (uint8_t) d; at++; }
	}
}

extern "C" int vdjx_index_generate(vdjx_ctx* c, const uint32_t* anchors, size_t n_anchors, uint64_t start, uint64_t end, int max_dist,
                                   uint64_t cap, uint64_t* n_rows, uint32_t* codes, uint8_t* dists) {
	if (!c || !n_rows || (n_anchors && !anchors)) { vdjx_set_error("vdjx_index_generate: NULL argument"); return VDJX_EINVAL; }
	if (cap && (!codes || !dists)) { vdjx_set_error("vdjx_index_generate: NULL result array"); return VDJX_EINVAL; }
	if (n_anchors >= (1ull << 31)) { vdjx_set_error("vdjx_index_generate: too many anchors"); return VDJX_ELIMIT; }
	*n_rows = 0;
	if (end > 0xFFFFFFFFull) end = 0xFFFFFFFFull;       // codes are 16 bases; the reference's loop simply runs on (unsigned long)
	if (start > end) return VDJX_OK;
	if (max_dist < 0) return VDJX_OK;                   // min_dist <= MAX_DIST never holds
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	const u64 CH = 1ull << 24;
	const u32 nblk_max = (u32) (CH / IDX_BLOCK);
	u32 *d_anchors, *d_dist4, *d_bcnt, *d_bstart, *d_codes;
	uint8_t* d_dists;
	HIP_TRY(db.alloc(&d_anchors, n_anchors));
	HIP_TRY(db.alloc(&d_dist4, CH / IDX_PER));
	HIP_TRY(db.alloc(&d_bcnt, nblk_max));
	HIP_TRY(db.alloc(&d_bstart, nblk_max + 1));
	HIP_TRY(db.alloc(&d_codes, CH));
	HIP_TRY(db.alloc(&d_dists, CH));
	if (n_anchors) HIP_TRY(hipMemcpyAsync(d_anchors, anchors, n_anchors * 4, hipMemcpyHostToDevice, st));
	u64 total = 0;
	const u32 md = (u32) max_dist;
	for (u64 s0 = start; s0 <= end; s0 += CH) {
		const u64 count = end - s0 + 1 < CH ? end - s0 + 1 : CH;
		const u32 nblk = (u32) ((count + IDX_BLOCK - 1) / IDX_BLOCK);
		{
			vdjx_prof_scope ps(c, "k_index_dist");
			hipLaunchKernelGGL(k_index_dist, dim3(nblk), dim3(IDX_THREADS), 0, st, d_anchors, (u32) n_anchors, s0, count, md, d_dist4, d_bcnt);
		}
		hipLaunchKernelGGL(k_index_scan, dim3(1), dim3(1024), 0, st, d_bcnt, nblk, d_bstart);
		u32 rows = 0;
		HIP_TRY(hipMemcpyAsync(&rows, d_bstart + nblk, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (rows && total < cap) {
			hipLaunchKernelGGL(k_index_emit, dim3(nblk), dim3(IDX_THREADS), 0, st, d_dist4, s0, count, md, d_bstart, d_codes, d_dists);
			const u64 take = cap - total < rows ? cap - total : rows;
			HIP_TRY(hipMemcpyAsync(codes + total, d_codes, take * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(dists + total, d_dists, take, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
		}
		total += rows;
		if (s0 + CH < s0) break;                        // (cannot wrap: end <= 2^32-1)
	}
	HIP_TRY(hipGetLastError());
	*n_rows = total;
	return VDJX_OK;
}

// The membership sets straight from the anchors: {code : min distance <= am} is the union of the Hamming balls of
// radius am around the anchors: sum_d C(16,d) 3^d codes per anchor (1.2 M at am = 5), enumerated directly: one thread per
// (anchor, ball member) unranks "which d positions, which of the 3 other bases at each" and sets one bit.
__constant__ u32 c_binom[17][7];

__global__ void k_ball_bits(const u32* __restrict__ anchors, u32 n_anchors, u32 ball, u32 thr, const u32* __restrict__ level_start,
                            u32* __restrict__ bits) {
	const u64 idx = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (u64) n_anchors * ball) return;
	const u32 a = (u32) (idx / ball);
	u32 r = (u32) (idx % ball);
	u32 d = 0;
	while (d < thr && r >= level_start[d + 1]) d++;
	r -= level_start[d];
	u32 p3 = 1;
	for (u32 i = 0; i < d; i++) p3 *= 3;
	u32 sub = r % p3, comb = r / p3;
	u32 code = anchors[a];
	u32 left = d;
	for (u32 p = 0; p < 16 && left; p++) {
		const u32 with_p = c_binom[15 - p][left - 1];           // subsets that take position p
		if (comb < with_p) {
			code ^= (sub % 3 + 1) << (2 * (15 - p));             // one of the three other bases
			sub /= 3;
			left--;
		} else
			comb -= with_p;
	}
	if (code) atomicOr(&bits[code >> 5], 1u << (code & 31));      // code 0 is the sets' empty key (vj_filter.c:317-318)
}

static int ball_set(vdjx_ctx* c, const u32* anchors, size_t n, u32 thr, u32** d_bits) {
	const size_t words = (size_t) 1 << 27;
	if (!*d_bits) HIP_TRY(hipMalloc(d_bits, words * 4));
	HIP_TRY(hipMemsetAsync(*d_bits, 0, words * 4, c->stream));
	if (!n) return VDJX_OK;
	u32 binom[17][7] = {};
	for (int i = 0; i <= 16; i++)
		for (int j = 0; j <= 6; j++) binom[i][j] = j == 0 ? 1u : (i == 0 ? 0u : binom[i - 1][j - 1] + binom[i - 1][j]);
	u32 level[8] = {};
	u32 p3 = 1;
	for (u32 d = 0; d <= thr; d++) { level[d + 1] = level[d] + binom[16][d] * p3; p3 *= 3; }
	const u32 ball = level[thr + 1];
	vdjx_work db(c);
	u32 *d_a, *d_level;
	HIP_TRY(db.alloc(&d_a, n));
	HIP_TRY(db.alloc(&d_level, 8));
	HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_binom), binom, sizeof(binom), 0, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(d_a, anchors, n * 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(d_level, level, sizeof(level), hipMemcpyHostToDevice, c->stream));
	const u64 total = (u64) n * ball;
	{
		vdjx_prof_scope ps(c, "k_ball_bits");
		hipLaunchKernelGGL(k_ball_bits, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, d_a, (u32) n, ball, thr, d_level, *d_bits);
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

extern "C" int vdjx_anchor_sets_from_anchors(vdjx_ctx* c, const uint32_t* v_anchors, size_t nv, const uint32_t* j_anchors, size_t nj, int am) {
	if (!c || (nv && !v_anchors) || (nj && !j_anchors)) { vdjx_set_error("vdjx_anchor_sets_from_anchors: NULL argument"); return VDJX_EINVAL; }
	if (nv > 3000 || nj > 3000) { vdjx_set_error("vdjx_anchor_sets_from_anchors: more than 3000 anchors in a set"); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	if (am < 0) {                                            // no row has a distance <= am: empty sets
		int rc = load_set(c, nullptr, 0, &c->d_vbits, 0);
		if (!rc) rc = load_set(c, nullptr, 0, &c->d_jbits, 0);
		if (rc) return rc;
	} else {
		const u32 thr = (u32) (am > 5 ? 5 : am);             // the index files stop at MAX_DIST 5 (seq_dist.c:9)
		int rc = ball_set(c, v_anchors, nv, thr, &c->d_vbits);
		if (!rc) rc = ball_set(c, j_anchors, nj, thr, &c->d_jbits);
		if (rc) return rc;
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->anchors_loaded = true;
	return VDJX_OK;
}
