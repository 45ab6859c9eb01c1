// vdjx_score.hip -- the batched scorers (SURVEY §8a rows a-7 ... a-10).
//
//   K7  k_seed_count / k_root_dp      root (V-region homology) scorer         seq_score.c:92-156
//   K8  (vdjx_rindex.hip)             read index                              quick_map3.c:126-149
//       k_window_pairs / k_window_cover  read->window mapper + coverage test  quick_map3.c:188-266, coverage.c:10-130
//   K10 k_map_emit                    mapped pairs of final contigs in order  quick_map3.c:152-181, 311-340
//
// None of this is a dense contraction: the DP is a max-plus recurrence on int8 cells, the mapper is
// exact-match hashing, the validator is counting.  They run on the VALU/LDS; MFMA does not apply.
#include "vdjx_common.h"

#include <algorithm>
#include <numeric>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

#define NONE32 0xFFFFFFFFu

namespace {
template <typename T> void free_set(T*& p) { if (p) (void) hipFree(p); p = nullptr; }
}  // namespace

// ==============================================================================================
// a-7 root scorer
// ==============================================================================================
// seq_to_kmer.c:6-29: A0 T1 C2 G3, anything else -1.  Branch-free: bits 1-2 of the ASCII code tell A(00) C(01) T(10) G(11) apart
__host__ __device__ inline int base_code(char ch) {
	const unsigned c = (unsigned char) ch;
	const int code = (int) ((0xD8u >> (((c >> 1) & 3u) * 2u)) & 3u);
	return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? code : -1;
}

extern "C" int vdjx_vregion_load(vdjx_ctx* c, const char* const* lines, size_t n_lines, int vk) {
	if (!c || (n_lines && !lines)) { vdjx_set_error("vdjx_vregion_load: NULL argument"); return VDJX_EINVAL; }
	if (vk < 2 || vk > 16) { vdjx_set_error("vregion k-mer size %d outside [2,16]", vk); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	free_set(c->d_vtext); free_set(c->d_line_off); free_set(c->d_seed_code); free_set(c->d_seed_pos);
	std::vector<u32> off(n_lines + 1, 0);
	std::string text;
	for (size_t i = 0; i < n_lines; i++) {
		text += lines[i];
		off[i + 1] = (u32) text.size();
	}
	// seq_score.c:36-48: every vk-mer start i < len - vk of every line, keyed by content; all lines share the map
	std::vector<std::pair<u32, u32>> seeds;
	for (size_t li = 0; li < n_lines; li++) {
		const char* s = text.data() + off[li];
		long len = (long) (off[li + 1] - off[li]);
		for (long i = 0; i < len - vk; i++) {
			u32 code = 0;
			bool ok = true;
			for (int j = 0; j < vk; j++) {
				int b = base_code(s[i + j]);
				if (b < 0) { ok = false; break; }
				code = (code << 2) | (u32) b;
			}
			if (ok) seeds.push_back({code, (u32) i});
		}
	}
	std::sort(seeds.begin(), seeds.end());
	seeds.erase(std::unique(seeds.begin(), seeds.end()), seeds.end());
	std::vector<u32> sc(seeds.size()), sp(seeds.size());
	for (size_t i = 0; i < seeds.size(); i++) { sc[i] = seeds[i].first; sp[i] = seeds[i].second; }
	HIP_TRY(hipMalloc(&c->d_vtext, text.size() + 16));
	HIP_TRY(hipMalloc(&c->d_line_off, off.size() * 4));
	HIP_TRY(hipMalloc(&c->d_seed_code, sc.size() * 4 + 4));
	HIP_TRY(hipMalloc(&c->d_seed_pos, sp.size() * 4 + 4));
	HIP_TRY(hipMemcpy(c->d_vtext, text.data(), text.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c->d_line_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
	if (!sc.empty()) {
		HIP_TRY(hipMemcpy(c->d_seed_code, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(c->d_seed_pos, sp.data(), sp.size() * 4, hipMemcpyHostToDevice));
	}
	c->n_lines = n_lines;
	c->n_seeds = sc.size();
	c->vk = vk;
	c->h_line_off = off;
	return VDJX_OK;
}

__device__ inline u32 lower_bound_u32(const u32* __restrict__ a, u32 n, u32 key) {
	u32 lo = 0, hi = n;
	while (lo < hi) {
		u32 mid = (lo + hi) >> 1;
		if (a[mid] < key) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// per (root, seed offset): the range of index hits (seq_score.c:124-131)
__global__ void k_seed_count(const char* __restrict__ kmers, u32 n, int k, int vk, const u32* __restrict__ seed_code, u32 n_seeds,
                             u32* __restrict__ hit_lo, u32* __restrict__ hit_cnt) {
	const u32 stop = (u32) (k - vk);
	u32 w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n * stop) return;
	const u32 root = w / stop, i = w - root * stop;
	const char* s = kmers + (size_t) root * k + i;
	u32 code = 0;
	bool ok = true;
	for (int j = 0; j < vk; j++) {
		int b = base_code(s[j]);
		if (b < 0) ok = false;
		code = (code << 2) | (u32) (b & 3);
	}
	u32 lo = 0, cnt = 0;
	if (ok) {
		lo = lower_bound_u32(seed_code, n_seeds, code);
		u32 hi = lo;
		while (hi < n_seeds && seed_code[hi] == code) hi++;
		cnt = hi - lo;
	}
	hit_lo[w] = lo;
	hit_cnt[w] = cnt;
}

// exclusive scan of cnt[n] -> pre[n+1], one 1024-thread workgroup (n up to a few million)
__global__ __launch_bounds__(1024) void k_scan_u32(const u32* __restrict__ cnt, u32 n, u32* __restrict__ pre) {
	__shared__ u32 part[1024];
	const u32 per = (n + 1023) / 1024;
	const u32 lo = threadIdx.x * per;
	const u32 hi = lo + per < n ? lo + per : n;
	u32 s = 0;
	for (u32 i = lo; i < hi; i++) s += cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for (u32 i = lo; i < hi; i++) { pre[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) pre[n] = part[1023];
}

// coalesced two-level exclusive scan: per-workgroup local scan of 2048 elements + block sums
__global__ __launch_bounds__(256) void k_scan_local(const u32* __restrict__ cnt, u32 n, u32* __restrict__ pre, u32* __restrict__ bsum) {
	__shared__ u32 part[256];
	const u32 base = blockIdx.x * 2048u + threadIdx.x * 8u;
	u32 v[8], s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) { v[i] = base + i < n ? cnt[base + i] : 0u; s += v[i]; }
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 256; d <<= 1) {
		u32 x = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += x;
		__syncthreads();
	}
	u32 run = part[threadIdx.x] - s;
#pragma unroll
	for (int i = 0; i < 8; i++) { if (base + i < n) pre[base + i] = run; run += v[i]; }
	if (threadIdx.x == 255) bsum[blockIdx.x] = part[255];
}

__global__ void k_scan_add(u32* __restrict__ pre, u32 n, const u32* __restrict__ bpre) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) pre[i] += bpre[i / 2048u];
	if (i == 0) pre[n] = bpre[(n + 2047u) / 2048u];
}

// one thread per (root, seed offset, hit): for every line run the (k+1) x (2k+1) DP (seq_score.c:76-116) on
// root x line[start, start+2k), start as seq_score.c:139-146.  int8 cells, match +1 / mismatch 0 / gap -1,
// first row and column 0; out[root] = 1 as soon as any cell >= threshold.
#define DP_THREADS 128
__global__ __launch_bounds__(DP_THREADS) void k_root_dp(const char* __restrict__ kmers, int k, int threshold,
                                                        const u32* __restrict__ hit_lo, const u32* __restrict__ hit_pre,
                                                        u32 n_groups, u32 stop, u32 total, const u32* __restrict__ seed_pos,
                                                        const char* __restrict__ vtext, const u32* __restrict__ line_off, u32 n_lines,
                                                        uint8_t* __restrict__ out) {
	// The DP column and the root live in REGISTERS (the row loop is fully unrolled to the longest k, guarded by the uniform
	// r <= k): a cell is a handful of integer ops instead of a round trip through LDS per cell (the kernel is a dependent chain
	// per thread and far too small to hide LDS latency by occupancy).  Only the reference segment sits in LDS, one read per column.
	__shared__ char refc[2 * VDJX_MAX_KMER * DP_THREADS];
	const u32 tid = threadIdx.x;
	const u32 w = blockIdx.x * DP_THREADS + tid;
	if (w >= total) return;
	// group = (root, seed offset) holding work item w: last group with hit_pre[g] <= w
	u32 lo = 0, hi = n_groups;
	while (hi - lo > 1) {
		u32 mid = (lo + hi) >> 1;
		if (hit_pre[mid] <= w) lo = mid; else hi = mid;
	}
	const u32 g = lo;
	const u32 root = g / stop;
	if (out[root]) return;
	const int pos = (int) seed_pos[hit_lo[g] + (w - hit_pre[g])];
	char q[VDJX_MAX_KMER];
#pragma unroll
	for (int r = 0; r < VDJX_MAX_KMER; r++) q[r] = r < k ? kmers[(size_t) root * k + r] : (char) 0;
	for (u32 li = 0; li < n_lines; li++) {
		const int len = (int) (line_off[li + 1] - line_off[li]);
		int start = pos - k;
		if (start < 0) start = 0;
		if (start >= len - 2 * k) start = len - 2 * k - 1;
		const char* ref = vtext + line_off[li] + start;
		for (int cidx = 0; cidx < 2 * k; cidx++) refc[cidx * DP_THREADS + tid] = ref[cidx];
		int col[VDJX_MAX_KMER + 1];
#pragma unroll
		for (int r = 0; r <= VDJX_MAX_KMER; r++) col[r] = 0;
		for (int cidx = 1; cidx <= 2 * k; cidx++) {
			const char rc = refc[(cidx - 1) * DP_THREADS + tid];
			int diag = 0, up = 0, best = 0;
#pragma unroll
			for (int r = 1; r <= VDJX_MAX_KMER; r++) {
				if (r <= k) {
					const int left = col[r];
					int v = left - 1;
					v = v > up - 1 ? v : up - 1;
					const int d = diag + (q[r - 1] == rc ? 1 : 0);
					v = v > d ? v : d;
					diag = left;
					up = v;
					col[r] = v;                                 // (0 <= v <= k: the reference's int8 cells never wrap)
					best = best > v ? best : v;
				}
			}
			if (best >= threshold) { out[root] = 1; return; }   // any cell of the matrix at or above the threshold accepts (seq_score.c:103-112)
		}
	}
}

// the scorer proper: d_k = n*k ASCII on the device, out = n host bytes
static int root_score_device(vdjx_ctx* c, vdjx_work& db, const char* d_k, size_t n, int k, int threshold, uint8_t* out) {
	hipStream_t st = c->stream;
	const int stop = k - c->vk;
	u32 *d_lo, *d_cnt, *d_pre;
	uint8_t* d_out;
	const u32 ng = (u32) (n * stop);
	HIP_TRY(db.alloc(&d_lo, ng));
	HIP_TRY(db.alloc(&d_cnt, ng));
	HIP_TRY(db.alloc(&d_pre, ng + 1));
	HIP_TRY(db.alloc(&d_out, n));
	HIP_TRY(hipMemsetAsync(d_out, 0, n, st));
	{
		vdjx_prof_scope ps(c, "k_seed_count");
		hipLaunchKernelGGL(k_seed_count, dim3((ng + 255) / 256), dim3(256), 0, st, d_k, (u32) n, k, c->vk, c->d_seed_code, (u32) c->n_seeds, d_lo, d_cnt);
	}
	{
		const u32 nb = (ng + 2047u) / 2048u;
		u32 *d_bsum, *d_bpre;
		HIP_TRY(db.alloc(&d_bsum, nb));
		HIP_TRY(db.alloc(&d_bpre, nb + 1));
		hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, st, d_cnt, ng, d_pre, d_bsum);
		hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, d_bsum, nb, d_bpre);
		hipLaunchKernelGGL(k_scan_add, dim3((ng + 255) / 256), dim3(256), 0, st, d_pre, ng, d_bpre);
	}
	u32 run = 0;
	HIP_TRY(hipMemcpyAsync(&run, d_pre + ng, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	c->stats["root_dp_items"] = run;
	if (run >= (1u << 31)) { vdjx_set_error("too many seed hits in one call (%u)", run); return VDJX_ELIMIT; }
	if (threshold <= 0) {
		// cells of row/column 0 are 0 and are tested too (seq_score.c:103-112): any seed hit accepts
		std::vector<u32> pre(ng + 1);
		HIP_TRY(hipMemcpy(pre.data(), d_pre, (size_t) (ng + 1) * 4, hipMemcpyDeviceToHost));
		for (size_t r = 0; r < n; r++) out[r] = pre[(r + 1) * stop] > pre[r * stop];
		return VDJX_OK;
	}
	if (run) {
		vdjx_prof_scope ps(c, "k_root_dp");
		hipLaunchKernelGGL(k_root_dp, dim3((unsigned) ((run + DP_THREADS - 1) / DP_THREADS)), dim3(DP_THREADS), 0, st, d_k, k, threshold,
		                   d_lo, d_pre, ng, (u32) stop, (u32) run, c->d_seed_pos, c->d_vtext, c->d_line_off, (u32) c->n_lines, d_out);
	}
	HIP_TRY(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c);
	return VDJX_OK;
}

static int root_score_check(vdjx_ctx* c, const char* who, size_t n, int k) {
	if (!c->d_vtext) { vdjx_set_error("%s: call vdjx_vregion_load first", who); return VDJX_ESTATE; }
	if (k < 1 || k > VDJX_MAX_KMER) { vdjx_set_error("k=%d outside [1,%d]", k, VDJX_MAX_KMER); return VDJX_ELIMIT; }
	for (size_t li = 0; li < c->n_lines; li++) {
		if ((long) (c->h_line_off[li + 1] - c->h_line_off[li]) <= 2L * k) {
			vdjx_set_error("v_region line %zu is not longer than 2k=%d (the reference reads out of bounds there)", li, 2 * k);
			return VDJX_EINVAL;
		}
	}
	const int stop = k - c->vk;
	if (stop > 0 && n * (size_t) stop >= (1ull << 31)) { vdjx_set_error("too many roots in one call"); return VDJX_ELIMIT; }
	return VDJX_OK;
}

extern "C" int vdjx_root_score(vdjx_ctx* c, const char* kmers, size_t n, int k, int threshold, uint8_t* out) {
	if (!c || (n && (!kmers || !out))) { vdjx_set_error("vdjx_root_score: NULL argument"); return VDJX_EINVAL; }
	int rc = root_score_check(c, "vdjx_root_score", n, k);
	if (rc) return rc;
	if (n == 0) return VDJX_OK;
	memset(out, 0, n);
	if (k - c->vk <= 0 || c->n_seeds == 0) return VDJX_OK;          // no seed can hit: score_seq returns 0
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	vdjx_work db(c);
	char* d_k;
	HIP_TRY(db.alloc(&d_k, n * k));
	HIP_TRY(hipMemcpyAsync(d_k, kmers, n * k, hipMemcpyHostToDevice, c->stream));
	return root_score_device(c, db, d_k, n, k, threshold, out);
}

__global__ void k_root_gather(const char* __restrict__ kmers, const u32* __restrict__ roots, u32 first, u32 stride, u32 n_sel, int k,
                              char* __restrict__ out, u32* __restrict__ ids) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_sel) return;
	const u32 node = roots[first + (size_t) i * stride];
	ids[i] = node + 1;
	for (int j = 0; j < k; j++) out[(size_t) i * k + j] = kmers[(size_t) node * k + j];
}

extern "C" size_t vdjx_root_part(const vdjx_graph* g, uint32_t first, uint32_t stride) {
	if (!g || stride == 0 || first >= g->n_roots) return 0;
	return (g->n_roots - first + stride - 1) / stride;
}

extern "C" int vdjx_root_score_graph(vdjx_ctx* c, const vdjx_graph* g, int threshold, uint32_t first, uint32_t stride,
                                     uint32_t* root_ids, uint8_t* out) {
	if (!c || !g) { vdjx_set_error("vdjx_root_score_graph: NULL argument"); return VDJX_EINVAL; }
	if (g->ctx != c) { vdjx_set_error("vdjx_root_score_graph: graph belongs to another context"); return VDJX_EINVAL; }
	if (stride == 0) { vdjx_set_error("vdjx_root_score_graph: stride 0"); return VDJX_EINVAL; }
	const size_t n = vdjx_root_part(g, first, stride);
	const int k = g->k;
	int rc = root_score_check(c, "vdjx_root_score_graph", n, k);
	if (rc) return rc;
	if (n == 0) return VDJX_OK;
	if (!root_ids || !out) { vdjx_set_error("vdjx_root_score_graph: NULL result array"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	vdjx_work db(c);
	char* d_k;
	u32* d_ids;
	HIP_TRY(db.alloc(&d_k, n * k));
	HIP_TRY(db.alloc(&d_ids, n));
	hipLaunchKernelGGL(k_root_gather, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, g->d_kmers, g->d_roots, first, stride, (u32) n, k, d_k, d_ids);
	HIP_TRY(hipMemcpyAsync(root_ids, d_ids, n * 4, hipMemcpyDeviceToHost, c->stream));
	memset(out, 0, n);
	if (k - c->vk <= 0 || c->n_seeds == 0) { HIP_TRY(hipStreamSynchronize(c->stream)); return VDJX_OK; }
	return root_score_device(c, db, d_k, n, k, threshold, out);
}

// (a-8 read index: vdjx_rindex.hip)

// ==============================================================================================
// a-8/a-9/a-10 mapper core shared by k_window_hits, k_window_pairs and k_map_emit
// ==============================================================================================
#define MAP_THREADS 512
#define MAP_MAXOFF 1024          // window/contig length - rl  <= MAP_MAXOFF
#define WT_SLOTS 2048            // LDS table: read class -> last offset of the window where it occurs
#define MAP_PRESENT_LOG2 15
#define MAP_PRESENT_WORDS (1u << (MAP_PRESENT_LOG2 - 5))

struct ReadIndexDev {
	const u64* bases;
	const u32* slots; u32 mask;
	const u32* rep; const u32* start; const u32* cnt1; const u32* recs;
	const uint4* csr_info; const u32* pair_r2;    // csr_info[i] = record info of recs[i]: one coalesced 16-byte load per hit
	const u32* dstart; const uint4* dinfo;        // distinct read-1 infos per class with multiplicities (window scoring)
	int rl;
};

struct MapLds {
	u32 cls[MAP_MAXOFF];                            // read class at the offset or NONE32
	u32 cstart[MAP_MAXOFF];                         // CSR start of that class
	u32 hpre[MAP_MAXOFF + 1];                       // prefix of class sizes (hits enumerate in reference order)
	u32 wt_key[WT_SLOTS];                           // class id + 1
	u32 wt_last[WT_SLOTS];                          // last offset + 1 with that class
	u32 scan[MAP_THREADS];
	u32 present[MAP_PRESENT_WORDS];                 // one bit per hashed class id seen in the window: "is this mate class here at all?" is one LDS
	                                                // word for the 98 % of the hits whose mate lies elsewhere (a V gene is shared by many clones)
	u32 inst_total;                                 // weighted mode: read-1 instances behind the distinct hits
};

__device__ inline u32 map_present_bit(u32 cls) { return (cls * 2654435761u) >> (32 - MAP_PRESENT_LOG2); }

// classify every offset o in [0, len-rl) (quick_map3.c:200: the last offset is never looked at), build the
// class -> last offset table, prefix the class sizes.  Returns the hit count H (uniform).
// `weighted`: hits enumerate the DISTINCT read-1 infos of a class (ix.dstart/dinfo) instead of its members
__device__ inline u32 map_prepare(MapLds& L, const ReadIndexDev& ix, const char* __restrict__ w, int len, bool weighted = false) {
	const int rl = ix.rl;
	const int noff = len - rl;
	const u32 tid = threadIdx.x;
	for (u32 i = tid; i < WT_SLOTS; i += MAP_THREADS) { L.wt_key[i] = 0; L.wt_last[i] = 0; }
	for (u32 i = tid; i < MAP_PRESENT_WORDS; i += MAP_THREADS) L.present[i] = 0;
	if (tid == 0) L.inst_total = 0;
	__syncthreads();
	for (int o = tid; o < noff; o += MAP_THREADS) {
		u128 b = 0;
		bool ok = true;
		for (int j = 0; j < rl; j++) {
			int cde = base_code(w[o + j]);
			if (cde < 0) ok = false;
			b = (b << 2) | (u32) (cde & 3);
		}
		const u64 hi = (u64) (b >> 64), lo = (u64) b;
		u32 cls = NONE32, cs = 0, sz = 0;
		if (ok) {
			u32 slot = (u32) (vdjx_mix(lo, hi) >> 17) & ix.mask;
			for (;;) {
				const u32 v = ix.slots[slot];
				if (!v) break;
				const u32 rr = ix.rep[v - 1];
				const ulonglong2 k = ((const ulonglong2*) ix.bases)[rr];
				if (k.x == hi && k.y == lo) { cls = v - 1; break; }
				slot = (slot + 1) & ix.mask;
			}
		}
		if (cls != NONE32) {
			if (weighted) {
				cs = ix.dstart[cls];
				sz = ix.dstart[cls + 1] - cs;
				atomicAdd(&L.inst_total, ix.cnt1[cls]);
			} else {
				cs = ix.start[cls];
				sz = ix.cnt1[cls];                  // read-1 members come first
			}
			{
				const u32 pb = map_present_bit(cls);
				atomicOr(&L.present[pb >> 5], 1u << (pb & 31));
			}
			// class -> last offset ("read2[id] = m_info": the last writer wins, quick_map3.c:214)
			u32 slot = (cls * 2654435761u) & (WT_SLOTS - 1);
			for (;;) {
				u32 cur = L.wt_key[slot];
				if (cur == 0) {
					cur = atomicCAS(&L.wt_key[slot], 0u, cls + 1);
					if (cur == 0) cur = cls + 1;
				}
				if (cur == cls + 1) { atomicMax(&L.wt_last[slot], (u32) o + 1); break; }
				slot = (slot + 1) & (WT_SLOTS - 1);
			}
		}
		L.cls[o] = cls;
		L.cstart[o] = cs;
		L.hpre[o] = sz;
	}
	__syncthreads();
	// exclusive prefix of the class sizes: wave 0, 16 consecutive offsets per lane
	if (tid < 64) {
		const int per = (noff + 63) / 64;
		const int a = (int) tid * per;
		const int b2 = a + per < noff ? a + per : noff;
		u32 sum = 0;
		for (int o = a; o < b2; o++) sum += L.hpre[o];
		const u32 incl = (u32) vdjx_wave_scan_add((int) sum);
		u32 run = incl - sum;
		for (int o = a; o < b2; o++) { const u32 sz = L.hpre[o]; L.hpre[o] = run; run += sz; }
		if (tid == 63) L.hpre[noff] = incl;
	}
	__syncthreads();
	return L.hpre[noff];
}

// last offset+1 at which a read of class `cls` occurs in the window, or 0
__device__ inline u32 map_last_occurrence(const MapLds& L, u32 cls) {
	if (cls == NONE32) return 0;
	u32 slot = (cls * 2654435761u) & (WT_SLOTS - 1);
	for (;;) {
		const u32 cur = L.wt_key[slot];
		if (cur == 0) return 0;
		if (cur == cls + 1) return L.wt_last[slot];
		slot = (slot + 1) & (WT_SLOTS - 1);
	}
}

struct Hit {
	bool pair;          // a mapped pair (quick_map3.c:223-245)
	u32 pair_id, rec1;  // (weighted mode: pair_id = how many identical read pairs the entry stands for)
	int which;          // 0: read-2 record A won, 1: B
	int pos1, pos2, insert;
	uint8_t rc1, rc2;
};

// hit h (reference order: offset-major, registration order inside a class) -> mapped pair or not
template <bool WEIGHTED = false>
__device__ inline Hit map_eval_hit(const MapLds& L, const ReadIndexDev& ix, int noff, u32 h) {
	Hit r;
	r.pair = false;
	int lo = 0, hi = noff;                          // offset holding hit h: last o with hpre[o] <= h
	while (hi - lo > 1) {
		int mid = (lo + hi) >> 1;
		if (L.hpre[mid] <= h) lo = mid; else hi = mid;
	}
	const int o = lo;
	const u32 ci = L.cstart[o] + (h - L.hpre[o]);
	const uint4 info = WEIGHTED ? ix.dinfo[ci] : ix.csr_info[ci];
	if (!(info.w & RI_R1)) return r;                // read-2 instances only feed the read2 map
	// read2[id]: among the pair's read-2 records the one written last = largest offset, then latest registration
	const u32 ba = map_present_bit(info.y), bb = map_present_bit(info.z);
	const bool pa = (L.present[ba >> 5] >> (ba & 31)) & 1u, pb = (L.present[bb >> 5] >> (bb & 31)) & 1u;
	if (!pa && !pb) return r;
	const u32 la = pa ? map_last_occurrence(L, info.y) : 0u, lb = pb ? map_last_occurrence(L, info.z) : 0u;
	if (!la && !lb) return r;
	const int which = (lb && lb >= la) ? 1 : 0;
	const u32 best = which ? lb : la;
	const uint8_t rc1 = (info.w & RI_RC) ? 1 : 0;
	const uint8_t rc2 = (info.w & (which ? RI_RCB : RI_RCA)) ? 1 : 0;
	if (rc1 == rc2) return r;                       // quick_map3.c:227
	const int pos1 = o + 1, pos2 = (int) best;
	const int d = pos1 - pos2;
	const int insert = (int) (short) ((d < 0 ? -d : d) + ix.rl);
	if (insert < 50 || insert > 400) return r;      // MIN_INSERT / MAX_INSERT, quick_map3.c:23-24
	r.pair = true;
	r.pair_id = info.x; r.rec1 = WEIGHTED ? 0u : ix.recs[ci]; r.which = which;
	r.pos1 = pos1; r.pos2 = pos2; r.insert = insert; r.rc1 = rc1; r.rc2 = rc2;
	return r;
}

// hits per window/contig: sizes the pair scratch and orders the work largest-first
__global__ __launch_bounds__(MAP_THREADS) void k_window_hits(ReadIndexDev ix, const char* __restrict__ windows, u32 n, int len,
                                                             bool weighted, u32* __restrict__ out_hits, u32* __restrict__ out_inst) {
	__shared__ MapLds L;
	for (u32 wi = blockIdx.x; wi < n; wi += gridDim.x) {
		const u32 H = map_prepare(L, ix, windows + (size_t) wi * len, len, weighted);
		if (threadIdx.x == 0) { out_hits[wi] = H; if (out_inst) out_inst[wi] = weighted ? L.inst_total : H; }
		__syncthreads();
	}
}

// ----------------------------------------------------------------------------------------------
// K8+K9: one workgroup per window (largest first).
//
// coverage.c restated for counting hardware.  The start list holds (pos1,pos2) and
// (pos2,pos1) per mapped pair (quick_map3.c:241-242).
//  rule 1 (coverage.c:76-121) only looks at the sorted `first` values: it is evaluated from their
//    cumulative histogram.
//  rule 2 (coverage.c:10-61) rebuilds, for every position pos, a coverage array from the entries whose
//    first lies in (pos-rl, pos] and tests every j = pos+delta, delta in [clo, chi).  For a fixed delta
//    an entry (f, s) is counted at pos iff both f and s-delta lie in (pos-rl, pos], i.e. for pos in
//    [max(f, s-delta), min(f, s-delta)+rl): one +1/-1 pair in a difference array over pos.  The kernel
//    keeps DB difference arrays (one per delta of the current batch) in LDS, replays the window's mapped
//    pairs (stored once in a scratch list) for every batch, prefixes, and tests.
// ----------------------------------------------------------------------------------------------
#define COV_WORDS 8192
#define HIT_CHUNK 524288u        // hits per workgroup of k_window_pairs: only the very deepest windows are split (every piece pays one
                                 // preparation of the window; measured at 10 M pairs: 65536 -> 5.2 ms, 262144 and above -> 3.5 ms)

// one (weighted) entry of the class at offset o -> mapped pair or not (quick_map3.c:223-245)
__device__ inline bool map_eval_entry(const MapLds& L, int rl, int o, const uint4 info, u32& pos2_out) {
	if (!(info.w & RI_R1)) return false;
	const u32 ba = map_present_bit(info.y), bb = map_present_bit(info.z);
	const bool pa = (L.present[ba >> 5] >> (ba & 31)) & 1u, pb = (L.present[bb >> 5] >> (bb & 31)) & 1u;
	if (!pa && !pb) return false;
	const u32 la = pa ? map_last_occurrence(L, info.y) : 0u, lb = pb ? map_last_occurrence(L, info.z) : 0u;
	if (!la && !lb) return false;
	const int which = (lb && lb >= la) ? 1 : 0;
	const u32 best = which ? lb : la;
	const u32 rc1 = (info.w & RI_RC) ? 1u : 0u;
	const u32 rc2 = (info.w & (which ? RI_RCB : RI_RCA)) ? 1u : 0u;
	if (rc1 == rc2) return false;                  // quick_map3.c:227
	const int d = (o + 1) - (int) best;
	const int insert = (int) (short) ((d < 0 ? -d : d) + rl);
	if (insert < 50 || insert > 400) return false; // MIN_INSERT / MAX_INSERT, quick_map3.c:23-24
	pos2_out = best;
	return true;
}

// K8: mapped pairs of a slice [h0, h1) of a window's (distinct) hits, appended (any order) to the window's list as
// (multiplicity << 32 | pos1 << 16 | pos2); pair_np = mapped pairs counted with multiplicity.
// The waves of the workgroup take the offsets of the slice one at a time (LDS ticket): the entries of an offset's read class
// are consecutive, so a wave streams them with coalesced 16-byte loads and no search; pairs (a few per cent of the entries) are
// staged per wave in LDS and leave with one global cursor bump per 128.
#define WP_STAGE 64
#define WP_UNR 4
__global__ __launch_bounds__(MAP_THREADS) void k_window_pairs(ReadIndexDev ix, const char* __restrict__ windows, int len,
                                                              const uint4* __restrict__ work /* {window, h0, h1, -} */,
                                                              const u64* __restrict__ pair_off, u64* __restrict__ pair_buf,
                                                              u32* __restrict__ pair_cnt, u32* __restrict__ pair_np) {
	__shared__ MapLds L;
	__shared__ u64 stg[MAP_THREADS / 64][WP_STAGE];
	__shared__ int s_next, s_ohi;
	const uint4 wk = work[blockIdx.x];
	const u32 wi = wk.x;
	const int noff = len - ix.rl;
	const u32 H = map_prepare(L, ix, windows + (size_t) wi * len, len, true);
	const u32 h0 = wk.y, h1 = wk.z < H ? wk.z : H;
	if (h0 >= h1) return;
	if (threadIdx.x == 0) {
		int lo = 0, hi = noff;                          // last offset with hpre[o] <= h0
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.hpre[mid] <= h0) lo = mid; else hi = mid; }
		s_next = lo;
		lo = 0; hi = noff;                              // last offset with hpre[o] < h1
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.hpre[mid] < h1) lo = mid; else hi = mid; }
		s_ohi = lo;
	}
	__syncthreads();
	const int ohi = s_ohi;
	const int lane = __lane_id();
	u64* mystg = stg[threadIdx.x >> 6];
	u64* pairs = pair_buf + pair_off[wi];
	u32 fill = 0, mine = 0;
	for (;;) {
		int o = 0;
		if (lane == 0) o = atomicAdd(&s_next, 1);
		o = __builtin_amdgcn_readlane(o, 0);
		if (o > ohi) break;
		const u32 a = L.hpre[o], bnd = L.hpre[o + 1];
		if (a == bnd) continue;
		const u32 e0 = (h0 > a ? h0 : a) - a, e1 = (h1 < bnd ? h1 : bnd) - a;
		const u32 cs = L.cstart[o];
		// four 16-byte loads are in flight per lane before the first is used: the loop is bound by their latency, not their bytes
		for (u32 eb = e0; eb < e1; eb += 64 * WP_UNR) {
			uint4 info[WP_UNR];
#pragma unroll
			for (int u = 0; u < WP_UNR; u++) {
				const u32 e = eb + (u32) (u * 64 + lane);
				info[u] = e < e1 ? ix.dinfo[cs + e] : make_uint4(0, 0, 0, 0);        // (flags 0: not a read-1 entry)
			}
#pragma unroll
			for (int u = 0; u < WP_UNR; u++) {
				if (eb + (u32) (u * 64) >= e1) break;                                // (wave-uniform)
				u32 pos2 = 0;
				const bool pr = map_eval_entry(L, ix.rl, o, info[u], pos2);
				const u64 m = __ballot(pr);
				if (!m) continue;
				const u32 cnt = (u32) __popcll(m);
				if (fill + cnt > WP_STAGE) {                   // (wave-uniform) flush
					u32 base = 0;
					if (lane == 0) base = atomicAdd(&pair_cnt[wi], fill);
					base = (u32) __builtin_amdgcn_readlane((int) base, 0);
					for (u32 i = (u32) lane; i < fill; i += 64) pairs[base + i] = mystg[i];
					fill = 0;
				}
				if (pr) {
					mystg[fill + (u32) __popcll(m & ((1ull << lane) - 1ull))] = ((u64) info[u].x << 32) | ((u32) (o + 1) << 16) | pos2;
					mine += info[u].x;
				}
				fill += cnt;
			}
		}
	}
	if (fill) {
		u32 base = 0;
		if (lane == 0) base = atomicAdd(&pair_cnt[wi], fill);
		base = (u32) __builtin_amdgcn_readlane((int) base, 0);
		for (u32 i = (u32) lane; i < fill; i += 64) pairs[base + i] = mystg[i];
	}
	mine = (u32) __builtin_amdgcn_readlane(vdjx_wave_scan_add((int) mine), 63);
	if (lane == 0 && mine) atomicAdd(&pair_np[wi], mine);
}

// K9: coverage verdict of a window from its pair list
__global__ __launch_bounds__(MAP_THREADS) void k_window_cover(int len, int rl, vdjx_cov_params cp, const u32* __restrict__ order,
                                                              const u64* __restrict__ pair_off, const u64* __restrict__ pair_buf,
                                                              const u32* __restrict__ pair_cnt, uint8_t* __restrict__ out_valid,
                                                              u64* __restrict__ dbg) {
	const long long t_begin = dbg ? clock64() : 0;
	__shared__ u32 hf[MAP_MAXOFF + 64 + 2];         // histogram of firsts -> inclusive prefix "cum"
	__shared__ int diff[COV_WORDS];
	__shared__ u32 s_bad, s_ok;
	__shared__ int s_firstv;
	const u32 tid = threadIdx.x;
	const int D = len + 1;
	const int e0 = cp.eval_start, e1 = cp.eval_stop, fl = cp.floor;
	const int gap = rl - cp.read_span;
	const int clo = cp.insert_low - rl - cp.mate_span / 2;
	const int chi = cp.insert_high - rl + cp.mate_span / 2;
	const int npos = e1 - e0;                        // positions e0 .. e1-1
	const int stride = npos + 1;
	const int DB = COV_WORDS / stride;               // deltas per batch (host guarantees >= 1)
	const u32 wi = order[blockIdx.x];
	const u64* pairs = pair_buf + pair_off[wi];      // (multiplicity << 32 | pos1 << 16 | pos2): identical read pairs come once
	const u32 npairs = pair_cnt[wi];
	if (fl == 0) { if (tid == 0) out_valid[wi] = 1; return; }
	// histogram of the start entries' first values: one private copy per wave in `diff` (deep windows put
	// hundreds of thousands of increments on a few hundred positions), then summed into hf
	const u32 wv = tid >> 6, NW = MAP_THREADS / 64;
	u32* hp = (u32*) diff;
	const bool priv = (u32) (D + 1) * NW <= COV_WORDS;
	for (u32 i = tid; i < (u32) D + 1; i += MAP_THREADS) hf[i] = 0;
	if (priv) for (u32 i = tid; i < (u32) (D + 1) * NW; i += MAP_THREADS) hp[i] = 0;
	if (tid == 0) { s_bad = 0; s_ok = 0; }
	__syncthreads();
	u32* myh = priv ? hp + wv * (u32) (D + 1) : hf;
	for (u32 q = tid; q < npairs; q += MAP_THREADS) {
		const u64 pe = pairs[q];
		const u32 pr = (u32) pe, mult = (u32) (pe >> 32);
		atomicAdd(&myh[pr >> 16], mult);
		atomicAdd(&myh[pr & 0xFFFFu], mult);
	}
	__syncthreads();
	if (priv) {
		for (u32 i = tid; i < (u32) D + 1; i += MAP_THREADS) {
			u32 sum = 0;
			for (u32 w2 = 0; w2 < NW; w2++) sum += hp[w2 * (u32) (D + 1) + i];
			hf[i] = sum;
		}
		__syncthreads();
	}
	// ---- rule 1 (coverage.c:76-121) from the cumulative histogram cum[p] = #entries with first <= p.
	// Wave 0 prefixes; then every candidate value v is tested by its own thread (any failure invalidates, so the
	// reference's first-failure order does not matter); thread 0 does the two end-of-list checks.
	if (tid < 64) {
		const int per = (D + 63) / 64;
		const int a0 = (int) tid * per;
		const int b0 = a0 + per < D ? a0 + per : D;
		u32 sum = 0;
		for (int q = a0; q < b0; q++) sum += hf[q];
		const u32 incl = (u32) vdjx_wave_scan_add((int) sum);
		u32 run = incl - sum;
		for (int q = a0; q < b0; q++) { run += hf[q]; hf[q] = run; }
	}
	if (tid == 0) s_firstv = 0x7FFFFFFF;
	__syncthreads();
	{
		const int n_ent = (int) hf[D - 1];
		const int vmax = e1 - cp.read_span + 1;
		const int vlo = e0 > 1 ? e0 : 1;
		for (int v = vlo + (int) tid; v <= vmax && v < D; v += MAP_THREADS) {
			const int cv = (int) hf[v], cv1 = (int) hf[v - 1];
			if (cv == cv1) continue;
			atomicMin(&s_firstv, v);
			const int i0 = cv1;                                   // smallest index holding value v
			if (i0 < fl) s_bad = 1;                               // coverage.c:80-84
			const int pg = v - gap - 1;
			const int cg = pg < 0 ? 0 : (pg >= D ? n_ent : (int) hf[pg]);
			if (cg > i0 - fl) s_bad = 1;                          // :96-100
		}
		__syncthreads();
		if (tid == 0) {
			bool ok = true;
			auto cum = [&](int p) -> int { return p < 0 ? 0 : (p >= D ? n_ent : (int) hf[p]); };
			auto value_at = [&](int idx) -> int {       // idx-th smallest first (0-based): smallest p with cum[p] > idx
				int lo2 = 0, hi2 = D - 1;
				while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (cum(mid) > idx) hi2 = mid; else lo2 = mid + 1; }
				return lo2;
			};
			if (s_firstv != 0x7FFFFFFF && s_firstv > e0 + gap) ok = false;      // :86-93
			int i_fin = cum(vmax);
			if (i_fin >= n_ent) i_fin = n_ent - 1;                    // :106-108
			if (n_ent > 0) {
				const int v_fin = value_at(i_fin);
				if (i_fin > fl && value_at(i_fin - fl) < v_fin - gap) ok = false;   // :111-113
				if (!(i_fin > 0 && v_fin > e1 - rl)) ok = false;                   // :116-120
			} else {
				ok = false;
			}
			if (!ok) s_bad = 1;
		}
	}
	__syncthreads();
	// ---- rule 2.  Coverage only grows with more entries, so the pair list is replayed in doubling chunks and
	// a batch of deltas is done as soon as every tested (pos, delta) already reaches the floor.
	for (int d0 = clo; d0 < chi && !s_bad; d0 += DB) {
		const int nd = chi - d0 < DB ? chi - d0 : DB;
		for (int i = tid; i < nd * stride; i += MAP_THREADS) diff[i] = 0;
		__syncthreads();
		u32 done = 0, chunk = 4096;
		const u32 tot = 2 * npairs;
		const u64 perm_stride = (tot % 1000003u) ? 1000003ull : 1ull;      /* prime: a bijection on [0, tot) */
		for (;;) {
			const u32 end = done + chunk < 2 * npairs ? done + chunk : 2 * npairs;
			for (u32 q0 = done + tid; q0 < end; q0 += MAP_THREADS) {
				// the list is roughly sorted by position (hits are enumerated offset-major): replay it in a scattered
				// order so that every chunk samples the whole window and the early exit can fire
				const u32 q = (u32) (((u64) q0 * perm_stride) % tot);
				const u64 pe = pairs[q >> 1];
				const u32 pr = (u32) pe;
				const int mult = (int) (pe >> 32);
				const int p1 = (int) (pr >> 16), p2 = (int) (pr & 0xFFFFu);
				const int f = (q & 1) ? p2 : p1, sx = (q & 1) ? p1 : p2;
				// deltas of this batch with |f - (sx - delta)| < rl
				int dlo = sx - f - rl + 1, dhi = sx - f + rl - 1;
				if (dlo < d0) dlo = d0;
				if (dhi > d0 + nd - 1) dhi = d0 + nd - 1;
				for (int dl = dlo; dl <= dhi; dl++) {
					const int v = sx - dl;
					int lo = f > v ? f : v;                       // first pos that sees both
					int hi = (f < v ? f : v) + rl;                // one past the last
					if (lo < e0) lo = e0;
					if (hi > e1) hi = e1;
					if (lo < hi) {
						atomicAdd(&diff[(dl - d0) * stride + (lo - e0)], mult);
						atomicAdd(&diff[(dl - d0) * stride + (hi - e0)], -mult);
					}
				}
			}
			if (tid == 0) s_ok = 1;
			__syncthreads();
			// one wave per delta row: chunked inclusive scan of the difference array, test on the fly
			for (int row_i = (int) (tid >> 6); row_i < nd; row_i += MAP_THREADS / 64) {
				const int dl = d0 + row_i;
				const int* row = diff + row_i * stride;
				const int lane = (int) (tid & 63);
				int carry = 0;
				bool ok = true;
				for (int p0 = 0; p0 < npos; p0 += 64) {
					const int p = p0 + lane;
					const int v = vdjx_wave_scan_add(p < npos ? row[p] : 0);
					const int run = carry + v;
					carry += __builtin_amdgcn_readlane(v, 63);
					if (p < npos) {
						const int pos = e0 + p;
						const bool evaluated = pos == e0 || (pos - 1 + clo) < e1;    // the loop tests the previous mate_low (coverage.c:25)
						int mh = pos + chi;
						if (mh > e1) mh = e1 + 1;                                    // coverage.c:36-38
						const int j = pos + dl;
						if (evaluated && j < mh) {
							if (j < 0 || j > len + 1023) s_bad = 1;                   // outside the reference's array: undefined there
							else if (run < fl) ok = false;
						}
					}
				}
				if (!ok) s_ok = 0;
			}
			__syncthreads();
			done = end;
			if (s_ok || s_bad || done >= 2 * npairs) break;
			chunk *= 2;
			__syncthreads();
		}
		if (!s_ok) s_bad = 1;          // every entry counted and some (pos, delta) is still short
		__syncthreads();
	}
	__syncthreads();
	if (tid == 0) {
		out_valid[wi] = s_bad ? 0 : 1;
		if (dbg) dbg[wi] = (u64) (clock64() - t_begin);
	}
}

// ----------------------------------------------------------------------------------------------
// K10: mapped pairs of a contig in the reference's order.  A contig's hits are cut into slices (one workgroup each): a slice
// writes its pairs, in hit order, at the start of its own region (region offset = hit offset: pairs <= hits) and reports how
// many; the gather below lays the slices end to end, which is the reference's order (offset-major, registration order inside
// a class).  One workgroup per contig left half the GPU idle and walked ~20 k hits sequentially.
// ----------------------------------------------------------------------------------------------
#define MAP_SLICE 4096u          // hits per slice at least ...
#define MAP_SLICE_MAX 32768u     // ... and at most (map_emit_impl)
// slice b belongs to the contig ci with slice_start[ci] <= b < slice_start[ci + 1] (the slices are never listed: a binary search
// over the contigs' slice prefix replaces the work list the host used to build and upload)
__device__ inline u32 slice_contig(const u32* __restrict__ slice_start, u32 n, u32 b) {
	u32 lo = 0, hi = n;
	while (hi - lo > 1) {
		const u32 mid = (lo + hi) >> 1;
		if (slice_start[mid] <= b) lo = mid; else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(MAP_THREADS) void k_map_emit(ReadIndexDev ix, const char* __restrict__ contigs, u32 n, int len, u32 slice_hits,
                                                          const u32* __restrict__ slice_start, const u64* __restrict__ region_off,
                                                          vdjx_pair* __restrict__ pairs, u32* __restrict__ slice_cnt) {
	__shared__ MapLds L;
	__shared__ u32 s_base;
	const u32 tid = threadIdx.x;
	const int noff = len - ix.rl;
	const u32 ci = slice_contig(slice_start, n, blockIdx.x);
	const u32 w0 = (blockIdx.x - slice_start[ci]) * slice_hits;
	vdjx_pair* out = pairs + region_off[ci] + w0;
	const u32 H = map_prepare(L, ix, contigs + (size_t) ci * len, len);
	const u32 h1 = w0 + slice_hits < H ? w0 + slice_hits : H;
	if (tid == 0) s_base = 0;
	__syncthreads();
	for (u32 h0 = w0; h0 < h1; h0 += MAP_THREADS) {
		const u32 h = h0 + tid;
		Hit r;
		r.pair = false;
		if (h < h1) r = map_eval_hit(L, ix, noff, h);
		// ordered compaction: inclusive scan of the flags (wave ballots + one LDS word per wave)
		const u64 m = __ballot(r.pair);
		const u32 lane = tid & 63, wv = tid >> 6;
		const u32 before = __popcll(m & ((1ull << lane) - 1ull));
		if (lane == 0) L.scan[wv] = (u32) __popcll(m);
		__syncthreads();
		u32 wbase = 0, total = 0;
		for (u32 i = 0; i < MAP_THREADS / 64; i++) {
			const u32 v = L.scan[i];
			if (i < wv) wbase += v;
			total += v;
		}
		const u32 base = s_base;
		if (r.pair) {
			vdjx_pair* o = out + base + wbase + before;
			o->pair_id = r.pair_id; o->rec1 = r.rec1; o->rec2 = ix.pair_r2[2 * (size_t) r.pair_id + r.which];
			o->pos1 = (int16_t) r.pos1; o->pos2 = (int16_t) r.pos2; o->insert = (int16_t) r.insert;
			o->rc1 = r.rc1; o->rc2 = r.rc2;
		}
		__syncthreads();
		if (tid == 0) s_base = base + total;
		__syncthreads();
	}
	if (tid == 0) slice_cnt[blockIdx.x] = s_base;
}

// exclusive u64 prefix of the slice counts (one workgroup) and, from it, the pairs of every contig
__global__ __launch_bounds__(1024) void k_slice_scan(const u32* __restrict__ cnt, u32 n, u64* __restrict__ pre) {
	__shared__ u64 part[1024];
	const u32 per = (n + 1023) / 1024;
	const u32 lo = threadIdx.x * per;
	const u32 hi = lo + per < n ? lo + per : n;
	u64 s = 0;
	for (u32 i = lo; i < hi; i++) s += cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		const u64 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u64 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for (u32 i = lo; i < hi; i++) { pre[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) pre[n] = part[1023];
}
__global__ void k_contig_counts(const u64* __restrict__ slice_pre, const u32* __restrict__ slice_start, u32 n, u64* __restrict__ out) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = slice_pre[slice_start[i + 1]] - slice_pre[slice_start[i]];
}

// lay the slices end to end in the caller's dense layout
__global__ void k_gather_pairs(const vdjx_pair* __restrict__ src, u32 n, u32 slice_hits, const u32* __restrict__ slice_start,
                               const u64* __restrict__ region_off, const u64* __restrict__ slice_pre, vdjx_pair* __restrict__ dst) {
	const u32 ci = slice_contig(slice_start, n, blockIdx.x);
	const u64 so = region_off[ci] + (u64) (blockIdx.x - slice_start[ci]) * slice_hits;
	const u64 dof = slice_pre[blockIdx.x], cnt = slice_pre[blockIdx.x + 1] - dof;
	const uint32_t* s = (const uint32_t*) (src + so);
	uint32_t* d = (uint32_t*) (dst + dof);
	for (u64 i = threadIdx.x; i < cnt * (sizeof(vdjx_pair) / 4); i += blockDim.x) d[i] = s[i];
}

static int make_index_view(vdjx_ctx* c, ReadIndexDev* ix, int len, const char* who) {
	if (!c->ri_pool) { vdjx_set_error("%s: call vdjx_read_index_build first", who); return VDJX_ESTATE; }
	const vdjx_pool* p = c->ri_pool;
	if (len <= p->rl) { vdjx_set_error("%s: len=%d must exceed the read length %d", who, len, p->rl); return VDJX_EINVAL; }
	if (len - p->rl > MAP_MAXOFF) { vdjx_set_error("%s: len=%d too long (max %d)", who, len, MAP_MAXOFF + p->rl); return VDJX_ELIMIT; }
	ix->bases = p->d_bases;
	ix->slots = c->d_ri_slots; ix->mask = c->ri_nslots - 1;
	ix->rep = c->d_ri_rep; ix->start = c->d_ri_start; ix->cnt1 = c->d_ri_cnt1; ix->recs = c->d_ri_recs;
	ix->csr_info = c->d_rec_info; ix->pair_r2 = c->d_pair_r2;
	ix->dstart = c->d_ri_dstart; ix->dinfo = c->d_ri_dinfo;
	ix->rl = p->rl;
	return VDJX_OK;
}

// largest-first processing order without a comparison sort: by descending bit length of the size (64 counting buckets; inside a
// bucket sizes differ by less than 2x, which is all the tail of a launch cares about).  A stable_sort of 20,000 windows through
// an index indirection cost more host time than the coverage kernel ran.
static void order_by_size_desc(const std::vector<u64>& off, size_t n, std::vector<u32>& order) {
	order.resize(n);
	u32 cnt[66] = {0};
	auto key = [&](size_t i) { const u64 sz = off[i + 1] - off[i]; return sz ? 64u - (u32) __builtin_clzll(sz) : 0u; };     // 0..64
	for (size_t i = 0; i < n; i++) cnt[64 - key(i) + 1]++;
	for (int b = 0; b < 65; b++) cnt[b + 1] += cnt[b];
	for (size_t i = 0; i < n; i++) order[cnt[64 - key(i)]++] = (u32) i;
}

// hits per string, exclusive offsets, and the largest-first processing order
static int plan_windows(vdjx_ctx* c, vdjx_work& db, const ReadIndexDev& ix, const char* d_w, size_t n, int len, bool weighted,
                        std::vector<u64>& off, u32** d_order, u64** d_off, u64* inst_total = nullptr, u64* inst_max = nullptr,
                        std::vector<u32>* order_out = nullptr) {
	hipStream_t st = c->stream;
	vdjx_laps lp(c);
	u32 *d_hits, *d_inst;
	HIP_TRY(db.alloc(&d_hits, n));
	HIP_TRY(db.alloc(&d_inst, n));
	{
		vdjx_prof_scope ps(c, "k_window_hits");
		hipLaunchKernelGGL(k_window_hits, dim3((u32) std::min<size_t>(n, 4096)), dim3(MAP_THREADS), 0, st, ix, d_w, (u32) n, len, weighted, d_hits, d_inst);
	}
	// hit counts down, order and offsets up through ONE page-locked scratch buffer (pageable vectors cost a staging copy each way)
	const size_t need = n * 4 * 3 + (n + 1) * 8 + 64;
	if (need > c->h_plan_cap) {
		if (c->h_plan) (void) hipHostFree(c->h_plan);
		c->h_plan = nullptr; c->h_plan_cap = 0;
		HIP_TRY(hipHostMalloc(&c->h_plan, need + need / 2, hipHostMallocDefault));
		c->h_plan_cap = need + need / 2;
	}
	u64* h_off = (u64*) c->h_plan;
	u32* hits = (u32*) (h_off + n + 1);
	u32* inst = hits + n;
	u32* h_order = inst + n;
	HIP_TRY(hipMemcpyAsync(hits, d_hits, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(inst, d_inst, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	lp.mark("plan_hits_wait");
	if (inst_total) { *inst_total = 0; for (size_t i = 0; i < n; i++) *inst_total += inst[i]; }
	if (inst_max) { *inst_max = 0; for (size_t i = 0; i < n; i++) *inst_max = std::max<u64>(*inst_max, inst[i]); }
	off.assign(n + 1, 0);
	for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + hits[i];
	std::vector<u32> order;
	order_by_size_desc(off, n, order);
	if (order_out) *order_out = order;
	memcpy(h_order, order.data(), n * 4);
	memcpy(h_off, off.data(), (n + 1) * 8);
	HIP_TRY(db.alloc(d_order, n));
	HIP_TRY(db.alloc(d_off, n + 1));
	HIP_TRY(hipMemcpyAsync(*d_order, h_order, n * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(*d_off, h_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
	// (no wait: the scratch is rewritten by the next plan only, and every caller waits for the stream before it returns)
	lp.mark("plan_order_upload");
	return VDJX_OK;
}

// u64 lists laid end to end: move[i] = {source element, destination element, count}
__global__ void k_gather_u64(const u64* __restrict__ src, const u64* __restrict__ move, u64* __restrict__ dst) {
	const u64 so = move[3 * (size_t) blockIdx.x], dof = move[3 * (size_t) blockIdx.x + 1], cnt = move[3 * (size_t) blockIdx.x + 2];
	for (u64 i = threadIdx.x; i < cnt; i += blockDim.x) dst[dof + i] = src[so + i];
}

static int cov_params_check(const vdjx_cov_params* p, const char* who) {
	if (p->eval_start < 1 || p->eval_stop <= p->eval_start) { vdjx_set_error("%s: bad eval range", who); return VDJX_EINVAL; }
	if (p->eval_stop - p->eval_start + 1 > COV_WORDS) { vdjx_set_error("%s: eval range too long", who); return VDJX_ELIMIT; }
	return VDJX_OK;
}

// K8 over n windows: the (weighted) mapped-pair list of every window in the context's pair buffer (c->wp_*); the list of window i
// is wp_buf[wp_off[i] .. + wp_cnt[i])
static int window_pairs_run(vdjx_ctx* c, vdjx_work& db, const ReadIndexDev& ix, const char* windows, size_t n, int len, u32** d_np_out,
                            u32** d_cnt_out, u64** d_off_out, u32** d_order_out) {
	hipStream_t st = c->stream;
	char* d_w;
	u32 *d_np, *d_order, *d_cnt;
	u64* d_off;
	vdjx_laps lp(c);
	HIP_TRY(db.alloc(&d_w, n * len));
	HIP_TRY(db.alloc(&d_np, n));
	HIP_TRY(hipMemcpyAsync(d_w, windows, n * len, hipMemcpyHostToDevice, st));
	lp.mark("wp_upload");
	std::vector<u64>& off = c->wp_off;
	u64 inst_total = 0, inst_max = 0;
	std::vector<u32> ord;
	int rc = plan_windows(c, db, ix, d_w, n, len, true, off, &d_order, &d_off, &inst_total, &inst_max, &ord);
	if (rc) return rc;
	lp.mark("wp_plan");
	if ((size_t) off[n] + 1 > c->wp_cap) {
		free_set(c->wp_buf);
		c->wp_cap = 0;
		const size_t want = (size_t) off[n] + (size_t) off[n] / 4 + 1024;
		HIP_TRY(hipMalloc(&c->wp_buf, want * 8));
		c->wp_cap = want;
	}
	HIP_TRY(db.alloc(&d_cnt, n));
	c->stats["window_hits"] = inst_total;                 // read-1 instances matched (what the reference enumerates one by one)
	c->stats["window_hits_max"] = inst_max;
	c->stats["window_hits_distinct"] = off[n];             // weighted entries actually evaluated
	// work list: deep windows are cut into slices of HIT_CHUNK hits (largest windows first)
	std::vector<uint4> work;
	{
		static const u32 hit_chunk = getenv("VDJX_HIT_CHUNK") && atol(getenv("VDJX_HIT_CHUNK")) > 0 ? (u32) atol(getenv("VDJX_HIT_CHUNK")) : HIT_CHUNK;
		for (u32 wi : ord) {
			const u32 H = (u32) (off[wi + 1] - off[wi]);
			for (u32 h0 = 0; h0 < H || h0 == 0; h0 += hit_chunk) {
				work.push_back(make_uint4(wi, h0, std::min(H, h0 + hit_chunk), 0));
				if (H == 0) break;
			}
		}
	}
	uint4* d_work;
	lp.mark("wp_worklist");
	HIP_TRY(db.alloc(&d_work, work.size()));
	HIP_TRY(hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(uint4), hipMemcpyHostToDevice, st));
	lp.mark("wp_work_upload");
	HIP_TRY(hipMemsetAsync(d_np, 0, n * 4, st));
	HIP_TRY(hipMemsetAsync(d_cnt, 0, n * 4, st));
	{
		vdjx_prof_scope ps(c, "k_window_pairs");
		hipLaunchKernelGGL(k_window_pairs, dim3((u32) work.size()), dim3(MAP_THREADS), 0, st, ix, d_w, len, d_work, d_off, (u64*) c->wp_buf, d_cnt, d_np);
	}
	HIP_TRY(hipStreamSynchronize(st));       // `work` staging dies with this frame
	lp.mark("wp_kernel_wait");
	c->stats["window_work_items"] = work.size();
	*d_np_out = d_np; *d_cnt_out = d_cnt; *d_off_out = d_off; *d_order_out = d_order;
	return VDJX_OK;
}

extern "C" int vdjx_window_score(vdjx_ctx* c, const char* windows, size_t n, int len, const vdjx_cov_params* p,
                                 uint8_t* out_valid, uint32_t* out_npairs) {
	if (!c || !p || (n && (!windows || !out_valid || !out_npairs))) { vdjx_set_error("vdjx_window_score: NULL argument"); return VDJX_EINVAL; }
	if (n == 0) return VDJX_OK;
	ReadIndexDev ix;
	int rc = make_index_view(c, &ix, len, "vdjx_window_score");
	if (rc) return rc;
	if ((rc = cov_params_check(p, "vdjx_window_score"))) return rc;
	if (n >= (1ull << 31)) { vdjx_set_error("vdjx_window_score: too many windows"); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	u32 *d_np, *d_order, *d_cnt;
	u64* d_off;
	uint8_t* d_valid;
	HIP_TRY(db.alloc(&d_valid, n));
	rc = window_pairs_run(c, db, ix, windows, n, len, &d_np, &d_cnt, &d_off, &d_order);
	if (rc) return rc;
	c->wp_n = 0;                                              // (the lists are not offered to vdjx_window_pairs_fetch)
	{
		vdjx_prof_scope ps(c, "k_window_cover");
		hipLaunchKernelGGL(k_window_cover, dim3((u32) n), dim3(MAP_THREADS), 0, st, len, ix.rl, *p, d_order, d_off, (const u64*) c->wp_buf, d_cnt, d_valid, (u64*) nullptr);
	}
	HIP_TRY(hipMemcpyAsync(out_valid, d_valid, n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(out_npairs, d_np, n * 4, hipMemcpyDeviceToHost, st));
	vdjx_laps lp(c);
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	lp.mark("ws_cover_wait");
	vdjx_prof_collect(c);
	lp.mark("ws_prof_collect");
	{
		u64 tot = 0;
		for (size_t i = 0; i < n; i++) tot += out_npairs[i];
		c->stats["window_pairs"] = tot;
	}
	return VDJX_OK;
}

// ---- the two halves of vdjx_window_score for a pool sharded BY PAIR over several GPUs: every rank maps every window against its
// own reads (a mapped pair needs both mates in one index, so the mates of a pair must live on the same rank), the lists of a
// window meet on the rank that owns the window, which runs the coverage test on their union (the validator only counts entries)
extern "C" int vdjx_window_pairs(vdjx_ctx* c, const char* windows, size_t n, int len, uint32_t* out_entries, uint32_t* out_npairs) {
	if (!c || (n && (!windows || !out_entries || !out_npairs))) { vdjx_set_error("vdjx_window_pairs: NULL argument"); return VDJX_EINVAL; }
	if (c) c->wp_n = 0;
	if (n == 0) return VDJX_OK;
	ReadIndexDev ix;
	int rc = make_index_view(c, &ix, len, "vdjx_window_pairs");
	if (rc) return rc;
	if (n >= (1ull << 31)) { vdjx_set_error("vdjx_window_pairs: too many windows"); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	u32 *d_np, *d_order, *d_cnt;
	u64* d_off;
	rc = window_pairs_run(c, db, ix, windows, n, len, &d_np, &d_cnt, &d_off, &d_order);
	if (rc) return rc;
	c->wp_cnt.resize(n);
	HIP_TRY(hipMemcpyAsync(c->wp_cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(out_npairs, d_np, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c);
	memcpy(out_entries, c->wp_cnt.data(), n * 4);
	c->wp_n = n;
	return VDJX_OK;
}

extern "C" int vdjx_window_pairs_fetch(vdjx_ctx* c, const uint32_t* window_ids, size_t m, void* d_out) {
	if (!c || (m && !window_ids)) { vdjx_set_error("vdjx_window_pairs_fetch: NULL argument"); return VDJX_EINVAL; }
	if (!c->wp_n) { vdjx_set_error("vdjx_window_pairs_fetch: call vdjx_window_pairs first"); return VDJX_ESTATE; }
	if (m == 0) return VDJX_OK;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	std::vector<u64> move(3 * m);
	u64 at = 0;
	for (size_t i = 0; i < m; i++) {
		const u32 w = window_ids[i];
		if (w >= c->wp_n) { vdjx_set_error("vdjx_window_pairs_fetch: window %u of %zu", w, c->wp_n); return VDJX_EINVAL; }
		move[3 * i] = c->wp_off[w]; move[3 * i + 1] = at; move[3 * i + 2] = c->wp_cnt[w];
		at += c->wp_cnt[w];
	}
	if (at && !d_out) { vdjx_set_error("vdjx_window_pairs_fetch: NULL buffer"); return VDJX_EINVAL; }
	if (at) {
		u64* d_move;
		HIP_TRY(db.alloc(&d_move, move.size()));
		HIP_TRY(hipMemcpyAsync(d_move, move.data(), move.size() * 8, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_gather_u64, dim3((u32) m), dim3(256), 0, st, (const u64*) c->wp_buf, d_move, (u64*) d_out);
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}

extern "C" int vdjx_window_cover(vdjx_ctx* c, size_t n, int len, int rl, const vdjx_cov_params* p, const void* d_lists, size_t nsrc,
                                 const uint32_t* counts, uint8_t* out_valid) {
	if (!c || !p || (n && (!counts || !out_valid))) { vdjx_set_error("vdjx_window_cover: NULL argument"); return VDJX_EINVAL; }
	if (n == 0) return VDJX_OK;
	if (nsrc == 0 || len <= rl || len - rl > MAP_MAXOFF || rl < 1) { vdjx_set_error("vdjx_window_cover: bad geometry"); return VDJX_EINVAL; }
	int rc = cov_params_check(p, "vdjx_window_cover");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	// source-major lists -> one list per window
	std::vector<u64> off(n + 1, 0), move(3 * n * nsrc);
	std::vector<u32> cnt(n), order(n);
	for (size_t w = 0; w < n; w++) {
		u64 t = 0;
		for (size_t s = 0; s < nsrc; s++) t += counts[s * n + w];
		if (t >= (1ull << 32)) { vdjx_set_error("vdjx_window_cover: window %zu has too many pairs", w); return VDJX_ELIMIT; }
		cnt[w] = (u32) t;
		off[w + 1] = off[w] + t;
	}
	{
		u64 src_at = 0;
		std::vector<u64> fill(off.begin(), off.end() - 1);
		for (size_t s = 0; s < nsrc; s++)
			for (size_t w = 0; w < n; w++) {
				const size_t i = s * n + w;
				move[3 * i] = src_at; move[3 * i + 1] = fill[w]; move[3 * i + 2] = counts[i];
				src_at += counts[i];
				fill[w] += counts[i];
			}
	}
	if (off[n] && !d_lists) { vdjx_set_error("vdjx_window_cover: NULL lists"); return VDJX_EINVAL; }
	order_by_size_desc(off, n, order);
	u64 *d_move, *d_merged, *d_off;
	u32 *d_cnt, *d_order;
	uint8_t* d_valid;
	HIP_TRY(db.alloc(&d_move, move.size()));
	HIP_TRY(db.alloc(&d_merged, (size_t) off[n] + 1));
	HIP_TRY(db.alloc(&d_off, n + 1));
	HIP_TRY(db.alloc(&d_cnt, n));
	HIP_TRY(db.alloc(&d_order, n));
	HIP_TRY(db.alloc(&d_valid, n));
	HIP_TRY(hipMemcpyAsync(d_move, move.data(), move.size() * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_cnt, cnt.data(), n * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_order, order.data(), n * 4, hipMemcpyHostToDevice, st));
	if (off[n]) hipLaunchKernelGGL(k_gather_u64, dim3((u32) (n * nsrc)), dim3(256), 0, st, (const u64*) d_lists, d_move, d_merged);
	{
		vdjx_prof_scope ps(c, "k_window_cover");
		hipLaunchKernelGGL(k_window_cover, dim3((u32) n), dim3(MAP_THREADS), 0, st, len, rl, *p, d_order, d_off, (const u64*) d_merged, d_cnt, d_valid, (u64*) nullptr);
	}
	HIP_TRY(hipMemcpyAsync(out_valid, d_valid, n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c);
	return VDJX_OK;
}

// identity of a contig batch between the counting and the writing call of vdjx_map_emit: FNV-1a over its head, its tail and a
// sparse sample in between (hashing every byte of a few MB twice per call cost more than the mapping kernel)
static uint64_t fnv1a(const char* p, size_t n, uint64_t h) {
	auto eat = [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) { h ^= (unsigned char) p[i]; h *= 0x100000001b3ull; } };
	if (n <= 4096) { eat(0, n); return h; }
	eat(0, 1024);
	for (size_t i = 1024; i + 8 <= n - 1024; i += 509) eat(i, i + 8);
	eat(n - 1024, n);
	return h;
}

static int map_emit_impl(vdjx_ctx* c, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs, bool async) {
	if (!c || !offsets || (n && !contigs)) { vdjx_set_error("vdjx_map_emit: NULL argument"); return VDJX_EINVAL; }
	if (n == 0) { offsets[0] = 0; return VDJX_OK; }
	if (n >= (1ull << 31)) { vdjx_set_error("vdjx_map_emit: too many contigs"); return VDJX_ELIMIT; }
	ReadIndexDev ix;
	int rc = make_index_view(c, &ix, len, "vdjx_map_emit");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	vdjx_laps lp(c);
	// the mapping runs once: the counting call keeps its pairs (and the slice bookkeeping) on the device for the writing call
	uint64_t key = fnv1a(contigs, n * (size_t) len, 0xcbf29ce484222325ull ^ (uint64_t) n * 1315423911ull ^ (uint64_t) len);
	if (!key) key = 1;
	lp.mark("me_key");
	static const u32 slice_env = getenv("VDJX_MAP_SLICE") && atol(getenv("VDJX_MAP_SLICE")) > 0 ? (u32) atol(getenv("VDJX_MAP_SLICE")) : 0u;
	if (c->me_key != key || c->me_cnt.size() != n) {
		c->me_key = 0;
		char* d_c;
		u32* d_order;
		u64* d_off;
		HIP_TRY(db.alloc(&d_c, n * len));
		HIP_TRY(hipMemcpyAsync(d_c, contigs, n * len, hipMemcpyHostToDevice, st));
		std::vector<u64> off;
		rc = plan_windows(c, db, ix, d_c, n, len, false, off, &d_order, &d_off);
		if (rc) return rc;
		lp.mark("me_plan");
		if (off[n] > c->me_cap) {
			free_set(c->me_pairs);
			c->me_cap = 0;
			HIP_TRY(hipMalloc(&c->me_pairs, (size_t) off[n] * sizeof(vdjx_pair)));
			c->me_cap = (size_t) off[n];
		}
		// slices of `slice_hits` hits, contig after contig: only their prefix over the contigs goes to the device.  Every slice pays
		// one preparation of its contig (a few hundred index probes): as long as the slices are, while ~8 k of them remain
		u32 slice_hits = slice_env;
		if (!slice_hits) {
			slice_hits = MAP_SLICE;
			while (slice_hits < MAP_SLICE_MAX && off[n] / (slice_hits * 2) >= 8192) slice_hits *= 2;
		}
		c->me_slice_hits = slice_hits;
		std::vector<u32> sstart(n + 1, 0);
		for (size_t ci = 0; ci < n; ci++) {
			const u64 sl = (off[ci + 1] - off[ci] + slice_hits - 1) / slice_hits;
			if (sstart[ci] + sl >= (1ull << 31)) { vdjx_set_error("vdjx_map_emit: too many hit slices"); return VDJX_ELIMIT; }
			sstart[ci + 1] = sstart[ci] + (u32) sl;
		}
		const size_t nsl = sstart[n];
		// persistent bookkeeping: slice prefix of the contigs, hit offsets of the contigs, pairs per slice and their prefix
		const size_t need = (n + 1) * 4 + 8 + (n + 1) * 8 + nsl * 4 + 8 + (nsl + 1) * 8 + n * 8 + 64;
		if (need > c->me_book_cap) {
			free_set(c->me_book);
			c->me_book_cap = 0;
			HIP_TRY(hipMalloc(&c->me_book, need + need / 4));
			c->me_book_cap = need + need / 4;
		}
		uint8_t* bk = (uint8_t*) c->me_book;
		u64* b_off = (u64*) bk;                      bk += (n + 1) * 8;
		u64* b_pre = (u64*) bk;                      bk += (nsl + 1) * 8;
		u64* b_cnt = (u64*) bk;                      bk += n * 8;
		u32* b_sstart = (u32*) bk;                   bk += ((n + 1) * 4 + 7) / 8 * 8;
		u32* b_scnt = (u32*) bk;
		c->me_nsl = nsl;
		c->me_cnt.assign(n, 0);
		HIP_TRY(hipMemcpyAsync(b_sstart, sstart.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(b_off, d_off, (n + 1) * 8, hipMemcpyDeviceToDevice, st));
		lp.mark("me_worklist");
		if (nsl) {
			{
				vdjx_prof_scope ps(c, "k_map_emit");
				hipLaunchKernelGGL(k_map_emit, dim3((u32) nsl), dim3(MAP_THREADS), 0, st, ix, d_c, (u32) n, len, slice_hits, b_sstart, b_off, (vdjx_pair*) c->me_pairs, b_scnt);
			}
			hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, st, b_scnt, (u32) nsl, b_pre);
			hipLaunchKernelGGL(k_contig_counts, dim3((u32) (n + 255) / 256), dim3(256), 0, st, b_pre, b_sstart, (u32) n, b_cnt);
			HIP_TRY(hipMemcpyAsync(c->me_cnt.data(), b_cnt, n * 8, hipMemcpyDeviceToHost, st));
		}
		HIP_TRY(hipStreamSynchronize(st));           // (also: `sstart` staging dies with this frame)
		HIP_TRY(hipGetLastError());
		lp.mark("me_kernel_wait");
		c->me_key = key;
		c->stats["map_hits"] = off[n];
	}
	offsets[0] = 0;
	for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + c->me_cnt[i];
	if (!pairs) return VDJX_OK;
	const u64 total = offsets[n];
	if (total) {
		const size_t nsl = c->me_nsl;
		const u32 slice_hits = c->me_slice_hits;
		uint8_t* bk = (uint8_t*) c->me_book;
		const u64* b_off = (const u64*) bk;          bk += (n + 1) * 8;
		const u64* b_pre = (const u64*) bk;          bk += (nsl + 1) * 8;
		bk += n * 8;
		const u32* b_sstart = (const u32*) bk;
		// the dense copy outlives this call when the transfer to the host is asynchronous: a buffer of its own, not the arena
		lp.mark("me_second_call");
		HIP_TRY(hipStreamSynchronize(c->pairs_stream));          // an earlier asynchronous copy may still be reading the buffer
		lp.mark("me_prev_copy_wait");
		if (total > c->me_dense_cap) {
			free_set(c->me_dense);
			c->me_dense_cap = 0;
			HIP_TRY(hipMalloc(&c->me_dense, (size_t) (total + total / 4) * sizeof(vdjx_pair)));
			c->me_dense_cap = (size_t) (total + total / 4);
		}
		vdjx_pair* d_dense = (vdjx_pair*) c->me_dense;
		if (nsl) {
			vdjx_prof_scope ps(c, "k_gather_pairs");
			hipLaunchKernelGGL(k_gather_pairs, dim3((u32) nsl), dim3(256), 0, st, (const vdjx_pair*) c->me_pairs, (u32) n, slice_hits, b_sstart, b_off, b_pre, d_dense);
		}
		if (async) {
			// the pairs cross PCIe on the copy stream beside whatever the caller does next (vdjx_map_emit_end waits for them); the
			// copy stream waits for the gather through an event, the host does not
			HIP_TRY(hipEventRecord(c->ev_gathered, st));
			HIP_TRY(hipStreamWaitEvent(c->pairs_stream, c->ev_gathered, 0));
			HIP_TRY(hipMemcpyAsync(pairs, d_dense, (size_t) total * sizeof(vdjx_pair), hipMemcpyDeviceToHost, c->pairs_stream));
			lp.mark("me_copy_issue");
		} else {
			HIP_TRY(hipMemcpyAsync(pairs, d_dense, (size_t) total * sizeof(vdjx_pair), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			HIP_TRY(hipGetLastError());
			vdjx_prof_collect(c);          // (asynchronous: the gather's timing is collected by the next call that waits for the stream)
		}
	}
	c->me_key = 0;           // one counting call serves one writing call
	return VDJX_OK;
}

extern "C" int vdjx_map_emit(vdjx_ctx* c, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs) {
	return map_emit_impl(c, contigs, n, len, offsets, pairs, false);
}

extern "C" int vdjx_map_emit_begin(vdjx_ctx* c, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs) {
	if (!pairs) { vdjx_set_error("vdjx_map_emit_begin: pairs is NULL (count with vdjx_map_emit first)"); return VDJX_EINVAL; }
	return map_emit_impl(c, contigs, n, len, offsets, pairs, true);
}

extern "C" int vdjx_map_emit_end(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_map_emit_end: ctx is NULL"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->pairs_stream));
	return VDJX_OK;
}
