// vdjx_score.hip -- the batched scorers (SURVEY §8a rows a-7 ... a-10).
//
//   K7  k_seed_count / k_root_dp      root (V-region homology) scorer         seq_score.c:92-156
//   K8  (vdjx_rindex.hip)             read index                              quick_map3.c:126-149
//       k_map_classify / k_plan / k_window_pairs / k_window_cover  read->window mapper + coverage test  quick_map3.c:188-266, coverage.c:10-130
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
	// the four arrays are kept from load to load and only replaced when one needs more (a --config4 step loads a chain's V region with
	// every pool: four hipFree + four hipMalloc + four waits each time were most of the call); one wait for the four copies
	HIP_TRY(hipStreamSynchronize(c->stream));                // nothing in flight may still read the region that is replaced
	auto keep = [](auto** p, size_t* cap, size_t bytes) -> hipError_t {
		if (*p && *cap >= bytes) return hipSuccess;
		if (*p) (void) hipFree(*p);
		*p = nullptr; *cap = 0;
		const hipError_t e = hipMalloc((void**) p, bytes + bytes / 4 + 64);
		if (e == hipSuccess) *cap = bytes + bytes / 4 + 64;
		return e;
	};
	HIP_TRY(keep(&c->d_vtext, &c->vtext_cap, text.size() + 16));
	HIP_TRY(keep(&c->d_line_off, &c->line_off_cap, off.size() * 4));
	{
		size_t cap2 = c->seed_cap;
		HIP_TRY(keep(&c->d_seed_code, &c->seed_cap, sc.size() * 4 + 4));
		HIP_TRY(keep(&c->d_seed_pos, &cap2, sc.size() * 4 + 4));
	}
	HIP_TRY(hipMemcpyAsync(c->d_vtext, text.data(), text.size(), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_line_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
	if (!sc.empty()) {
		HIP_TRY(hipMemcpyAsync(c->d_seed_code, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->d_seed_pos, sp.data(), sp.size() * 4, hipMemcpyHostToDevice, c->stream));
	}
	HIP_TRY(hipStreamSynchronize(c->stream));                // (the vectors die with this frame)
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

// (the tiling of k_scan_local: 2,048 elements per workgroup, eight consecutive ones per thread, one block prefix per workgroup; a
// thread per element with its own bpre[i / 2048] took 197 us for 1.8 M elements)
__global__ __launch_bounds__(256) void k_scan_add(u32* __restrict__ pre, u32 n, const u32* __restrict__ bpre) {
	const u32 add = bpre[blockIdx.x];
	const u32 base = blockIdx.x * 2048u + threadIdx.x * 8u;
	if (base + 8u <= n) {
		uint4* p = (uint4*) (pre + base);
		uint4 a = p[0], b = p[1];
		a.x += add; a.y += add; a.z += add; a.w += add; b.x += add; b.y += add; b.z += add; b.w += add;
		p[0] = a; p[1] = b;
	} else
		for (u32 i = base; i < n; i++) pre[i] += add;
	if (blockIdx.x == 0 && threadIdx.x == 0) pre[n] = bpre[(n + 2047u) / 2048u];
}

// one thread per (root, seed offset, hit): for every line run the (k+1) x (2k+1) DP (seq_score.c:76-116) on
// root x line[start, start+2k), start as seq_score.c:139-146.  int8 cells, match +1 / mismatch 0 / gap -1,
// first row and column 0; out[root] = 1 as soon as any cell >= threshold.
#define DP_THREADS 128
__global__ __launch_bounds__(DP_THREADS) void k_root_dp(const char* __restrict__ kmers, int k, int threshold,
                                                        const u32* __restrict__ hit_lo, const u32* __restrict__ hit_pre,
                                                        u32 n_groups, u32 stop, u32 base, const u32* __restrict__ seed_pos,
                                                        const char* __restrict__ vtext, const u32* __restrict__ line_off, u32 n_lines,
                                                        uint8_t* __restrict__ out) {
	// The DP column and the root live in REGISTERS (the row loop is fully unrolled to the longest k, guarded by the uniform
	// r <= k): a cell is a handful of integer ops instead of a round trip through LDS per cell (the kernel is a dependent chain
	// per thread and far too small to hide LDS latency by occupancy).  Only the reference segment sits in LDS, one read per column.
	__shared__ char refc[2 * VDJX_MAX_KMER * DP_THREADS];
	const u32 tid = threadIdx.x;
	const u32 w = base + blockIdx.x * DP_THREADS + tid;      // (the number of work items is read on the device: the kernel may be launched before the host has it)
	if (w >= hit_pre[n_groups]) return;
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

// The same DP, ONE WAVE per work item, for calls with few items (a small pool: a few thousand, every thread of k_root_dp then walks
// its 2,450 cells alone on a SIMD -- 0.11 ms, the longest kernel of a 100 k-pair step).  Lane r - 1 owns row r and the cells are
// taken by anti-diagonals: at step t lane r computes cell (r, t - r) from its own last value (left), its upper neighbour's last
// value (up) and that neighbour's value before (diag); the reference characters move down the lanes one per step the same way.
// k + 2k steps of a dozen instructions instead of k x 2k cells of half a dozen.
#define DPW_THREADS 256
// (The anti-diagonal sweep below moves cells down the lanes with DPP wave_shr:1 -- control 0x138, lane 0 keeps `old` --, a GFX9-family
// wave-64 control: gfx90a / gfx942 / gfx950.  This library is written for gfx950 alone; the check makes a build for anything else
// stop here instead of assembling a different shift.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "k_root_dp_wave uses the GFX9 DPP control wave_shr:1 (0x138): build with --offload-arch=gfx950"
#endif
__global__ __launch_bounds__(DPW_THREADS) void k_root_dp_wave(const char* __restrict__ kmers, int k, int threshold,
                                                             const u32* __restrict__ hit_lo, const u32* __restrict__ hit_pre,
                                                             u32 n_groups, u32 stop, u32 base, const u32* __restrict__ seed_pos,
                                                             const char* __restrict__ vtext, const u32* __restrict__ line_off, u32 n_lines,
                                                             uint8_t* __restrict__ out) {
	const int lane = __lane_id();
	const u32 w = base + blockIdx.x * (DPW_THREADS / 64) + (threadIdx.x >> 6);      // (wave-uniform)
	if (w >= hit_pre[n_groups]) return;
	u32 lo = 0, hi = n_groups;                               // last group with hit_pre[g] <= w: 64 probes a step (three dependent loads for
	while (hi - lo > 1) {                                    // 34 k groups where a binary search takes fifteen)
		const u32 step = (hi - lo + 63u) / 64u;
		const u32 idx = lo + (u32) lane * step;
		const bool le = idx < hi && hit_pre[idx] <= w;         // (monotone: true for lanes 0 .. j)
		const u32 j = (u32) __popcll(__ballot(le)) - 1u;        // (lane 0 probes `lo`, which holds)
		lo += j * step;
		hi = lo + step < hi ? lo + step : hi;
	}
	const u32 g = lo;
	const u32 root = g / stop;
	if (out[root]) return;
	const int pos = (int) seed_pos[hit_lo[g] + (w - hit_pre[g])];
	const int r = lane + 1;                                   // this lane's row (1 .. k; the others idle)
	const char qc = r <= k ? kmers[(size_t) root * k + (r - 1)] : (char) 0;
	for (u32 li = 0; li < n_lines; li++) {
		const int len = (int) (line_off[li + 1] - line_off[li]);
		int start = pos - k;
		if (start < 0) start = 0;
		if (start >= len - 2 * k) start = len - 2 * k - 1;
		const char* ref = vtext + line_off[li] + start;
		const int refA = lane < 2 * k ? (int) ref[lane] : 0, refB = 64 + lane < 2 * k ? (int) ref[64 + lane] : 0;      // the 2k reference characters, one or two per lane
		int cur = 0, up_prev = 0, best = 0;                   // this row's last cell; the upper row's cell of the step before (= this step's diagonal)
		int rc = 0;                                           // the reference character of this lane's current column
		for (int t = 2; t <= 3 * k; t++) {                    // cell (r, c = t - r): c runs 1 .. 2k for every row r <= k
			// (one DPP move each -- wave_shr:1, lane 0 gets 0 -- instead of a trip through the LDS crossbar: the two are this loop's whole dependency chain)
			const int up = __builtin_amdgcn_update_dpp(0, cur, 0x138, 0xf, 0xf, false);      // lane r - 2 holds row r - 1: its cell (r - 1, c) of the last step
			const int rc_up = __builtin_amdgcn_update_dpp(0, rc, 0x138, 0xf, 0xf, false);
			const int c = t - r;
			const int ci = t - 2;                             // (uniform) row 1's column - 1
			const int rc_new = ci < 2 * k ? (ci < 64 ? __builtin_amdgcn_readlane(refA, ci) : __builtin_amdgcn_readlane(refB, ci - 64)) : 0;
			rc = lane == 0 ? rc_new : rc_up;
			const int upv = lane == 0 ? 0 : up;               // (row 0 is zero)
			const int diag = lane == 0 ? 0 : up_prev;
			if (r <= k && c >= 1 && c <= 2 * k) {
				const int left = c == 1 ? 0 : cur;            // (column 0 is zero)
				const int dg = c == 1 ? 0 : diag;
				int v = left - 1;
				v = v > upv - 1 ? v : upv - 1;
				const int d = dg + ((int) qc == rc ? 1 : 0);
				v = v > d ? v : d;
				cur = v;
				best = best > v ? best : v;
			}
			up_prev = up;
		}
		if (__ballot(best >= threshold)) { if (lane == 0) out[root] = 1; return; }      // any cell at or above the threshold accepts (seq_score.c:103-112)
	}
}

// the scorer proper: d_k = n*k ASCII on the device, out = n host bytes
// begin: queue everything, wait for nothing -- possible when the last call's item count is at hand as a guess (root_dp_hint) and the
// threshold is positive; *begun says whether it was (else the call ran to its end as ever).  vdjx_root_score_graph_end waits, and
// repeats the call the ordinary way should the guess have fallen short.
static int root_score_device(vdjx_ctx* c, vdjx_work& db, const char* d_k, size_t n, int k, int threshold, uint8_t* out, bool* begun = nullptr, hipStream_t st = nullptr) {
	if (!st) st = c->stream;
	if (begun) *begun = false;
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
		vdjx_prof_scope ps(c, "k_seed_count", st);
		hipLaunchKernelGGL(k_seed_count, dim3((ng + 255) / 256), dim3(256), 0, st, d_k, (u32) n, k, c->vk, c->d_seed_code, (u32) c->n_seeds, d_lo, d_cnt);
	}
	{
		const u32 nb = (ng + 2047u) / 2048u;
		u32 *d_bsum, *d_bpre;
		HIP_TRY(db.alloc(&d_bsum, nb));
		HIP_TRY(db.alloc(&d_bpre, nb + 1));
		hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, st, d_cnt, ng, d_pre, d_bsum);
		hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, d_bsum, nb, d_bpre);
		hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, st, d_pre, ng, d_bpre);
	}
	// the DP is launched for as many items as the last call had before the host knows this call's number (its copy is queued first:
	// the host waits for that event only, and adds a launch for what is beyond the guess)
	u32* h_run = begun ? (u32*) c->h_pin + VDJX_HPIN_ROOT_RUN : (u32*) c->h_pin;          // (a begun call's number waits in a place no other call writes: vdjx_common.h)
	HIP_TRY(hipMemcpyAsync(h_run, d_pre + ng, 4, hipMemcpyDeviceToHost, st));
	if (!begun) HIP_TRY(hipEventRecord(c->ev_plan, st));
	// few items: a wave each (k_root_dp_wave); many: a thread each
	static const u32 dp_wave_max = getenv("VDJX_DP_WAVE_MAX") ? (u32) atol(getenv("VDJX_DP_WAVE_MAX")) : 32768u;      // (measured: 10 k items 0.043 against 0.110 ms, 25 k 0.087 / 0.111, 77 k 0.234 / 0.130, 252 k 0.74 / 0.18)
	auto launch_dp = [&](u32 first, u32 count) -> u32 {        // -> items covered from `first` on (whole workgroups)
		vdjx_prof_scope ps(c, "k_root_dp", st);
		if (count <= dp_wave_max && k <= 64) {
			const u32 per = DPW_THREADS / 64;
			hipLaunchKernelGGL(k_root_dp_wave, dim3((count + per - 1) / per), dim3(DPW_THREADS), 0, st, d_k, k, threshold,
			                   d_lo, d_pre, ng, (u32) stop, first, c->d_seed_pos, c->d_vtext, c->d_line_off, (u32) c->n_lines, d_out);
			return (count + per - 1) / per * per;
		}
		hipLaunchKernelGGL(k_root_dp, dim3((unsigned) ((count + DP_THREADS - 1) / DP_THREADS)), dim3(DP_THREADS), 0, st, d_k, k, threshold,
		                   d_lo, d_pre, ng, (u32) stop, first, c->d_seed_pos, c->d_vtext, c->d_line_off, (u32) c->n_lines, d_out);
		return (count + DP_THREADS - 1) / DP_THREADS * DP_THREADS;
	};
	u32 ahead = 0;
	if (threshold > 0 && c->root_dp_hint) ahead = launch_dp(0u, c->root_dp_hint);
	// (the verdicts follow the guessed launch at once: when the guess covered the call -- every call but the first of a size -- the host
	// wakes up once, with the verdicts there, instead of once for the number and again for the verdicts)
	if (ahead) HIP_TRY(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, st));
	if (begun && ahead && threshold > 0) { c->root_pending_ahead = ahead; *begun = true; return VDJX_OK; }      // (nothing waited for)
	if (begun) HIP_TRY(hipEventRecord(c->ev_plan, st));
	HIP_TRY(hipEventSynchronize(c->ev_plan));
	const u32 run = *h_run;
	c->root_dp_hint = run + run / 4 + 1024;
	c->stats["root_dp_items"] = run;
	if (run >= (1u << 31)) { vdjx_set_error("too many seed hits in one call (%u)", run); return VDJX_ELIMIT; }
	if (ahead && threshold > 0 && run <= ahead) {
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		vdjx_prof_collect(c, false);
		return VDJX_OK;
	}
	if (threshold <= 0) {
		// cells of row/column 0 are 0 and are tested too (seq_score.c:103-112): any seed hit accepts
		std::vector<u32> pre(ng + 1);
		HIP_TRY(hipMemcpy(pre.data(), d_pre, (size_t) (ng + 1) * 4, hipMemcpyDeviceToHost));
		for (size_t r = 0; r < n; r++) out[r] = pre[(r + 1) * stop] > pre[r * stop];
		return VDJX_OK;
	}
	if (run > ahead) (void) launch_dp(ahead, run - ahead);
	HIP_TRY(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c, false);
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

// the root k-mers laid end to end for the scorer: 16 LANES per root, lane j copies the characters j, j + 16, ... of its root -- a root's k bytes
// are a few consecutive loads and stores of a quarter wave (one thread per root was k one-byte loads and stores 35 bytes apart from lane to
// lane).  0.011 ms at 10 M pairs by HIP events.  (rocprofv3's kernel trace shows this kernel at 1.2-1.4 ms whenever the graph's copy to
// the host runs beside it on the copy stream -- its timestamps, not its time: profiles/README.md, round 6.)
__global__ void k_root_gather(const char* __restrict__ kmers, const u32* __restrict__ roots, u32 first, u32 stride, u32 n_sel, int k,
                              char* __restrict__ out, u32* __restrict__ ids) {
	const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 i = t >> 4, j0 = t & 15u;
	if (i >= n_sel) return;
	const u32 node = roots[first + (size_t) i * stride];
	if (j0 == 0) ids[i] = node + 1;
	const char* src = kmers + (size_t) node * k;
	char* dst = out + (size_t) i * k;
	for (int j = (int) j0; j < k; j += 16) dst[j] = src[j];
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
	hipLaunchKernelGGL(k_root_gather, dim3((unsigned) ((n * 16 + 255) / 256)), dim3(256), 0, c->stream, g->d_kmers, g->d_roots, first, stride, (u32) n, k, d_k, d_ids);
	HIP_TRY(hipMemcpyAsync(root_ids, d_ids, n * 4, hipMemcpyDeviceToHost, c->stream));
	memset(out, 0, n);
	if (k - c->vk <= 0 || c->n_seeds == 0) { HIP_TRY(hipStreamSynchronize(c->stream)); return VDJX_OK; }
	return root_score_device(c, db, d_k, n, k, threshold, out);
}

// The same without waiting: kernels and result copies are queued on the context's stream and the call returns; the two arrays
// (page-locked: vdjx_host_alloc) are valid after vdjx_root_score_graph_end.  Calls made in between run BEHIND it on the same stream
// (they may reuse its workspace: stream order), so a caller with other work for the device -- the window scorer does not need the
// root verdicts -- keeps it busy instead of waiting for a few kilobytes.  The graph must stay alive until _end.
extern "C" int vdjx_root_score_graph_begin(vdjx_ctx* c, const vdjx_graph* g, int threshold, uint32_t first, uint32_t stride,
                                           uint32_t* root_ids, uint8_t* out) {
	if (!c || !g) { vdjx_set_error("vdjx_root_score_graph_begin: NULL argument"); return VDJX_EINVAL; }
	if (c->root_pending) { vdjx_set_error("vdjx_root_score_graph_begin: a begun call has not been ended"); return VDJX_ESTATE; }
	if (g->ctx != c) { vdjx_set_error("vdjx_root_score_graph_begin: graph belongs to another context"); return VDJX_EINVAL; }
	if (stride == 0) { vdjx_set_error("vdjx_root_score_graph_begin: stride 0"); return VDJX_EINVAL; }
	const size_t n = vdjx_root_part(g, first, stride);
	const int k = g->k;
	int rc = root_score_check(c, "vdjx_root_score_graph_begin", n, k);
	if (rc) return rc;
	if (n == 0) return VDJX_OK;
	if (!root_ids || !out) { vdjx_set_error("vdjx_root_score_graph_begin: NULL result array"); return VDJX_EINVAL; }
	if (k - c->vk <= 0 || c->n_seeds == 0 || threshold <= 0 || !c->root_dp_hint) return vdjx_root_score_graph(c, g, threshold, first, stride, root_ids, out);
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	// VDJX_ROOT_STREAM=1: on a stream and out of a workspace of its own, behind everything the context's stream holds so far, BESIDE what the
	// caller queues on the context afterwards (the window scorer).  Measured at 10 M pairs (round 6): the step got SLOWER, 10.87 -> 11.10 ms --
	// the begin call costs the host 0.25 ms instead of 0.07 (event record + cross-stream wait + a second queue to feed) and the window
	// scorer gains 0.09 ms: the root kernels are 0.23 ms of small launches that the scorer's kernels, which fill the machine, do not
	// run beside but around.  1 M pairs: 1.955 -> 1.964 ms.  So the default stays the context's stream (the calls in between run behind it)
	static const bool own_stream = getenv("VDJX_ROOT_STREAM") && getenv("VDJX_ROOT_STREAM")[0] == '1';
	hipStream_t st = own_stream ? c->root_stream : c->stream;
	if (own_stream) {
		HIP_TRY(hipEventRecord(c->ev_root_go, c->stream));
		HIP_TRY(hipStreamWaitEvent(st, c->ev_root_go, 0));
	}
	vdjx_work db(c, own_stream ? &c->root_arena : &c->arena);      // (its own workspace: the calls in between reset and reuse the context's)
	char* d_k;
	u32* d_ids;
	HIP_TRY(db.alloc(&d_k, n * k));
	HIP_TRY(db.alloc(&d_ids, n));
	{
		vdjx_prof_scope ps(c, "k_root_gather", st);
		hipLaunchKernelGGL(k_root_gather, dim3((unsigned) ((n * 16 + 255) / 256)), dim3(256), 0, st, g->d_kmers, g->d_roots, first, stride, (u32) n, k, d_k, d_ids);
	}
	HIP_TRY(hipMemcpyAsync(root_ids, d_ids, n * 4, hipMemcpyDeviceToHost, st));
	bool begun = false;
	rc = root_score_device(c, db, d_k, n, k, threshold, out, &begun, st);
	if (rc || !begun) return rc;
	HIP_TRY(hipEventRecord(c->ev_root_done, st));
	c->root_pending = true;
	c->root_pending_g = g; c->root_pending_thr = threshold; c->root_pending_first = first; c->root_pending_stride = stride;
	c->root_pending_ids = root_ids; c->root_pending_out = out;
	return VDJX_OK;
}
extern "C" int vdjx_root_score_graph_end(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_root_score_graph_end: NULL argument"); return VDJX_EINVAL; }
	if (!c->root_pending) return VDJX_OK;
	c->root_pending = false;
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipEventSynchronize(c->ev_root_done));
	HIP_TRY(hipGetLastError());
	const u32 run = *((const u32*) c->h_pin + VDJX_HPIN_ROOT_RUN);
	c->root_dp_hint = run + run / 4 + 1024;
	c->stats["root_dp_items"] = run;
	if (run <= c->root_pending_ahead) return VDJX_OK;
	// the guess fell short (a pool unlike the last one): the call again, the ordinary way
	return vdjx_root_score_graph(c, c->root_pending_g, c->root_pending_thr, c->root_pending_first, c->root_pending_stride, c->root_pending_ids, c->root_pending_out);
}

// (a-8 read index: vdjx_rindex.hip)

// ==============================================================================================
// a-8/a-9/a-10 mapper core shared by k_window_pairs and k_map_emit
// ==============================================================================================
#define MAP_THREADS 512
#define MAP_MAXOFF 1024          // window/contig length - rl  <= MAP_MAXOFF
#define WT_SLOTS 2048            // LDS table: read class -> last offset of the window where it occurs
#define MAP_PRESENT_LOG2 15
#define MAP_PRESENT_WORDS (1u << (MAP_PRESENT_LOG2 - 5))

struct ReadIndexDev {                             // vdjx_rindex.hip
	const u64* tab; u32 mask;                     // per slot: the read sequence (W words), then the class's data (k_ri_tab, vdjx_rindex.hip)
	const u32* start; const u32* cnt1; const u32* recs;
	const u64* csr8; const u32* csr_pair; const u32* pair_r2;     // per CSR member: 8-byte entry, pair id; per pair: its read-2 records
	const u32* dstart; const u64* d8;             // distinct read-1 entries per class with multiplicities (window scoring)
	int rl;
	bool canon;                                   // the table's slots are pairs {sequence, reverse complement} (k_ri_tab_canon)
	u32 epoch;                                    // ... taken iff their claim word's top bits hold this number
};
__device__ inline u32 ent_a(u64 e) { return (u32) e & RI_ENT_NONE; }
__device__ inline u32 ent_b(u64 e) { return (u32) (e >> 26) & RI_ENT_NONE; }
__device__ inline u32 ent_flags(u64 e) { return (u32) (e >> 52) & 15u; }
__device__ inline u32 ent_count(u64 e) { return (u32) (e >> 56); }

// the image of one classified string in LDS, sized for at most NOFF offsets (512: windows of 486 and contigs of 360 bases at any read
// length; 1024: the limit of the interface)
template <int NOFF> struct MapImg {
	u32 cstart[NOFF];                               // first entry of the class at the offset
	u32 hpre[NOFF + 1];                             // prefix of class sizes (hits enumerate in reference order)
	u32 wt_key[2 * NOFF];                           // class id + 1 -> ...
	u32 wt_last[2 * NOFF];                          // ... last offset + 1 with that class
	u32 present[MAP_PRESENT_WORDS];                 // one bit per (low bits of the) class id seen in the string: "is this mate class here at all?" is one
	                                                // LDS word for the 98 % of the hits whose mate lies elsewhere (a V gene is shared by many clones)
};

// class ids are ranks in the order of a hash table's slots: their low bits are as good as a hash (and cost no multiplication)
__device__ inline u32 map_present_bit(u32 cls) { return cls & ((1u << MAP_PRESENT_LOG2) - 1u); }
// ... and a second bit per class from a multiplicative hash (round 4): an entry goes to the full test only if BOTH bits of a mate class
// are set -- with ~440 classes in 32 Kbit one bit alone is wrong for 1.3 % of the absent classes, which at two mates per entry was
// as many queue entries as the true ones; both bits: 0.07 %
__device__ inline u32 map_present_bit2(u32 cls) { return (cls * 0x9E3779B1u) >> (32 - MAP_PRESENT_LOG2); }

// ---- classification: every offset o in [0, len-rl) of every string (quick_map3.c:200: the last offset is never looked at) -> its
// read class, the class's entries and how many (weighted: the DISTINCT read-1 entries, else the read-1 members), kept in HBM
// (16 bytes per offset) for the kernels that evaluate the hits: a string is looked up in the index ONCE per call.
// One workgroup per string (looping), one thread per offset; the string is staged in LDS.
template <int W>
__global__ __launch_bounds__(MAP_THREADS) void k_map_classify(ReadIndexDev ix, const char* __restrict__ strings, u32 n, int len, bool weighted,
                                                              uint4* __restrict__ prep, u32* __restrict__ out_hits, u32* __restrict__ out_inst,
                                                              u64* __restrict__ out_gkey, u32* __restrict__ out_gidx, u32 w0) {
	constexpr int TXT = MAP_MAXOFF + VDJX_MAX_READ_LEN;
	__shared__ u64 wimg[TXT / 32 + 8];                      // the string as 2-bit codes, 32 bases per word, first base most significant
	__shared__ u64 bimg[TXT / 64 + 4];                      // bit i: character i is not ACGT
	__shared__ u32 s_h[MAP_THREADS / 64], s_i[MAP_THREADS / 64];
	__shared__ u64 s_b1[MAP_THREADS / 64], s_b2[MAP_THREADS / 64];
	constexpr int SW = VDJX_RI_SLOT_WORDS(W);
	const int rl = ix.rl, noff = len - rl;
	const u32 tid = threadIdx.x;
	for (u32 wi = w0 + blockIdx.x; wi < n; wi += gridDim.x) {         // strings [w0, n)
		const char* w = strings + (size_t) wi * len;
		for (u32 i = tid; i < TXT / 32 + 8; i += MAP_THREADS) wimg[i] = 0;
		for (u32 i = tid; i < TXT / 64 + 4; i += MAP_THREADS) bimg[i] = 0;
		__syncthreads();
		// every character once: its code into the word image (a per-offset loop over rl characters did this 436 times over)
		for (int i = tid; i < len; i += MAP_THREADS) {
			const int cde = base_code(w[i]);
			if (cde < 0) atomicOr((unsigned long long*) &bimg[i >> 6], 1ull << (i & 63));
			else if (cde) atomicOr((unsigned long long*) &wimg[i >> 5], (unsigned long long) cde << (62 - 2 * (i & 31)));
		}
		__syncthreads();
		u32 hs = 0, is = 0;
		u64 b1 = 0, b2 = 0;                                    // deepest class of the string's first / second half: entries << 26 | ~class
		for (int o = tid; o < noff; o += MAP_THREADS) {
			// any character of [o, o + rl) that is not ACGT?
			bool ok = true;
			for (int a = 0; a < rl; a += 64) {
				const int i = (o + a) >> 6;
				const u32 sh = (u32) (o + a) & 63u;
				u64 m = (bimg[i] >> sh) | ((bimg[i + 1] << 1) << (63u - sh));
				if (rl - a < 64) m &= (1ull << (rl - a)) - 1ull;
				if (m) ok = false;
			}
			// the read-length string at the offset in the pool's record format (vdjx_pool): the words from base o on
			u64 key[W];
			{
				const int wi0 = o >> 5;
				const u32 s2 = 2u * ((u32) o & 31u);
				if (W == 2) {                       // right-aligned 2*rl-bit integer in (hi, lo)
					const u64 x0 = wimg[wi0], x1 = wimg[wi0 + 1], x2 = wimg[wi0 + 2];
					const u64 hi = (x0 << s2) | ((x1 >> 1) >> (63u - s2)), lo = (x1 << s2) | ((x2 >> 1) >> (63u - s2));
					const u32 r = 128u - 2u * (u32) rl;                  // 0 .. 126 (uniform)
					if (r >= 64u) { key[1] = hi >> (r - 64u); key[0] = 0; }
					else if (r) { key[1] = (lo >> r) | (hi << (64u - r)); key[0] = hi >> r; }
					else { key[1] = lo; key[0] = hi; }
				} else {                            // W words, left-aligned, nothing behind base rl
#pragma unroll
					for (int q = 0; q < W; q++) {
						u64 x = (wimg[wi0 + q] << s2) | ((wimg[wi0 + q + 1] >> 1) >> (63u - s2));
						const int left = 2 * rl - 64 * q;                // bits of this word that belong to the read
						if (left <= 0) x = 0; else if (left < 64) x &= ~0ull << (64 - left);
						key[q] = x;
					}
				}
			}
			u32 cls = NONE32, cs = 0, sz = 0, inst = 0;
			if (ok && W == 2 && ix.canon) {
				// pools of couples: the slot of the pair {string, reverse complement} lies under the smaller of the two; the odd class is the larger one's
				u64 rh, rlo;
				vdjx_read_rc(key[0], key[1], rl, rh, rlo);
				const bool odd = rh < key[0] || (rh == key[0] && rlo < key[1]);
				u64 ck[2];
				ck[0] = odd ? rh : key[0];
				ck[1] = odd ? rlo : key[1];
				u32 slot = (u32) (ri_hash<2>(ck) >> 17) & ix.mask;
				for (;;) {
					const u64* sl = ix.tab + (size_t) slot * 8;
					const ulonglong2 kk = ((const ulonglong2*) sl)[0];
					const ulonglong2 w23 = ((const ulonglong2*) sl)[1];
					if (((u32) w23.x >> RI_TAB_EPOCH_SHIFT) != ix.epoch) break;
					if (kk.x == ck[0] && kk.y == ck[1]) {
						const u64 a = sl[odd ? 5 : 4];
						cls = 2u * (((u32) w23.x & ((1u << RI_TAB_EPOCH_SHIFT) - 1u)) - 1u) + (odd ? 1u : 0u);
						inst = odd ? (u32) (w23.y >> 32) : (u32) w23.y;
						cs = (u32) a;
						sz = weighted ? (u32) (a >> 32) : inst;
						break;
					}
					slot = (slot + 1) & ix.mask;
				}
			} else if (ok) {
				u32 slot = (u32) (ri_hash<W>(key) >> 17) & ix.mask;
				for (;;) {
					const u64* sl = ix.tab + (size_t) slot * SW;
					const u64 cw = sl[W];
					if (!(u32) cw) break;
					u64 d = 0;
#pragma unroll
					for (int q = 0; q < W; q++) d |= sl[q] ^ key[q];
					if (!d) {
						const u64 a = sl[W + 1];
						cls = (u32) cw - 1; inst = (u32) (cw >> 32);
						cs = (u32) a;                                  // (the weighted entries lie where the class's CSR members do)
						sz = weighted ? (u32) (a >> 32) : inst;
						break;
					}
					slot = (slot + 1) & ix.mask;
				}
			}
			prep[(size_t) wi * noff + o] = make_uint4(cls, cs, sz, inst);
			hs += sz; is += inst;
			if (sz) {
				const u64 v = ((u64) sz << 26) | (u64) (RI_ENT_NONE - cls);
				if (2 * o < noff) b1 = v > b1 ? v : b1; else b2 = v > b2 ? v : b2;
			}
		}
		b1 = vdjx_wave_max64(b1); b2 = vdjx_wave_max64(b2);
		hs = (u32) __builtin_amdgcn_readlane(vdjx_wave_scan_add((int) hs), 63);
		is = (u32) __builtin_amdgcn_readlane(vdjx_wave_scan_add((int) is), 63);
		if ((tid & 63u) == 0) { s_h[tid >> 6] = hs; s_i[tid >> 6] = is; s_b1[tid >> 6] = b1; s_b2[tid >> 6] = b2; }
		__syncthreads();
		if (tid == 0) {
			u32 a = 0, bb = 0;
			u64 m1 = 0, m2 = 0;
			for (u32 v = 0; v < MAP_THREADS / 64; v++) { a += s_h[v]; bb += s_i[v]; m1 = s_b1[v] > m1 ? s_b1[v] : m1; m2 = s_b2[v] > m2 ? s_b2[v] : m2; }
			out_hits[wi] = a;
			out_inst[wi] = bb;
			if (out_gkey) {
				// strings that share their deepest classes (the reads of a V gene, of a J gene: hundreds of clones) are neighbours under
				// this key, the deepest ones first: 6 bits of depth, 16 bits of a hash of the two classes (k_group_pairs)
				const u64 deep = (m1 > m2 ? m1 : m2) >> 26;
				const u32 depth = deep ? 64u - (u32) __builtin_clzll(deep) : 0u;            // 0 .. 32
				const u32 c1 = (u32) m1 & RI_ENT_NONE, c2 = (u32) m2 & RI_ENT_NONE;
				out_gkey[wi] = ((u64) (63u - depth) << 16) | (vdjx_mix(c1, c2) >> 48);
				out_gidx[wi] = wi;
			}
		}
		__syncthreads();
	}
}

// the classified offsets of one string -> the workgroup's LDS image: first entry / prefix of sizes per offset, the class -> last offset
// table ("read2[id] = m_info": the last writer wins, quick_map3.c:214) and the presence bits.  Returns the hit count.
template <int NOFF>
__device__ inline u32 map_load_prep(MapImg<NOFF>& L, const uint4* __restrict__ prow, int noff) {
	const u32 tid = threadIdx.x;
	for (u32 i = tid; i < 2 * NOFF; i += MAP_THREADS) { L.wt_key[i] = 0; L.wt_last[i] = 0; }
	for (u32 i = tid; i < MAP_PRESENT_WORDS; i += MAP_THREADS) L.present[i] = 0;
	__syncthreads();
	for (int o = tid; o < noff; o += MAP_THREADS) {
		const uint4 p = prow[o];
		const u32 cls = p.x;
		if (cls != NONE32) {
			const u32 pb = map_present_bit(cls), pb2 = map_present_bit2(cls);
			atomicOr(&L.present[pb >> 5], 1u << (pb & 31));
			atomicOr(&L.present[pb2 >> 5], 1u << (pb2 & 31));
			u32 slot = cls & (2 * NOFF - 1);
			for (;;) {
				u32 cur = L.wt_key[slot];
				if (cur == 0) {
					cur = atomicCAS(&L.wt_key[slot], 0u, cls + 1);
					if (cur == 0) cur = cls + 1;
				}
				if (cur == cls + 1) { atomicMax(&L.wt_last[slot], (u32) o + 1); break; }
				slot = (slot + 1) & (2 * NOFF - 1);
			}
		}
		L.cstart[o] = p.y;
		L.hpre[o] = p.z;
	}
	__syncthreads();
	// exclusive prefix of the class sizes: wave 0, consecutive offsets per lane
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

// last offset+1 at which a read of class `cls` occurs in the string, or 0
template <int NOFF>
__device__ inline u32 map_last_occurrence(const MapImg<NOFF>& L, u32 cls) {
	u32 slot = cls & (2 * NOFF - 1);
	for (;;) {
		const u32 cur = L.wt_key[slot];
		if (cur == 0) return 0;
		if (cur == cls + 1) return L.wt_last[slot];
		slot = (slot + 1) & (2 * NOFF - 1);
	}
}

// the cheap half of the test of an entry: is either mate class in the string at all?  (A "none" class tests some bit like any other:
// a false positive the full test sorts out.)
template <int NOFF>
__device__ inline bool map_entry_present(const MapImg<NOFF>& L, const u64 e) {
	const u32 ca = ent_a(e), cb = ent_b(e);
	const u32 ba = map_present_bit(ca), bb = map_present_bit(cb);
	const u32 ha = (L.present[ba >> 5] >> (ba & 31)) & 1u, hb = (L.present[bb >> 5] >> (bb & 31)) & 1u;
	if (!(ha | hb)) return false;                          // (nearly every entry leaves here: one look per mate)
	const u32 ba2 = map_present_bit2(ca), bb2 = map_present_bit2(cb);
	return ((ha & (L.present[ba2 >> 5] >> (ba2 & 31))) | (hb & (L.present[bb2 >> 5] >> (bb2 & 31)))) & 1u;
}
// the full test: the entry of the class at offset o -> mapped pair or not (quick_map3.c:223-245).  read2[id]: among the pair's read-2
// records the one written last = largest offset, then latest registration (B)
template <int NOFF>
__device__ inline bool map_eval_entry(const MapImg<NOFF>& L, int rl, int o, const u64 e, u32& pos2_out, u32& which_out) {
	const u32 ca = ent_a(e), cb = ent_b(e);
	const u32 la = ca != RI_ENT_NONE ? map_last_occurrence(L, ca) : 0u, lb = cb != RI_ENT_NONE ? map_last_occurrence(L, cb) : 0u;
	if (!(la | lb)) return false;
	const u32 which = (lb && lb >= la) ? 1u : 0u;
	const u32 best = which ? lb : la;
	const u32 fl = ent_flags(e);
	const u32 rc1 = (fl & RI_RC) ? 1u : 0u;
	const u32 rc2 = (fl & (which ? RI_RCB : RI_RCA)) ? 1u : 0u;
	if (rc1 == rc2) return false;                  // quick_map3.c:227
	const int d = (o + 1) - (int) best;
	const int insert = (int) (short) ((d < 0 ? -d : d) + rl);
	if (insert < 50 || insert > 400) return false; // MIN_INSERT / MAX_INSERT, quick_map3.c:23-24
	pos2_out = best;
	which_out = which;
	return true;
}

// ----------------------------------------------------------------------------------------------
// planning on the device: offsets of the strings' hit lists, processing order (largest first), work items
// ----------------------------------------------------------------------------------------------
// entry b of a prefix array belongs to the position j with start[j] <= b < start[j + 1]
__device__ inline u32 slice_contig(const u32* __restrict__ slice_start, u32 n, u32 b) {
	u32 lo = 0, hi = n;
	while (hi - lo > 1) {
		const u32 mid = (lo + hi) >> 1;
		if (slice_start[mid] <= b) lo = mid; else hi = mid;
	}
	return lo;
}

#define MAP_SLICE 4096u          // hits per slice of k_map_emit at least ...
#define MAP_SLICE_MAX 32768u     // ... and at most: as long as ~8 k slices remain (every slice loads its contig's image once)
struct PlanOut { u64 total_hits, inst_total; u32 inst_max, chunk, nwork, pad; };

__device__ inline u32 plan_block_scan(u32 v, u32* tmp, u32& total) {       // exclusive prefix over the threads of the workgroup (all call it; at most 1024)
	const u32 incl = (u32) vdjx_wave_scan_add((int) v);
	__syncthreads();
	if ((threadIdx.x & 63u) == 63u) tmp[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 base = 0, tot = 0;
	for (u32 w = 0; w < blockDim.x / 64; w++) { const u32 x = tmp[w]; if (w < (threadIdx.x >> 6)) base += x; tot += x; }
	total = tot;
	return base + incl - v;
}

// One workgroup: off[i] = hits before string i (u64), order[] = strings by descending bit length of their hit count (sorted) or as
// they come, wstart[j] = work items before position j of that order, where string i has ceil(hits / chunk) of them (at least one if
// min_one).  chunk_fixed == 0: the slice length of k_map_emit, chosen from the total.  A stable_sort of 20,000 windows on the host,
// the list of their slices and three transfers cost more than the coverage kernel ran.
// (One workgroup; every thread owns a run of consecutive strings and walks it five times.  `staged`: the hit counts are first copied
// into dynamic LDS with consecutive lanes on consecutive addresses -- read from global memory a thread's run was a chain of line
// fills per pass: 105 us for 20,000 windows.)
__global__ __launch_bounds__(1024) void k_plan(const u32* __restrict__ hits_g, const u32* __restrict__ inst, u32 n, u32 chunk_fixed, int sorted, int min_one,
                                               u64* __restrict__ off, u32* __restrict__ order, u32* __restrict__ wstart, PlanOut* __restrict__ out, int staged) {
	extern __shared__ u32 hits_l[];
	if (staged) {
		for (u32 i = threadIdx.x; i < n; i += 1024) hits_l[i] = hits_g[i];
		__syncthreads();
	}
	const u32* hits = staged ? hits_l : hits_g;
	__shared__ u64 part[1024];
	__shared__ u32 tmp[16];
	__shared__ u64 s_inst;
	__shared__ u32 s_imax, s_chunk;
	const u32 tid = threadIdx.x;
	const u32 per = (n + 1023) / 1024;
	const u32 lo = tid * per < n ? tid * per : n;
	const u32 hi = lo + per < n ? lo + per : n;
	if (tid == 0) { s_inst = 0; s_imax = 0; }
	u64 s = 0, is = 0;
	u32 im = 0;
	for (u32 i = lo; i < hi; i++) s += hits[i];
	for (u32 i = tid; i < n; i += 1024) { const u32 x = inst[i]; is += x; im = im > x ? im : x; }      // (sum and maximum: any order)
	part[tid] = s;
	__syncthreads();
	for (u32 d = 1; d < 1024; d <<= 1) {
		const u64 v = tid >= d ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	const u64 total = part[1023];
	u64 run = part[tid] - s;
	for (u32 i = lo; i < hi; i++) { off[i] = run; run += hits[i]; }
	if (is) atomicAdd((unsigned long long*) &s_inst, (unsigned long long) is);
	if (im) atomicMax(&s_imax, im);
	if (tid == 0) {
		off[n] = total;
		u32 chunk = chunk_fixed;
		if (!chunk) {
			chunk = MAP_SLICE;
			while (chunk < MAP_SLICE_MAX && total / ((u64) chunk * 2) >= 8192) chunk *= 2;
		}
		s_chunk = chunk;
	}
	__syncthreads();
	const u32 chunk = s_chunk;
	// order: by descending bit length of the hit count (33 buckets; inside a bucket any order serves: the order only decides which
	// workgroup starts first, never a result)
	if (sorted) {
		__shared__ u32 bcnt[33], bcur[33];
		if (tid < 33) bcnt[tid] = 0;
		__syncthreads();
		for (u32 i = lo; i < hi; i++) { const u32 h = hits[i]; atomicAdd(&bcnt[32 - (h ? 32 - __clz(h) : 0)], 1u); }
		__syncthreads();
		if (tid == 0) { u32 run = 0; for (int b = 0; b < 33; b++) { bcur[b] = run; run += bcnt[b]; } }
		__syncthreads();
		for (u32 i = lo; i < hi; i++) { const u32 h = hits[i]; order[atomicAdd(&bcur[32 - (h ? 32 - __clz(h) : 0)], 1u)] = i; }
		__threadfence_block();
		__syncthreads();
	} else
		for (u32 i = lo; i < hi; i++) order[i] = i;
	__syncthreads();
	// work items by position
	u32 wc = 0;
	for (u32 j = lo; j < hi; j++) {
		const u32 h = hits[sorted ? order[j] : j];
		const u32 it = (h + chunk - 1) / chunk;
		wc += (min_one && !it) ? 1u : it;
	}
	u32 wtot;
	u32 wrun = plan_block_scan(wc, tmp, wtot);
	for (u32 j = lo; j < hi; j++) {
		const u32 h = hits[sorted ? order[j] : j];
		const u32 it = (h + chunk - 1) / chunk;
		wstart[j] = wrun;
		wrun += (min_one && !it) ? 1u : it;
	}
	if (tid == 0) {
		wstart[n] = wtot;
		out->total_hits = total; out->inst_total = s_inst; out->inst_max = s_imax; out->chunk = chunk; out->nwork = wtot; out->pad = 0;
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
#ifndef COVER_CHUNK0
#define COVER_CHUNK0 512u               // first replayed chunk of rule 2 (then doubling): a few hundred scattered entries cover a valid window (4096: 0.82 ms, 512: 0.77)
#endif
#ifndef COVER_RULE2
#define COVER_RULE2 true             // (ablation: -DCOVER_RULE2=false)
#endif
#ifndef COVER_BITS
#define COVER_BITS true              // (-DCOVER_BITS=false: floor 1 through the difference arrays too)
#endif
#define HIT_CHUNK 524288u        // hits per workgroup of k_window_pairs: only the very deepest windows are split (every piece loads the
                                 // window's image once; measured at 10 M pairs: 65536 -> 5.2 ms, 262144 and above -> 3.5 ms)

// K8: mapped pairs of a slice of a window's (distinct) hits, appended (any order) to the window's list as
// (multiplicity << 32 | pos1 << 16 | pos2); pair_np = mapped pairs counted with multiplicity.  Work item b is slice b - wstart[j] of
// the window at position j of the processing order (binary search over the prefix: the slices are never listed).
// The waves of the workgroup take the offsets of the slice one at a time (LDS ticket): the entries of an offset's read class
// are consecutive, so a wave streams them with coalesced 8-byte loads and no search; pairs (a few per cent of the entries) are
// staged per wave in LDS and leave with one global cursor bump per 64.
#define WP_K 16                  // rows of 64 consecutive hits per wave and round: 16 loads in flight per lane, in 64 registers (8 waves per SIMD:
                                 // with 96 registers and 5 waves the same kernel ran 1.7x longer)
#define WP_Q 128                 // per-wave queue of the entries that passed the presence test
// work item -> position of its window in the processing order (what slice_contig finds by a search over wstart: with the windows of
// the groups done, tens of thousands of k_window_pairs' workgroups are there only to find that out)
__global__ void k_work_items(const u32* __restrict__ wstart, u32 n, u32* __restrict__ witem) {
	const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	for (u32 b = wstart[j]; b < wstart[j + 1]; b++) witem[b] = j;
}
template <int NOFF>
__global__ __launch_bounds__(MAP_THREADS, 8) void k_window_pairs(ReadIndexDev ix, const uint4* __restrict__ prep, u32 n, int len, u32 chunk,
                                                              const u32* __restrict__ order, const u32* __restrict__ wstart,
                                                              const u64* __restrict__ pair_off, u64* __restrict__ pair_buf,
                                                              u32* __restrict__ pair_cnt, u32* __restrict__ pair_np, const u32* __restrict__ done,
                                                              const unsigned long long* __restrict__ groups_left, const u32* __restrict__ witem) {
	__shared__ MapImg<NOFF> L;
	__shared__ u64 q_ent[MAP_THREADS / 64][WP_Q];
	__shared__ unsigned short q_off[MAP_THREADS / 64][WP_Q];
	__shared__ u32 s_next;
	if (groups_left && *groups_left == 0) return;       // (k_group_pairs did every window: one cached word instead of a search per workgroup)
	const u32 j = witem ? witem[blockIdx.x] : slice_contig(wstart, n, blockIdx.x);
	const u32 wi = order[j];
	if (done && done[wi]) return;                       // (k_group_pairs)
	const int noff = len - ix.rl;
	const u32 tid = threadIdx.x;
	const u32 h0 = (blockIdx.x - wstart[j]) * chunk;
	if (tid == 0) s_next = h0;
	const u32 H = map_load_prep(L, prep + (size_t) wi * noff, noff);
	const u32 h1 = h0 + chunk < H ? h0 + chunk : H;
	if (h0 >= h1) return;
	const int lane = __lane_id();
	const u32 wv = tid >> 6;
	u64* pairs = pair_buf + pair_off[wi];
	u32 qn = 0, mine = 0;                               // (qn: wave-uniform)
	// the queued entries, 64 at a time with every lane busy: the full test and the pairs out
	auto drain = [&]() {
		vdjx_wave_lds_fence();                          // (the queue was written by other lanes of the wave)
		for (u32 q0 = 0; q0 < qn; q0 += 64) {
			const u32 qi = q0 + (u32) lane;
			bool pr = false;
			u64 e = 0;
			u32 pos2 = 0, which = 0, o = 0;
			if (qi < qn) {
				e = q_ent[wv][qi]; o = q_off[wv][qi];
				pr = map_eval_entry(L, ix.rl, (int) o, e, pos2, which);
			}
			const u64 m = __ballot(pr);
			if (m) {
				u32 base = 0;
				if (lane == 0) base = atomicAdd(&pair_cnt[wi], (u32) __popcll(m));
				base = (u32) __builtin_amdgcn_readlane((int) base, 0);
				if (pr) {
					const u32 mult = ent_count(e);
					pairs[base + (u32) __popcll(m & ((1ull << lane) - 1ull))] = ((u64) mult << 32) | ((o + 1) << 16) | pos2;
					mine += mult;
				}
			}
		}
		vdjx_wave_lds_fence();
		qn = 0;
	};
	// The hits of the slice are one flat sequence (offset-major); a wave takes WP_K rows of 64 consecutive hits per round: WP_K
	// loads in flight per lane whatever the class sizes (a round per OFFSET waited for memory 436 times per window).  98 % of the
	// entries fail the presence test "is either mate class in the window at all?" (a V gene is shared by hundreds of clones): that
	// test is all the sparse sweep does, straight-line; the few that pass are queued and finished 64 at a time (the rest of the
	// test inline ran for one or two lanes of nearly every row: the kernel was bound by instruction issue).
	for (;;) {
		u32 hb = 0;
		if (lane == 0) hb = atomicAdd(&s_next, 64u * WP_K);
		hb = (u32) __builtin_amdgcn_readlane((int) hb, 0);
		if (hb >= h1) break;
		// the offset of every row's first hit: lanes 0 .. WP_K-1 search, one row each
		int orow = 0;
		{
			const u32 hr = hb + (u32) lane * 64u;
			if (lane < WP_K && hr < h1) {
				int lo = 0, hi = noff;                      // last offset with hpre[o] <= hr
				while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.hpre[mid] <= hr) lo = mid; else hi = mid; }
				orow = lo;
			}
		}
		u64 ent[WP_K];
		u32 oo[WP_K / 2];                               // the hits' offsets, two per register
#pragma unroll
		for (int u = 0; u < WP_K; u++) {
			const u32 h = hb + (u32) (u * 64 + lane);
			int o = __builtin_amdgcn_readlane(orow, u);
			u64 e = ~0ull;
			if (h < h1) {
				while (h >= L.hpre[o + 1]) o++;                 // (few steps: a row of 64 hits spans few classes; empty ones are skipped)
				e = ix.d8[L.cstart[o] + (h - L.hpre[o])];
			}
			ent[u] = e;
			if (u & 1) oo[u >> 1] |= (u32) o << 16; else oo[u >> 1] = (u32) o;
		}
#pragma unroll
		for (int u = 0; u < WP_K; u++) {
			if (hb + (u32) (u * 64) >= h1) break;                                // (wave-uniform)
			const bool pass = map_entry_present(L, ent[u]) && hb + (u32) (u * 64 + lane) < h1;
			const u64 m = __ballot(pass);
			if (!m) continue;
			if (pass) {
				const u32 at = qn + (u32) __popcll(m & ((1ull << lane) - 1ull));
				q_ent[wv][at] = ent[u];
				q_off[wv][at] = (unsigned short) ((oo[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
			}
			qn += (u32) __popcll(m);
			if (qn > WP_Q - 64) drain();
		}
	}
	drain();
	mine = (u32) __builtin_amdgcn_readlane(vdjx_wave_scan_add((int) mine), 63);
	if (lane == 0 && mine) atomicAdd(&pair_np[wi], mine);
}

// K8 for GP_G strings at a time ("class-major": VERDICT r2 item 4).  A V gene is shared by hundreds of clones and a J gene by thousands:
// the windows of such clones consist of the same read classes but for the CDR3 and a few mutations, and k_window_pairs streams the
// same entry lists -- the deep ones -- once per window: 1.85 G entries at 10 M pairs, a kernel bound by instruction issue.  Here
// a workgroup takes GP_G windows that are neighbours under k_map_classify's key, numbers their DISTINCT classes, streams every
// class's entries ONCE against the union of the windows' presence bits, and only the entries that pass (a few per cent) are
// taken to the windows that hold their class: the occurrences (window, offset) of every class are listed by window, so "the
// last offset of mate class A in window w" is a walk over A's list in step with the walk over the entry's own.  Pair lists are
// appended through LDS counters (a group is never split), any order (k_window_cover only counts).
// A group whose distinct classes do not fit (unrelated windows) is left alone: k_window_pairs, launched after this kernel,
// does every window that is not marked done.
#define GP_G 8
#ifndef GP_THREADS                  // measured at 10 M pairs (threads / waves per SIMD / rows per round): 1024/8/16 1.94 ms (21 registers spilled), 1024/8/8 1.47,
#define GP_THREADS 768              // 512/4/16 1.51, 768/6/8 1.25, 768/6/12 1.21: two workgroups per CU (LDS), 70 registers
#define GP_WAVES 6
#define GP_K 12
#define GP_Q 128                   // the queue of a wave: drained in full rows of 64 (GP_Q - 64 entries wait at most)
#endif
#define GP_TS 4096u                 // class table slots
#define GP_DMAX 2048u               // distinct classes of a group
#define GP_NOFF 512
#define GP_OCC (GP_G * GP_NOFF)
#define GP_PRESENT_LOG2 16
#ifndef GP_XCD_TILE
#define GP_XCD_TILE 1                // (0: groups to workgroups as they come: 1.22-1.27 ms against 1.19-1.20; the same for k_map_emit's slices: no difference)
#endif
static_assert(GP_Q >= 128, "a drain of full rows must find one");
struct GroupImg {
	u32 key[GP_TS];                                 // class + 1 -> ...
	unsigned short didx[GP_TS];                     // ... its number in the group
	u32 hpre[GP_DMAX + 1];                          // per numbered class: prefix of the entry counts (the flat hit sequence) ...
	u32 cstart[GP_DMAX];                            // ... and its first entry
	unsigned short occ_start[GP_DMAX + 2];          // its occurrences: occ[occ_start[d] .. occ_start[d + 1]), by window
	unsigned short wmask[GP_DMAX];                  // bit w: window w holds the class; bit 15: some window holds it at more than one offset
	unsigned short occ[GP_OCC];                     // window slot << 10 | offset
	u32 present[1u << (GP_PRESENT_LOG2 - 5)];
};
__device__ inline int gp_lookup(const GroupImg& L, u32 cls) {
	u32 slot = cls & (GP_TS - 1);
	for (;;) {
		const u32 cur = L.key[slot];
		if (cur == 0) return -1;
		if (cur == cls + 1) return (int) L.didx[slot];
		slot = (slot + 1) & (GP_TS - 1);
	}
}
__device__ inline u32 gp_present_bit2(u32 cls) { return (cls * 0x9E3779B1u) >> (32 - GP_PRESENT_LOG2); }      // (see map_present_bit2)
__device__ inline bool gp_entry_present(const GroupImg& L, const u64 e) {
	const u32 ca = ent_a(e), cb = ent_b(e);
	const u32 ba = ca & ((1u << GP_PRESENT_LOG2) - 1u), bb = cb & ((1u << GP_PRESENT_LOG2) - 1u);
	const u32 ha = (L.present[ba >> 5] >> (ba & 31)) & 1u, hb = (L.present[bb >> 5] >> (bb & 31)) & 1u;
	if (!(ha | hb)) return false;
	const u32 ba2 = gp_present_bit2(ca), bb2 = gp_present_bit2(cb);
	return ((ha & (L.present[ba2 >> 5] >> (ba2 & 31))) | (hb & (L.present[bb2 >> 5] >> (bb2 & 31)))) & 1u;
}

__global__ __launch_bounds__(GP_THREADS, GP_WAVES) void k_group_pairs(ReadIndexDev ix, const uint4* __restrict__ prep, const u32* __restrict__ hits, u32 n, int len,
                                                              const u32* __restrict__ gorder, const u64* __restrict__ pair_off, u64* __restrict__ pair_buf,
                                                              u32* __restrict__ pair_cnt, u32* __restrict__ pair_np, u32* __restrict__ done, unsigned long long* __restrict__ gstat, u32 dbg,
                                                              u64 cap) {
	if (pair_off[n] + 1 > cap) return;                  // (launched before the host knew the lists' total size: they do not fit, it will come again)
	const long long t_begin = clock64();
	__shared__ GroupImg L;
	__shared__ u64 q_ent[GP_THREADS / 64][GP_Q];
	__shared__ unsigned short q_d[GP_THREADS / 64][GP_Q];
	__shared__ u64 s_pbase[GP_G];
	__shared__ u32 s_w[GP_G], s_pcnt[GP_G], s_np[GP_G], s_tmp[GP_THREADS / 64];
	__shared__ u32 s_nd, s_next, s_over;
	const int noff = len - ix.rl;
	const u32 tid = threadIdx.x;
	const int lane = __lane_id();
	const u32 wv = tid >> 6;
	// workgroups go to the XCDs round-robin (workgroup b to XCD b % 8, each with its own L2): inside every tile of 64 workgroups an XCD
	// gets 8 groups that are neighbours under the key -- they stream the same deep classes at the same time
	u32 gblk = blockIdx.x;
#ifndef GP_XCD_M
#define GP_XCD_M 32u                 // neighbours per XCD and tile (tile = 8 XCDs x GP_XCD_M workgroups; measured 8: 1.12 ms, 16: 1.11, 32: 1.09, 64: 1.08)
#endif
	if (GP_XCD_TILE && (gblk | (8u * GP_XCD_M - 1u)) < gridDim.x) gblk = (gblk & ~(8u * GP_XCD_M - 1u)) | ((gblk & 7u) * GP_XCD_M) | ((gblk >> 3) & (GP_XCD_M - 1u));
	if (tid < GP_G) {
		const u32 idx = gblk * GP_G + tid;
		const u32 wi = idx < n ? gorder[idx] : NONE32;
		s_w[tid] = wi;
		s_pbase[tid] = wi != NONE32 ? pair_off[wi] : 0ull;
		s_pcnt[tid] = 0; s_np[tid] = 0;
	}
	if (tid == 0) { s_nd = 0; s_next = 0; s_over = 0; }
	for (u32 i = tid; i < GP_TS; i += GP_THREADS) L.key[i] = 0;
	for (u32 i = tid; i < (1u << (GP_PRESENT_LOG2 - 5)); i += GP_THREADS) L.present[i] = 0;
	__syncthreads();
	if (tid == 0) {                                       // (the flat hit sequence is indexed with 32 bits)
		u64 tot = 0;
		for (int w = 0; w < GP_G; w++) if (s_w[w] != NONE32) tot += hits[s_w[w]];
		if (tot >> 32) s_over = 1;
	}
	const u32 items = (u32) GP_G * (u32) noff;
	// ---- the distinct classes of the group, numbered as they come
	for (u32 it = tid; it < items; it += GP_THREADS) {
		const u32 w = it / (u32) noff, o = it - w * (u32) noff;
		if (s_w[w] == NONE32) continue;
		const uint4 p = prep[(size_t) s_w[w] * noff + o];
		const u32 cls = p.x;
		if (cls == NONE32) continue;
		const u32 pb = cls & ((1u << GP_PRESENT_LOG2) - 1u), pb2 = gp_present_bit2(cls);
		atomicOr(&L.present[pb >> 5], 1u << (pb & 31));
		atomicOr(&L.present[pb2 >> 5], 1u << (pb2 & 31));
		u32 slot = cls & (GP_TS - 1);
		for (u32 probes = 0;; probes++) {
			u32 cur = L.key[slot];
			if (cur == 0) {
				cur = atomicCAS(&L.key[slot], 0u, cls + 1);
				if (cur == 0) {
					const u32 d = atomicAdd(&s_nd, 1u);
					if (d < GP_DMAX) { L.didx[slot] = (unsigned short) d; L.cstart[d] = p.y; L.hpre[d] = p.z; }
					else s_over = 1;
					break;
				}
			}
			if (cur == cls + 1) break;
			slot = (slot + 1) & (GP_TS - 1);
			if (probes >= GP_TS) { s_over = 1; break; }
		}
	}
	__syncthreads();
	if (s_over) { if (tid == 0) atomicAdd(&gstat[1], 1ull); return; }       // (k_window_pairs takes these windows)
	const u32 nd = s_nd;
	u32* ocnt = (u32*) &q_ent[0][0];                      // (the queues are idle while the image is built)
	for (u32 i = tid; i < nd; i += GP_THREADS) ocnt[i] = 0;
	__syncthreads();
	for (u32 it = tid; it < items; it += GP_THREADS) {
		const u32 w = it / (u32) noff, o = it - w * (u32) noff;
		if (s_w[w] == NONE32) continue;
		const u32 cls = prep[(size_t) s_w[w] * noff + o].x;
		if (cls != NONE32) atomicAdd(&ocnt[gp_lookup(L, cls)], 1u);
	}
	__syncthreads();
	{	// both prefixes: GP_DMAX / GP_THREADS consecutive numbered classes per thread
		constexpr u32 PER = (GP_DMAX + GP_THREADS) / GP_THREADS;      // (covers index nd <= GP_DMAX as well)
		u32 hv[PER], cv[PER], hs = 0, cs = 0;
#pragma unroll
		for (u32 j = 0; j < PER; j++) {
			const u32 a = PER * tid + j;
			hv[j] = a < nd ? L.hpre[a] : 0u; cv[j] = a < nd ? ocnt[a] : 0u;
			hs += hv[j]; cs += cv[j];
		}
		u32 htot, ctot;
		u32 hx = plan_block_scan(hs, s_tmp, htot);
		u32 cx = plan_block_scan(cs, s_tmp, ctot);
		__syncthreads();
#pragma unroll
		for (u32 j = 0; j < PER; j++) {
			const u32 a = PER * tid + j;
			if (a <= nd) { L.hpre[a] = hx; L.occ_start[a] = (unsigned short) cx; }
			if (a < nd) ocnt[a] = 0;
			hx += hv[j]; cx += cv[j];
		}
	}
	__syncthreads();
	for (int w = 0; w < GP_G; w++) {                      // window by window: a class's occurrences lie in runs by window
		if (s_w[w] != NONE32)
			for (int o = (int) tid; o < noff; o += GP_THREADS) {
				const u32 cls = prep[(size_t) s_w[w] * noff + o].x;
				if (cls == NONE32) continue;
				const int d = gp_lookup(L, cls);
				L.occ[L.occ_start[d] + (atomicAdd(&ocnt[d], 1u) & 0xFFFFu)] = (unsigned short) ((u32) w << 10 | (u32) o);
				if (atomicOr(&ocnt[d], 0x10000u << w) & (0x10000u << w)) atomicOr(&ocnt[d], 0x80000000u);      // (twice in this window)
			}
		__syncthreads();
	}
	for (u32 i = tid; i < nd; i += GP_THREADS) L.wmask[i] = (unsigned short) (ocnt[i] >> 16);
	__syncthreads();
	const u32 H = dbg == 2 ? 0u : L.hpre[nd];
	if (tid == 0) { atomicAdd(&gstat[0], (unsigned long long) H); atomicAdd(&gstat[2], (unsigned long long) nd); }
	u32 qn = 0, nq = 0;                                     // (wave-uniform)
	// the queued entries, one per lane: the windows that hold the entry's class, each with its own last offsets of the mate classes
	auto drain = [&](bool all) {
		vdjx_wave_lds_fence();
		const u32 upto = all ? qn : qn & ~63u;                  // (full rows of 64 only, until the last call: every lane busy)
		for (u32 q0 = 0; q0 < (dbg == 1 ? 0u : upto); q0 += 64) {
			const u32 qi = q0 + (u32) lane;
			if (qi >= qn) continue;
			const u64 e = q_ent[wv][qi];
			const u32 d = q_d[wv][qi];
			const u32 ca = ent_a(e), cb = ent_b(e);
			u32 as = 0, ae = 0, bs = 0, be = 0;
			int xa = -1, xb = -1;
			if (ca != RI_ENT_NONE) { xa = gp_lookup(L, ca); if (xa >= 0) { as = L.occ_start[xa]; ae = L.occ_start[xa + 1]; } }
			if (cb != RI_ENT_NONE) { xb = gp_lookup(L, cb); if (xb >= 0) { bs = L.occ_start[xb]; be = L.occ_start[xb + 1]; } }
			if (as == ae && bs == be) continue;
			const u32 fl = ent_flags(e), mult = ent_count(e);
			const u32 rc1 = (fl & RI_RC) ? 1u : 0u;
			const u32 mc = L.wmask[d], ma = as < ae ? (u32) L.wmask[xa] : 0u, mb = bs < be ? (u32) L.wmask[xb] : 0u;
			const u32 cs = L.occ_start[d];
			auto emit = [&](u32 w, u32 o, u32 la, u32 lb) {               // (quick_map3.c:223-245, as map_eval_entry)
				const u32 which = (lb && lb >= la) ? 1u : 0u;
				const u32 best = which ? lb : la;
				const u32 rc2 = (fl & (which ? RI_RCB : RI_RCA)) ? 1u : 0u;
				if (rc1 == rc2) return;
				const int dd = (int) (o + 1) - (int) best;
				const int insert = (int) (short) ((dd < 0 ? -dd : dd) + ix.rl);
				if (insert < 50 || insert > 400) return;
				const u32 pos = atomicAdd(&s_pcnt[w], 1u);
				pair_buf[s_pbase[w] + pos] = ((u64) mult << 32) | ((o + 1) << 16) | best;
				atomicAdd(&s_np[w], mult);
			};
			if (!((mc | ma | mb) & 0x8000u)) {
				// every class at most once per window: the occurrence in window w is the popcount(mask below w)-th of its list
				for (u32 m = mc & (ma | mb); m; m &= m - 1) {
					const u32 w = (u32) __ffs((int) m) - 1u, below = (1u << w) - 1u;
					const u32 o = (u32) L.occ[cs + (u32) __popc(mc & below)] & 1023u;
					const u32 la = (ma >> w) & 1u ? ((u32) L.occ[as + (u32) __popc(ma & below)] & 1023u) + 1u : 0u;
					const u32 lb = (mb >> w) & 1u ? ((u32) L.occ[bs + (u32) __popc(mb & below)] & 1023u) + 1u : 0u;
					emit(w, o, la, lb);
				}
				continue;
			}
			// a class repeats inside a window: the lists are walked in step (runs by window)
			u32 i = cs;
			const u32 ce = L.occ_start[d + 1];
			while (i < ce) {
				const u32 w = (u32) L.occ[i] >> 10;
				u32 la = 0, lb = 0;
				while (as < ae && ((u32) L.occ[as] >> 10) < w) as++;
				while (as < ae && ((u32) L.occ[as] >> 10) == w) { const u32 x = ((u32) L.occ[as] & 1023u) + 1u; la = x > la ? x : la; as++; }
				while (bs < be && ((u32) L.occ[bs] >> 10) < w) bs++;
				while (bs < be && ((u32) L.occ[bs] >> 10) == w) { const u32 x = ((u32) L.occ[bs] & 1023u) + 1u; lb = x > lb ? x : lb; bs++; }
				for (; i < ce && ((u32) L.occ[i] >> 10) == w; i++)
					if (la | lb) emit(w, (u32) L.occ[i] & 1023u, la, lb);
			}
		}
		// what is left moves to the front
		const u32 rem = qn - upto;
		u64 me = 0; unsigned short md = 0;
		if ((u32) lane < rem) { me = q_ent[wv][upto + (u32) lane]; md = q_d[wv][upto + (u32) lane]; }
		vdjx_wave_lds_fence();
		if ((u32) lane < rem && upto) { q_ent[wv][lane] = me; q_d[wv][lane] = md; }
		vdjx_wave_lds_fence();
		qn = rem;
	};
	// the sweep of k_window_pairs over the classes of the group instead of the offsets of a window
	for (;;) {
		u32 hb = 0;
		if (lane == 0) hb = atomicAdd(&s_next, 64u * GP_K);
		hb = (u32) __builtin_amdgcn_readlane((int) hb, 0);
		if (hb >= H) break;
		int drow = 0;
		{
			const u32 hr = hb + (u32) lane * 64u;
			if (lane < GP_K && hr < H) {
				int lo = 0, hi = (int) nd;                  // last class with hpre[d] <= hr
				while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.hpre[mid] <= hr) lo = mid; else hi = mid; }
				drow = lo;
			}
		}
		u64 ent[GP_K];
		u32 dd[GP_K / 2];
#pragma unroll
		for (int u = 0; u < GP_K; u++) {
			const u32 h = hb + (u32) (u * 64 + lane);
			int d = __builtin_amdgcn_readlane(drow, u);
			u64 e = ~0ull;
			if (h < H) {
				while (h >= L.hpre[d + 1]) d++;
				e = ix.d8[L.cstart[d] + (h - L.hpre[d])];
			}
			ent[u] = e;
			if (u & 1) dd[u >> 1] |= (u32) d << 16; else dd[u >> 1] = (u32) d;
		}
#pragma unroll
		for (int u = 0; u < GP_K; u++) {
			if (hb + (u32) (u * 64) >= H) break;                                 // (wave-uniform)
			const bool pass = gp_entry_present(L, ent[u]) && hb + (u32) (u * 64 + lane) < H;
			const u64 m = __ballot(pass);
			if (!m) continue;
			if (pass) {
				const u32 at = qn + (u32) __popcll(m & ((1ull << lane) - 1ull));
				q_ent[wv][at] = ent[u];
				q_d[wv][at] = (unsigned short) ((dd[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
			}
			qn += (u32) __popcll(m); nq += (u32) __popcll(m);
			if (qn > GP_Q - 64) drain(false);
		}
	}
	drain(true);
	if (lane == 0 && nq) atomicAdd(&gstat[3], (unsigned long long) nq);
	__syncthreads();
	if (tid < GP_G && s_w[tid] != NONE32) { pair_cnt[s_w[tid]] = s_pcnt[tid]; pair_np[s_w[tid]] = s_np[tid]; done[s_w[tid]] = 1u; }
	if (tid == 0) {                                       // (statistics: is the kernel its work or its longest workgroup?  vdjx_stat "group_clocks_*")
		const unsigned long long dt = (unsigned long long) (clock64() - t_begin);
		atomicAdd(&gstat[4], dt);
		atomicMax(&gstat[5], dt);
	}
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
	// Floor 1 (the reference's default, params.c:64) asks whether every tested (position, delta) is covered AT ALL -- one bit each -- and
	// that needs no pass over (entry, position, delta): an entry (read at f, mate at s) is ONE bit of a grid, row f, column s - f; the
	// mate sees position pos + delta iff s lies in (pos + delta - rl, pos + delta], the read sees pos iff f lies in (pos - rl, pos].  So
	//   covered(pos, delta)  =  OR over t = pos - f in [0, rl)  of  "row pos - t has a bit in columns [t + i, t + i + rl)",  i = delta - clo,
	// i.e. with every row smeared over rl columns once (log2(rl) shifts): the word of deltas of a position is the OR of rl shifted
	// row words.  Cost per window: one LDS atomic per entry, then (positions + rl) x a few words -- whatever the depth.  (The
	// difference arrays below pay two LDS atomics per entry AND delta plus a scan per delta, in batches of 22 deltas, and replay the
	// list in growing chunks until the window is covered: 0.36 ms for ONE window of 15,000 entries, the tail of the whole kernel.)
	const int nd_all = chi - clo;
	const int fmin = e0 - rl + 1 > 0 ? e0 - rl + 1 : 0;            // rows: reads at fmin .. e1 - 1
	const int nrows = e1 - fmin;
	const int nbits = nd_all + 2 * rl;                             // columns: s - f - (clo - rl + 1) in [0, nd_all + 2 rl - 2)
	const int nwb = (nbits + 63) >> 6;
	const bool by_bits = fl == 1 && nd_all >= 1 && nd_all <= 64 && nrows >= 1 && nrows * nwb * 2 <= COV_WORDS && COVER_RULE2 && COVER_BITS;
	if (by_bits && !s_bad) {
		u64* grid = (u64*) diff;
		const int dmin = clo - rl + 1;
		for (int i = tid; i < nrows * nwb; i += MAP_THREADS) grid[i] = 0;
		__syncthreads();
		for (u32 q = tid; q < 2 * npairs; q += MAP_THREADS) {
			const u32 pr = (u32) pairs[q >> 1];
			const int p1 = (int) (pr >> 16), p2 = (int) (pr & 0xFFFFu);
			const int f = (q & 1) ? p2 : p1, sx = (q & 1) ? p1 : p2;
			const int y = sx - f - dmin;
			if (f < fmin || f >= e1 || y < 0 || y >= nbits) continue;
			u64* w = &grid[(f - fmin) * nwb + (y >> 6)];
			const u64 bit = 1ull << (y & 63);
			if (!(vdjx_peek(w) & bit)) atomicOr((unsigned long long*) w, (unsigned long long) bit);
		}
		__syncthreads();
		// smear: bit z of a row <- OR of its bits z .. z + rl - 1 (a thread per row, words in ascending order: a word only reads itself
		// and the words above it)
		for (int r = tid; r < nrows; r += MAP_THREADS) {
			u64* row = grid + r * nwb;
			auto or_shifted = [&](int sh) {
				const int ws = sh >> 6, bs = sh & 63;
				for (int j = 0; j + ws < nwb; j++) {
					const u64 lo = row[j + ws], hi = j + ws + 1 < nwb ? row[j + ws + 1] : 0ull;
					row[j] |= bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
				}
			};
			int cur = 1;
			while (cur * 2 <= rl) { or_shifted(cur); cur *= 2; }
			if (rl > cur) or_shifted(rl - cur);
		}
		__syncthreads();
		for (int p = tid; p < npos; p += MAP_THREADS) {
			const int pos = e0 + p;
			const bool evaluated = pos == e0 || (pos - 1 + clo) < e1;    // the loop tests the previous mate_low (coverage.c:25)
			int mh = pos + chi;
			if (mh > e1) mh = e1 + 1;                                    // coverage.c:36-38
			const int nreq = (mh - pos < chi ? mh - pos : chi) - clo;    // deltas clo .. clo + nreq - 1 are tested at pos
			if (!evaluated || nreq <= 0) continue;
			if (pos + clo < 0 || pos + clo + nreq - 1 > len + 1023) { s_bad = 1; continue; }      // outside the reference's array: undefined there
			const u64 req = nreq >= 64 ? ~0ull : (1ull << nreq) - 1ull;
			u64 acc = 0;
			for (int t = 0; t < rl && pos - t >= fmin; t++) {
				const u64* row = grid + (pos - t - fmin) * nwb;
				const int ws = t >> 6, bs = t & 63;
				const u64 lo = row[ws], hi = ws + 1 < nwb ? row[ws + 1] : 0ull;
				acc |= bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
			}
			if ((acc & req) != req) s_bad = 1;
		}
		__syncthreads();
	}
	for (int d0 = clo; d0 < chi && !s_bad && COVER_RULE2 && !by_bits; d0 += DB) {
		const int nd = chi - d0 < DB ? chi - d0 : DB;
		for (int i = tid; i < nd * stride; i += MAP_THREADS) diff[i] = 0;
		__syncthreads();
		u32 done = 0, chunk = COVER_CHUNK0;
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
#ifndef ME_K
#define ME_K 8                    // rows of 64 consecutive hits per wave and round (10: 0.77 ms, 12: 0.82 with 5 spilled registers, 16: 1.30)
#define ME_WAVES 6
#endif
#define ME_Q 128
// A slice writes its mapped pairs at the start of its own region (region offset = hit offset: pairs <= hits) in ANY order, each with
// the number of its hit, and marks the hit in the slice's bitmap; k_gather_pairs puts every pair at the rank of its hit among the
// marked ones -- the reference's order (offset-major, registration order inside a class) comes out of the hit numbering, not out
// of the order of evaluation.  The sweep is the one of k_window_pairs: presence test over the flat hits, dense queue for the full
// test.
template <int NOFF>
__global__ __launch_bounds__(MAP_THREADS, ME_WAVES) void k_map_emit(ReadIndexDev ix, const uint4* __restrict__ prep, u32 n, int len, u32 slice_hits,
                                                          const u32* __restrict__ slice_start, const u64* __restrict__ region_off,
                                                          vdjx_pair* __restrict__ pairs, u32* __restrict__ pair_hit, u32* __restrict__ slice_bits,
                                                          u32* __restrict__ slice_cnt) {
	__shared__ MapImg<NOFF> L;
	__shared__ u64 q_ent[MAP_THREADS / 64][ME_Q];
	__shared__ u32 q_hit[MAP_THREADS / 64][ME_Q];
	__shared__ unsigned short q_off[MAP_THREADS / 64][ME_Q];
	__shared__ u32 bits[MAP_SLICE_MAX / 32];
	__shared__ u32 s_next, s_cnt;
	const u32 tid = threadIdx.x, wv = tid >> 6;
	const int lane = __lane_id();
	const int noff = len - ix.rl;
	const u32 ci = slice_contig(slice_start, n, blockIdx.x);
	const u32 w0 = (blockIdx.x - slice_start[ci]) * slice_hits;
	vdjx_pair* out = pairs + region_off[ci] + w0;
	u32* out_hit = pair_hit + region_off[ci] + w0;
	const u32 nwords = (slice_hits + 31) / 32;
	for (u32 i = tid; i < nwords; i += MAP_THREADS) bits[i] = 0;
	if (tid == 0) { s_next = w0; s_cnt = 0; }
	const u32 H = map_load_prep(L, prep + (size_t) ci * noff, noff);
	const u32 h1 = w0 + slice_hits < H ? w0 + slice_hits : H;
	u32 qn = 0;
	auto drain = [&]() {
		vdjx_wave_lds_fence();                          // (the queue was written by other lanes of the wave)
		for (u32 q0 = 0; q0 < qn; q0 += 64) {
			const u32 qi = q0 + (u32) lane;
			bool pr = false;
			u64 e = 0;
			u32 o = 0, h = 0, pos2 = 0, which = 0;
			if (qi < qn) {
				e = q_ent[wv][qi]; o = q_off[wv][qi]; h = q_hit[wv][qi];
				pr = map_eval_entry(L, ix.rl, (int) o, e, pos2, which);
			}
			const u64 m = __ballot(pr);
			if (!m) continue;
			u32 base = 0;
			if (lane == 0) base = atomicAdd(&s_cnt, (u32) __popcll(m));
			base = (u32) __builtin_amdgcn_readlane((int) base, 0);
			if (pr) {
				const u32 at = base + (u32) __popcll(m & ((1ull << lane) - 1ull));
				const u32 ci2 = L.cstart[o] + (h - L.hpre[o]);
				const u32 pid = ix.csr_pair[ci2];
				const u32 fl = ent_flags(e);
				const int d = (int) (o + 1) - (int) pos2;
				vdjx_pair* q = out + at;
				q->pair_id = pid; q->rec1 = ix.recs[ci2]; q->rec2 = ix.pair_r2[2 * (size_t) pid + which];
				q->pos1 = (int16_t) (o + 1); q->pos2 = (int16_t) pos2; q->insert = (int16_t) ((d < 0 ? -d : d) + ix.rl);
				q->rc1 = (fl & RI_RC) ? 1 : 0; q->rc2 = (fl & (which ? RI_RCB : RI_RCA)) ? 1 : 0;
				out_hit[at] = h - w0;
				atomicOr(&bits[(h - w0) >> 5], 1u << ((h - w0) & 31));
			}
		}
		vdjx_wave_lds_fence();
		qn = 0;
	};
	if (w0 < h1) for (;;) {
		u32 hb = 0;
		if (lane == 0) hb = atomicAdd(&s_next, 64u * ME_K);
		hb = (u32) __builtin_amdgcn_readlane((int) hb, 0);
		if (hb >= h1) break;
		int orow = 0;
		{
			const u32 hr = hb + (u32) lane * 64u;
			if (lane < ME_K && hr < h1) {
				int lo = 0, hi = noff;                      // last offset with hpre[o] <= hr
				while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.hpre[mid] <= hr) lo = mid; else hi = mid; }
				orow = lo;
			}
		}
		u64 ent[ME_K];
		u32 oo[ME_K / 2];
#pragma unroll
		for (int u = 0; u < ME_K; u++) {
			const u32 h = hb + (u32) (u * 64 + lane);
			int o = __builtin_amdgcn_readlane(orow, u);
			u64 e = ~0ull;
			if (h < h1) {
				while (h >= L.hpre[o + 1]) o++;                 // (few steps: a row of 64 hits spans few classes; empty ones are skipped)
				e = ix.csr8[L.cstart[o] + (h - L.hpre[o])];
			}
			ent[u] = e;
			if (u & 1) oo[u >> 1] |= (u32) o << 16; else oo[u >> 1] = (u32) o;
		}
#pragma unroll
		for (int u = 0; u < ME_K; u++) {
			if (hb + (u32) (u * 64) >= h1) break;                                // (wave-uniform)
			const u32 h = hb + (u32) (u * 64 + lane);
			const bool pass = map_entry_present(L, ent[u]) && h < h1;
			const u64 m = __ballot(pass);
			if (!m) continue;
			if (pass) {
				const u32 at = qn + (u32) __popcll(m & ((1ull << lane) - 1ull));
				q_ent[wv][at] = ent[u];
				q_hit[wv][at] = h;
				q_off[wv][at] = (unsigned short) ((oo[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
			}
			qn += (u32) __popcll(m);
			if (qn > ME_Q - 64) drain();
		}
	}
	drain();
	__syncthreads();
	u32* gb = slice_bits + (size_t) blockIdx.x * nwords;
	for (u32 i = tid; i < nwords; i += MAP_THREADS) gb[i] = bits[i];
	if (tid == 0) slice_cnt[blockIdx.x] = s_cnt;
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

// every pair of a slice to the rank of its hit among the slice's mapped hits, slice after slice, in the caller's dense layout: hit order
// = the reference's order
__global__ __launch_bounds__(256) void k_gather_pairs(const vdjx_pair* __restrict__ src, const u32* __restrict__ pair_hit, u32 n, u32 slice_hits,
                                                      const u32* __restrict__ slice_start, const u64* __restrict__ region_off,
                                                      const u32* __restrict__ slice_bits, const u64* __restrict__ slice_pre, vdjx_pair* __restrict__ dst, u64 cap) {
	__shared__ u32 wpre[MAP_SLICE_MAX / 32 + 1], wbit[MAP_SLICE_MAX / 32];
	__shared__ u32 part[4];
	if (slice_pre[gridDim.x] > cap) return;          // (launched before the host knew the total: the buffer is the last call's)
	const u64 dof = slice_pre[blockIdx.x];
	const u32 cnt = (u32) (slice_pre[blockIdx.x + 1] - dof);
	if (!cnt) return;
	const u32 ci = slice_contig(slice_start, n, blockIdx.x);
	const u64 so = region_off[ci] + (u64) (blockIdx.x - slice_start[ci]) * slice_hits;
	const u32 nwords = (slice_hits + 31) / 32;
	const u32* gb = slice_bits + (size_t) blockIdx.x * nwords;
	// exclusive prefix of the words' popcounts (nwords <= 1024: four words per thread)
	const u32 per = (nwords + 255) / 256;
	const u32 lo = threadIdx.x * per < nwords ? threadIdx.x * per : nwords, hi = lo + per < nwords ? lo + per : nwords;
	u32 sum = 0;
	for (u32 i = lo; i < hi; i++) { const u32 w = gb[i]; wbit[i] = w; sum += __popc(w); }
	const u32 incl = (u32) vdjx_wave_scan_add((int) sum);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 run = incl - sum;
	for (u32 w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
	for (u32 i = lo; i < hi; i++) { wpre[i] = run; run += __popc(wbit[i]); }
	__syncthreads();
	constexpr u32 PW = sizeof(vdjx_pair) / 4;
	const uint32_t* s4 = (const uint32_t*) (src + so);
	uint32_t* d4 = (uint32_t*) (dst + dof);
	// dword q of pair i: consecutive threads read consecutive dwords (the slice's pairs are contiguous)
	for (u32 t = threadIdx.x; t < cnt * PW; t += 256) {
		const u32 i = t / PW, q = t - i * PW;
		const u32 h = pair_hit[so + i];
		const u32 rank = wpre[h >> 5] + (u32) __popc(wbit[h >> 5] & ((1u << (h & 31)) - 1u));
		d4[(size_t) rank * PW + q] = s4[t];
	}
}

static int make_index_view(vdjx_ctx* c, ReadIndexDev* ix, int len, const char* who) {
	if (c->ri_job) { const int jr = vdjx_ri_join(c); if (jr) return jr; }      // a begun index build ends at the first call that needs the index
	if (!c->ri_pool) { vdjx_set_error("%s: call vdjx_read_index_build first", who); return VDJX_ESTATE; }
	const vdjx_pool* p = c->ri_pool;
	if (len <= p->rl) { vdjx_set_error("%s: len=%d must exceed the read length %d", who, len, p->rl); return VDJX_EINVAL; }
	if (len - p->rl > MAP_MAXOFF) { vdjx_set_error("%s: len=%d too long (max %d)", who, len, MAP_MAXOFF + p->rl); return VDJX_ELIMIT; }
	ix->tab = (const u64*) c->d_ri_tab; ix->mask = c->ri_tab_mask;
	ix->start = c->d_ri_start; ix->cnt1 = c->d_ri_cnt1; ix->recs = c->d_ri_recs;
	ix->csr8 = c->d_ri_csr8; ix->csr_pair = c->d_ri_csr_pair; ix->pair_r2 = c->d_pair_r2;
	ix->dstart = c->d_ri_dstart; ix->d8 = c->d_ri_d8;
	ix->rl = p->rl;
	ix->canon = c->ri_canon;
	ix->epoch = c->ri_tab_epoch;
	return VDJX_OK;
}

// largest-first processing order without a comparison sort: by descending bit length of the size (host version: vdjx_window_cover,
// whose list sizes arrive as host numbers)
static void order_by_size_desc(const std::vector<u64>& off, size_t n, std::vector<u32>& order) {
	order.resize(n);
	u32 cnt[66] = {0};
	auto key = [&](size_t i) { const u64 sz = off[i + 1] - off[i]; return sz ? 64u - (u32) __builtin_clzll(sz) : 0u; };     // 0..64
	for (size_t i = 0; i < n; i++) cnt[64 - key(i) + 1]++;
	for (int b = 0; b < 65; b++) cnt[b + 1] += cnt[b];
	for (size_t i = 0; i < n; i++) order[cnt[64 - key(i)]++] = (u32) i;
}

// the strings on the device, classified (k_map_classify) and planned (k_plan); the plan's totals come back through the context's
// page-locked scratch: the one host wait of a scorer call before its result
struct MapPlan {
	uint4* d_prep = nullptr;
	u64* d_off = nullptr;
	u32 *d_order = nullptr, *d_wstart = nullptr;
	u32 *d_hits = nullptr, *d_gorder = nullptr;       // hits per string; the strings in k_group_pairs' order (if asked for)
	bool gstat = false;
	PlanOut tot{};
};
int vdjx_sort_pairs(vdjx_work& db, hipStream_t st, u64* k_in, u64* k_out, u32* v_in, u32* v_out, u32 n, unsigned end_bit);     // vdjx_rindex.hip
// the plan's totals once they have come down: the host waits for the EVENT behind their copy, not for the stream (work launched behind
// the plan keeps the device busy meanwhile)
static int plan_finish(vdjx_ctx* c, MapPlan* mp) {
	vdjx_laps lp(c);
	HIP_TRY(hipEventSynchronize(c->ev_plan));
	HIP_TRY(hipGetLastError());
	lp.mark("plan_wait");
	mp->tot = *(const PlanOut*) c->h_plan;
	if (mp->tot.total_hits >= (1ull << 40)) { vdjx_set_error("too many hits in one scorer call (%llu)", (unsigned long long) mp->tot.total_hits); return VDJX_ELIMIT; }
	return VDJX_OK;
}
static int classify_and_plan(vdjx_ctx* c, vdjx_work& db, const ReadIndexDev& ix, const char* strings, size_t n, int len, bool weighted,
                             u32 chunk_fixed, MapPlan* mp, bool grouped = false, bool wait = true) {
	hipStream_t st = c->stream;
	vdjx_laps lp(c);
	const int noff = len - ix.rl;
	char* d_s;
	u32 *d_hits, *d_inst;
	PlanOut* d_tot;
	HIP_TRY(db.alloc(&d_s, n * len));
	HIP_TRY(db.alloc(&mp->d_prep, n * (size_t) noff));
	HIP_TRY(db.alloc(&d_hits, n));
	HIP_TRY(db.alloc(&d_inst, n));
	HIP_TRY(db.alloc(&mp->d_off, n + 1));
	HIP_TRY(db.alloc(&mp->d_order, n));
	HIP_TRY(db.alloc(&mp->d_wstart, n + 1));
	HIP_TRY(db.alloc(&d_tot, 1));
	u64 *d_gkey = nullptr, *d_gkey2 = nullptr;
	u32* d_gidx = nullptr;
	if (grouped) {
		HIP_TRY(db.alloc(&d_gkey, n));
		HIP_TRY(db.alloc(&d_gkey2, n));
		HIP_TRY(db.alloc(&d_gidx, n));
		HIP_TRY(db.alloc(&mp->d_gorder, n));
	}
	mp->d_hits = d_hits;
	// the strings cross PCIe in pieces on a stream of their own, each piece classified as soon as it is there (8.7 MB of windows: 0.18 ms
	// of which the first quarter is waited for)
	// ... unless they lie in page-locked host memory of the context's own (vdjx_host_alloc): then the
	// kernel reads them where they are -- every character is read exactly once, 64 consecutive bytes per wave, and the other workgroups
	// of a CU classify while one waits for its string; no copy, no pieces, nothing for the stream to wait for
	static const bool zero_copy = !(getenv("VDJX_STRINGS_IN_PLACE") && atoi(getenv("VDJX_STRINGS_IN_PLACE")) == 0);
	const char* d_src = nullptr;
	if (zero_copy && vdjx_host_block_holds(c, strings, n * (size_t) len)) d_src = strings;      // (asking the runtime about a foreign pointer costs more than the copy saves)
	const u32 pieces = d_src ? 1u : n * (size_t) len >= (2u << 20) ? 4u : 1u;
	if (pieces > 1) {
		HIP_TRY(hipEventRecord(c->ev_up[0], st));                  // (the arena's last users are on `st`)
		HIP_TRY(hipStreamWaitEvent(c->up_stream, c->ev_up[0], 0));
	}
	for (u32 pc = 0; pc < pieces; pc++) {
		const size_t a = n * (size_t) pc / pieces, b = n * (size_t) (pc + 1) / pieces;
		if (pieces > 1) {
			HIP_TRY(hipMemcpyAsync(d_s + a * len, strings + a * len, (b - a) * len, hipMemcpyHostToDevice, c->up_stream));
			HIP_TRY(hipEventRecord(c->ev_up[1 + pc], c->up_stream));
			HIP_TRY(hipStreamWaitEvent(st, c->ev_up[1 + pc], 0));
		} else if (!d_src)
			HIP_TRY(hipMemcpyAsync(d_s, strings, n * len, hipMemcpyHostToDevice, st));
		vdjx_prof_scope ps(c, "k_map_classify");
		const dim3 grid((u32) std::min<size_t>(b - a, 8192));
		if (d_src) d_s = const_cast<char*>(d_src);
		if (c->ri_pool->W == 2) hipLaunchKernelGGL(k_map_classify<2>, grid, dim3(MAP_THREADS), 0, st, ix, d_s, (u32) b, len, weighted, mp->d_prep, d_hits, d_inst, d_gkey, d_gidx, (u32) a);
		else hipLaunchKernelGGL(k_map_classify<VDJX_LONG_W>, grid, dim3(MAP_THREADS), 0, st, ix, d_s, (u32) b, len, weighted, mp->d_prep, d_hits, d_inst, d_gkey, d_gidx, (u32) a);
	}
	{
		vdjx_prof_scope ps(c, "k_plan");
		const bool staged = n * 4 <= 96 * 1024;
		if (staged) HIP_TRY(hipFuncSetAttribute((const void*) k_plan, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
		hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), staged ? n * 4 : 0, st, d_hits, d_inst, (u32) n, chunk_fixed, weighted ? 1 : 0, weighted ? 1 : 0, mp->d_off, mp->d_order, mp->d_wstart, d_tot,
		                   staged ? 1 : 0);
	}
	if (grouped) {
		vdjx_prof_scope ps(c, "group_sort");
		const int rc = vdjx_sort_pairs(db, st, d_gkey, d_gkey2, d_gidx, mp->d_gorder, (u32) n, 22);
		if (rc) return rc;
	}
	if (!c->h_plan) {
		HIP_TRY(hipHostMalloc(&c->h_plan, 256, hipHostMallocDefault));
		c->h_plan_cap = 256;
	}
	HIP_TRY(hipMemcpyAsync(c->h_plan, d_tot, sizeof(PlanOut), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(c->ev_plan, st));
	lp.mark("plan_issue");
	return wait ? plan_finish(c, mp) : VDJX_OK;
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

// K8 over n windows: the (weighted) mapped-pair list of every window in the context's pair buffer; the list of window i
// is wp_buf[d_off[i] .. + d_cnt[i])
// budget (bytes, 0: none): pair lists that would need more are not built -- *need = their bytes, nothing else done (the caller takes
// fewer windows at a time)
static int window_pairs_run(vdjx_ctx* c, vdjx_work& db, const ReadIndexDev& ix, const char* windows, size_t n, int len, u32** d_np_out,
                            u32** d_cnt_out, MapPlan* mp, u64 budget = 0, u64* need = nullptr) {
	hipStream_t st = c->stream;
	u32 *d_np, *d_cnt;
	static const u32 hit_chunk = getenv("VDJX_HIT_CHUNK") && atol(getenv("VDJX_HIT_CHUNK")) > 0 ? (u32) atol(getenv("VDJX_HIT_CHUNK")) : HIT_CHUNK;
	// VDJX_WINDOW_GROUP=0: every window on its own (k_window_pairs only)
	static const u32 gp_dbg = getenv("VDJX_GP_DBG") ? (u32) atol(getenv("VDJX_GP_DBG")) : 0u;      // ablation (profiles/): 1 no full tests, 2 images only
	static const bool group_on = !(getenv("VDJX_WINDOW_GROUP") && atol(getenv("VDJX_WINDOW_GROUP")) == 0);
	// (few windows are mapped one by one: a group is one workgroup for eight windows, and below a few thousand windows the groups
	// leave most of the chip idle -- 200 windows: 0.14 ms in groups + left-overs, 0.08 one by one; 2,000: 0.29 against 0.23; 20,000: 1.1 against 3.5)
	static const size_t group_min = getenv("VDJX_GROUP_MIN") ? (size_t) atol(getenv("VDJX_GROUP_MIN")) : 4096;
	const bool grouped = group_on && len - ix.rl <= GP_NOFF && n >= group_min;
	// With a pair buffer from an earlier call the groups are mapped BEFORE the host knows the plan's totals (the kernel returns at once
	// if the lists would not fit): the host's wait for the totals and its next launches hide behind that kernel.
	// (grouped: k_window_pairs only gets the windows of the few groups whose classes did not fit one image -- unrelated windows, each as
	// deep as a window gets, and nothing else is running by then: their hits in small slices over many workgroups.  Two such groups of
	// the bench workload: 0.135 ms in slices of 524,288 hits)
	static const u32 left_chunk = getenv("VDJX_LEFT_CHUNK") && atol(getenv("VDJX_LEFT_CHUNK")) > 0 ? (u32) atol(getenv("VDJX_LEFT_CHUNK")) : 32768u;
	int rc = classify_and_plan(c, db, ix, windows, n, len, true, grouped && left_chunk < hit_chunk ? left_chunk : hit_chunk, mp, grouped, false);
	if (rc) return rc;
	// (one block, cleared by one call: a fill is 5 us of a small pool's step whatever its size)
	const size_t nq = (n + 3) & ~(size_t) 3;                  // the statistics are 64-bit words behind three arrays of n
	u32* d_zero;
	HIP_TRY(db.alloc(&d_zero, (grouped ? 3 : 2) * nq + (grouped ? 16 : 0)));
	d_np = d_zero; d_cnt = d_zero + nq;
	u32* d_done = nullptr;
	unsigned long long* d_gstat = nullptr;
	if (grouped) { d_done = d_zero + 2 * nq; d_gstat = (unsigned long long*) (d_zero + 3 * nq); }
	auto map_groups = [&]() -> int {
		HIP_TRY(hipMemsetAsync(d_zero, 0, ((grouped ? 3 : 2) * nq + (grouped ? 16 : 0)) * 4, st));
		if (!grouped) return VDJX_OK;
		vdjx_prof_scope ps(c, "k_group_pairs");
		hipLaunchKernelGGL(k_group_pairs, dim3((u32) ((n + GP_G - 1) / GP_G)), dim3(GP_THREADS), 0, st, ix, mp->d_prep, mp->d_hits, (u32) n, len, mp->d_gorder,
		                   mp->d_off, (u64*) c->wp_buf, d_cnt, d_np, d_done, d_gstat, gp_dbg, (u64) c->wp_cap);
		return VDJX_OK;
	};
	const bool early = grouped && c->wp_cap > 0;
	if (early && (rc = map_groups())) return rc;
	if ((rc = plan_finish(c, mp))) return rc;
	const u64 total = mp->tot.total_hits;
	bool again = !early;
	if (need) *need = 0;
	if ((size_t) total + 1 > c->wp_cap && budget && n > 1 && (total + total / 4 + 1024) * 8 > budget) {
		HIP_TRY(hipStreamSynchronize(st));                  // (the early launch has returned without a write)
		*need = (total + total / 4 + 1024) * 8;
		return VDJX_OK;
	}
	if ((size_t) total + 1 > c->wp_cap) {
		HIP_TRY(hipStreamSynchronize(st));                  // (the early launch, if any, has returned without a write: wait before the buffer goes)
		free_set(c->wp_buf);
		c->wp_cap = 0;
		const size_t want = (size_t) total + (size_t) total / 4 + 1024;
		HIP_TRY(hipMalloc(&c->wp_buf, want * 8));
		c->wp_cap = want;
		again = true;
	}
	if (again && (rc = map_groups())) return rc;
	c->stats["window_hits"] = mp->tot.inst_total;          // read-1 instances matched (what the reference enumerates one by one)
	c->stats["window_hits_max"] = mp->tot.inst_max;
	c->stats["window_hits_distinct"] = total;              // weighted entries actually evaluated
	c->stats["window_work_items"] = mp->tot.nwork;
	if (mp->tot.nwork) {
		u32* d_witem = nullptr;
		if (grouped) HIP_TRY(db.alloc(&d_witem, (size_t) mp->tot.nwork + 1));
		vdjx_prof_scope ps(c, "k_window_pairs");
		if (grouped) hipLaunchKernelGGL(k_work_items, dim3((u32) (n / 256 + 1)), dim3(256), 0, st, mp->d_wstart, (u32) n, d_witem);
		if (len - ix.rl <= 512)
			hipLaunchKernelGGL(k_window_pairs<512>, dim3(mp->tot.nwork), dim3(MAP_THREADS), 0, st, ix, mp->d_prep, (u32) n, len, mp->tot.chunk, mp->d_order, mp->d_wstart,
			                   mp->d_off, (u64*) c->wp_buf, d_cnt, d_np, d_done, d_gstat ? d_gstat + 1 : nullptr, (const u32*) d_witem);
		else
			hipLaunchKernelGGL(k_window_pairs<MAP_MAXOFF>, dim3(mp->tot.nwork), dim3(MAP_THREADS), 0, st, ix, mp->d_prep, (u32) n, len, mp->tot.chunk, mp->d_order, mp->d_wstart,
			                   mp->d_off, (u64*) c->wp_buf, d_cnt, d_np, d_done, d_gstat ? d_gstat + 1 : nullptr, (const u32*) d_witem);
	}
	if (d_gstat) HIP_TRY(hipMemcpyAsync((char*) c->h_plan + 128, d_gstat, 64, hipMemcpyDeviceToHost, st));        // (read after the caller's wait)
	mp->gstat = d_gstat != nullptr;
	*d_np_out = d_np; *d_cnt_out = d_cnt;
	return VDJX_OK;
}

// one slice of windows; *need != 0 on return: its pair lists do not fit the budget, nothing was written
static int window_score_slice(vdjx_ctx* c, const ReadIndexDev& ix, const char* windows, size_t n, int len, const vdjx_cov_params* p,
                              uint8_t* out_valid, uint32_t* out_npairs, u64 budget, u64* need) {
	hipStream_t st = c->stream;
	vdjx_work db(c);
	u32 *d_np, *d_cnt;
	uint8_t* d_valid;
	MapPlan mp;
	HIP_TRY(db.alloc(&d_valid, n));
	int rc = window_pairs_run(c, db, ix, windows, n, len, &d_np, &d_cnt, &mp, budget, need);
	if (rc || *need) return rc;
	c->wp_n = 0;                                              // (the lists are not offered to vdjx_window_pairs_fetch)
	{
		vdjx_prof_scope ps(c, "k_window_cover");
		u64* d_clk = nullptr;
		static const bool cover_clocks = getenv("VDJX_COVER_CLOCKS") != nullptr;       // profiling aid: the slowest windows of the call to stderr
		if (cover_clocks) HIP_TRY(db.alloc(&d_clk, n));
		hipLaunchKernelGGL(k_window_cover, dim3((u32) n), dim3(MAP_THREADS), 0, st, len, ix.rl, *p, mp.d_order, mp.d_off, (const u64*) c->wp_buf, d_cnt, d_valid, d_clk);
		if (cover_clocks) {
			std::vector<u64> clk(n);
			std::vector<u32> cnt(n), ord(n);
			std::vector<uint8_t> val(n);
			HIP_TRY(hipStreamSynchronize(st));
			HIP_TRY(hipMemcpy(clk.data(), d_clk, n * 8, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(ord.data(), mp.d_order, n * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(val.data(), d_valid, n, hipMemcpyDeviceToHost));
			std::vector<u32> by(n);
			for (size_t i = 0; i < n; i++) by[i] = (u32) i;
			std::sort(by.begin(), by.end(), [&](u32 a, u32 b) { return clk[a] > clk[b]; });
			u64 sum = 0;
			for (size_t i = 0; i < n; i++) sum += clk[i];
			fprintf(stderr, "cover clocks: %zu windows, sum %llu, slowest:", n, (unsigned long long) sum);
			for (size_t i = 0; i < n && i < 8; i++) {
				size_t at = 0;
				while (at < n && ord[at] != by[i]) at++;
				fprintf(stderr, " [w%u blk%zu entries %u valid %d clk %llu]", by[i], at, cnt[by[i]], (int) val[by[i]], (unsigned long long) clk[by[i]]);
			}
			fprintf(stderr, "\n");
		}
	}
	// verdicts, pair counts and list lengths through the context's page-locked buffer: three asynchronous copies and one wait
	const size_t nres = ((n + 15) & ~(size_t) 15) + 8 * n;
	if (nres > c->h_res_cap) {
		if (c->h_res) (void) hipHostFree(c->h_res);
		c->h_res = nullptr; c->h_res_cap = 0;
		HIP_TRY(hipHostMalloc(&c->h_res, nres + nres / 4, hipHostMallocDefault));
		c->h_res_cap = nres + nres / 4;
	}
	uint8_t* hr_valid = (uint8_t*) c->h_res;
	u32* hr_np = (u32*) (hr_valid + ((n + 15) & ~(size_t) 15));
	u32* hr_cnt = hr_np + n;
	HIP_TRY(hipMemcpyAsync(hr_valid, d_valid, n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hr_np, d_np, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hr_cnt, d_cnt, n * 4, hipMemcpyDeviceToHost, st));
	vdjx_laps lp(c);
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	lp.mark("ws_cover_wait");
	memcpy(out_valid, hr_valid, n);
	memcpy(out_npairs, hr_np, n * 4);
	c->wp_cnt.assign(hr_cnt, hr_cnt + n);
	vdjx_prof_collect(c, false);
	if (mp.gstat) {
		const unsigned long long* g = (const unsigned long long*) ((const char*) c->h_plan + 128);
		c->stats["group_hits_distinct"] = g[0];                // entries streamed by k_group_pairs (each class of a group once)
		c->stats["group_overflows"] = g[1];                    // groups left to k_window_pairs
		c->stats["group_classes"] = g[2];
		c->stats["group_queued"] = g[3];                       // entries that passed the presence test of their group
		c->stats["group_clocks_sum"] = g[4];                   // shader clocks over all workgroups of k_group_pairs ...
		c->stats["group_clocks_max"] = g[5];                   // ... and of the longest one
	}
	{
		u64 tot = 0, ent = 0;
		for (size_t i = 0; i < n; i++) { tot += out_npairs[i]; ent += c->wp_cnt[i]; }
		c->stats["window_pairs"] = tot;                    // mapped pairs (with multiplicity)
		c->stats["window_pairs_entries"] = ent;            // entries of the weighted lists
	}
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
	// The pair lists of a call are sized by its windows' hits (8 bytes each; reads of a shared V germline hit every window that has
	// it: the sum grows with windows x pool).  What does not fit the device is done in slices of windows, every slice like a call of
	// its own (400 k windows over 10 M pairs ask for 600 GB at once).  VDJX_WP_BUDGET_MB: the tests' way into the slices.
	static const u64 budget_env = getenv("VDJX_WP_BUDGET_MB") && atol(getenv("VDJX_WP_BUDGET_MB")) > 0 ? (u64) atol(getenv("VDJX_WP_BUDGET_MB")) << 20 : 0;
	u64 budget = budget_env;
	if (!budget) {
		size_t fr = 0, tt = 0;
		HIP_TRY(hipMemGetInfo(&fr, &tt));
		budget = ((u64) fr + (u64) c->wp_cap * 8) / 10 * 8;            // (the buffer in hand is given back before a larger one is asked for)
	}
	static const char* const summed[] = {"window_hits", "window_hits_distinct", "window_work_items", "group_hits_distinct", "group_overflows", "group_classes",
	                                     "group_queued", "window_pairs", "window_pairs_entries"};
	std::map<std::string, uint64_t> acc;
	u64 hits_max = 0, slices = 0;
	size_t at = 0, step = n;
	while (at < n) {
		const size_t m = std::min(step, n - at);
		u64 need = 0;
		rc = window_score_slice(c, ix, windows + at * (size_t) len, m, len, p, out_valid + at, out_npairs + at, budget, &need);
		if (rc) return rc;
		if (need) {                                          // fewer windows: in proportion, with a margin
			step = std::max<size_t>(1, std::min<size_t>(m - 1, (size_t) ((double) m * (double) budget / (double) need * 0.8)));
			continue;
		}
		for (const char* k_ : summed) acc[k_] += c->stats[k_];
		hits_max = std::max<u64>(hits_max, c->stats["window_hits_max"]);
		slices++;
		at += m;
	}
	if (slices > 1) {
		for (const char* k_ : summed) c->stats[k_] = acc[k_];
		c->stats["window_hits_max"] = hits_max;
	}
	c->stats["window_slices"] = slices;
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
	u32 *d_np, *d_cnt;
	MapPlan mp;
	rc = window_pairs_run(c, db, ix, windows, n, len, &d_np, &d_cnt, &mp);
	if (rc) return rc;
	c->wp_cnt.resize(n);
	c->wp_off.resize(n + 1);
	HIP_TRY(hipMemcpyAsync(c->wp_cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(c->wp_off.data(), mp.d_off, (n + 1) * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(out_npairs, d_np, n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c, false);
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
	vdjx_prof_collect(c, false);
	return VDJX_OK;
}

// identity of a contig batch between the counting and the writing call of vdjx_map_emit: FNV-1a over its head, its tail and a
// sparse sample in between (hashing every byte of a few MB twice per call cost more than the mapping kernel).  Only a guard: the
// counting call always maps afresh, and a writing call that does not follow the counting call of the same batch maps again.
static uint64_t fnv1a(const char* p, size_t n, uint64_t h) {
	auto eat = [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) { h ^= (unsigned char) p[i]; h *= 0x100000001b3ull; } };
	if (n <= 4096) { eat(0, n); return h; }
	eat(0, 1024);
	for (size_t i = 1024; i + 8 <= n - 1024; i += 509) eat(i, i + 8);
	eat(n - 1024, n);
	return h;
}

static int map_emit_impl(vdjx_ctx* c, const char* contigs, size_t n, int len, uint64_t* offsets, vdjx_pair* pairs, bool async, bool device_only = false) {
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
	static const u32 slice_env = getenv("VDJX_MAP_SLICE") && atol(getenv("VDJX_MAP_SLICE")) > 0 ? (u32) std::min<long>(atol(getenv("VDJX_MAP_SLICE")), (long) MAP_SLICE_MAX) : 0u;
	if (!pairs) c->me_key = 0;               // a counting call never reuses an earlier mapping
	if (c->me_key != key || c->me_cnt.size() != n || c->me_src != (const void*) contigs) {
		c->me_key = 0;
		MapPlan mp;
		rc = classify_and_plan(c, db, ix, contigs, n, len, false, slice_env, &mp);
		if (rc) return rc;
		lp.mark("me_plan");
		const u64 total_hits = mp.tot.total_hits;
		if (total_hits > c->me_cap) {
			free_set(c->me_pairs);
			free_set(c->me_hit);
			c->me_cap = 0;
			HIP_TRY(hipMalloc(&c->me_pairs, (size_t) total_hits * sizeof(vdjx_pair)));
			HIP_TRY(hipMalloc(&c->me_hit, (size_t) total_hits * 4 + 4));
			c->me_cap = (size_t) total_hits;
		}
		const u32 slice_hits = mp.tot.chunk;
		const size_t nsl = mp.tot.nwork;
		c->me_slice_hits = slice_hits;
		// persistent bookkeeping: hit offsets of the contigs, pairs per slice and their prefix, pairs per contig, slice prefix of the
		// contigs, the slices' bitmaps of mapped hits
		const size_t nwords = ((size_t) slice_hits + 31) / 32;
		const size_t need = (n + 1) * 4 + 8 + (n + 1) * 8 + nsl * 4 + 8 + (nsl + 1) * 8 + n * 8 + nsl * nwords * 4 + 64;
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
		u32* b_scnt = (u32*) bk;                     bk += (nsl * 4 + 7) / 8 * 8;
		u32* b_bits = (u32*) bk;
		c->me_nsl = nsl;
		c->me_cnt.assign(n, 0);
		HIP_TRY(hipMemcpyAsync(b_sstart, mp.d_wstart, (n + 1) * 4, hipMemcpyDeviceToDevice, st));
		HIP_TRY(hipMemcpyAsync(b_off, mp.d_off, (n + 1) * 8, hipMemcpyDeviceToDevice, st));
		if (nsl) {
			{
				vdjx_prof_scope ps(c, "k_map_emit");
				if (len - ix.rl <= 512)
					hipLaunchKernelGGL(k_map_emit<512>, dim3((u32) nsl), dim3(MAP_THREADS), 0, st, ix, mp.d_prep, (u32) n, len, slice_hits, b_sstart, b_off, (vdjx_pair*) c->me_pairs, (u32*) c->me_hit, b_bits, b_scnt);
				else
					hipLaunchKernelGGL(k_map_emit<MAP_MAXOFF>, dim3((u32) nsl), dim3(MAP_THREADS), 0, st, ix, mp.d_prep, (u32) n, len, slice_hits, b_sstart, b_off, (vdjx_pair*) c->me_pairs, (u32*) c->me_hit, b_bits, b_scnt);
			}
			hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, st, b_scnt, (u32) nsl, b_pre);
			hipLaunchKernelGGL(k_contig_counts, dim3((u32) (n + 255) / 256), dim3(256), 0, st, b_pre, b_sstart, (u32) n, b_cnt);
			HIP_TRY(hipMemcpyAsync(c->me_cnt.data(), b_cnt, n * 8, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipEventRecord(c->ev_plan, st));
			// the pairs are laid end to end while the host still waits for their number: into the buffer of the last call if it is
			// large enough (the kernel looks), again by the writing call if it was not
			c->me_gathered_cap = 0;
			if (c->me_dense_cap) {
				HIP_TRY(hipStreamWaitEvent(st, c->ev_pairs_copied, 0));      // an earlier asynchronous copy may still be reading the buffer
				vdjx_prof_scope ps(c, "k_gather_pairs");
				hipLaunchKernelGGL(k_gather_pairs, dim3((u32) nsl), dim3(256), 0, st, (const vdjx_pair*) c->me_pairs, (const u32*) c->me_hit, (u32) n, slice_hits, b_sstart, b_off, b_bits, b_pre,
				                   (vdjx_pair*) c->me_dense, (u64) c->me_dense_cap);
				c->me_gathered_cap = c->me_dense_cap;
			}
			HIP_TRY(hipEventSynchronize(c->ev_plan));
		} else
			HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		lp.mark("me_kernel_wait");
		c->me_key = key;
		c->me_src = (const void*) contigs;
		c->stats["map_hits"] = total_hits;
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
		const u32* b_sstart = (const u32*) bk;       bk += ((n + 1) * 4 + 7) / 8 * 8;
		bk += (nsl * 4 + 7) / 8 * 8;
		const u32* b_bits = (const u32*) bk;
		// the dense copy outlives this call when the transfer to the host is asynchronous: a buffer of its own, not the arena
		lp.mark("me_second_call");
		HIP_TRY(hipStreamSynchronize(c->pairs_stream));          // an earlier asynchronous copy may still be reading the buffer
		lp.mark("me_prev_copy_wait");
		const bool gathered = total <= c->me_gathered_cap;
		c->me_gathered_cap = 0;
		if (total > c->me_dense_cap) {
			HIP_TRY(hipStreamSynchronize(st));           // (a gather launched ahead that found the buffer too small may still be queued)
			free_set(c->me_dense);
			c->me_dense_cap = 0;
			HIP_TRY(hipMalloc(&c->me_dense, (size_t) (total + total / 4) * sizeof(vdjx_pair)));
			c->me_dense_cap = (size_t) (total + total / 4);
		}
		vdjx_pair* d_dense = (vdjx_pair*) c->me_dense;
		if (nsl && !gathered) {
			vdjx_prof_scope ps(c, "k_gather_pairs");
			hipLaunchKernelGGL(k_gather_pairs, dim3((u32) nsl), dim3(256), 0, st, (const vdjx_pair*) c->me_pairs, (const u32*) c->me_hit, (u32) n, slice_hits, b_sstart, b_off, b_bits, b_pre, d_dense,
			                   (u64) c->me_dense_cap);
		}
		if (device_only) {
			// (vdjx_sam_text: the pairs stay in c->me_dense, the stream is not waited for)
		} else if (async) {
			// the pairs cross PCIe on the copy stream beside whatever the caller does next (vdjx_map_emit_end waits for them); the
			// copy stream waits for the gather through an event, the host does not
			HIP_TRY(hipEventRecord(c->ev_gathered, st));
			HIP_TRY(hipStreamWaitEvent(c->pairs_stream, c->ev_gathered, 0));
			HIP_TRY(hipMemcpyAsync(pairs, d_dense, (size_t) total * sizeof(vdjx_pair), hipMemcpyDeviceToHost, c->pairs_stream));
			HIP_TRY(hipEventRecord(c->ev_pairs_copied, c->pairs_stream));
			lp.mark("me_copy_issue");
		} else {
			HIP_TRY(hipMemcpyAsync(pairs, d_dense, (size_t) total * sizeof(vdjx_pair), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			HIP_TRY(hipGetLastError());
			vdjx_prof_collect(c, false);          // (asynchronous: the gather's timing is collected by the next call that waits for the stream)
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

// ==============================================================================================
// a-10, the text: SAM records of the mapped pairs, formatted on the device
//   replaces output_mapping (quick_map3.c:152-181): per pair two lines
//     "%s\t%d\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n"  name (leading '@' dropped), flag, contig id, pos, read length, mate pos, insert, seq, qual
//   with the stored sequence and qualities of the record that matched (the pool's record: not-ACGT bases read as N, like everywhere here)
// ==============================================================================================
__device__ inline u32 dec_digits(u32 v) { return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : 10u; }
__device__ inline u32 put_dec(char* dst, u32 v) {
	const u32 n = dec_digits(v);
	for (u32 i = n; i-- > 0;) { dst[i] = (char) ('0' + v % 10u); v /= 10u; }
	return n;
}
struct SamSrc {
	const vdjx_pair* pairs; const u64* offs; u32 n_contigs;        // pairs of contig c: [offs[c], offs[c+1])
	const char* ids; const u32* id_off;                            // contig ids, concatenated
	const char* names; const u64* name_off;                        // read names by pair id, concatenated
	const u64* bases; const u64* nmask; vdjx_qrows quals; int rl, W, M;
};
__device__ inline u32 sam_contig_of(const SamSrc& s, u64 i) {
	u32 lo = 0, hi = s.n_contigs;
	while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s.offs[mid] <= i) lo = mid; else hi = mid; }
	return lo;
}
__device__ inline u32 sam_name(const SamSrc& s, u32 pid, const char*& nm) {
	const u64 a = s.name_off[pid], b = s.name_off[pid + 1];
	nm = s.names + a;
	u32 n = (u32) (b - a);
	if (n && nm[0] == '@') { nm++; n--; }
	return n;
}
__global__ void k_sam_len(SamSrc s, u64 total, u32* __restrict__ len) {
	const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total) return;
	const vdjx_pair q = s.pairs[i];
	const u32 c = sam_contig_of(s, i);
	const char* nm;
	const u32 nl = sam_name(s, q.pair_id, nm), cl = s.id_off[c + 1] - s.id_off[c];
	const u32 f1 = 1u | 2u | (q.rc1 ? 0x10u : 0x20u) | 0x40u, f2 = 1u | 2u | (q.rc2 ? 0x10u : 0x20u) | 0x80u;
	const u32 per = nl + cl + dec_digits((u32) q.pos1) + dec_digits((u32) q.pos2) + dec_digits((u32) s.rl) + dec_digits((u32) q.insert) + 2u * (u32) s.rl + 16u;
	len[i] = 2u * per + dec_digits(f1) + dec_digits(f2);
}
__device__ inline char sam_base(const SamSrc& s, u32 rec, int i) {
	if (s.W == 2) {
		if ((s.nmask[rec] >> i) & 1ull) return 'N';
		const ulonglong2 b = ((const ulonglong2*) s.bases)[rec];
		const int sh = 2 * (s.rl - 1 - i);
		const u32 cde = (u32) (sh < 64 ? b.y >> sh : b.x >> (sh - 64)) & 3u;
		return "ATCG"[cde];
	}
	if ((s.nmask[(size_t) rec * s.M + (i >> 6)] >> (i & 63)) & 1ull) return 'N';
	return "ATCG"[(u32) (s.bases[(size_t) rec * s.W + (i >> 5)] >> (62 - 2 * (i & 31))) & 3u];
}
// one wave per pair
__global__ __launch_bounds__(256) void k_sam_write(SamSrc s, u64 total, const u64* __restrict__ at, char* __restrict__ text) {
	__shared__ char hdr[4][2][64];
	const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const u64 i = (u64) blockIdx.x * 4u + wv;
	if (i >= total) return;
	const vdjx_pair q = s.pairs[i];
	const u32 c = sam_contig_of(s, i);
	const char* nm;
	const u32 nl = sam_name(s, q.pair_id, nm);
	const char* cid = s.ids + s.id_off[c];
	const u32 cl = s.id_off[c + 1] - s.id_off[c];
	char* dst = text + at[i];
	for (int line = 0; line < 2; line++) {
		const u32 flag = 1u | 2u | ((line ? q.rc2 : q.rc1) ? 0x10u : 0x20u) | (line ? 0x80u : 0x40u);
		const u32 pos = (u32) (line ? q.pos2 : q.pos1), mpos = (u32) (line ? q.pos1 : q.pos2), rec = line ? q.rec2 : q.rec1;
		char* ha = hdr[wv][0];          // "\t<flag>\t"
		char* hb = hdr[wv][1];          // "\t<pos>\t255\t<rl>M\t=\t<mate pos>\t<insert>\t"
		u32 la = 0, lb = 0;
		if (lane == 0) {
			ha[la++] = '\t'; la += put_dec(ha + la, flag); ha[la++] = '\t';
			hb[lb++] = '\t'; lb += put_dec(hb + lb, pos);
			hb[lb++] = '\t'; hb[lb++] = '2'; hb[lb++] = '5'; hb[lb++] = '5'; hb[lb++] = '\t';
			lb += put_dec(hb + lb, (u32) s.rl); hb[lb++] = 'M'; hb[lb++] = '\t'; hb[lb++] = '='; hb[lb++] = '\t';
			lb += put_dec(hb + lb, mpos); hb[lb++] = '\t';
			lb += put_dec(hb + lb, (u32) q.insert); hb[lb++] = '\t';
		}
		vdjx_wave_lds_fence();         // lane 0's header bytes are read by the other lanes of the wave
		la = (u32) __builtin_amdgcn_readlane((int) la, 0);
		lb = (u32) __builtin_amdgcn_readlane((int) lb, 0);
		for (u32 j = lane; j < nl; j += 64) dst[j] = nm[j];
		dst += nl;
		if (lane < la) dst[lane] = ha[lane];
		dst += la;
		for (u32 j = lane; j < cl; j += 64) dst[j] = cid[j];
		dst += cl;
		if (lane < lb) dst[lane] = hb[lane];
		dst += lb;
		for (int j = (int) lane; j < s.rl; j += 64) dst[j] = sam_base(s, rec, j);
		dst += s.rl;
		if (lane == 0) dst[0] = '\t';
		dst += 1;
		const uint8_t* qr = s.quals.row(rec);
		for (int j = (int) lane; j < s.rl; j += 64) dst[j] = (char) qr[j];
		dst += s.rl;
		if (lane == 0) dst[0] = '\n';
		dst += 1;
		vdjx_wave_lds_fence();         // the next line rewrites the header buffers
	}
}

// read names by pair id for vdjx_sam_text: name_off[n_pairs + 1] into `names`
extern "C" int vdjx_sam_names_load(vdjx_ctx* c, const char* names, const uint64_t* name_off, uint32_t n_pairs) {
	if (!c || !name_off || (n_pairs && !names)) { vdjx_set_error("vdjx_sam_names_load: NULL argument"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	HIP_TRY(hipStreamSynchronize(c->stream));
	free_set(c->d_sam_names); free_set(c->d_sam_noff);
	c->sam_pairs = 0;
	for (uint32_t i = 0; i < n_pairs; i++) if (name_off[i + 1] < name_off[i]) { vdjx_set_error("vdjx_sam_names_load: offsets must not decrease"); return VDJX_EINVAL; }
	const size_t nb = (size_t) name_off[n_pairs];
	HIP_TRY(hipMalloc(&c->d_sam_names, nb + 16));
	HIP_TRY(hipMalloc(&c->d_sam_noff, ((size_t) n_pairs + 1) * 8));
	if (nb) HIP_TRY(hipMemcpy(c->d_sam_names, names, nb, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c->d_sam_noff, name_off, ((size_t) n_pairs + 1) * 8, hipMemcpyHostToDevice));
	c->sam_pairs = n_pairs;
	return VDJX_OK;
}

// per mapped pair the key that orders the SAM records of a pool sharded by pair (vdjx_sam_blocks): contig, then read-1 position,
// then the registration rank of the read-1 record -- the order quick_map_process_contig walks its lists in (quick_map3.c:199-245:
// offsets ascending, the instances of a read sequence in registration order)
#define SAM_KEY_POS_BITS 12
#define SAM_KEY_CONTIG_BITS 20
__global__ void k_sam_keys(SamSrc s, u64 total, const u32* __restrict__ reg_rank, u64* __restrict__ keys) {
	const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total) return;
	const vdjx_pair q = s.pairs[i];
	keys[i] = ((u64) sam_contig_of(s, i) << (32 + SAM_KEY_POS_BITS)) | ((u64) (u32) q.pos1 << 32) | reg_rank[q.rec1];
}

// the SAM text of the mapped pairs of `contigs` in c->d_sam_text (device), the bytes of every pair's two lines in *d_len_out (arena of `db`)
static int sam_text_device(vdjx_ctx* c, vdjx_work& db, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off, SamSrc* src, u64* total_out,
                           u64* nbytes_out, u32** d_len_out) {
	*total_out = 0; *nbytes_out = 0; *d_len_out = nullptr;
	if (!c->d_sam_noff) { vdjx_set_error("vdjx_sam_text: call vdjx_sam_names_load first"); return VDJX_ESTATE; }
	if (c->ri_job) { const int jr = vdjx_ri_join(c); if (jr) return jr; }
	if (!c->ri_pool) { vdjx_set_error("vdjx_sam_text: call vdjx_read_index_build first"); return VDJX_ESTATE; }
	if (c->sam_pairs < c->n_pairs) { vdjx_set_error("vdjx_sam_text: %u names for %u pairs", c->sam_pairs, c->n_pairs); return VDJX_EINVAL; }
	std::vector<uint64_t> offs(n + 1);
	int rc = map_emit_impl(c, contigs, n, len, offs.data(), nullptr, false);
	if (rc) return rc;
	const u64 total = offs[n];
	if (!total) return VDJX_OK;
	rc = map_emit_impl(c, contigs, n, len, offs.data(), (vdjx_pair*) 1, false, true);         // -> c->me_dense, on the stream
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const vdjx_pool* p = c->ri_pool;
	u64 *d_offs, *d_at;
	u32 *d_idoff, *d_len;
	char* d_ids;
	const size_t idb = id_off[n];
	HIP_TRY(db.alloc(&d_offs, n + 1));
	HIP_TRY(db.alloc(&d_idoff, n + 1));
	HIP_TRY(db.alloc(&d_ids, idb + 16));
	HIP_TRY(db.alloc(&d_len, (size_t) total + 1));
	HIP_TRY(db.alloc(&d_at, (size_t) total + 2));
	HIP_TRY(hipMemcpyAsync(d_offs, offs.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_idoff, id_off, (n + 1) * 4, hipMemcpyHostToDevice, st));
	if (idb) HIP_TRY(hipMemcpyAsync(d_ids, ids, idb, hipMemcpyHostToDevice, st));
	SamSrc s{(const vdjx_pair*) c->me_dense, d_offs, (u32) n, d_ids, d_idoff, c->d_sam_names, c->d_sam_noff, p->d_bases, p->d_nmask,
	         vdjx_qrows{p->d_quals, p->d_quals2, p->q_split, p->qstride}, p->rl, p->W, p->M};
	if (total >= (1ull << 32)) { vdjx_set_error("vdjx_sam_text: too many pairs in one call"); return VDJX_ELIMIT; }
	u64 nbytes = 0;
	{
		vdjx_prof_scope ps(c, "k_sam_text");
		hipLaunchKernelGGL(k_sam_len, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, s, total, d_len);
		hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, st, d_len, (u32) total, d_at);
		HIP_TRY(hipMemcpyAsync(&nbytes, d_at + total, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));           // (also: `offs` and the ids have left the host)
		HIP_TRY(hipGetLastError());
		if (nbytes + 1 > c->sam_text_cap) {
			if (c->h_sam_text) (void) hipHostFree(c->h_sam_text);
			free_set(c->d_sam_text);
			c->h_sam_text = nullptr; c->sam_text_cap = 0;
			const size_t want = (size_t) nbytes + (size_t) nbytes / 8 + 4096;
			HIP_TRY(hipMalloc(&c->d_sam_text, want));
			HIP_TRY(hipHostMalloc(&c->h_sam_text, want, hipHostMallocDefault));
			c->sam_text_cap = want;
		}
		hipLaunchKernelGGL(k_sam_write, dim3((u32) ((total + 3) / 4)), dim3(256), 0, st, s, total, d_at, (char*) c->d_sam_text);
	}
	*src = s; *total_out = total; *nbytes_out = nbytes; *d_len_out = d_len;
	c->stats["sam_pairs"] = total;
	c->stats["sam_bytes"] = nbytes;
	return VDJX_OK;
}

extern "C" int vdjx_sam_text(vdjx_ctx* c, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off,
                             const char** out_text, uint64_t* out_bytes) {
	if (!c || !out_text || !out_bytes || (n && (!contigs || !ids || !id_off))) { vdjx_set_error("vdjx_sam_text: NULL argument"); return VDJX_EINVAL; }
	*out_text = ""; *out_bytes = 0;
	if (n == 0) return VDJX_OK;
	vdjx_work db(c);
	SamSrc s;
	u64 total = 0, nbytes = 0;
	u32* d_len;
	int rc = sam_text_device(c, db, contigs, n, len, ids, id_off, &s, &total, &nbytes, &d_len);
	if (rc || !total) return rc;
	hipStream_t st = c->stream;
	HIP_TRY(hipMemcpyAsync(c->h_sam_text, c->d_sam_text, (size_t) nbytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c, false);
	((char*) c->h_sam_text)[nbytes] = 0;
	*out_text = (const char*) c->h_sam_text;
	*out_bytes = nbytes;
	return VDJX_OK;
}

// ==============================================================================================
// the SAM records of a pool that is sharded BY PAIR over several GPUs (no counterpart in the reference): every rank formats the
// records of ITS pairs (vdjx_sam_blocks: text, bytes per pair, an ordering key per pair -- all left on the device), the caller
// brings them to one rank, which lays them out in the reference's order (vdjx_sam_merge: one sort of the keys, one copy)
// ==============================================================================================
extern "C" int vdjx_sam_blocks(vdjx_ctx* c, const char* contigs, size_t n, int len, const char* ids, const uint32_t* id_off, const uint32_t* d_reg_rank,
                               uint64_t* n_blocks, uint64_t* n_bytes, const void** d_keys, const void** d_lens, const void** d_text) {
	if (!c || !n_blocks || !n_bytes || !d_keys || !d_lens || !d_text || (n && (!contigs || !ids || !id_off || !d_reg_rank))) { vdjx_set_error("vdjx_sam_blocks: NULL argument"); return VDJX_EINVAL; }
	*n_blocks = 0; *n_bytes = 0; *d_keys = nullptr; *d_lens = nullptr; *d_text = nullptr;
	if (n == 0) return VDJX_OK;
	if (n >= (1ull << SAM_KEY_CONTIG_BITS) || len >= (1 << SAM_KEY_POS_BITS)) { vdjx_set_error("vdjx_sam_blocks: at most 2^%d contigs of fewer than %d bases per call", SAM_KEY_CONTIG_BITS, 1 << SAM_KEY_POS_BITS); return VDJX_ELIMIT; }
	vdjx_work db(c);
	SamSrc s;
	u64 total = 0, nbytes = 0;
	u32* d_len;
	int rc = sam_text_device(c, db, contigs, n, len, ids, id_off, &s, &total, &nbytes, &d_len);
	if (rc || !total) return rc;
	hipStream_t st = c->stream;
	if (total > c->sam_blk_cap) {
		free_set(c->d_sam_keys); free_set(c->d_sam_lens);
		c->sam_blk_cap = 0;
		HIP_TRY(hipMalloc(&c->d_sam_keys, (size_t) (total + total / 8 + 64) * 8));
		HIP_TRY(hipMalloc(&c->d_sam_lens, (size_t) (total + total / 8 + 64) * 4));
		c->sam_blk_cap = (size_t) (total + total / 8 + 64);
	}
	hipLaunchKernelGGL(k_sam_keys, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, s, total, d_reg_rank, (u64*) c->d_sam_keys);
	HIP_TRY(hipMemcpyAsync(c->d_sam_lens, d_len, (size_t) total * 4, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c, false);
	*n_blocks = total; *n_bytes = nbytes;
	*d_keys = c->d_sam_keys; *d_lens = c->d_sam_lens; *d_text = c->d_sam_text;
	return VDJX_OK;
}

__global__ void k_iota_u32(u32* __restrict__ v, u32 n) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = i;
}
__global__ void k_take_u32(const u32* __restrict__ src, const u32* __restrict__ idx, u32 n, u32* __restrict__ dst) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}
// one wave per block of text: bytes [from[idx[j]], + len) of `src` to `at[j]` of `dst`
__global__ __launch_bounds__(256) void k_sam_move(const char* __restrict__ src, const u64* __restrict__ from, const u32* __restrict__ lens, const u32* __restrict__ idx,
                                                  const u64* __restrict__ at, u32 n, char* __restrict__ dst) {
	const u32 j = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (j >= n) return;
	const u32 b = idx[j], l = lens[b];
	const char* sp = src + from[b];
	char* dp = dst + at[j];
	for (u32 i = lane; i < l; i += 64) dp[i] = sp[i];
}

extern "C" int vdjx_sam_merge(vdjx_ctx* c, uint64_t n_blocks, uint64_t n_bytes, const void* d_keys, const void* d_lens, const void* d_text,
                              const char** out_text, uint64_t* out_bytes) {
	if (!c || !out_text || !out_bytes) { vdjx_set_error("vdjx_sam_merge: NULL argument"); return VDJX_EINVAL; }
	*out_text = ""; *out_bytes = 0;
	if (!n_blocks) return VDJX_OK;
	if (!d_keys || !d_lens || !d_text) { vdjx_set_error("vdjx_sam_merge: NULL buffer"); return VDJX_EINVAL; }
	if (n_blocks >= (1ull << 32)) { vdjx_set_error("vdjx_sam_merge: too many pairs in one call"); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipStream_t st = c->stream;
	vdjx_work db(c);
	const u32 nb = (u32) n_blocks;
	u64 *d_from, *d_at, *d_k2, *d_kin;
	u32 *d_idx, *d_idx2, *d_l2;
	char* d_out;
	HIP_TRY(db.alloc(&d_from, (size_t) nb + 2));
	HIP_TRY(db.alloc(&d_at, (size_t) nb + 2));
	HIP_TRY(db.alloc(&d_kin, (size_t) nb + 1));
	HIP_TRY(db.alloc(&d_k2, (size_t) nb + 1));
	HIP_TRY(db.alloc(&d_idx, (size_t) nb + 1));
	HIP_TRY(db.alloc(&d_idx2, (size_t) nb + 1));
	HIP_TRY(db.alloc(&d_l2, (size_t) nb + 1));
	HIP_TRY(db.alloc(&d_out, (size_t) n_bytes + 16));
	// where every block starts in the text as it came (source after source, every source's blocks in its own order: one running sum)
	hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, st, (const u32*) d_lens, nb, d_from);
	HIP_TRY(hipMemcpyAsync(d_kin, d_keys, (size_t) nb * 8, hipMemcpyDeviceToDevice, st));       // (the sort may use its input as scratch)
	hipLaunchKernelGGL(k_iota_u32, dim3(nb / 256 + 1), dim3(256), 0, st, d_idx, nb);
	int rc = vdjx_sort_pairs(db, st, d_kin, d_k2, d_idx, d_idx2, nb, 64u);
	if (rc) return rc;
	hipLaunchKernelGGL(k_take_u32, dim3(nb / 256 + 1), dim3(256), 0, st, (const u32*) d_lens, d_idx2, nb, d_l2);
	hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, st, d_l2, nb, d_at);
	u64 tot_in = 0, tot_out = 0;
	HIP_TRY(hipMemcpyAsync(&tot_in, d_from + nb, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot_out, d_at + nb, 8, hipMemcpyDeviceToHost, st));
	hipLaunchKernelGGL(k_sam_move, dim3((nb + 3) / 4), dim3(256), 0, st, (const char*) d_text, d_from, (const u32*) d_lens, d_idx2, d_at, nb, d_out);
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (tot_in != n_bytes || tot_out != n_bytes) { vdjx_set_error("vdjx_sam_merge: the blocks' lengths add up to %llu bytes, the text has %llu", (unsigned long long) tot_in, (unsigned long long) n_bytes); return VDJX_EINVAL; }
	if (n_bytes + 1 > c->sam_merge_cap) {
		if (c->h_sam_merge) (void) hipHostFree(c->h_sam_merge);
		c->h_sam_merge = nullptr; c->sam_merge_cap = 0;
		const size_t want = (size_t) n_bytes + (size_t) n_bytes / 8 + 4096;
		HIP_TRY(hipHostMalloc(&c->h_sam_merge, want, hipHostMallocDefault));
		c->sam_merge_cap = want;
	}
	HIP_TRY(hipMemcpyAsync(c->h_sam_merge, d_out, (size_t) n_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	((char*) c->h_sam_merge)[n_bytes] = 0;
	*out_text = (const char*) c->h_sam_merge;
	*out_bytes = n_bytes;
	return VDJX_OK;
}

// rows of `row` bytes: row pos[i] of dst = row i of src (the records a rank receives for its slice of the scan order, each with its place)
__global__ __launch_bounds__(256) void k_rows_scatter(char* __restrict__ dst, const char* __restrict__ src, const u32* __restrict__ pos, u64 n, u32 row) {
	const u64 i = (u64) blockIdx.x * 4u + (threadIdx.x >> 6);
	if (i >= n) return;
	const char* sp = src + i * row;
	char* dp = dst + (u64) pos[i] * row;
	for (u32 b = threadIdx.x & 63u; b < row; b += 64) dp[b] = sp[b];
}
extern "C" int vdjx_rows_scatter(vdjx_ctx* c, void* d_dst, const void* d_src, const uint32_t* d_pos, size_t n, size_t row) {
	if (!c || (n && (!d_dst || !d_src || !d_pos)) || row == 0 || row >= (1u << 20)) { vdjx_set_error("vdjx_rows_scatter: bad argument"); return VDJX_EINVAL; }
	if (!n) return VDJX_OK;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	hipLaunchKernelGGL(k_rows_scatter, dim3((u32) ((n + 3) / 4)), dim3(256), 0, c->stream, (char*) d_dst, (const char*) d_src, d_pos, (u64) n, (u32) row);
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipGetLastError());
	return VDJX_OK;
}
