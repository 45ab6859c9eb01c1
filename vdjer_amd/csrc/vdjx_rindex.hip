// vdjx_rindex.hip -- the read index of the read->contig mapper, built on the device.
//
// replaces add_read_info (quick_map3.c:126-149): a map "read sequence -> its instances in registration order", filled once per
// record by extract (bam_read.c:228,243).  What the mapper kernels (vdjx_score.hip) read:
//   tab[]            open-addressing table over the distinct read sequences (classes), 64-byte slots {sequence, class + 1, members,
//                    CSR start, first weighted entry, weighted entries}: everything the classification of a window offset needs
//                    in one probe (the build's own table names records and is dropped)
//   start[cls]       CSR of the class's READ-1 members in registration order (read-2 instances only ever feed the "read2" map,
//                    quick_map3.c:211-215, which the kernels replace by a class -> last offset table: they are not listed)
//   recs[i], csr_pair[i], csr8[i]  record, pair id and the 8-byte entry of CSR member i
//   pair_r2[2p..]    the pair's read-2 records in registration order (at most two: as-is and reverse complement, bam_read.c:206-244)
//   dstart/d8        per class its DISTINCT read-1 entries with multiplicities (window scoring counts pairs, it does not name them)
// An entry is 8 bytes: class of the pair's read-2 record A (26 bits) | class of B (26) | flags (4) | multiplicity (8); the mapper
// kernels stream billions of them per step, so their size is the kernels' time.
// Everything is counting, scanning and two key sorts (class-major: registration order inside a class for the CSR, a hash of the
// info for the folding); no host pass over the records.  Round 2 did this on the host: 3.8 s at 10 M pairs.
#include "vdjx_common.h"

#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#define NONE32 0xFFFFFFFFu
#define NONE64 0xFFFFFFFFFFFFFFFFull

namespace {
template <typename T> void free_set(T*& p) { if (p) (void) hipFree(p); p = nullptr; }

// ---- exclusive scan of u32 counts (n up to 2^31): out[0..n], out[n] = total -------------------------------------------------
#define RS_BLOCK 2048u
__global__ __launch_bounds__(256) void k_rs_local(const u32* __restrict__ cnt, u32 n, u32* __restrict__ pre, u32* __restrict__ bsum) {
	__shared__ u32 part[4];
	const u32 base = blockIdx.x * RS_BLOCK + threadIdx.x * 8u;
	u32 v[8], s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) { v[i] = base + i < n ? cnt[base + i] : 0u; s += v[i]; }
	const u32 incl = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 run = incl - s;
	for (u32 w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
#pragma unroll
	for (int i = 0; i < 8; i++) { if (base + i < n) pre[base + i] = run; run += v[i]; }
	if (threadIdx.x == 255) bsum[blockIdx.x] = run;
}
// block sums -> their exclusive prefix, one workgroup (up to a few million block sums)
__global__ __launch_bounds__(1024) void k_rs_top(const u32* __restrict__ cnt, u32 n, u32* __restrict__ pre) {
	__shared__ u32 part[16];
	const u32 per = (n + 1023) / 1024;
	const u32 lo = threadIdx.x * per;
	const u32 hi = lo + per < n ? lo + per : n;
	u32 s = 0;
	for (u32 i = lo; i < hi && lo < n; i++) s += cnt[i];
	const u32 incl = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 run = incl - s;
	for (u32 w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
	for (u32 i = lo; i < hi && lo < n; i++) { pre[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) pre[n] = run;
}
__global__ void k_rs_add(u32* __restrict__ pre, u32 n, const u32* __restrict__ bpre) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) pre[i] += bpre[i / RS_BLOCK];
	if (i == 0) pre[n] = bpre[(n + RS_BLOCK - 1) / RS_BLOCK];
}

int scan_u32(vdjx_work& db, hipStream_t st, const u32* d_cnt, u32 n, u32* d_pre) {
	const u32 nb = (n + RS_BLOCK - 1) / RS_BLOCK;
	u32 *bsum, *bpre;
	HIP_TRY(db.alloc(&bsum, nb + 1));
	HIP_TRY(db.alloc(&bpre, nb + 1));
	if (nb) hipLaunchKernelGGL(k_rs_local, dim3(nb), dim3(256), 0, st, d_cnt, n, d_pre, bsum);
	hipLaunchKernelGGL(k_rs_top, dim3(1), dim3(1024), 0, st, bsum, nb, bpre);
	hipLaunchKernelGGL(k_rs_add, dim3(n / 256 + 1), dim3(256), 0, st, d_pre, n, bpre);
	return VDJX_OK;
}

// ---- the table ------------------------------------------------------------------------------------------------------------------
// slots name a representative record (claiming a slot is one 32-bit CAS).  Records holding an 'N' are left out: contigs are
// ACGT-only and can never match them.  W words per read (vdjx_pool: 2, or 5 for reads of more than 64 bases), M per mask.
template <int W, int M>
__global__ void k_ri_insert(const u64* __restrict__ bases, const u64* __restrict__ nmask, u32 R,
                            u32* __restrict__ slots, u32 mask, u32* __restrict__ rec_slot) {
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= R) return;
	u64 nm = 0;
#pragma unroll
	for (int i = 0; i < M; i++) nm |= nmask[(size_t) r * M + i];
	if (nm) { rec_slot[r] = NONE32; return; }
	u64 b[W];
#pragma unroll
	for (int i = 0; i < W; i++) b[i] = bases[(size_t) r * W + i];
	u32 slot = (u32) (ri_hash<W>(b) >> 17) & mask;
	for (;;) {
		u32 cur = slots[slot];
		if (cur == 0) {
			cur = atomicCAS(&slots[slot], 0u, r + 1);
			if (cur == 0) break;
		}
		u64 d = 0;
#pragma unroll
		for (int i = 0; i < W; i++) d |= bases[(size_t) (cur - 1) * W + i] ^ b[i];
		if (!d) break;
		slot = (slot + 1) & mask;
	}
	rec_slot[r] = slot;
}

// classes are numbered in slot order (any numbering serves: a class id is an identity, never an order)
__global__ __launch_bounds__(256) void k_ri_occ(const u32* __restrict__ slots, u32 nslots, u32* __restrict__ bcnt) {
	__shared__ u32 part[4];
	const u32 base = blockIdx.x * RS_BLOCK + threadIdx.x * 8u;
	u32 s = 0;
	if (base + 8 <= nslots) {
		const uint4 a = *(const uint4*) &slots[base], b = *(const uint4*) &slots[base + 4];
		s = (a.x != 0) + (a.y != 0) + (a.z != 0) + (a.w != 0) + (b.x != 0) + (b.y != 0) + (b.z != 0) + (b.w != 0);
	} else
		for (u32 i = base; i < nslots; i++) s += slots[i] != 0;
	const u32 incl = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	if (threadIdx.x == 0) bcnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_ri_number(u32* __restrict__ slots, u32 nslots, const u32* __restrict__ bpre, u32* __restrict__ rep) {
	__shared__ u32 part[4];
	const u32 base = blockIdx.x * RS_BLOCK + threadIdx.x * 8u;
	u32 v[8], s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) { v[i] = base + i < nslots ? slots[base + i] : 0u; s += v[i] != 0; }
	const u32 incl = (u32) vdjx_wave_scan_add((int) s);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	u32 run = bpre[blockIdx.x] + incl - s;
	for (u32 w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
#pragma unroll
	for (int i = 0; i < 8; i++) if (v[i]) { rep[run] = v[i] - 1; slots[base + i] = run + 1; run++; }
}

// ---- per record ---------------------------------------------------------------------------------------------------------------------
#define RI_ERR_PAIR 0      // err[0]: records whose pair id is out of range
#define RI_ERR_R2 1        // err[1]: pairs with more than two read-2 records
// record -> class; the pair's read-2 records in registration order: the two smallest (reg_rank << 32 | record) of the pair, kept
// by a chain of two atomic minima (what loses at the first slot moves on to the second; a third arrival is an error)
__global__ void k_ri_records(const u32* __restrict__ rec_slot, const u32* __restrict__ slots, u32 R, const u32* __restrict__ pair_id,
                             const uint8_t* __restrict__ read_num, const u32* __restrict__ reg_rank, u32 n_pairs,
                             u32* __restrict__ rec_cls, unsigned long long* __restrict__ r2key, u32* __restrict__ err) {
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= R) return;
	const u32 s = rec_slot[r];
	rec_cls[r] = s == NONE32 ? NONE32 : slots[s] - 1;
	const u32 p = pair_id[r];
	if (p >= n_pairs) { atomicAdd(&err[RI_ERR_PAIR], 1u); return; }
	if (read_num[r] == 1) return;
	const unsigned long long v = ((unsigned long long) reg_rank[r] << 32) | r;
	const unsigned long long old = atomicMin(&r2key[2 * (size_t) p], v);
	const unsigned long long y = old > v ? old : v;                  // what does not stay in the first slot
	if (y == NONE64) return;
	const unsigned long long z = atomicMin(&r2key[2 * (size_t) p + 1], y);
	if (z != NONE64) atomicAdd(&err[RI_ERR_R2], 1u);
}
__global__ void k_ri_r2(const unsigned long long* __restrict__ r2key, size_t n, u32* __restrict__ pair_r2) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) pair_r2[i] = r2key[i] == NONE64 ? NONE32 : (u32) r2key[i];
}

// the read-1 members of the classes: how many per class, and (class << 32 | registration rank, record) for the sort
__global__ void k_ri_members(const u32* __restrict__ rec_cls, const uint8_t* __restrict__ read_num, const u32* __restrict__ reg_rank, u32 R,
                             u32* __restrict__ cnt1, u64* __restrict__ keys, u32* __restrict__ vals, u32* __restrict__ n1) {
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 cls = r < R ? rec_cls[r] : NONE32;
	const bool mem = cls != NONE32 && read_num[r] == 1;
	const u32 pos = vdjx_wave_inc(n1, mem);
	if (mem) {
		atomicAdd(&cnt1[cls], 1u);
		keys[pos] = ((u64) cls << 32) | reg_rank[r];
		vals[pos] = r;
	}
}

// what a hit needs in one 8-byte load, in CSR order; and the key of the folding sort: class << 32 | hash of the entry
__global__ void k_ri_info(const u64* __restrict__ keys_sorted, const u32* __restrict__ recs, u32 n1, const u32* __restrict__ pair_id,
                          const uint8_t* __restrict__ is_rc, const u32* __restrict__ rec_cls, const u32* __restrict__ pair_r2,
                          u64* __restrict__ csr8, u32* __restrict__ csr_pair, u64* __restrict__ keys2, u32* __restrict__ vals2) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n1) return;
	const u32 r = recs[i];
	const u32 p = pair_id[r];
	const u32 ra = pair_r2[2 * (size_t) p], rb = pair_r2[2 * (size_t) p + 1];
	const u32 ca = ra != NONE32 ? rec_cls[ra] : NONE32, cb = rb != NONE32 ? rec_cls[rb] : NONE32;
	const u32 fl = (is_rc[r] ? RI_RC : 0u) | (ra != NONE32 && is_rc[ra] ? RI_RCA : 0u) | (rb != NONE32 && is_rc[rb] ? RI_RCB : 0u);
	const u64 e = ri_entry(ca == NONE32 ? RI_ENT_NONE : ca, cb == NONE32 ? RI_ENT_NONE : cb, fl, 1u);
	csr8[i] = e;
	csr_pair[i] = p;
	keys2[i] = (keys_sorted[i] & 0xFFFFFFFF00000000ull) | (u32) (vdjx_mix(e, 0) >> 32);
	vals2[i] = i;
}

// sorted by (class, hash): a member opens a new weighted entry when its class or its entry differs from its predecessor's (a hash
// collision between different entries of a class only splits a group in two: the multiplicities still add up to the members)
__global__ void k_ri_heads(const u64* __restrict__ keys2, const u32* __restrict__ vals2, u32 n1, const u64* __restrict__ csr8, u32* __restrict__ head) {
	const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n1) return;
	u32 h = 1;
	if (j && (keys2[j] >> 32) == (keys2[j - 1] >> 32)) h = csr8[vals2[j]] != csr8[vals2[j - 1]] ? 1u : 0u;
	head[j] = h;
}
__global__ void k_ri_head_pos(const u32* __restrict__ head, const u32* __restrict__ hpre, u32 n1, u32* __restrict__ hpos) {
	const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n1 && head[j]) hpos[hpre[j]] = j;
	if (j == 0) hpos[hpre[n1]] = n1;
}
// a group of m identical members becomes ceil(m / 255) entries (the multiplicity field has 8 bits)
__global__ void k_ri_group_size(const u32* __restrict__ hpos, u32 ng, u32* __restrict__ ne) {
	const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
	if (d < ng) ne[d] = (hpos[d + 1] - hpos[d] + RI_ENT_MAXCNT - 1) / RI_ENT_MAXCNT;
	if (d == ng) ne[d] = 0;
}
__global__ void k_ri_d8(const u32* __restrict__ hpos, const u32* __restrict__ epre, u32 ng, const u32* __restrict__ vals2, const u64* __restrict__ csr8,
                        u64* __restrict__ d8) {
	const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
	if (d >= ng) return;
	const u32 j = hpos[d];
	u32 m = hpos[d + 1] - j;
	const u64 e = csr8[vals2[j]] & ((1ull << 56) - 1ull);
	for (u32 at = epre[d]; m; at++) {
		const u32 c = m < RI_ENT_MAXCNT ? m : RI_ENT_MAXCNT;
		d8[at] = e | ((u64) c << 56);
		m -= c;
	}
}
// the folding sort is class-major like the CSR: the weighted entries of a class start with the group its CSR segment starts with
__global__ void k_ri_dstart(const u32* __restrict__ start, u32 ncls, const u32* __restrict__ hpre, const u32* __restrict__ epre, u32* __restrict__ dstart) {
	const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c <= ncls) dstart[c] = epre[hpre[start[c]]];
}

// the lookup table the mapper reads: one slot per class = the sequence (W words), then {class + 1 | read-1 members << 32}, {CSR start |
// first weighted entry << 32}, {weighted entries}: W + 3 words in a 64-byte slot (one line per probe)
template <int W>
__global__ void k_ri_tab(const u32* __restrict__ rep, u32 ncls, const u64* __restrict__ bases, const u32* __restrict__ cnt1, const u32* __restrict__ start,
                         const u32* __restrict__ dstart, u64* __restrict__ tab, u32 mask) {
	constexpr int SW = VDJX_RI_SLOT_WORDS(W);
	const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= ncls) return;
	u64 b[W];
#pragma unroll
	for (int i = 0; i < W; i++) b[i] = bases[(size_t) rep[c] * W + i];
	u32 slot = (u32) (ri_hash<W>(b) >> 17) & mask;
	while (atomicCAS((u32*) &tab[(size_t) slot * SW + W], 0u, c + 1) != 0u) slot = (slot + 1) & mask;
	u64* sl = tab + (size_t) slot * SW;
#pragma unroll
	for (int i = 0; i < W; i++) sl[i] = b[i];
	((u32*) &sl[W])[1] = cnt1[c];                         // (the low half was claimed by the CAS)
	sl[W + 1] = (u64) start[c] | ((u64) dstart[c] << 32);
	sl[W + 2] = (u64) (dstart[c + 1] - dstart[c]);
}

int sort_pairs(vdjx_work& db, hipStream_t st, u64* k_in, u64* k_out, u32* v_in, u32* v_out, u32 n, unsigned end_bit) {
	size_t tb = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, (size_t) n, 0u, end_bit, st));
	char* tmp;
	HIP_TRY(db.alloc(&tmp, tb + 256));
	HIP_TRY(rocprim::radix_sort_pairs((void*) tmp, tb, k_in, k_out, v_in, v_out, (size_t) n, 0u, end_bit, st));
	return VDJX_OK;
}

inline unsigned bits_for(u64 x) { unsigned b = 1; while (b < 64 && (1ull << b) <= x) b++; return b; }

// the build proper; the four per-record arrays are on the device
int read_index_build_dev(vdjx_ctx* c, vdjx_work& db, const vdjx_pool* pool, const u32* d_pair, const uint8_t* d_rnum, const uint8_t* d_rc,
                         const u32* d_reg, u32 n_pairs) {
	hipStream_t st = c->stream;
	const u32 R = (u32) pool->n_records;
	u32 mask = 1023;
	while ((size_t) mask + 1 < (size_t) R * 2) mask = mask * 2 + 1;
	const u32 nslots = mask + 1;
	const dim3 gR(R / 256 + 1), b256(256);
	u32 *d_rec_slot, *d_rec_cls, *d_err, *d_n1;
	unsigned long long* d_r2key;
	u32* d_slots;                                 // the build's own table: slot -> a record of the class, then -> class + 1
	HIP_TRY(db.alloc(&d_slots, (size_t) nslots));
	HIP_TRY(db.alloc(&d_rec_slot, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_rec_cls, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_err, 4));
	d_n1 = d_err + 2;
	HIP_TRY(db.alloc(&d_r2key, (size_t) n_pairs * 2 + 2));
	HIP_TRY(hipMemsetAsync(d_slots, 0, (size_t) nslots * 4, st));
	HIP_TRY(hipMemsetAsync(d_err, 0, 16, st));
	HIP_TRY(hipMemsetAsync(d_r2key, 0xFF, ((size_t) n_pairs * 2 + 2) * 8, st));
	{
		vdjx_prof_scope ps(c, "k_ri_insert");
		if (pool->W == 2) hipLaunchKernelGGL((k_ri_insert<2, 1>), gR, b256, 0, st, pool->d_bases, pool->d_nmask, R, d_slots, mask, d_rec_slot);
		else hipLaunchKernelGGL((k_ri_insert<VDJX_LONG_W, VDJX_LONG_M>), gR, b256, 0, st, pool->d_bases, pool->d_nmask, R, d_slots, mask, d_rec_slot);
	}
	// classes
	const u32 nsb = (nslots + RS_BLOCK - 1) / RS_BLOCK;
	u32 *d_bcnt, *d_bpre;
	HIP_TRY(db.alloc(&d_bcnt, nsb + 1));
	HIP_TRY(db.alloc(&d_bpre, nsb + 1));
	hipLaunchKernelGGL(k_ri_occ, dim3(nsb), b256, 0, st, d_slots, nslots, d_bcnt);
	hipLaunchKernelGGL(k_rs_top, dim3(1), dim3(1024), 0, st, d_bcnt, nsb, d_bpre);
	u32 ncls = 0;
	HIP_TRY(hipMemcpyAsync(&ncls, d_bpre + nsb, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (ncls >= RI_ENT_NONE) { vdjx_set_error("vdjx_read_index_build: %u distinct read sequences on one GPU (limit 2^26 - 1): shard the pool by pair", ncls); return VDJX_ELIMIT; }
	u32 tmask = 1023;
	while ((size_t) tmask + 1 < (size_t) ncls * 2) tmask = tmask * 2 + 1;
	u32* d_rep;
	HIP_TRY(db.alloc(&d_rep, (size_t) ncls + 1));
	const size_t slot_bytes = (size_t) (pool->W == 2 ? VDJX_RI_SLOT_WORDS(2) : VDJX_RI_SLOT_WORDS(VDJX_LONG_W)) * 8;
	HIP_TRY(hipMalloc(&c->d_ri_tab, ((size_t) tmask + 1) * slot_bytes));
	HIP_TRY(hipMalloc(&c->d_ri_start, ((size_t) ncls + 2) * 4));
	HIP_TRY(hipMalloc(&c->d_ri_cnt1, ((size_t) ncls + 2) * 4));
	HIP_TRY(hipMalloc(&c->d_ri_dstart, ((size_t) ncls + 2) * 4));
	HIP_TRY(hipMalloc(&c->d_pair_r2, ((size_t) n_pairs * 2 + 2) * 4));
	HIP_TRY(hipMemsetAsync(c->d_ri_cnt1, 0, ((size_t) ncls + 2) * 4, st));
	HIP_TRY(hipMemsetAsync(c->d_ri_tab, 0, ((size_t) tmask + 1) * slot_bytes, st));
	hipLaunchKernelGGL(k_ri_number, dim3(nsb), b256, 0, st, d_slots, nslots, d_bpre, d_rep);
	// records: class, read-2 records of the pairs; read-1 members
	u64 *d_keys, *d_keys_s;
	u32 *d_vals, *d_vals_s;
	HIP_TRY(db.alloc(&d_keys, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_keys_s, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_vals, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_vals_s, (size_t) R + 1));
	{
		vdjx_prof_scope ps(c, "k_ri_records");
		hipLaunchKernelGGL(k_ri_records, gR, b256, 0, st, d_rec_slot, d_slots, R, d_pair, d_rnum, d_reg, n_pairs, d_rec_cls, d_r2key, d_err);
		hipLaunchKernelGGL(k_ri_r2, dim3((unsigned) (((size_t) n_pairs * 2 + 2) / 256 + 1)), b256, 0, st, d_r2key, (size_t) n_pairs * 2 + 2, c->d_pair_r2);
		hipLaunchKernelGGL(k_ri_members, gR, b256, 0, st, d_rec_cls, d_rnum, d_reg, R, c->d_ri_cnt1, d_keys, d_vals, d_n1);
	}
	int rc = scan_u32(db, st, c->d_ri_cnt1, ncls + 1, c->d_ri_start);         // (cnt1[ncls] = 0: start[ncls] = start[ncls + 1] = members)
	if (rc) return rc;
	u32 h_err[4] = {0, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (h_err[RI_ERR_PAIR]) { vdjx_set_error("vdjx_read_index_build: %u records with a pair id >= n_pairs=%u", h_err[RI_ERR_PAIR], n_pairs); return VDJX_EINVAL; }
	if (h_err[RI_ERR_R2]) { vdjx_set_error("a pair has more than two read-2 records (read names must be unique per pair)"); return VDJX_EINVAL; }
	const u32 n1 = h_err[2];
	// CSR order = (class, registration rank)
	HIP_TRY(hipMalloc(&c->d_ri_recs, ((size_t) n1 + 1) * 4));
	HIP_TRY(hipMalloc(&c->d_ri_csr8, ((size_t) n1 + 1) * 8));
	HIP_TRY(hipMalloc(&c->d_ri_csr_pair, ((size_t) n1 + 1) * 4));
	u32 nd = 0;
	if (n1) {
		{
			vdjx_prof_scope ps(c, "ri_sort_members");
			rc = sort_pairs(db, st, d_keys, d_keys_s, d_vals, d_vals_s, n1, 32u + bits_for(ncls));
			if (rc) return rc;
		}
		HIP_TRY(hipMemcpyAsync(c->d_ri_recs, d_vals_s, (size_t) n1 * 4, hipMemcpyDeviceToDevice, st));
		const dim3 g1(n1 / 256 + 1);
		// (the unsorted key/value buffers are free again: they take the folding sort's input)
		hipLaunchKernelGGL(k_ri_info, g1, b256, 0, st, d_keys_s, c->d_ri_recs, n1, d_pair, d_rc, d_rec_cls, c->d_pair_r2, c->d_ri_csr8, c->d_ri_csr_pair, d_keys, d_vals);
		{
			vdjx_prof_scope ps(c, "ri_sort_infos");
			rc = sort_pairs(db, st, d_keys, d_keys_s, d_vals, d_vals_s, n1, 32u + bits_for(ncls));
			if (rc) return rc;
		}
		u32 *d_head, *d_hpre, *d_hpos, *d_ne, *d_epre;
		HIP_TRY(db.alloc(&d_head, (size_t) n1 + 1));
		HIP_TRY(db.alloc(&d_hpre, (size_t) n1 + 2));
		HIP_TRY(db.alloc(&d_hpos, (size_t) n1 + 2));
		HIP_TRY(db.alloc(&d_ne, (size_t) n1 + 2));
		HIP_TRY(db.alloc(&d_epre, (size_t) n1 + 3));
		hipLaunchKernelGGL(k_ri_heads, g1, b256, 0, st, d_keys_s, d_vals_s, n1, c->d_ri_csr8, d_head);
		rc = scan_u32(db, st, d_head, n1, d_hpre);
		if (rc) return rc;
		u32 ng = 0;
		HIP_TRY(hipMemcpyAsync(&ng, d_hpre + n1, 4, hipMemcpyDeviceToHost, st));
		hipLaunchKernelGGL(k_ri_head_pos, g1, b256, 0, st, d_head, d_hpre, n1, d_hpos);
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		hipLaunchKernelGGL(k_ri_group_size, dim3(ng / 256 + 1), b256, 0, st, d_hpos, ng, d_ne);
		rc = scan_u32(db, st, d_ne, ng + 1, d_epre);          // (ne[ng] = 0: epre[ng] = epre[ng + 1] = entries)
		if (rc) return rc;
		HIP_TRY(hipMemcpyAsync(&nd, d_epre + ng, 4, hipMemcpyDeviceToHost, st));
		hipLaunchKernelGGL(k_ri_dstart, dim3(ncls / 256 + 1), b256, 0, st, c->d_ri_start, ncls, d_hpre, d_epre, c->d_ri_dstart);
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMalloc(&c->d_ri_d8, ((size_t) nd + 1) * 8));
		hipLaunchKernelGGL(k_ri_d8, dim3(ng / 256 + 1), b256, 0, st, d_hpos, d_epre, ng, d_vals_s, c->d_ri_csr8, c->d_ri_d8);
	} else {
		HIP_TRY(hipMalloc(&c->d_ri_d8, 8));
		HIP_TRY(hipMemsetAsync(c->d_ri_dstart, 0, ((size_t) ncls + 2) * 4, st));
	}
	{	// the mapper's table, now that the classes' sizes and starts are known
		vdjx_prof_scope ps(c, "k_ri_tab");
		if (pool->W == 2) hipLaunchKernelGGL(k_ri_tab<2>, dim3(ncls / 256 + 1), b256, 0, st, d_rep, ncls, pool->d_bases, c->d_ri_cnt1, c->d_ri_start, c->d_ri_dstart, (u64*) c->d_ri_tab, tmask);
		else hipLaunchKernelGGL(k_ri_tab<VDJX_LONG_W>, dim3(ncls / 256 + 1), b256, 0, st, d_rep, ncls, pool->d_bases, c->d_ri_cnt1, c->d_ri_start, c->d_ri_dstart, (u64*) c->d_ri_tab, tmask);
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	vdjx_prof_collect(c, false);
	c->stats["read_index_r1_members"] = n1;
	c->stats["read_index_r1_distinct"] = nd;
	c->stats["read_index_classes"] = ncls;
	c->ri_tab_mask = tmask;
	c->n_pairs = n_pairs;
	c->n_classes = ncls;
	c->ri_pool = pool;
	return VDJX_OK;
}

void drop_index(vdjx_ctx* c) {
	free_set(c->d_ri_tab); free_set(c->d_ri_start); free_set(c->d_ri_recs); free_set(c->d_ri_cnt1);
	free_set(c->d_pair_r2); free_set(c->d_ri_csr8); free_set(c->d_ri_csr_pair); free_set(c->d_ri_dstart); free_set(c->d_ri_d8);
	c->ri_pool = nullptr;
	c->me_key = 0;
}

int check_args(vdjx_ctx* c, const vdjx_pool* pool, const void* a, const void* b, const void* d, const void* e, const char* who) {
	if (!c || !pool || !a || !b || !d || !e) { vdjx_set_error("%s: NULL argument", who); return VDJX_EINVAL; }
	if (pool->ctx != c) { vdjx_set_error("%s: pool belongs to another context", who); return VDJX_EINVAL; }
	if (pool->pending_bad) { vdjx_set_error("%s: the pool is still loading (vdjx_pool_wait)", who); return VDJX_ESTATE; }
	return VDJX_OK;
}
}  // namespace

// the same sort for a caller with its own allocator (vdjx_kmer.hip: node numbering): tmp == nullptr asks for the scratch size
int vdjx_sort_pairs_raw(void* tmp, size_t* tmp_bytes, hipStream_t st, u64* k_in, u64* k_out, u32* v_in, u32* v_out, u32 n, unsigned end_bit) {
	HIP_TRY(rocprim::radix_sort_pairs(tmp, *tmp_bytes, k_in, k_out, v_in, v_out, (size_t) n, 0u, end_bit, st));
	return VDJX_OK;
}

// (the scorers sort their strings with the same helper: vdjx_score.hip)
int vdjx_sort_pairs(vdjx_work& db, hipStream_t st, u64* k_in, u64* k_out, u32* v_in, u32* v_out, u32 n, unsigned end_bit) {
	return sort_pairs(db, st, k_in, k_out, v_in, v_out, n, end_bit);
}

extern "C" int vdjx_read_index_build_device(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* d_pair_id, const uint8_t* d_read_num,
                                            const uint8_t* d_is_rc, const uint32_t* d_reg_rank, uint32_t n_pairs) {
	int rc = check_args(c, pool, d_pair_id, d_read_num, d_is_rc, d_reg_rank, "vdjx_read_index_build_device");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	HIP_TRY(hipStreamSynchronize(c->stream));           // nothing in flight may still read the index that is replaced
	drop_index(c);
	vdjx_work db(c);
	rc = read_index_build_dev(c, db, pool, d_pair_id, d_read_num, d_is_rc, d_reg_rank, n_pairs);
	if (rc) drop_index(c);
	return rc;
}

extern "C" int vdjx_read_index_build(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* pair_id,
                                     const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs) {
	int rc = check_args(c, pool, pair_id, read_num, is_rc, reg_rank, "vdjx_read_index_build");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	HIP_TRY(hipStreamSynchronize(c->stream));
	drop_index(c);
	vdjx_work db(c);
	const size_t R = pool->n_records;
	u32 *d_pair, *d_reg;
	uint8_t *d_rnum, *d_rc;
	HIP_TRY(db.alloc(&d_pair, R + 1));
	HIP_TRY(db.alloc(&d_reg, R + 1));
	HIP_TRY(db.alloc(&d_rnum, R + 1));
	HIP_TRY(db.alloc(&d_rc, R + 1));
	if (R) {
		HIP_TRY(hipMemcpyAsync(d_pair, pair_id, R * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(d_reg, reg_rank, R * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(d_rnum, read_num, R, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(d_rc, is_rc, R, hipMemcpyHostToDevice, c->stream));
	}
	rc = read_index_build_dev(c, db, pool, d_pair, d_rnum, d_rc, d_reg, n_pairs);
	if (rc) drop_index(c);
	return rc;
}
