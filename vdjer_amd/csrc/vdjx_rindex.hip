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
#include <thread>
#include <condition_variable>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_merge.hpp>

#define NONE32 0xFFFFFFFFu
#define NONE64 0xFFFFFFFFFFFFFFFFull

namespace {
template <typename T> void free_set(T*& p) { if (p) (void) hipFree(p); p = nullptr; }
// *p holds at least `bytes` bytes (contents undefined): kept if large enough, replaced (with a margin) if not
template <typename T> hipError_t ri_keep(T** p, size_t* cap, size_t bytes) {
	if (*p && *cap >= bytes) return hipSuccess;
	if (*p) (void) hipFree(*p);
	*p = nullptr; *cap = 0;
	const size_t want = bytes + bytes / 8 + 256;
	const hipError_t e = hipMalloc((void**) p, want);
	if (e == hipSuccess) *cap = want;
	return e;
}

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

// ---- pools made of couples (vdjx_pool::sym: record 2i + 1 is the reverse complement of record 2i -- what add_to_buffer writes,
// bam_read.c:206-244; reads of up to 64 bases).  A couple is inserted ONCE, under the smaller of its two sequences: half the
// insertions into a table of half the size.  The sequence under the key gets class 2c, its reverse complement class 2c + 1 (c: the
// key's number in slot order); a read that is its own reverse complement (even lengths) has class 2c only.
#define RI_CS_STR 0x80000000u         // couple_slot: record 2i holds the LARGER sequence (its class is the odd one, record 2i + 1's the even one)
#define RI_CS_PAL 0x40000000u         // ... the two are the same sequence
#define RI_CS_SLOT 0x3FFFFFFFu
#define RI_SLOT_PAL 0x80000000u       // slots[]: the representative record + 1 | this while the table is built (k_ri_number_sym: the key's number + 1)
__global__ void k_ri_insert_sym(const u64* __restrict__ bases, const u64* __restrict__ nmask, u32 R2, int rl,
                                u32* __restrict__ slots, u32 mask, u32* __restrict__ couple_slot) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= R2) return;
	if (nmask[2 * (size_t) i]) { couple_slot[i] = NONE32; return; }          // (the mask of record 2i + 1 is this one reversed)
	const ulonglong2 f = ((const ulonglong2*) bases)[2 * (size_t) i];
	u64 rh, rlo;
	vdjx_read_rc(f.x, f.y, rl, rh, rlo);
	const bool flip = rh < f.x || (rh == f.x && rlo < f.y);
	const bool pal = rh == f.x && rlo == f.y;
	u64 b[2];
	b[0] = flip ? rh : f.x;
	b[1] = flip ? rlo : f.y;
	const u32 rep = 2u * i + (flip ? 1u : 0u);                                // the record that holds the key
	u32 slot = (u32) (ri_hash<2>(b) >> 17) & mask;
	for (;;) {
		u32 cur = slots[slot];
		if (cur == 0) {
			cur = atomicCAS(&slots[slot], 0u, (rep + 1) | (pal ? RI_SLOT_PAL : 0u));
			if (cur == 0) break;
		}
		const ulonglong2 o = ((const ulonglong2*) bases)[(cur & ~RI_SLOT_PAL) - 1];
		if (o.x == b[0] && o.y == b[1]) break;
		slot = (slot + 1) & mask;
	}
	couple_slot[i] = slot | (flip ? RI_CS_STR : 0u) | (pal ? RI_CS_PAL : 0u);
}
// the keys numbered in slot order; rep[2c], rep[2c + 1]: a record that holds class 2c's sequence, one that holds class 2c + 1's (its
// neighbour; none for a sequence that is its own reverse complement)
__global__ __launch_bounds__(256) void k_ri_number_sym(u32* __restrict__ slots, u32 nslots, const u32* __restrict__ bpre, u32* __restrict__ rep) {
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
	for (int i = 0; i < 8; i++) if (v[i]) {
		const u32 r = (v[i] & ~RI_SLOT_PAL) - 1u;
		*(uint2*) &rep[2 * (size_t) run] = make_uint2(r, (v[i] & RI_SLOT_PAL) ? NONE32 : r ^ 1u);
		slots[base + i] = run + 1;
		run++;
	}
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
#define RI_ERR_ORDER 2     // err[2]: records registered before their predecessor: none -- the records are in registration order; one -- two runs
#define RI_ERR_RUNS 3      // err[3]: ... that meet at this record (a merge); more -- the members are sorted by rank
#define RI_RB 2048u        // records per workgroup of the record passes: 8 per thread, all of a thread's loads in flight together
#define RI_RCBIT 0x80000000u
// record -> class (| is_rc in the top bit: what the entry of a pair needs of its mates in ONE load); the pair's read-2 records in
// registration order: the two smallest (reg_rank << 32 | record) of the pair, kept by a chain of two atomic minima (what loses at the
// first slot moves on to the second; a third arrival is an error); read-1 members per workgroup (their compaction follows)
// (SYM: rec_slot is per COUPLE, k_ri_insert_sym; the two records of a couple ask for the same slot)
template <bool SYM>
__global__ __launch_bounds__(256) void k_ri_records(const u32* __restrict__ rec_slot, const u32* __restrict__ slots, u32 R,
                                                    const u32* __restrict__ pair_id, const uint8_t* __restrict__ read_num, const uint8_t* __restrict__ is_rc,
                                                    const u32* __restrict__ reg_rank, u32 n_pairs, u32* __restrict__ rec_cls,
                                                    unsigned long long* __restrict__ r2key, u32* __restrict__ err, u32* __restrict__ bcnt) {
	__shared__ u32 part[4];
	const u32 base = blockIdx.x * RI_RB + threadIdx.x;
	u32 s[8], p[8], reg[8], regp[8];
	uint8_t rn[8], rc[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const u32 r = base + (u32) i * 256u;
		const bool in = r < R;
		s[i] = in ? rec_slot[SYM ? r >> 1 : r] : NONE32;
		p[i] = in ? pair_id[r] : 0u;
		rn[i] = in ? read_num[r] : (uint8_t) 0;
		rc[i] = in ? is_rc[r] : (uint8_t) 0;
		reg[i] = in ? reg_rank[r] : 0u;
		regp[i] = in && r ? reg_rank[r - 1] : 0u;
	}
	unsigned long long mate[8];                            // SYM: the neighbouring record's (rank, record) if it is a read-2 record of the same pair (in range), else all-ones
#pragma unroll
	for (int i = 0; i < 8; i++) {
		mate[i] = NONE64;
		if (!SYM) continue;
		const u32 r = base + (u32) i * 256u;
		const u32 pp = (u32) __shfl_xor((int) p[i], 1), rp = (u32) __shfl_xor((int) reg[i], 1), np = (u32) __shfl_xor((int) rn[i], 1);
		if ((r ^ 1u) < R && rn[i] == 2 && np == 2u && pp == p[i] && p[i] < n_pairs) mate[i] = ((unsigned long long) rp << 32) | (r ^ 1u);
	}
	u32 cls[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		if (!SYM) { cls[i] = s[i] == NONE32 ? RI_ENT_NONE : slots[s[i]] - 1u; continue; }
		const u32 r = base + (u32) i * 256u;
		// the record's own side of its couple: the odd class if it holds the larger sequence
		const u32 odd = (s[i] & RI_CS_PAL) ? 0u : ((s[i] >> 31) ^ (r & 1u));
		cls[i] = s[i] == NONE32 ? RI_ENT_NONE : 2u * (slots[s[i] & RI_CS_SLOT] - 1u) + odd;
	}
	u32 mine = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const u32 r = base + (u32) i * 256u;
		if (r >= R) continue;
		rec_cls[r] = cls[i] | (rc[i] ? RI_RCBIT : 0u);
		if (r && reg[i] < regp[i]) { atomicAdd(&err[RI_ERR_ORDER], 1u); atomicMax(&err[RI_ERR_RUNS], r); }
		if (p[i] >= n_pairs) { atomicAdd(&err[RI_ERR_PAIR], 1u); continue; }
		if (rn[i] == 1) { mine += cls[i] != RI_ENT_NONE; continue; }
		const unsigned long long v = ((unsigned long long) reg[i] << 32) | r;
		// one record into the pair's two places: the smaller stays in the first, what loses moves on to the second, a third arrival is an error
		auto chain = [&](unsigned long long x) {
			const unsigned long long old = atomicMin(&r2key[2 * (size_t) p[i]], x);
			const unsigned long long y = old > x ? old : x;
			if (y == NONE64) return;
			if (atomicMin(&r2key[2 * (size_t) p[i] + 1], y) != NONE64) atomicAdd(&err[RI_ERR_R2], 1u);
		};
		if (SYM && mate[i] != NONE64) {
			// couples: the read-2 records of a pair are a read and its reverse complement next to each other -- the even lane brings both,
			// two atomics instead of four (someone else in the first place already: one by one after all)
			if (r & 1u) continue;
			const unsigned long long v0 = v < mate[i] ? v : mate[i], v1 = v < mate[i] ? mate[i] : v;
			if (atomicMin(&r2key[2 * (size_t) p[i]], v0) == NONE64) { if (atomicMin(&r2key[2 * (size_t) p[i] + 1], v1) != NONE64) atomicAdd(&err[RI_ERR_R2], 1u); }
			else { chain(v0); chain(v1); }
			continue;
		}
		chain(v);
	}
	const u32 incl = (u32) vdjx_wave_scan_add((int) mine);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	if (threadIdx.x == 0) bcnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// the read-1 members of the classes, compacted IN RECORD ORDER (a workgroup's members start at bpre[workgroup]: no global counter):
// per member its record, pair, registration rank, the 8-byte entry a hit needs (the classes and orientations of the pair's read-2
// records) and the sort key class << 32 | member; members per class
// (the pair's read-2 records come straight from the two minima of k_ri_records and are written, as records, where the mapper looks
// for them: pair_r2 -- only the pairs of members are ever looked up there)
__global__ __launch_bounds__(256) void k_ri_members(const u32* __restrict__ rec_cls, const uint8_t* __restrict__ read_num, const u32* __restrict__ reg_rank,
                                                    const u32* __restrict__ pair_id, const unsigned long long* __restrict__ r2key, u32* __restrict__ pair_r2, u32 R, u32 n_pairs,
                                                    const u32* __restrict__ bpre, u32* __restrict__ cnt1, u64* __restrict__ mkey, u32* __restrict__ m_reg,
                                                    ulonglong2* __restrict__ m_pack) {
	__shared__ u32 wcnt[8][4];
	const u32 base = blockIdx.x * RI_RB + threadIdx.x;
	const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	u32 cr[8], p[8];
	u64 bal[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const u32 r = base + (u32) i * 256u;
		const bool in = r < R;
		cr[i] = in ? rec_cls[r] : RI_ENT_NONE;
		p[i] = in ? pair_id[r] : NONE32;
		const bool mem = in && (cr[i] & RI_ENT_NONE) != RI_ENT_NONE && read_num[r] == 1 && p[i] < n_pairs;
		if (!mem) p[i] = NONE32;
		bal[i] = __ballot(mem);
		if (lane == 0) wcnt[i][wv] = (u32) __popcll(bal[i]);
	}
	__syncthreads();
	u32 run = bpre[blockIdx.x];
	uint2 r2[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		r2[i] = make_uint2(NONE32, NONE32);
		if (p[i] == NONE32) continue;
		const ulonglong2 kk = *(const ulonglong2*) &r2key[2 * (size_t) p[i]];
		r2[i] = make_uint2(kk.x == NONE64 ? NONE32 : (u32) kk.x, kk.y == NONE64 ? NONE32 : (u32) kk.y);
		*(uint2*) &pair_r2[2 * (size_t) p[i]] = r2[i];
	}
	u32 ca[8], cb[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		ca[i] = r2[i].x != NONE32 ? rec_cls[r2[i].x] : RI_ENT_NONE;
		cb[i] = r2[i].y != NONE32 ? rec_cls[r2[i].y] : RI_ENT_NONE;
	}
#pragma unroll
	for (int i = 0; i < 8; i++) {
		u32 at = run;
		for (u32 w = 0; w < wv; w++) at += wcnt[i][w];
		run += wcnt[i][0] + wcnt[i][1] + wcnt[i][2] + wcnt[i][3];
		if (p[i] == NONE32) continue;
		const u32 r = base + (u32) i * 256u;
		const u32 pos = at + (u32) __popcll(bal[i] & ((1ull << lane) - 1ull));
		const u32 cls = cr[i] & RI_ENT_NONE;
		const u32 fl = (cr[i] & RI_RCBIT ? RI_RC : 0u) | (ca[i] != RI_ENT_NONE && (ca[i] & RI_RCBIT) ? RI_RCA : 0u) | (cb[i] != RI_ENT_NONE && (cb[i] & RI_RCBIT) ? RI_RCB : 0u);
		atomicAdd(&cnt1[cls], 1u);
		mkey[pos] = ((u64) cls << 32) | pos;
		m_reg[pos] = reg_rank[r];
		// record, pair and entry in ONE 16-byte word: the CSR order gathers them by member (three separate arrays were three random
		// line fills per member: 7.7 GB of traffic for 0.3 GB of payload)
		m_pack[pos] = make_ulonglong2((u64) r | ((u64) p[i] << 32), ri_entry(ca[i] & RI_ENT_NONE, cb[i] & RI_ENT_NONE, fl, 1u));
	}
}
// the read-1 members before record err[RI_ERR_RUNS] -- where the second run of registration ranks starts, if the records are two runs:
// the members of the workgroups before that record's (bpre) + those of its own workgroup up to the record.  One workgroup, right behind
// k_ri_records: the number reaches the host with the other counts (a search over the finished members cost a wait of its own).
__global__ __launch_bounds__(256) void k_ri_split(const u32* __restrict__ err, const u32* __restrict__ rec_cls, const uint8_t* __restrict__ read_num,
                                                  const u32* __restrict__ pair_id, u32 R, u32 n_pairs, const u32* __restrict__ bpre, u32* __restrict__ out) {
	__shared__ u32 part[4];
	const u32 rs = err[RI_ERR_RUNS] < R ? err[RI_ERR_RUNS] : R;
	const u32 blk = rs / RI_RB;
	u32 mine = 0;
	for (u32 r = blk * RI_RB + threadIdx.x; r < rs; r += 256u)
		mine += (rec_cls[r] & RI_ENT_NONE) != RI_ENT_NONE && read_num[r] == 1 && pair_id[r] < n_pairs;
	const u32 incl = (u32) vdjx_wave_scan_add((int) mine);
	if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
	__syncthreads();
	if (threadIdx.x == 0) *out = bpre[blk] + part[0] + part[1] + part[2] + part[3];
}

// CSR order = (class, registration rank): the sorted key names the member
__global__ void k_ri_csr(const u64* __restrict__ mkey_sorted, u32 n1, const ulonglong2* __restrict__ m_pack,
                         u32* __restrict__ recs, u32* __restrict__ csr_pair, u64* __restrict__ csr8) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n1) return;
	const ulonglong2 v = m_pack[(u32) mkey_sorted[i]];
	recs[i] = (u32) v.x;
	csr_pair[i] = (u32) (v.x >> 32);
	csr8[i] = v.y;
}

// ---- the weighted entries: identical read-1 entries of a class folded into one with a multiplicity ------------------------------------
// (window scoring counts pairs, it does not name them.)  Class c's entries go to d8[start[c] ...): the region of its CSR members,
// of which they use the first dcnt[c] -- no scan, no second pass; ANY grouping whose multiplicities add up to the members is a
// valid result, which is what the overflow paths rely on.  The classes of up to RI_FOLD_SMALL members are folded by their members'
// own threads (k_ri_fold_members); the larger ones (reads of deep clones: hundreds of members, tens of distinct entries) are left to
// the waves of a workgroup, one class at a time, through a table in LDS.
#define RI_FOLD_SMALL 8u
#define RI_FM_PER 1024u               // members per workgroup of k_ri_fold_members
#define RI_FOLD_WAVE 256u             // up to here a wave folds the class through its own LDS table (it cannot fill: 512 slots)
#define RI_FOLD_SLOTS 512u
#define RI_FOLD_BIG_SLOTS 8192u       // beyond, a workgroup per class (k_ri_fold_big); a class with more distinct entries than 3/4 of this
#define RI_FOLD_BIG_THREADS 512u      // writes the rest unfolded
__device__ inline u32 ri_fold_slot(u64 e, u32 slots) { return (u32) (vdjx_mix(e, 0) >> 40) & (slots - 1u); }
// insert e into an LDS table of (key, count); returns true if this lane made the key
__device__ inline bool ri_fold_insert(unsigned long long* tk, u32* tc, u32 slots, u64 e) {
	u32 slot = ri_fold_slot(e, slots);
	bool fresh = false;
	for (;;) {
		unsigned long long cur = vdjx_peek(&tk[slot]);
		if (cur == NONE64) { cur = atomicCAS(&tk[slot], NONE64, (unsigned long long) e); if (cur == NONE64) { fresh = true; break; } }
		if (cur == (unsigned long long) e) break;
		slot = (slot + 1) & (slots - 1u);
	}
	atomicAdd(&tc[slot], 1u);
	return fresh;
}
// the classes of up to RI_FOLD_SMALL members -- nearly all of them, most with one -- a thread per MEMBER in CSR order: its class's
// entries are its neighbours' (a thread per class walked its members one dependent load after the other: 0.77 ms at 10 M pairs,
// 92 % of the wave-cycles waiting).  A member whose entry no earlier member of the class holds writes it, with the number of members
// that hold it, at its rank among such; the class's first member writes their number.
__global__ __launch_bounds__(256) void k_ri_fold_members(const u64* __restrict__ by_class, u32 n1, const u32* __restrict__ start, const u32* __restrict__ cnt1,
                                                         const u64* __restrict__ csr8, u64* __restrict__ d8, u32* __restrict__ dcnt,
                                                         unsigned long long* __restrict__ n_entries, u32* __restrict__ giant, u32* __restrict__ n_giant) {
	// the larger classes that START among this workgroup's members (at most RI_FM_PER / (RI_FOLD_SMALL + 1) + 1 of them) are folded by its
	// waves afterwards, one class at a time through a table in LDS; those of more than RI_FOLD_WAVE members go on the list of k_ri_fold_big.
	// (RI_FM_PER members per workgroup, four per thread: with one per thread the kernel was 74,000 workgroups of 25 KB of LDS at 10 M pairs
	// and their dispatch, not their work, its 0.65 ms)
	__shared__ u32 s_big[RI_FM_PER / (RI_FOLD_SMALL + 1) + 2];
	__shared__ u32 s_nbig, tot;
	__shared__ unsigned long long tkey[4][RI_FOLD_SLOTS];
	__shared__ u32 tcnt[4][RI_FOLD_SLOTS];
	if (threadIdx.x == 0) { s_nbig = 0; tot = 0; }
	__syncthreads();
	u32 made = 0;
	for (u32 rep = 0; rep < RI_FM_PER / 256u; rep++) {
	const u32 i = blockIdx.x * RI_FM_PER + rep * 256u + threadIdx.x;
	bool is_big = false, is_giant = false;
	u32 cls = 0;
	if (i < n1) {
		cls = (u32) (by_class[i] >> 32);
		const u32 s = start[cls], m = cnt1[cls];
		if (m > RI_FOLD_SMALL && i == s) { is_giant = m > RI_FOLD_WAVE; is_big = !is_giant; }
		if (m <= RI_FOLD_SMALL) {
			const u32 j = i - s;
			u64 e[RI_FOLD_SMALL];
#pragma unroll
			for (u32 q = 0; q < RI_FOLD_SMALL; q++) e[q] = q < m ? csr8[s + q] & ((1ull << 56) - 1ull) : 0ull;
			u32 lead = 0;                              // bit p: member p's entry is not held by an earlier member
#pragma unroll
			for (u32 p_ = 0; p_ < RI_FOLD_SMALL; p_++) {
				bool first = p_ < m;
#pragma unroll
				for (u32 q = 0; q < p_; q++) first = first && e[q] != e[p_];
				lead |= (u32) first << p_;
			}
			if ((lead >> j) & 1u) {
				u64 mine = 0;
				u32 mult = 0;
#pragma unroll
				for (u32 q = 0; q < RI_FOLD_SMALL; q++) if (q == j) mine = e[q];
#pragma unroll
				for (u32 q = 0; q < RI_FOLD_SMALL; q++) mult += q < m && e[q] == mine;
				d8[s + (u32) __popc(lead & ((1u << j) - 1u))] = mine | ((u64) mult << 56);
				made++;
			}
			if (j == 0) dcnt[cls] = (u32) __popc(lead);
		}
	}
	if (is_big) s_big[atomicAdd(&s_nbig, 1u)] = cls;
	{
		const u32 ag = vdjx_wave_inc(n_giant, is_giant);
		if (is_giant) giant[ag] = cls;
	}
	}
	// (the statistic: one bump per workgroup -- a bump per wave was 300,000 atomics on one address, 2 ms of the kernel's 2.2; so was a
	// global list of the larger classes)
	const u32 incl = (u32) vdjx_wave_scan_add((int) made);
	if ((threadIdx.x & 63u) == 63u && incl) atomicAdd(&tot, incl);
	__syncthreads();
	const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	unsigned long long* tk = tkey[wv];
	u32* tc = tcnt[wv];
	for (u32 b = wv; b < s_nbig; b += 4) {
		const u32 cc = s_big[b], m = cnt1[cc], s = start[cc];
		for (u32 q = lane; q < RI_FOLD_SLOTS; q += 64) { tk[q] = NONE64; tc[q] = 0; }
		vdjx_wave_lds_fence();
		for (u32 at = 0; at < m; at += 64)
			if (at + lane < m) (void) ri_fold_insert(tk, tc, RI_FOLD_SLOTS, csr8[s + at + lane] & ((1ull << 56) - 1ull));
		vdjx_wave_lds_fence();
		u32 out = 0;
		for (u32 q = 0; q < RI_FOLD_SLOTS; q += 64) {
			const u32 n = tc[q + lane];                                    // (n <= 256: one entry, or two)
			const u32 pieces = (n + RI_ENT_MAXCNT - 1) / RI_ENT_MAXCNT;      // the multiplicity field has 8 bits
			const u32 inc2 = (u32) vdjx_wave_scan_add((int) pieces);
			u32 o = s + out + inc2 - pieces, left = n;
			const u64 e = (u64) tk[q + lane];
			while (left) { const u32 x = left < RI_ENT_MAXCNT ? left : RI_ENT_MAXCNT; d8[o++] = e | ((u64) x << 56); left -= x; }
			out += (u32) __builtin_amdgcn_readlane((int) inc2, 63);
		}
		if (lane == 0) { dcnt[cc] = out; atomicAdd(&tot, out); }
		vdjx_wave_lds_fence();
	}
	__syncthreads();
	if (threadIdx.x == 0 && tot) atomicAdd(n_entries, (unsigned long long) tot);
}

// the classes of more than RI_FOLD_WAVE members (the reads of the deepest clones: thousands of members, hundreds of distinct
// entries), a workgroup at a time; their number stays on the device (the workgroups take them in turns)
__global__ __launch_bounds__(RI_FOLD_BIG_THREADS) void k_ri_fold_big(const u32* __restrict__ giant, const u32* __restrict__ n_giant, const u32* __restrict__ start,
                                                                     const u32* __restrict__ cnt1, const u64* __restrict__ csr8, u64* __restrict__ d8,
                                                                     u32* __restrict__ dcnt, unsigned long long* __restrict__ n_entries) {
	__shared__ unsigned long long tk[RI_FOLD_BIG_SLOTS];
	__shared__ u32 tc[RI_FOLD_BIG_SLOTS];
	__shared__ u32 s_distinct, s_out;
	const u32 lane = threadIdx.x & 63u;
	for (u32 b = blockIdx.x; b < *n_giant; b += gridDim.x) {
	const u32 cc = giant[b], m = cnt1[cc], s = start[cc];
	__syncthreads();
	for (u32 i = threadIdx.x; i < RI_FOLD_BIG_SLOTS; i += RI_FOLD_BIG_THREADS) { tk[i] = NONE64; tc[i] = 0; }
	if (threadIdx.x == 0) { s_distinct = 0; s_out = 0; }
	__syncthreads();
	for (u32 at = 0; at < m; at += RI_FOLD_BIG_THREADS) {
		const bool full = s_distinct > RI_FOLD_BIG_SLOTS * 3 / 4;          // (the same for the whole workgroup: read between two barriers)
		__syncthreads();
		const bool have = at + threadIdx.x < m;
		const u64 e = have ? csr8[s + at + threadIdx.x] & ((1ull << 56) - 1ull) : 0ull;
		if (!full) {
			const bool fresh = have && ri_fold_insert(tk, tc, RI_FOLD_BIG_SLOTS, e);
			const u64 fm = __ballot(fresh);
			if (lane == 0 && fm) atomicAdd(&s_distinct, (u32) __popcll(fm));
		} else {                                                            // the table is full enough: this row leaves unfolded
			const u64 dm = __ballot(have);
			u32 base = 0;
			if (lane == 0 && dm) base = atomicAdd(&s_out, (u32) __popcll(dm));
			base = (u32) __builtin_amdgcn_readlane((int) base, 0);
			if (have) d8[s + base + (u32) __popcll(dm & ((1ull << lane) - 1ull))] = e | (1ull << 56);
		}
		__syncthreads();
	}
	for (u32 i = 0; i < RI_FOLD_BIG_SLOTS; i += RI_FOLD_BIG_THREADS) {
		u32 left = tc[i + threadIdx.x];
		const u32 pieces = (left + RI_ENT_MAXCNT - 1) / RI_ENT_MAXCNT;
		const u32 incl = (u32) vdjx_wave_scan_add((int) pieces);
		u32 base = 0;
		if (lane == 63 && incl) base = atomicAdd(&s_out, incl);
		base = (u32) __builtin_amdgcn_readlane((int) base, 63);
		u32 o = s + base + incl - pieces;
		const u64 e = (u64) tk[i + threadIdx.x];
		while (left) { const u32 q = left < RI_ENT_MAXCNT ? left : RI_ENT_MAXCNT; d8[o++] = e | ((u64) q << 56); left -= q; }
	}
	__syncthreads();
	if (threadIdx.x == 0) { dcnt[cc] = s_out; atomicAdd(n_entries, (unsigned long long) s_out); }
	}
}

// the lookup table the mapper reads: one slot per class = the sequence (W words), then {class + 1 | read-1 members << 32}, {CSR start |
// weighted entries << 32}: W + 2 words (32-byte slots for reads of up to 64 bases, 64-byte ones beyond: a probe is one line either way)
template <int W>
__global__ void k_ri_tab(const u32* __restrict__ rep, u32 ncls, const u64* __restrict__ bases, const u32* __restrict__ cnt1, const u32* __restrict__ start,
                         const u32* __restrict__ dcnt, u64* __restrict__ tab, u32 mask) {
	constexpr int SW = VDJX_RI_SLOT_WORDS(W);
	const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= ncls) return;
	if (rep[c] == NONE32) return;                          // (couples: the odd class of a read that is its own reverse complement)
	u64 b[W];
#pragma unroll
	for (int i = 0; i < W; i++) b[i] = bases[(size_t) rep[c] * W + i];
	u32 slot = (u32) (ri_hash<W>(b) >> 17) & mask;
	while (atomicCAS((u32*) &tab[(size_t) slot * SW + W], 0u, c + 1) != 0u) slot = (slot + 1) & mask;
	u64* sl = tab + (size_t) slot * SW;
#pragma unroll
	for (int i = 0; i < W; i++) sl[i] = b[i];
	((u32*) &sl[W])[1] = cnt1[c];                         // (the low half was claimed by the CAS)
	sl[W + 1] = (u64) start[c] | ((u64) dcnt[c] << 32);   // (the weighted entries of a class lie where its CSR members do: k_ri_fold)
}

// couples: ONE 64-byte slot per pair {sequence, reverse complement}, under the smaller of the two (what k_ri_insert_sym keyed the
// build's own table by): {key hi, key lo, key number + 1, members of class 2c | of class 2c + 1 << 32, CSR start | weighted entries << 32
// of class 2c, the same of class 2c + 1}.  Half the insertions of k_ri_tab into lines of their own; k_map_classify looks a string up
// under the smaller of itself and its reverse complement and takes its side of the slot.
// The claim word of a slot is build number << 25 | key number + 1: what an earlier build of the context left in the table reads as
// empty, so the table is cleared once per 127 builds (and when it is new), not per build.
__global__ void k_ri_tab_canon(const u32* __restrict__ rep, u32 ncanon, const u64* __restrict__ bases, const u32* __restrict__ cnt1, const u32* __restrict__ start,
                               const u32* __restrict__ dcnt, u64* __restrict__ tab, u32 mask, u32 epoch) {
	const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= ncanon) return;
	const ulonglong2 b = ((const ulonglong2*) bases)[rep[2 * (size_t) c]];
	u64 key[2] = {b.x, b.y};
	u32 slot = (u32) (ri_hash<2>(key) >> 17) & mask;
	for (;;) {
		u32* w = (u32*) &tab[(size_t) slot * 8 + 2];
		const u32 old = vdjx_peek(w);
		if ((old >> RI_TAB_EPOCH_SHIFT) == epoch) { slot = (slot + 1) & mask; continue; }       // taken in this build
		if (atomicCAS(w, old, (epoch << RI_TAB_EPOCH_SHIFT) | (c + 1)) == old) break;           // (lost the race for it: look again)
	}
	u64* sl = tab + (size_t) slot * 8;
	const uint2 m = *(const uint2*) &cnt1[2 * (size_t) c], st = *(const uint2*) &start[2 * (size_t) c], dc = *(const uint2*) &dcnt[2 * (size_t) c];
	((ulonglong2*) sl)[0] = b;
	((ulonglong2*) sl)[2] = make_ulonglong2((u64) st.x | ((u64) dc.x << 32), (u64) st.y | ((u64) dc.y << 32));
	sl[3] = (u64) m.x | ((u64) m.y << 32);                // (word 2's low half was claimed by the CAS)
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
// `st`: the context's stream (the waiting calls) or its index stream (a begun build: then nothing is bracketed for the profile --
// `pc` is null -- and the counts go to `stats`, handed to the context when the build is joined: the caller's thread owns c->stats)
int read_index_build_dev(vdjx_ctx* c, vdjx_work& db, const vdjx_pool* pool, const u32* d_pair, const uint8_t* d_rnum, const uint8_t* d_rc,
                         const u32* d_reg, u32 n_pairs, hipStream_t st, vdjx_ctx* pc, std::map<std::string, uint64_t>& stats) {
	if (pool->n_records > ((size_t) 1 << 29)) { vdjx_set_error("vdjx_read_index_build: %zu records on one GPU (limit 2^29): shard the pool by pair", pool->n_records); return VDJX_ELIMIT; }
	const u32 R = (u32) pool->n_records;
	const bool sym_on = getenv("VDJX_NO_SYM_INDEX") == nullptr;       // (read per build: the tests build one pool both ways)
	const bool sym = sym_on && pool->sym && pool->W == 2 && R % 2 == 0;      // couples: k_ri_insert_sym
	stats["read_index_sym"] = sym ? 1 : 0;
	const size_t RK = sym ? R / 2 : R;                     // what the build's table holds: records, or couples
	u32 mask = 1023;
	// (2 R slots.  1.5 R would do for the table itself and saves 0.3 ms of clearing and scanning at 10 M pairs -- but the classes are
	// numbered in slot order, the window mapper groups windows by a hash of their deepest classes' NUMBERS, and how evenly the deep
	// windows fall into groups decides k_group_pairs' longest workgroup: 1.17 ms with this numbering, 1.84 ms with that one, same work)
	while ((size_t) mask + 1 < RK * 2) mask = mask * 2 + 1;
	const size_t nslots = (size_t) mask + 1;
	const dim3 gR((u32) (RK / 256 + 1)), gB((R + RI_RB - 1) / RI_RB + 1), b256(256);
	u32 *d_rec_slot, *d_rec_cls, *d_err, *d_split;
	unsigned long long *d_r2key, *d_nent;
	u32* d_slots;                                 // the build's own table: slot -> a record of the class, then -> class + 1
	HIP_TRY(db.alloc(&d_slots, nslots));
	HIP_TRY(db.alloc(&d_rec_slot, RK + 1));
	HIP_TRY(db.alloc(&d_rec_cls, (size_t) R + 1));
	HIP_TRY(db.alloc(&d_err, 12));                     // errors [4] | split, giants | entries (u64) | larger classes
	d_split = d_err + 4;
	d_nent = (unsigned long long*) (d_err + 6);
	HIP_TRY(db.alloc(&d_r2key, (size_t) n_pairs * 2 + 2));
	HIP_TRY(hipMemsetAsync(d_slots, 0, nslots * 4, st));
	HIP_TRY(hipMemsetAsync(d_err, 0, 48, st));
	HIP_TRY(hipMemsetAsync(d_r2key, 0xFF, ((size_t) n_pairs * 2 + 2) * 8, st));
	{
		vdjx_prof_scope ps(pc, "k_ri_insert");
		if (sym) hipLaunchKernelGGL(k_ri_insert_sym, gR, b256, 0, st, pool->d_bases, pool->d_nmask, (u32) RK, pool->rl, d_slots, mask, d_rec_slot);
		else if (pool->W == 2) hipLaunchKernelGGL((k_ri_insert<2, 1>), gR, b256, 0, st, pool->d_bases, pool->d_nmask, R, d_slots, mask, d_rec_slot);
		else hipLaunchKernelGGL((k_ri_insert<VDJX_LONG_W, VDJX_LONG_M>), gR, b256, 0, st, pool->d_bases, pool->d_nmask, R, d_slots, mask, d_rec_slot);
	}
	// classes
	const u32 nsb = (u32) ((nslots + RS_BLOCK - 1) / RS_BLOCK);
	u32 *d_bcnt, *d_bpre;
	HIP_TRY(db.alloc(&d_bcnt, nsb + 1));
	HIP_TRY(db.alloc(&d_bpre, nsb + 1));
	{
		vdjx_prof_scope ps(pc, "k_ri_number");
		hipLaunchKernelGGL(k_ri_occ, dim3(nsb), b256, 0, st, d_slots, (u32) nslots, d_bcnt);
	}
	// (two levels: one workgroup over the 32,768 block counts of a 10 M-pair pool's table was 56 us)
	if (nsb > 4096) { const int rc_ = scan_u32(db, st, d_bcnt, nsb, d_bpre); if (rc_) return rc_; }
	else hipLaunchKernelGGL(k_rs_top, dim3(1), dim3(1024), 0, st, d_bcnt, nsb, d_bpre);
	u32 ncls = 0;
	HIP_TRY(hipMemcpyAsync(&ncls, d_bpre + nsb, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (sym) {
		if (ncls >= RI_ENT_NONE / 2) { vdjx_set_error("vdjx_read_index_build: %u distinct read sequences on one GPU (limit 2^26 - 1): shard the pool by pair", 2 * ncls); return VDJX_ELIMIT; }
		ncls *= 2;                                          // a key's sequence and its reverse complement
	}
	if (ncls >= RI_ENT_NONE) { vdjx_set_error("vdjx_read_index_build: %u distinct read sequences on one GPU (limit 2^26 - 1): shard the pool by pair", ncls); return VDJX_ELIMIT; }
	u32 tmask = 1023;
	while ((size_t) tmask + 1 < (size_t) (sym ? ncls / 2 : ncls) * 2) tmask = tmask * 2 + 1;
	u32* d_rep;
	HIP_TRY(db.alloc(&d_rep, (size_t) ncls + 1));
	const size_t slot_bytes = sym ? 64 : (size_t) (pool->W == 2 ? VDJX_RI_SLOT_WORDS(2) : VDJX_RI_SLOT_WORDS(VDJX_LONG_W)) * 8;
	// the index's arrays are kept from build to build and only replaced when one needs more (hipMalloc / hipFree of gigabytes per pool
	// cost more than the kernels that fill them)
	const void* tab_before = c->d_ri_tab;
	const size_t tab_cap_before = c->ri_cap[0];
	HIP_TRY(ri_keep(&c->d_ri_tab, &c->ri_cap[0], ((size_t) tmask + 1) * slot_bytes));
	if (c->d_ri_tab != tab_before || c->ri_cap[0] != tab_cap_before || !sym || c->ri_tab_epoch >= 127u) c->ri_tab_epoch = 0;        // a new buffer, another slot format, the numbers used up: clear
	HIP_TRY(ri_keep(&c->d_ri_start, &c->ri_cap[1], ((size_t) ncls + 2) * 4));
	HIP_TRY(ri_keep(&c->d_ri_cnt1, &c->ri_cap[2], ((size_t) ncls + 2) * 4));
	HIP_TRY(ri_keep(&c->d_ri_dstart, &c->ri_cap[3], ((size_t) ncls + 2) * 4));          // (weighted entries per class: k_ri_fold's dcnt)
	HIP_TRY(ri_keep(&c->d_pair_r2, &c->ri_cap[4], ((size_t) n_pairs * 2 + 2) * 4));
	HIP_TRY(hipMemsetAsync(c->d_ri_cnt1, 0, ((size_t) ncls + 2) * 4, st));
	if (c->ri_tab_epoch == 0) HIP_TRY(hipMemsetAsync(c->d_ri_tab, 0, c->ri_cap[0], st));          // (all of it: a later build may use more of the buffer)
	if (sym) c->ri_tab_epoch++;
	{
		vdjx_prof_scope ps(pc, "k_ri_number");
		if (sym) hipLaunchKernelGGL(k_ri_number_sym, dim3(nsb), b256, 0, st, d_slots, (u32) nslots, d_bpre, d_rep);
		else hipLaunchKernelGGL(k_ri_number, dim3(nsb), b256, 0, st, d_slots, (u32) nslots, d_bpre, d_rep);
	}
	// records: class, read-2 records of the pairs; the read-1 members in record order
	const u32 nrb = (R + RI_RB - 1) / RI_RB;
	u32 *d_rbcnt, *d_rbpre;
	HIP_TRY(db.alloc(&d_rbcnt, nrb + 1));
	HIP_TRY(db.alloc(&d_rbpre, nrb + 2));
	{
		vdjx_prof_scope ps(pc, "k_ri_records");
		if (nrb && sym) hipLaunchKernelGGL(k_ri_records<true>, dim3(nrb), b256, 0, st, d_rec_slot, d_slots, R, d_pair, d_rnum, d_rc, d_reg, n_pairs, d_rec_cls, d_r2key, d_err, d_rbcnt);
		else if (nrb) hipLaunchKernelGGL(k_ri_records<false>, dim3(nrb), b256, 0, st, d_rec_slot, d_slots, R, d_pair, d_rnum, d_rc, d_reg, n_pairs, d_rec_cls, d_r2key, d_err, d_rbcnt);
		hipLaunchKernelGGL(k_rs_top, dim3(1), dim3(1024), 0, st, d_rbcnt, nrb, d_rbpre);
		hipLaunchKernelGGL(k_ri_split, dim3(1), b256, 0, st, d_err, d_rec_cls, d_rnum, d_pair, R, n_pairs, d_rbpre, d_split);
	}
	u32 h_err[4] = {0, 0, 0, 0}, n1 = 0, n1p = 0;
	HIP_TRY(hipMemcpyAsync(&n1p, d_split, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&n1, d_rbpre + nrb, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (h_err[RI_ERR_PAIR]) { vdjx_set_error("vdjx_read_index_build: %u records with a pair id >= n_pairs=%u", h_err[RI_ERR_PAIR], n_pairs); return VDJX_EINVAL; }
	if (h_err[RI_ERR_R2]) { vdjx_set_error("a pair has more than two read-2 records (read names must be unique per pair)"); return VDJX_EINVAL; }
	u64 *d_mkey, *d_mkey2;
	ulonglong2* d_mpack;
	u32 *d_mreg, *d_mreg2;
	HIP_TRY(db.alloc(&d_mkey, (size_t) n1 + 1));
	HIP_TRY(db.alloc(&d_mkey2, (size_t) n1 + 1));
	HIP_TRY(db.alloc(&d_mpack, (size_t) n1 + 1));
	HIP_TRY(db.alloc(&d_mreg, (size_t) n1 + 1));
	HIP_TRY(db.alloc(&d_mreg2, (size_t) n1 + 1));
	{
		vdjx_prof_scope ps(pc, "k_ri_members");
		if (nrb) hipLaunchKernelGGL(k_ri_members, dim3(nrb), b256, 0, st, d_rec_cls, d_rnum, d_reg, d_pair, d_r2key, c->d_pair_r2, R, n_pairs, d_rbpre, c->d_ri_cnt1, d_mkey, d_mreg, d_mpack);
	}
	int rc = scan_u32(db, st, c->d_ri_cnt1, ncls + 1, c->d_ri_start);         // (cnt1[ncls] = 0: start[ncls] = start[ncls + 1] = members)
	if (rc) return rc;
	// CSR order = (class, registration rank)
	HIP_TRY(ri_keep(&c->d_ri_recs, &c->ri_cap[5], ((size_t) n1 + 1) * 4));
	HIP_TRY(ri_keep(&c->d_ri_csr8, &c->ri_cap[6], ((size_t) n1 + 1) * 8));
	HIP_TRY(ri_keep(&c->d_ri_csr_pair, &c->ri_cap[7], ((size_t) n1 + 1) * 4));
	HIP_TRY(ri_keep(&c->d_ri_d8, &c->ri_cap[8], ((size_t) n1 + 1) * 8));
	unsigned long long nd = 0;
	u64* d_by_class = nullptr;
	if (n1) {
		vdjx_prof_scope ps(pc, "ri_sort_members");
		// the members are in record order; inside a pool that IS registration order (add_to_buffer registers what it appends,
		// bam_read.c:206-244), so by rank they are two sorted runs (primary, secondary): a merge.  A caller whose ranks do not
		// follow its records gets a sort by rank instead.  Then a STABLE sort by class alone (the key's upper half; the member
		// rides in the lower one): three 8-bit passes over 8-byte keys instead of seven over key + value.
		u64* by_rank = d_mkey;
		if (h_err[RI_ERR_ORDER] > 1) {
			size_t tb = 0;
			const unsigned rb = 32;
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_mreg, d_mreg2, d_mkey, d_mkey2, (size_t) n1, 0u, rb, st));
			char* tmp;
			HIP_TRY(db.alloc(&tmp, tb + 256));
			HIP_TRY(rocprim::radix_sort_pairs((void*) tmp, tb, d_mreg, d_mreg2, d_mkey, d_mkey2, (size_t) n1, 0u, rb, st));
			by_rank = d_mkey2;
		} else if (h_err[RI_ERR_ORDER] == 1) {
			if (n1p > n1) n1p = n1;
			size_t tb = 0;
			HIP_TRY(rocprim::merge(nullptr, tb, d_mreg, d_mreg + n1p, d_mreg2, d_mkey, d_mkey + n1p, d_mkey2, (size_t) n1p, (size_t) (n1 - n1p), rocprim::less<u32>(), st));
			char* tmp;
			HIP_TRY(db.alloc(&tmp, tb + 256));
			HIP_TRY(rocprim::merge((void*) tmp, tb, d_mreg, d_mreg + n1p, d_mreg2, d_mkey, d_mkey + n1p, d_mkey2, (size_t) n1p, (size_t) (n1 - n1p), rocprim::less<u32>(), st));
			by_rank = d_mkey2;
		}
		u64* by_class = by_rank == d_mkey ? d_mkey2 : d_mkey;
		size_t tb = 0;
		HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, by_rank, by_class, (size_t) n1, 32u, 32u + bits_for(ncls), st));
		char* tmp;
		HIP_TRY(db.alloc(&tmp, tb + 256));
		HIP_TRY(rocprim::radix_sort_keys((void*) tmp, tb, by_rank, by_class, (size_t) n1, 32u, 32u + bits_for(ncls), st));
		hipLaunchKernelGGL(k_ri_csr, dim3(n1 / 256 + 1), b256, 0, st, by_class, n1, d_mpack, c->d_ri_recs, c->d_ri_csr_pair, c->d_ri_csr8);
		d_by_class = by_class;
	}
	{
		vdjx_prof_scope ps(pc, "k_ri_fold");
		const u32 max_giant = n1 / (RI_FOLD_WAVE + 1) + 1;
		u32* d_giant;
		HIP_TRY(db.alloc(&d_giant, (size_t) max_giant + 1));
		HIP_TRY(hipMemsetAsync(c->d_ri_dstart, 0, ((size_t) ncls + 2) * 4, st));          // (a class without read-1 members has no entries)
		if (n1) hipLaunchKernelGGL(k_ri_fold_members, dim3(n1 / RI_FM_PER + 1), b256, 0, st, (const u64*) d_by_class, n1, c->d_ri_start, c->d_ri_cnt1, c->d_ri_csr8, c->d_ri_d8, c->d_ri_dstart, d_nent,
		                           d_giant, d_split + 1);
		hipLaunchKernelGGL(k_ri_fold_big, dim3(max_giant < 2048u ? max_giant : 2048u), dim3(RI_FOLD_BIG_THREADS), 0, st, d_giant, d_split + 1, c->d_ri_start, c->d_ri_cnt1, c->d_ri_csr8, c->d_ri_d8, c->d_ri_dstart, d_nent);
	}
	HIP_TRY(hipMemcpyAsync(&nd, d_nent, 8, hipMemcpyDeviceToHost, st));
	{	// the mapper's table, now that the classes' sizes and starts are known
		vdjx_prof_scope ps(pc, "k_ri_tab");
		if (sym) hipLaunchKernelGGL(k_ri_tab_canon, dim3(ncls / 512 + 1), b256, 0, st, d_rep, ncls / 2, pool->d_bases, c->d_ri_cnt1, c->d_ri_start, c->d_ri_dstart, (u64*) c->d_ri_tab, tmask, c->ri_tab_epoch);
		else if (pool->W == 2) hipLaunchKernelGGL(k_ri_tab<2>, dim3(ncls / 256 + 1), b256, 0, st, d_rep, ncls, pool->d_bases, c->d_ri_cnt1, c->d_ri_start, c->d_ri_dstart, (u64*) c->d_ri_tab, tmask);
		else hipLaunchKernelGGL(k_ri_tab<VDJX_LONG_W>, dim3(ncls / 256 + 1), b256, 0, st, d_rep, ncls, pool->d_bases, c->d_ri_cnt1, c->d_ri_start, c->d_ri_dstart, (u64*) c->d_ri_tab, tmask);
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipGetLastError());
	if (pc) vdjx_prof_collect(pc, false);
	stats["read_index_r1_members"] = n1;
	stats["read_index_r1_distinct"] = nd;
	stats["read_index_classes"] = ncls;
	stats["read_index_rank_order"] = h_err[RI_ERR_ORDER] > 1 ? 2 : h_err[RI_ERR_ORDER];      // 0 as recorded, 1 merge of the two pools' runs, 2 sort by rank
	c->ri_tab_mask = tmask;
	c->ri_canon = sym;
	c->n_pairs = n_pairs;
	c->n_classes = ncls;
	c->ri_pool = pool;
	return VDJX_OK;
}

// the index is gone (its arrays stay for the next build: ri_keep)
void drop_index(vdjx_ctx* c) {
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

// ---- a build begun and ended (the index beside the k-mer build) -----------------------------------------------------------------
struct vdjx_ri_job {
	std::thread th;
	int rc = VDJX_OK;
	std::string err;
	std::map<std::string, uint64_t> stats;
	// the gate: the thread launches nothing before it opens.  Measured (profiles/overlap.py, 10 M pairs): beside phase A of the k-mer build
	// -- the streaming histogram / partition kernels, which fill the machine on their own -- both sides only stretch (k_part_records 2.5x,
	// k_gated_hist 2.5x, k_ri_fold_big 6x: 12.3 ms of work done in 11.4); the graph pass is where the device waits on dependent probes
	// with most of its bandwidth idle.  So the k-mer build opens the gate when its phase A is done (vdjx_ri_open_gate), and so does
	// whoever asks for the index first
	std::mutex mu;
	std::condition_variable cv;
	bool go = false;
};

void vdjx_ri_open_gate(vdjx_ctx* c) {
	vdjx_ri_job* j = c->ri_job;
	if (!j) return;
	{
		std::lock_guard<std::mutex> lk(j->mu);
		if (j->go) return;
		j->go = true;
	}
	j->cv.notify_all();
}

int vdjx_ri_join(vdjx_ctx* c) {
	vdjx_ri_job* j = c->ri_job;
	if (!j) return VDJX_OK;
	vdjx_ri_open_gate(c);
	if (j->th.joinable()) j->th.join();
	c->ri_job = nullptr;
	for (auto& kv : j->stats) c->stats[kv.first] = kv.second;
	const int rc = j->rc;
	if (rc) { vdjx_set_error("%s", j->err.c_str()); drop_index(c); }
	delete j;
	return rc;
}

namespace {
// host = true: the four arrays are the caller's host arrays (they go up on the index stream; they must stay valid until _end)
int ri_begin(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* pair_id, const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank,
             uint32_t n_pairs, bool host, const char* who) {
	int rc = check_args(c, pool, pair_id, read_num, is_rc, reg_rank, who);
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	rc = vdjx_ri_join(c);                               // (one build in flight per context; an unjoined one ends here)
	if (rc) return rc;
	drop_index(c);
	// what the context's stream holds so far -- the packing of this pool, the last scorer kernels that read the index being replaced --
	// comes first; nothing else orders the two streams until _end
	HIP_TRY(hipEventRecord(c->ev_ri_go, c->stream));
	HIP_TRY(hipStreamWaitEvent(c->ri_stream, c->ev_ri_go, 0));
	vdjx_ri_job* j = new vdjx_ri_job();
	c->ri_job = j;
	const int device = c->device;
	static const bool gated = !(getenv("VDJX_RI_GATE") && getenv("VDJX_RI_GATE")[0] == '0');      // (VDJX_RI_GATE=0: start at once, beside whatever comes)
	j->go = !gated;
	j->th = std::thread([=]() {
		{
			std::unique_lock<std::mutex> lk(j->mu);
			j->cv.wait(lk, [&] { return j->go; });
		}
		if (hipSetDevice(device) != hipSuccess) { j->rc = VDJX_EHIP; j->err = "vdjx_read_index_build_begin: hipSetDevice failed on the index thread"; return; }
		vdjx_work db(c, &c->ri_arena);
		hipStream_t st = c->ri_stream;
		const uint32_t *d_pair = pair_id, *d_reg = reg_rank;
		const uint8_t *d_rnum = read_num, *d_rc = is_rc;
		int r = VDJX_OK;
		if (host) {
			const size_t R = pool->n_records;
			u32 *dp = nullptr, *dr = nullptr;
			uint8_t *dn = nullptr, *dc = nullptr;
			if (db.alloc(&dp, R + 1) != hipSuccess || db.alloc(&dr, R + 1) != hipSuccess || db.alloc(&dn, R + 1) != hipSuccess || db.alloc(&dc, R + 1) != hipSuccess) r = VDJX_EHIP;
			if (!r && R) {
				hipError_t e = hipMemcpyAsync(dp, pair_id, R * 4, hipMemcpyHostToDevice, st);
				if (e == hipSuccess) e = hipMemcpyAsync(dr, reg_rank, R * 4, hipMemcpyHostToDevice, st);
				if (e == hipSuccess) e = hipMemcpyAsync(dn, read_num, R, hipMemcpyHostToDevice, st);
				if (e == hipSuccess) e = hipMemcpyAsync(dc, is_rc, R, hipMemcpyHostToDevice, st);
				if (e != hipSuccess) { vdjx_set_error("vdjx_read_index_build_begin: upload: %s", hipGetErrorString(e)); r = VDJX_EHIP; }
			}
			d_pair = dp; d_reg = dr; d_rnum = dn; d_rc = dc;
		}
		if (!r) r = read_index_build_dev(c, db, pool, d_pair, d_rnum, d_rc, d_reg, n_pairs, st, nullptr, j->stats);
		if (r) { (void) hipStreamSynchronize(st); j->err = vdjx_last_error(); }      // (nothing of a failed build may still be running when its workspace is reset)
		j->rc = r;
	});
	return VDJX_OK;
}
}  // namespace

extern "C" int vdjx_read_index_build_device_begin(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* d_pair_id, const uint8_t* d_read_num,
                                                  const uint8_t* d_is_rc, const uint32_t* d_reg_rank, uint32_t n_pairs) {
	return ri_begin(c, pool, d_pair_id, d_read_num, d_is_rc, d_reg_rank, n_pairs, false, "vdjx_read_index_build_device_begin");
}

extern "C" int vdjx_read_index_build_begin(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* pair_id, const uint8_t* read_num,
                                           const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs) {
	return ri_begin(c, pool, pair_id, read_num, is_rc, reg_rank, n_pairs, true, "vdjx_read_index_build_begin");
}

extern "C" int vdjx_read_index_build_end(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_read_index_build_end: ctx is NULL"); return VDJX_EINVAL; }
	return vdjx_ri_join(c);
}

extern "C" int vdjx_read_index_build_device(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* d_pair_id, const uint8_t* d_read_num,
                                            const uint8_t* d_is_rc, const uint32_t* d_reg_rank, uint32_t n_pairs) {
	int rc = check_args(c, pool, d_pair_id, d_read_num, d_is_rc, d_reg_rank, "vdjx_read_index_build_device");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	(void) vdjx_ri_join(c);
	HIP_TRY(hipStreamSynchronize(c->stream));           // nothing in flight may still read the index that is replaced
	drop_index(c);
	vdjx_work db(c);
	rc = read_index_build_dev(c, db, pool, d_pair_id, d_read_num, d_is_rc, d_reg_rank, n_pairs, c->stream, c, c->stats);
	if (rc) drop_index(c);
	return rc;
}

extern "C" int vdjx_read_index_build(vdjx_ctx* c, const vdjx_pool* pool, const uint32_t* pair_id,
                                     const uint8_t* read_num, const uint8_t* is_rc, const uint32_t* reg_rank, uint32_t n_pairs) {
	int rc = check_args(c, pool, pair_id, read_num, is_rc, reg_rank, "vdjx_read_index_build");
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	(void) vdjx_ri_join(c);
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
	rc = read_index_build_dev(c, db, pool, d_pair, d_rnum, d_rc, d_reg, n_pairs, c->stream, c, c->stats);
	if (rc) drop_index(c);
	return rc;
}
