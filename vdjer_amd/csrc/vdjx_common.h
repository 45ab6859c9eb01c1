// vdjx_common.h -- shared host/device definitions of libvdjx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <chrono>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/vdjx.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

#define VDJX_WAVE 64

// flags of a read-index entry (csr_info / dinfo .w, vdjx_rindex.hip)
#define RI_R1 1u        // the record is a read-1 instance
#define RI_RC 2u        // its is_rc flag
#define RI_RCA 4u       // is_rc of the pair's read-2 record A (registered first)
#define RI_RCB 8u       // is_rc of the pair's read-2 record B (registered last)
// words per slot of the read index's lookup table: the read (W words) + three words of class data
#define VDJX_RI_SLOT_WORDS(W) ((W) == 2 ? 4 : 8)     // the read (W words) + {class + 1 | members << 32} + {CSR start | weighted entries << 32}: 32 bytes
                                                     // for reads of up to 64 bases (two slots per 64-byte line, a 1 GB table at 10 M pairs), 64 for longer ones
// an 8-byte entry: class of the pair's read-2 record A (26 bits, all ones = none) | class of B (26) | flags (4) | multiplicity (8)
#define RI_TAB_EPOCH_SHIFT 25        // k_ri_tab_canon's claim word: build number << 25 | key number + 1 (keys: half of the 2^26 - 1 classes)
#define RI_ENT_NONE 0x3FFFFFFu
#define RI_ENT_MAXCNT 255u
__host__ __device__ inline unsigned long long ri_entry(unsigned y, unsigned z, unsigned fl, unsigned cnt) {
	return (unsigned long long) y | ((unsigned long long) z << 26) | ((unsigned long long) fl << 52) | ((unsigned long long) cnt << 56);
}

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
void vdjx_set_error(const char* fmt, ...);
// drop a stale error left in the runtime's per-thread slot by an earlier, unrelated call
inline void vdjx_clear_errors() { (void) hipGetLastError(); }

#define HIP_TRY(expr)                                                                            \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			vdjx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return VDJX_EHIP;                                                                    \
		}                                                                                        \
	} while (0)

// ----------------------------------------------------------------------------------------------
// host-side objects behind the opaque handles
// ----------------------------------------------------------------------------------------------
struct vdjx_prof_entry {
	double ms = 0;
	uint64_t launches = 0;
};

// Grow-only device workspace: one bump allocator per context, reset at the end of every API call.
// hipMalloc/hipFree cost milliseconds each and hipFree synchronises the device; the hot path allocates ~25
// buffers per call, so they come out of this arena instead.
struct vdjx_arena {
	// ONE growing range of device addresses (hipMemAddressReserve), backed piece by piece as the calls need it (hipMemCreate +
	// hipMemMap at the end of what is mapped): allocation is a bump of `used`, a call that needs more than any before adds pieces
	// behind the others, nothing is ever moved, traded or re-sized, and what a context holds is the PEAK of its calls -- not the sum
	// of chunk lists that different call sequences left behind.  (Round 4: the arena used to be a list of hipMalloc'ed chunks traded
	// for one chunk of their total after every call; that one large hipMalloc -- 38 GB -- took 0.3 ms on some runs and 1.0-2.4 s
	// on others: the "first step" of bench.py at k = 25.  Keeping the chunks instead grew 8 contexts on one device out of memory
	// over three chains of the configs[4] rehearsal.)  Where the runtime has no virtual memory management the chunk list remains.
	char* base = nullptr;            // the reserved range (nullptr: chunk mode)
	size_t reserved = 0, mapped = 0; // bytes of the range / backed so far
	std::vector<std::pair<void*, size_t>> pieces;     // allocation handles (hipMemGenericAllocationHandle_t) and sizes, in address order
	size_t gran = 0;
	int mode = 0;                    // 0 undecided, 1 one range, 2 chunks
	struct chunk { char* p; size_t cap; };
	std::vector<chunk> chunks;
	size_t cur = 0;                  // chunk mode: the chunk allocations come out of
	size_t used = 0;                 // bytes used (of the range; of chunk `cur`)
	void* alloc(size_t bytes);       // 256-byte aligned; nullptr (and the error set) on failure
	void reset();                    // forget all allocations (the memory stays)
	void release(bool for_good = false);     // give the memory back (for_good: the address range too)
	// stack discipline inside one call sequence: everything allocated after mark() is given up by release_to(mark) and its space is
	// taken again by the next allocations (a build's early temporaries do not add to its later phases' footprint).  Nothing on the
	// stream may still use what is released.
	struct mark_t { size_t cur, used; };
	mark_t mark() const { return {cur, used}; }
	void release_to(mark_t m);
	bool grow(size_t upto);
};

struct vdjx_shard;

// vdjx_ctx::h_pin: 16 KB of page-locked scratch.  Every synchronous call reads its small numbers back into the first bytes (the
// largest: stage_gated_reduce's (64 * 16 + 2) * 8 = 8,208); a call that is BEGUN and ended later (vdjx_root_score_graph_begin) keeps
// its number in a word behind all of those, because other calls run in between (a k-mer build between begin and end
// overwrote a word at byte 8,192)
#define VDJX_HPIN_BYTES 16384
#define VDJX_HPIN_SYNC_BYTES ((64 * 16 + 2) * 8)       // the largest read-back of a synchronous call (vdjx_kmer.hip stage_gated_reduce)
#define VDJX_HPIN_ROOT_RUN 3072                        // u32 index: byte 12,288
static_assert(VDJX_HPIN_ROOT_RUN * 4 >= VDJX_HPIN_SYNC_BYTES + 64 && VDJX_HPIN_ROOT_RUN * 4 + 4 <= VDJX_HPIN_BYTES, "h_pin layout");

// device blocks that outlive a call (packed pools, exported graphs): freed blocks are kept for the next call of the
// same size class instead of going back to hipFree (which synchronises the device)
struct vdjx_block_cache {
	struct blk { char* p; size_t cap; };
	std::vector<blk> free_list;
	hipError_t acquire(size_t need, char** out, size_t* cap);
	void release(char* p, size_t cap);
	void drop();
};

struct vdjx_ctx {
	int device = 0;
	vdjx_arena arena;
	vdjx_block_cache blocks;
	vdjx_arena shard_arena;            // lives across the phases of one sharded build
	// vdjx_read_index_build[_device]_begin .. _end: the index is built by a thread of its own, on a stream of its own, out of a workspace of
	// its own, beside the k-mer build of the same pool (both only READ the packed records; the reference registers the reads during
	// extraction, quick_map3.c:126-149, and builds its k-mer table afterwards, A2:1388: nothing orders the two but the first scorer call)
	hipStream_t ri_stream = nullptr;
	hipEvent_t ev_ri_go = nullptr;
	vdjx_arena ri_arena;
	struct vdjx_ri_job* ri_job = nullptr;     // the build in flight (vdjx_rindex.hip); every call that looks at the index joins it first
	vdjx_shard* live_shard = nullptr;
	hipStream_t stream = nullptr;
	hipStream_t copy_stream = nullptr;   // result copies that may run beside the next kernels (vdjx_graph_export_begin)
	hipStream_t pairs_stream = nullptr;  // the mapped pairs' copy (vdjx_map_emit_begin): a stream of its own, so that waiting for one result is not waiting for the other
	void* h_pin = nullptr;                          // 16 KB of page-locked memory: where the small numbers a call waits for come down (a copy into
	                                                // pageable memory goes through a staging buffer of the runtime: tens of microseconds per read)
	void* d_stage[2] = {nullptr, nullptr};          // vdjx_pool_load: upload staging (two chunks in flight)
	size_t stage_cap = 0;
	hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_packed[2] = {nullptr, nullptr};
	bool profiling = false;
	std::string prof_only;            // vdjx_profile_only: the one scope name that is bracketed (empty: all)
	std::vector<std::string> prof_names;                 // insertion order
	std::map<std::string, vdjx_prof_entry> prof;
	struct pending_ev { std::string name; hipEvent_t a, b; };
	std::vector<pending_ev> prof_pending;
	std::vector<hipEvent_t> ev_free;                     // events are recycled: creating two per launch costs as much as recording them
	// a-6 anchor bitmaps (2^32 bits each)
	u32* d_vbits = nullptr;
	u32* d_jbits = nullptr;
	bool anchors_loaded = false;
	u32 sub_tuples_set = 0;           // the value of the device's g_sub_tuples this context has set (vdjx_kmer.hip: once, not per build)
	u32* d_anchor_tmp = nullptr;      // the codes of a set on their way into its bitmap (kept: a new chain's ref-dir comes with every pool at configs[4])
	size_t anchor_tmp_cap = 0, vtext_cap = 0, line_off_cap = 0, seed_cap = 0;      // bytes behind d_anchor_tmp / d_vtext / d_line_off / d_seed_code and d_seed_pos
	// a-7 V-region index
	char* d_vtext = nullptr;          // all lines concatenated
	u32* d_line_off = nullptr;        // [n_lines+1]
	u32* d_seed_code = nullptr;       // sorted vk-mer codes
	u32* d_seed_pos = nullptr;        // position within its line
	size_t n_lines = 0, n_seeds = 0;
	int vk = 0;
	std::vector<u32> h_line_off;
	// a-8 read index
	const vdjx_pool* ri_pool = nullptr;
	void* d_ri_tab = nullptr;         // slots {read sequence, class + 1 | members, CSR start | weighted entries} (k_ri_tab; VDJX_RI_SLOT_WORDS)
	u32 ri_tab_mask = 0;
	u32 ri_tab_epoch = 0;             // k_ri_tab_canon: a slot is taken iff the top 7 bits of its claim word hold the build's number (1 .. 127; 0: the table
	                                  // has to be cleared first) -- clearing a gigabyte per build was 0.15 ms of 4.8
	bool ri_canon = false;            // the table holds one 64-byte slot per pair {sequence, reverse complement} under the smaller of the two (k_ri_tab_canon: pools of couples)
	u32* d_ri_start = nullptr;        // class -> CSR start [ncls+1]
	u32* d_ri_cnt1 = nullptr;         // class -> number of read-1 members (the CSR lists those only)
	u32* d_ri_recs = nullptr;         // CSR: read-1 records in registration order
	u64* d_ri_csr8 = nullptr;         // CSR: the member's 8-byte entry (ri_entry)
	u32* d_ri_csr_pair = nullptr;     // CSR: the member's pair id
	u32* d_pair_r2 = nullptr;         // pair -> its two read-2 records in registration order (or ~0u)
	u32* d_ri_dstart = nullptr;       // class -> first of its DISTINCT read-1 entries (window scoring counts, it does not name pairs)
	u64* d_ri_d8 = nullptr;           // those entries, with multiplicities
	size_t ri_cap[9] = {};            // bytes behind the nine arrays above (kept from build to build, vdjx_rindex.hip ri_keep)
	std::vector<std::pair<const char*, size_t>> host_blocks;     // vdjx_host_alloc's page-locked blocks (the scorers read strings that lie in one of them in place)
	std::mutex host_blocks_mu;
	// cached result of the last vdjx_map_emit count call (the write call of the two-call protocol reuses it)
	uint64_t me_key = 0;
	const void* me_src = nullptr;     // the batch the cached mapping belongs to
	void* me_pairs = nullptr;         // vdjx_pair[me_cap], per-contig regions at the contigs' hit offsets
	void* me_hit = nullptr;           // u32[me_cap]: the hit (inside its slice) every stored pair belongs to
	size_t me_cap = 0;
	// the (weighted) mapped-pair lists of the last window batch: (multiplicity << 32 | pos1 << 16 | pos2), window i at wp_off[i]
	void* wp_buf = nullptr;
	size_t wp_cap = 0, wp_n = 0;
	std::vector<u64> wp_off;
	std::vector<u32> wp_cnt;
	void* me_dense = nullptr;         // the pairs laid end to end for the copy to the host (kept: the copy may be asynchronous)
	size_t me_dense_cap = 0;
	size_t me_gathered_cap = 0;       // capacity the counting call's gather (launched ahead of the total) ran against; 0: it did not run
	std::vector<u64> me_cnt;          // pairs per contig
	void* h_plan = nullptr;           // page-locked scratch: the plan's totals come down here (vdjx_score.hip classify_and_plan)
	size_t h_plan_cap = 0;
	void* h_res = nullptr;            // page-locked, grows: small per-call results come down here in one go (a copy into the caller's pageable
	size_t h_res_cap = 0;             // arrays blocks the host per copy) and are handed over after the call's one wait
	void* me_book = nullptr;          // device bookkeeping between the counting and the writing call (vdjx_score.hip map_emit_impl)
	size_t me_book_cap = 0, me_nsl = 0;
	u32 me_slice_hits = 0;
	hipEvent_t ev_gathered = nullptr;
	hipStream_t up_stream = nullptr;        // the scorers' strings on their way up, in pieces (classify_and_plan)
	hipEvent_t ev_up[5] = {};
	hipEvent_t ev_pairs_copied = nullptr;   // the last asynchronous copy of mapped pairs to the host has left me_dense
	uint32_t root_dp_hint = 0;        // work items of the last root scoring (+ a margin): the next call's DP is launched for that many ahead of its own count
	hipEvent_t ev_plan = nullptr;     // the scorers' plan totals have come down (the host waits for this, not for the stream)
	// vdjx_root_score_graph_begin .. _end: the call in flight
	bool root_pending = false;
	const vdjx_graph* root_pending_g = nullptr;
	int root_pending_thr = 0;
	uint32_t root_pending_first = 0, root_pending_stride = 1, root_pending_ahead = 0;
	uint32_t* root_pending_ids = nullptr;
	uint8_t* root_pending_out = nullptr;
	hipEvent_t ev_root_done = nullptr;
	// round 6: a begun root scoring runs on a stream and out of a workspace of its own -- its kernels are small (tens of thousands of roots,
	// a wave per seed hit) and wait on dependent loads; the window scorer's kernels, which do not need the verdicts, run beside them
	hipStream_t root_stream = nullptr;
	hipEvent_t ev_root_go = nullptr;
	vdjx_arena root_arena;
	// SAM text (vdjx_sam_text): read names by pair id on the device, the text buffers
	char* d_sam_names = nullptr;
	u64* d_sam_noff = nullptr;
	u32 sam_pairs = 0;
	void* d_sam_text = nullptr;
	void* h_sam_text = nullptr;
	size_t sam_text_cap = 0;
	void *d_sam_keys = nullptr, *d_sam_lens = nullptr;      // vdjx_sam_blocks: per mapped pair its ordering key and the bytes of its two lines
	size_t sam_blk_cap = 0;
	void* h_sam_merge = nullptr;      // vdjx_sam_merge: the merged text (page-locked)
	size_t sam_merge_cap = 0;
	u32 n_pairs = 0, n_classes = 0;
	std::map<std::string, uint64_t> stats;
};

// Two record formats.  Reads of up to 64 bases (W = 2, M = 1, ob = 6): the read as ONE 2*rl-bit integer in two words (hi, lo), first base
// most significant, right-aligned -- a k-mer is one 128-bit shift + mask.  Longer reads (up to VDJX_MAX_READ_LEN; W = 5, M = 3,
// ob = 8): W words, LEFT-aligned, base i in bits 63-2(i%32), 62-2(i%32) of word i/32; masks of M words, bit i%64 of word i/64.
// Instance ids are record << ob | offset.
#define VDJX_LONG_W 5
#define VDJX_LONG_M 3
struct vdjx_pool {
	vdjx_ctx* ctx = nullptr;
	int device = 0;              // copy: a pool may be freed after its context
	size_t n_primary = 0, n_records = 0;
	int rl = 0;
	int qstride = 0;
	int W = 2, M = 1, ob = 6;    // words per read, words per mask, offset bits of an instance id
	bool sym = false;            // every record 2i+1 is the reverse complement of record 2i with mirrored masks (checked by the packing, or made by it):
	                             // phase A of the k-mer build then moves half the tuples (vdjx_kmer.hip, SYM)
	char* d_block = nullptr;     // one device block holds the four arrays below
	size_t block_cap = 0;
	u64* d_bases = nullptr;      // [R][W]
	u64* d_nmask = nullptr;      // [R][M] bit i = base i is not ACGT
	u64* d_lowq = nullptr;       // [R][M] the GATE mask: bit i = base i is not ACGT or (uint8)(q-33) < 20 (include_kmer, A2:240-259, needs nothing else: the
	                             // gating kernels read this word and not d_nmask)
	const uint8_t* d_quals = nullptr;   // quality rows (Phred+33 characters), `qstride` bytes apart: packed [R][qstride] inside d_block, or -- for a pool
	const uint8_t* d_quals2 = nullptr;  // loaded from ASCII records that stay resident in device memory (vdjx_pool_load_device) -- the records' own
	size_t q_split = ~(size_t) 0;       // quality characters, never copied: records below q_split in d_quals (primary), the others in d_quals2
	u32* pending_bad = nullptr;  // vdjx_pool_load_forward_begin: the load is still running on the copy stream (vdjx_pool_wait)
};

// the finished graph stays on the device until vdjx_graph_export copies it straight into the caller's arrays
struct vdjx_graph {
	vdjx_ctx* ctx = nullptr;
	int device = 0;
	int k = 0;
	size_t n = 0, pre_nodes = 0;
	size_t export_bytes = 0;         // the arrays of vdjx_graph_export lie in the first export_bytes bytes of d_block (vdjx_graph_block_layout)
	char* d_block = nullptr;
	size_t block_cap = 0;
	u64* d_first_inst = nullptr;
	u32 *d_gcnt = nullptr, *d_freq = nullptr, *d_to_ids = nullptr, *d_from_ids = nullptr;
	uint8_t *d_hv = nullptr, *d_hj = nullptr, *d_to_deg = nullptr, *d_from_deg = nullptr;
	char* d_kmers = nullptr;         // n*k ASCII
	u32* d_roots = nullptr;          // 0-based indices of the nodes without predecessor, ascending
	size_t n_roots = 0;
};

bool vdjx_ctx_alive(const vdjx_ctx* c);   // a pool or graph may be freed after its context
int vdjx_ri_join(vdjx_ctx* c);            // waits for a begun read-index build (vdjx_rindex.hip); its status, VDJX_OK if none is in flight
void vdjx_ri_open_gate(vdjx_ctx* c);      // a begun build that waits for the k-mer build's graph pass may start (no-op without one)
bool vdjx_host_block_holds(vdjx_ctx* c, const void* p, size_t bytes);   // [p, p + bytes) inside a block of vdjx_host_alloc (vdjx_core.hip)

// scoped workspace allocations out of the context's arena
struct vdjx_work {
	vdjx_ctx* c;
	vdjx_arena* ar;
	explicit vdjx_work(vdjx_ctx* ctx) : c(ctx), ar(&ctx->arena) {}
	vdjx_work(vdjx_ctx* ctx, vdjx_arena* arena) : c(ctx), ar(arena) {}      // (the begun read-index build: a workspace of its own)
	~vdjx_work() { ar->reset(); }
	template <typename T> hipError_t alloc(T** out, size_t n) {
		*out = (T*) ar->alloc((n ? n : 1) * sizeof(T));
		return *out ? hipSuccess : hipErrorOutOfMemory;
	}
	vdjx_arena::mark_t mark() const { return ar->mark(); }
	void release_to(vdjx_arena::mark_t m) { ar->release_to(m); }
};

// profiling: bracket a launch with events on the context's stream
struct vdjx_prof_scope {
	vdjx_ctx* c;
	const char* name;
	hipEvent_t a = nullptr, b = nullptr;
	hipStream_t st = nullptr;                             // the stream the launch goes to (null: the context's)
	vdjx_prof_scope(vdjx_ctx* ctx, const char* nm, hipStream_t stream = nullptr);      // ctx == nullptr: nothing is bracketed (another THREAD's launches)
	~vdjx_prof_scope();
};
void vdjx_prof_collect(vdjx_ctx* ctx, bool force = true);

// host-side laps (VDJX_LAPS=1): microseconds between two marks of a call, summed into vdjx_stat("us_<name>")
struct vdjx_laps {
	vdjx_ctx* c;
	bool on;
	std::chrono::steady_clock::time_point t;
	explicit vdjx_laps(vdjx_ctx* ctx) : c(ctx) { static const bool e = getenv("VDJX_LAPS") != nullptr; on = e; if (on) t = std::chrono::steady_clock::now(); }
	void mark(const char* name) {
		if (!on) return;
		const auto n = std::chrono::steady_clock::now();
		c->stats[std::string("us_") + name] += (uint64_t) std::chrono::duration_cast<std::chrono::microseconds>(n - t).count();
		t = n;
	}
};

// ----------------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------------
// where the quality characters of a record are (see vdjx_pool::d_quals)
struct vdjx_qrows {
	const uint8_t* a; const uint8_t* b; size_t split; int stride;
	__host__ __device__ inline const uint8_t* row(size_t rec) const { return rec < split ? a + rec * (size_t) stride : b + (rec - split) * (size_t) stride; }
};
__host__ __device__ inline u64 vdjx_mix(u64 lo, u64 hi) {
	u64 x = lo ^ (hi * 0x9E3779B97F4A7C15ull) ^ 0x2545F4914F6CDD1Dull;
	x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
	x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
	x ^= x >> 32;
	return x;
}

// the partition's hash of a k-mer: only where a k-mer's tuples MEET depends on it (its bucket; in the sharded build its owner), no
// result does.  Every gated instance is hashed five times on its way into its bucket (histogram, both placements of both partition
// passes) and the multiplier runs at a quarter of the ALU rate: four 32-bit multiplications here against the twelve the three 64-bit
// products of vdjx_mix compile to.  32 result bits on top of the word (the buckets use at most 20); every input bit flips every one
// of them with probability 0.49-0.51 (profiles/micro/bucket_hash.py), bucket occupancies of real k-mer sets as Poisson as vdjx_mix's.
__host__ __device__ inline u64 vdjx_bucket_mix(u64 lo, u64 hi) {
	u32 c = (u32) hi ^ ((u32) (hi >> 32) << 26);            // (2k - 64 <= 36 bits)
	u32 h = (c ^ (c >> 15)) * 0x9E3779B1u;
	h ^= h >> 15;
	h = (h ^ (u32) lo) * 0x85EBCA6Bu;
	h ^= h >> 13;
	const u32 b = (u32) (lo >> 32);
	h = (h ^ b ^ (b >> 16)) * 0xC2B2AE35u;
	h ^= h >> 16;
	h *= 0x27D4EB2Fu;
	h ^= h >> 15;
	return (u64) h << 32;
}

// hash of a read of W words (the read index: vdjx_rindex.hip, k_map_classify)
template <int W>
__device__ inline u64 ri_hash(const u64* __restrict__ w) {
	if (W == 2) return vdjx_mix(w[1], w[0]);
	u64 h = vdjx_mix(w[1], w[0]);
#pragma unroll
	for (int i = 2; i < W; i += 2) h = vdjx_mix(h ^ w[i], i + 1 < W ? w[i + 1] : 0ull);
	return h;
}

// k-mer at offset o of a packed read (first base most significant), as a 2k-bit integer
__host__ __device__ inline void vdjx_kmer_at(u64 bhi, u64 blo, int rl, int k, int o, u64& khi, u64& klo) {
	u128 b = ((u128) bhi << 64) | blo;
	int sh = 2 * (rl - k - o);
	u128 v = b >> sh;
	if (k < 64) v &= (((u128) 1) << (2 * k)) - 1;
	khi = (u64) (v >> 64);
	klo = (u64) v;
}

// LDS written by one lane of a wave and read by another without a workgroup barrier (the dense listings): the hardware keeps a
// wave's LDS operations in order, this keeps the COMPILER from moving or caching them across the hand-over
__device__ inline void vdjx_wave_lds_fence() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// the same for an offset that differs from lane to lane, written with masks and double shifts only.  Round 2 blamed the plain
// 128-bit shift by a per-lane amount (v_cmp into an SGPR pair + v_cndmask) for a few wrong k-mers in ten thousand when a wave's
// amounts lay on both sides of 64 (reads of more than 32 offsets).  The standalone reproducer (tests/test_gpu_platform.py) does NOT
// show it: 5 x 4 M random shifts, 0 mismatches on this stack -- so that diagnosis stands unconfirmed, and the likelier cause is the
// dense listing it was found in handing LDS data from lane to lane without telling the compiler (vdjx_wave_lds_fence, added in
// round 3).  This form is kept: it is what the randomised differential tests (tests/test_gpu_fuzz.py) have been green with, and it
// costs nothing over the shift.
__device__ inline void vdjx_kmer_at_lane(u64 bhi, u64 blo, int rl, int k, int o, u64& khi, u64& klo) {
	const u32 sh = (u32) (2 * (rl - k - o));               // 0 .. 126
	const u64 big = 0ull - (u64) (sh >> 6);                 // all ones: shift by 64 or more
	const u64 x_lo = (bhi & big) | (blo & ~big), x_hi = bhi & ~big;
	const u32 s = sh & 63u;
	u64 lo = (x_lo >> s) | ((x_hi << 1) << (63u - s));
	u64 hi = x_hi >> s;
	if (k < 32) { lo &= (1ull << (2 * k)) - 1ull; hi = 0; }          // (k is uniform: scalar branches)
	else if (k < 64) hi &= (1ull << (2 * k - 64)) - 1ull;
	khi = hi;
	klo = lo;
}

// ---- long reads (left-aligned words; see vdjx_pool) ----
// k-mer at offset o of a read given as words w[0 .. W) followed by at least two readable words (their content never reaches the
// result): the 128 bits from bit 2o of the stream, then the top 2k of them.  o may differ from lane to lane: 64-bit shifts only.
__device__ inline void vdjx_kmer_at_words(const u64* __restrict__ w, int k, int o, u64& khi, u64& klo) {
	const int wi = o >> 5;
	const u32 s = 2u * ((u32) o & 31u);                     // 0 .. 62
	const u64 x0 = w[wi], x1 = w[wi + 1], x2 = w[wi + 2];
	const u64 hi = (x0 << s) | ((x1 >> 1) >> (63u - s));
	const u64 lo = (x1 << s) | ((x2 >> 1) >> (63u - s));
	const u32 r = 128u - 2u * (u32) k;                       // uniform; k <= 50: r >= 28
	if (r >= 64u) { klo = hi >> (r - 64u); khi = 0; }
	else { klo = (lo >> r) | (hi << (64u - r)); khi = hi >> r; }
}
// 32 bits = the 16 bases from base index bi on (zeros past the words), first base most significant
__device__ inline u32 vdjx_bases16_words(const u64* __restrict__ w, int bi) {
	const int wi = bi >> 5;
	const u32 s = 2u * ((u32) bi & 31u);
	const u64 hi = (w[wi] << s) | ((w[wi + 1] >> 1) >> (63u - s));
	return (u32) (hi >> 32);
}
__device__ inline u32 vdjx_base_words(const u64* __restrict__ w, int i) { return (u32) (w[i >> 5] >> (62u - 2u * ((u32) i & 31u))) & 3u; }

// masks of up to three words (bit i%64 of word i/64), by value
struct vdjx_mask3 {
	u64 w0, w1, w2;
	__host__ __device__ inline u64 word(int i) const { return i == 0 ? w0 : (i == 1 ? w1 : (i == 2 ? w2 : 0ull)); }
	__device__ inline bool test(int i) const { return (word(i >> 6) >> (i & 63)) & 1ull; }
	__device__ inline void set(int i) { const u64 b = 1ull << (i & 63); if ((i >> 6) == 0) w0 |= b; else if ((i >> 6) == 1) w1 |= b; else w2 |= b; }
	__device__ inline bool any() const { return (w0 | w1 | w2) != 0ull; }
	// 64 bits from bit `from` on
	__device__ inline u64 slice(int from) const {
		const int i = from >> 6;
		const u32 s = (u32) from & 63u;
		return (word(i) >> s) | ((word(i + 1) << 1) << (63u - s));
	}
	// first set bit at or above `from`, or `none`
	__device__ inline int next(int from, int none) const {
		for (int i = from >> 6; i < 3; i++) {
			u64 x = word(i);
			if (i == (from >> 6)) x &= ~0ull << (from & 63);
			if (x) return 64 * i + __builtin_ctzll(x);
		}
		return none;
	}
};
// the 16 offsets base_o .. base_o+15 (base_o a multiple of 16) whose k bases are all clean: bit j set iff none of the bits
// base_o+j .. base_o+j+k-1 of `bad` is, and base_o+j < P
__device__ inline u32 vdjx_clean16(const vdjx_mask3& bad, int base_o, int k, int P) {
	u64 lo = bad.slice(base_o), hi = bad.slice(base_o + 64);
	int cur = 1;
	while (cur * 2 <= k) { lo |= (lo >> cur) | (hi << (64 - cur)); hi |= hi >> cur; cur *= 2; }     // (cur <= 32)
	const int rest = k - cur;                                                                       // 0 .. 31
	if (rest) lo |= (lo >> rest) | (hi << (64 - rest));
	const int left = P - base_o;
	return ~(u32) lo & 0xFFFFu & (left >= 16 ? 0xFFFFu : (left > 0 ? (1u << left) - 1u : 0u));
}

// 32 bits of the packed read from bit `pos` (0 .. 127) up, and one base, for positions that differ from lane to lane (see above)
__device__ inline u32 vdjx_bits_at_lane(u64 bhi, u64 blo, u32 pos) {
	const u64 big = 0ull - (u64) (pos >> 6);
	const u64 x_lo = (bhi & big) | (blo & ~big), x_hi = bhi & ~big;
	const u32 s = pos & 63u;
	return (u32) ((x_lo >> s) | ((x_hi << 1) << (63u - s)));
}
__device__ inline u32 vdjx_base_at_lane(u64 bhi, u64 blo, int rl, int i) { return vdjx_bits_at_lane(bhi, blo, (u32) (2 * (rl - 1 - i))) & 3u; }

// reverse complement of a packed k-mer (2k bits, right-aligned, first base most significant; A0 T1 C2 G3: the complement is bit 0
// of every code): the 2-bit groups in reverse order, complemented.  k is uniform.
__device__ inline void vdjx_kmer_rc(u64 hi, u64 lo, int k, u64& rhi, u64& rlo) {
	u64 a = __brevll(lo), b = __brevll(hi);            // 128-bit bit reversal: (b:a) -> new (hi:lo) = (a:b)
	a = ((a >> 1) & 0x5555555555555555ull) | ((a & 0x5555555555555555ull) << 1);      // the two bits of every group back in order
	b = ((b >> 1) & 0x5555555555555555ull) | ((b & 0x5555555555555555ull) << 1);
	const u32 s = 128u - 2u * (u32) k;                  // the reversed k-mer sits in the TOP 2k bits of (a:b): down by s (28 .. 126)
	u64 h, l;
	if (s >= 64u) { l = s == 64u ? a : a >> (s - 64u); h = 0; }
	else { l = (b >> s) | (a << (64u - s)); h = a >> s; }
	const u32 nb = 2u * (u32) k;
	const u64 m_lo = nb >= 64u ? ~0ull : (1ull << nb) - 1ull, m_hi = nb > 64u ? (1ull << (nb - 64u)) - 1ull : 0ull;
	rlo = l ^ (0x5555555555555555ull & m_lo);
	rhi = h ^ (0x5555555555555555ull & m_hi);
}

// reverse complement of a read of rl <= 64 bases (a right-aligned 2*rl-bit integer in hi:lo, vdjx_pool's short-read record)
__device__ inline void vdjx_read_rc(u64 hi, u64 lo, int rl, u64& rhi, u64& rlo) {
	if (rl < 64) { vdjx_kmer_rc(hi, lo, rl, rhi, rlo); return; }
	u64 a = __brevll(lo), b = __brevll(hi);
	a = ((a >> 1) & 0x5555555555555555ull) | ((a & 0x5555555555555555ull) << 1);
	b = ((b >> 1) & 0x5555555555555555ull) | ((b & 0x5555555555555555ull) << 1);
	rhi = a ^ 0x5555555555555555ull;
	rlo = b ^ 0x5555555555555555ull;
}

// ---- pools whose odd records are the reverse complements of the records before them (what add_to_buffer writes, bam_read.c:206-244:
// vdjx_pool::sym).  The gated instances of record 2i+1 mirror those of record 2i: k-mer X at offset o there is rc(X) at offset
// rl-k-o here.  Of the two tuples the pair of instances would make, phase A of the k-mer build keeps the one whose key is the smaller
// of X and rc(X) (k odd: they differ) -- half the tuples -- and restores the other side where the verdicts are made.
// the mirror of an instance id (record << ob | offset): the same bases read from the other record of the couple
__device__ inline u64 vdjx_inst_mirror(u64 inst, int ob, int rl, int k) {
	const u64 om = (1ull << ob) - 1ull;
	return (((inst >> ob) ^ 1ull) << ob) | ((u64) (rl - k) - (inst & om));
}

// Is the 2k-bit k-mer (hi:lo) equal to itself d bases further on, i.e. K[i] == K[i+d] for all i < k-d (0 < d < k)?  Two gated
// instances of one k-mer at offsets d apart in reads with identical sequences force that period on the k-mer, so a k-mer WITHOUT it
// proves the two reads different (compare_read, A2:142-144) without fetching them.  d differs from lane to lane: masks and double
// shifts only (see vdjx_kmer_at_lane).
__device__ inline bool vdjx_kmer_has_period(u64 hi, u64 lo, int k, u32 d) {
	const u32 sh = 2u * d;                                   // 2 .. 98
	const u64 big = 0ull - (u64) (sh >> 6);
	const u64 x_lo = (hi & big) | (lo & ~big), x_hi = hi & ~big;
	const u32 s = sh & 63u;
	const u64 s_lo = (x_lo >> s) | ((x_hi << 1) << (63u - s)), s_hi = x_hi >> s;
	const u32 bits = 2u * ((u32) k - d);                      // 2 .. 98: the low bits that must agree
	const u32 bl = bits < 64u ? bits : 64u, bh = bits > 64u ? bits - 64u : 0u;
	const u64 m_lo = ~0ull >> (64u - bl), m_hi = (1ull << bh) - 1ull;
	return (((s_lo ^ lo) & m_lo) | ((s_hi ^ hi) & m_hi)) == 0ull;
}

// wave-wide unsigned minimum through DPP row shifts and broadcasts (gfx9 family): no LDS traffic, unlike __shfl_xor
// (ds_bpermute); every lane gets the result.  All 64 lanes must be active (the result is read from lane 63).
__device__ inline u32 vdjx_wave_min(u32 v) {
	u32 t;
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:1
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:2
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:4
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:8: lane 15 of a row = row minimum
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;   // row_bcast:15 -> rows 1, 3
	t = (u32) __builtin_amdgcn_update_dpp((int) 0xFFFFFFFF, (int) v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;   // row_bcast:31 -> rows 2, 3
	return (u32) __builtin_amdgcn_readlane((int) v, 63);
}

// wave-wide maximum of a 64-bit value, in every lane (all 64 lanes active)
__device__ inline u64 vdjx_wave_max64(u64 v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const u64 t = (u64) __shfl_xor((unsigned long long) v, d, 64); v = t > v ? t : v; }
	return v;
}

// wave-wide inclusive prefix sum through DPP (row shifts inside the rows of 16 lanes, then the two row broadcasts): all 64 lanes
// must be active
__device__ inline int vdjx_wave_scan_add(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
	return v;
}

// A read / write of a word other lanes are changing (LDS tables): a relaxed atomic access of workgroup scope.  NOT `volatile`: the
// address-space inference of the compiler leaves volatile accesses alone, so a volatile access through a pointer that went through
// a function parameter stays a FLAT one with system-scope cache bits -- flat_load_dword ... sc0 sc1 and a wait on both the vector
// memory and the LDS counter, instead of one ds_read (round 3: 50 of them in k_gated_reduce's sweeps).
template <typename T> __device__ inline T vdjx_peek(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ inline void vdjx_poke(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a value that publishes data written before it / the read that such data is reached through (claim-and-publish slots in LDS)
template <typename T> __device__ inline T vdjx_peek_acquire(const T* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ inline void vdjx_poke_release(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// cnt[idx] += 1 and mn[idx] = min(mn[idx], val) on LDS arrays for the lanes with `active`.  Hot k-mers put most lanes of a wave on
// ONE address: the lanes that share the first active lane's index are combined into one add and one min; the others go one by one.
__device__ inline void vdjx_lds_count_min(u32* cnt, u32* mn, u32 idx, u32 val, bool active) {
	const u64 act = __ballot(active);
	if (act) {
		const int leader = __ffsll((long long) act) - 1;
		const u32 lidx = (u32) __builtin_amdgcn_readlane((int) idx, leader);
		const bool same = active && idx == lidx;
		const u64 m = __ballot(same);
		if (__popcll(m) >= 16) {
			const u32 mv = vdjx_wave_min(same ? val : 0xFFFFFFFFu);
			if (__lane_id() == leader) { atomicAdd(&cnt[lidx], (u32) __popcll(m)); atomicMin(&mn[lidx], mv); }
			active = active && !same;
		}
	}
	if (active) { atomicAdd(&cnt[idx], 1u); atomicMin(&mn[idx], val); }
}

// the same for a 64-bit minimum of values below 2^63 (instance ids).  The loop of the caller must be wave-uniform (ballots, DPP).
__device__ inline void vdjx_lds_count_min64(u32* cnt, u64* mn, u32 idx, u64 val, bool active) {
	const u64 act = __ballot(active);
	if (act) {
		const int leader = __ffsll((long long) act) - 1;
		const u32 lidx = (u32) __builtin_amdgcn_readlane((int) idx, leader);
		const bool same = active && idx == lidx;
		const u64 m = __ballot(same);
		if (__popcll(m) >= 8) {
			const u32 hi_min = vdjx_wave_min(same ? (u32) (val >> 32) : 0xFFFFFFFFu);
			const u32 lo_min = vdjx_wave_min(same && (u32) (val >> 32) == hi_min ? (u32) val : 0xFFFFFFFFu);
			if (__lane_id() == leader) {
				atomicAdd(&cnt[lidx], (u32) __popcll(m));
				atomicMin((unsigned long long*) &mn[lidx], ((unsigned long long) hi_min << 32) | lo_min);
			}
			active = active && !same;
		}
	}
	if (active) { atomicAdd(&cnt[idx], 1u); atomicMin((unsigned long long*) &mn[idx], (unsigned long long) val); }
}

__device__ inline u32 vdjx_wave_inc(u32* ctr, bool pred) {
	// wave-aggregated counter increment (LDS or global); returns this lane's slot, undefined if !pred
	const u64 m = __ballot(pred);
	if (!m) return 0;                                      // (wave-uniform)
	const int lane = __lane_id();
	const int leader = __ffsll((long long) m) - 1;
	u32 base = 0;
	if (lane == leader) base = atomicAdd(ctr, (u32) __popcll(m));
	base = (u32) __builtin_amdgcn_readlane((int) base, leader);     // (v_readlane: no LDS round trip, unlike __shfl)
	return base + (u32) __popcll(m & ((1ull << lane) - 1ull));
}
// the same with `each` places per lane
__device__ inline u32 vdjx_wave_inc(u32* ctr, bool pred, u32 each) {
	const u64 m = __ballot(pred);
	if (!m) return 0;
	const int lane = __lane_id();
	const int leader = __ffsll((long long) m) - 1;
	u32 base = 0;
	if (lane == leader) base = atomicAdd(ctr, each * (u32) __popcll(m));
	base = (u32) __builtin_amdgcn_readlane((int) base, leader);
	return base + each * (u32) __popcll(m & ((1ull << lane) - 1ull));
}
