// vdjx_core.hip -- context, error, profiling and pool-packing (K1) parts of libvdjx.
#include "vdjx_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <set>

// ----------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void vdjx_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

extern "C" const char* vdjx_last_error(void) { return g_err; }
extern "C" const char* vdjx_version(void) { return "vdjx 0.1 (gfx950)"; }

// ----------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------
// ---- live contexts (pools and graphs may outlive theirs) and the block cache
static std::mutex g_ctx_mu;
static std::set<const vdjx_ctx*> g_ctx_live;
bool vdjx_ctx_alive(const vdjx_ctx* c) {
	std::lock_guard<std::mutex> lk(g_ctx_mu);
	return g_ctx_live.count(c) != 0;
}

extern "C" int vdjx_init(int device, vdjx_ctx** out) {
	if (!out) { vdjx_set_error("vdjx_init: out is NULL"); return VDJX_EINVAL; }
	*out = nullptr;
	int n = 0;
	HIP_TRY(hipGetDeviceCount(&n));
	if (device < 0 || device >= n) { vdjx_set_error("vdjx_init: device %d of %d", device, n); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(device));
	vdjx_ctx* c = new vdjx_ctx();
	c->device = device;
	hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pairs_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
	for (auto& ev : c->ev_up) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pairs_copied, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_root_done, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->ri_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->root_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_root_go, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ri_go, hipEventDisableTiming);
	if (e == hipSuccess) e = hipHostMalloc(&c->h_pin, VDJX_HPIN_BYTES, hipHostMallocDefault);
	if (e != hipSuccess) { delete c; vdjx_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return VDJX_EHIP; }
	{
		std::lock_guard<std::mutex> lk(g_ctx_mu);
		g_ctx_live.insert(c);
	}
	*out = c;
	return VDJX_OK;
}

static void free_dev(void* p) { if (p) (void) hipFree(p); }

hipError_t vdjx_block_cache::acquire(size_t need, char** out, size_t* cap) {
	need = (need + 4095) & ~(size_t) 4095;
	int best = -1;
	for (size_t i = 0; i < free_list.size(); i++)
		if (free_list[i].cap >= need && free_list[i].cap <= 4 * need + (1u << 20) && (best < 0 || free_list[i].cap < free_list[best].cap)) best = (int) i;
	if (best >= 0) {
		*out = free_list[best].p; *cap = free_list[best].cap;
		free_list.erase(free_list.begin() + best);
		return hipSuccess;
	}
	const size_t want = need + need / 8;        // headroom: the next batch is rarely the same size to the byte
	hipError_t e = hipMalloc(out, want);
	if (e == hipSuccess) { *cap = want; return e; }
	(void) hipGetLastError();
	drop();                                     // memory pressure: give the cached blocks back and ask for the exact size
	e = hipMalloc(out, need);
	if (e == hipSuccess) *cap = need;
	return e;
}

void vdjx_block_cache::release(char* p, size_t cap) {
	if (!p) return;
	if (free_list.size() >= 6) {                // evict the smallest
		size_t m = 0;
		for (size_t i = 1; i < free_list.size(); i++) if (free_list[i].cap < free_list[m].cap) m = i;
		if (free_list[m].cap < cap) { (void) hipFree(free_list[m].p); free_list[m] = {p, cap}; }
		else (void) hipFree(p);
		return;
	}
	free_list.push_back({p, cap});
}

void vdjx_block_cache::drop() {
	for (auto& b : free_list) (void) hipFree(b.p);
	free_list.clear();
}

extern "C" void vdjx_shutdown(vdjx_ctx* c) {
	if (!c) return;
	(void) hipSetDevice(c->device);
	(void) vdjx_ri_join(c);
	(void) hipStreamSynchronize(c->stream);
	if (c->ri_stream) { (void) hipStreamSynchronize(c->ri_stream); (void) hipStreamDestroy(c->ri_stream); }
	if (c->ev_ri_go) (void) hipEventDestroy(c->ev_ri_go);
	c->ri_arena.release(true);
	if (c->root_stream) { (void) hipStreamSynchronize(c->root_stream); (void) hipStreamDestroy(c->root_stream); }
	if (c->ev_root_go) (void) hipEventDestroy(c->ev_root_go);
	c->root_arena.release(true);
	free_dev(c->d_vbits); free_dev(c->d_jbits); free_dev(c->d_anchor_tmp);
	free_dev(c->d_vtext); free_dev(c->d_line_off); free_dev(c->d_seed_code); free_dev(c->d_seed_pos);
	free_dev(c->d_ri_tab); free_dev(c->d_ri_start); free_dev(c->d_ri_recs); free_dev(c->d_ri_csr8); free_dev(c->d_ri_csr_pair);
	free_dev(c->d_pair_r2);
	free_dev(c->d_sam_keys); free_dev(c->d_sam_lens);
	if (c->h_sam_merge) (void) hipHostFree(c->h_sam_merge);
	free_dev(c->me_pairs); free_dev(c->me_hit); free_dev(c->me_dense); free_dev(c->me_book); free_dev(c->wp_buf); free_dev(c->d_ri_cnt1); free_dev(c->d_ri_dstart); free_dev(c->d_ri_d8);
	for (int i = 0; i < 2; i++) { free_dev(c->d_stage[i]); if (c->ev_copied[i]) (void) hipEventDestroy(c->ev_copied[i]); if (c->ev_packed[i]) (void) hipEventDestroy(c->ev_packed[i]); }
	c->arena.release(true);
	c->shard_arena.release(true);
	c->blocks.drop();
	{
		std::lock_guard<std::mutex> lk(g_ctx_mu);
		g_ctx_live.erase(c);
	}
	for (auto& p : c->prof_pending) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
	for (auto& e : c->ev_free) (void) hipEventDestroy(e);
	(void) hipStreamSynchronize(c->copy_stream);
	(void) hipStreamDestroy(c->copy_stream);
	if (c->pairs_stream) { (void) hipStreamSynchronize(c->pairs_stream); (void) hipStreamDestroy(c->pairs_stream); }
	if (c->ev_gathered) (void) hipEventDestroy(c->ev_gathered);
	if (c->up_stream) { (void) hipStreamSynchronize(c->up_stream); (void) hipStreamDestroy(c->up_stream); }
	for (auto& ev : c->ev_up) if (ev) (void) hipEventDestroy(ev);
	if (c->ev_pairs_copied) (void) hipEventDestroy(c->ev_pairs_copied);
	if (c->ev_plan) (void) hipEventDestroy(c->ev_plan);
	if (c->ev_root_done) (void) hipEventDestroy(c->ev_root_done);
	if (c->h_plan) (void) hipHostFree(c->h_plan);
	if (c->h_res) (void) hipHostFree(c->h_res);
	if (c->h_pin) (void) hipHostFree(c->h_pin);
	if (c->h_sam_text) (void) hipHostFree(c->h_sam_text);
	free_dev(c->d_sam_text); free_dev(c->d_sam_names); free_dev(c->d_sam_noff);
	(void) hipStreamDestroy(c->stream);
	delete c;
}

// Hands the workspaces' memory back to the device: the arenas hold the PEAK of the calls so far (tens of GB after a k-mer build),
// which a process that builds once and then serves scorer calls -- a rank of `vdjer --gpus N` -- has no further use for.  The address
// ranges stay reserved; a later call maps what it needs again (0.02 ms a piece).
extern "C" int vdjx_trim(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_trim: ctx is NULL"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	(void) vdjx_ri_join(c);
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipStreamSynchronize(c->copy_stream));
	if (c->pairs_stream) HIP_TRY(hipStreamSynchronize(c->pairs_stream));
	if (c->root_stream) HIP_TRY(hipStreamSynchronize(c->root_stream));
	c->arena.release(false);
	c->ri_arena.release(false);
	c->root_arena.release(false);
	if (!c->live_shard) c->shard_arena.release(false);
	c->blocks.drop();
	// the scorers' result buffers (grow-only between calls: the pair lists of the last window batch, the mapped pairs and the SAM
	// records of the last contigs): results the caller has taken; the next call allocates what it needs
	auto drop = [](void*& p, size_t& cap) { if (p) (void) hipFree(p); p = nullptr; cap = 0; };
	drop(c->wp_buf, c->wp_cap);
	c->wp_n = 0;
	drop(c->me_pairs, c->me_cap);
	if (c->me_hit) { (void) hipFree(c->me_hit); c->me_hit = nullptr; }
	drop(c->me_dense, c->me_dense_cap);
	c->me_gathered_cap = 0;
	drop(c->me_book, c->me_book_cap);
	if (c->d_sam_text) { (void) hipFree(c->d_sam_text); c->d_sam_text = nullptr; }
	if (c->h_sam_text) { (void) hipHostFree(c->h_sam_text); c->h_sam_text = nullptr; }
	c->sam_text_cap = 0;
	if (c->d_sam_keys) { (void) hipFree(c->d_sam_keys); c->d_sam_keys = nullptr; }
	if (c->d_sam_lens) { (void) hipFree(c->d_sam_lens); c->d_sam_lens = nullptr; }
	c->sam_blk_cap = 0;
	c->me_key = 0;          // (the cached counting call of vdjx_map_emit pointed into the workspace)
	c->me_src = nullptr;
	return VDJX_OK;
}

// The read index of the context is dropped and its arrays (kept from build to build otherwise: 2-5 GB at 10 M pairs) go back to the
// device.  The scorers need a new vdjx_read_index_build afterwards.
extern "C" int vdjx_read_index_drop(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_read_index_drop: ctx is NULL"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	(void) vdjx_ri_join(c);
	HIP_TRY(hipStreamSynchronize(c->stream));
	free_dev(c->d_ri_tab); free_dev(c->d_ri_start); free_dev(c->d_ri_cnt1); free_dev(c->d_ri_dstart); free_dev(c->d_pair_r2);
	free_dev(c->d_ri_recs); free_dev(c->d_ri_csr8); free_dev(c->d_ri_csr_pair); free_dev(c->d_ri_d8);
	c->d_ri_tab = nullptr; c->d_ri_start = nullptr; c->d_ri_cnt1 = nullptr; c->d_ri_dstart = nullptr; c->d_pair_r2 = nullptr;
	c->d_ri_recs = nullptr; c->d_ri_csr8 = nullptr; c->d_ri_csr_pair = nullptr; c->d_ri_d8 = nullptr;
	for (auto& cap : c->ri_cap) cap = 0;
	c->ri_pool = nullptr;
	c->me_key = 0;
	return VDJX_OK;
}

// bytes from one device buffer to another on the context's stream (test and bench drivers move library-owned results into buffers
// of their own with it; the C host calls the runtime directly)
extern "C" int vdjx_device_copy(vdjx_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
	if (!c) { vdjx_set_error("vdjx_device_copy: ctx is NULL"); return VDJX_EINVAL; }
	if (!bytes) return VDJX_OK;
	if (!d_dst || !d_src) { vdjx_set_error("vdjx_device_copy: NULL buffer"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return VDJX_OK;
}

static bool arena_trace() { static const bool t = getenv("VDJX_ARENA_TRACE") != nullptr; return t; }     // diagnostic: every growth of a workspace

// back the range up to `upto` bytes: pieces of at most 4 GB (a single 38 GB allocation is what took seconds)
bool vdjx_arena::grow(size_t upto) {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return false;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = dev;
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	// (equal pieces, each at a multiple of its size: hipMemSetAccess refused a 3,953 MB piece mapped behind one of 256 MB, "invalid argument")
	const size_t piece = (size_t) 1 << 30;
	while (mapped < upto) {
		const size_t want = piece;
		if (mapped + want > reserved) { vdjx_set_error("workspace: the reserved address range (%zu GB) is used up", reserved >> 30); return false; }
		const auto t0 = std::chrono::steady_clock::now();
		hipMemGenericAllocationHandle_t h;
		const char* step = "hipMemCreate";
		hipError_t e = hipMemCreate(&h, want, &prop, 0);
		if (e == hipSuccess) {
			step = "hipMemMap";
			e = hipMemMap(base + mapped, want, 0, h, 0);
			if (e == hipSuccess) { step = "hipMemSetAccess"; e = hipMemSetAccess(base + mapped, want, &acc, 1); if (e != hipSuccess) (void) hipMemUnmap(base + mapped, want); }
			if (e != hipSuccess) (void) hipMemRelease(h);
		}
		if (e != hipSuccess && arena_trace()) fprintf(stderr, "[vdjx] workspace: %s of %zu MB at %zu MB failed: %s\n", step, want >> 20, mapped >> 20, hipGetErrorString(e));
		if (e != hipSuccess) { vdjx_set_error("workspace: %zu MB more (behind %zu MB): %s", want >> 20, mapped >> 20, hipGetErrorString(e)); (void) hipGetLastError(); return false; }
		pieces.emplace_back((void*) h, want);
		mapped += want;
		if (arena_trace()) fprintf(stderr, "[vdjx] workspace grows by %zu MB to %zu MB: %.3f ms\n", want >> 20, mapped >> 20,
		                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	}
	return true;
}

void* vdjx_arena::alloc(size_t bytes) {
	bytes = (bytes + 255) & ~(size_t) 255;
	if (mode == 0) {
		// one range, if the runtime manages virtual memory on this device (VDJX_ARENA_CHUNKS=1: the chunk list, for comparison)
		mode = 2;
		int dev = 0, vmm = 0;
		static const bool force_chunks = getenv("VDJX_ARENA_CHUNKS") != nullptr;
		if (!force_chunks && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev) == hipSuccess && vmm) {
			hipMemAllocationProp prop = {};
			prop.type = hipMemAllocationTypePinned;
			prop.location.type = hipMemLocationTypeDevice;
			prop.location.id = dev;
			size_t total = 0, fr = 0, g = 0;
			if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && g && hipMemGetInfo(&fr, &total) == hipSuccess) {
				const size_t gb = (size_t) 1 << 30;
				const size_t want = (total + gb - 1) / gb * gb;            // (addresses, not memory: as much as the device has)
				void* r = nullptr;
				if (hipMemAddressReserve(&r, want, gb, nullptr, 0) == hipSuccess && r) { base = (char*) r; reserved = want; gran = g; mode = 1; }
			}
		}
		(void) hipGetLastError();
		if (arena_trace()) fprintf(stderr, "[vdjx] workspace: %s\n", mode == 1 ? "one reserved address range, backed as needed" : "list of chunks");
	}
	if (mode == 1) {
		if (used + bytes > mapped && !grow(used + bytes)) return nullptr;
		void* r = base + used;
		used += bytes;
		return r;
	}
	if (!chunks.empty() && used + bytes <= chunks[cur].cap) {
		void* r = chunks[cur].p + used;
		used += bytes;
		return r;
	}
	// a later chunk the arena already has (after release_to / reset), else a new one
	for (size_t i = chunks.empty() ? 0 : cur + 1; i < chunks.size(); i++)
		if (bytes <= chunks[i].cap) { cur = i; used = bytes; return chunks[i].p; }
	size_t cap = bytes > ((size_t) 64 << 20) ? bytes : ((size_t) 64 << 20);
	char* p = nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	hipError_t e = hipMalloc(&p, cap);
	if (arena_trace()) fprintf(stderr, "[vdjx] workspace grows by %zu MB (chunk %zu): hipMalloc %.3f ms\n", cap >> 20, chunks.size(),
	                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	if (e != hipSuccess) { vdjx_set_error("workspace alloc of %zu bytes: %s", cap, hipGetErrorString(e)); return nullptr; }
	chunks.push_back({p, cap});
	cur = chunks.size() - 1;
	used = bytes;
	return p;
}

void vdjx_arena::reset() {
	cur = 0;
	used = 0;
}

void vdjx_arena::release_to(mark_t m) {
	if (mode == 1) { used = m.used <= used ? m.used : used; return; }
	if (chunks.empty() || m.cur >= chunks.size()) { cur = 0; used = 0; return; }
	cur = m.cur;
	used = m.used;
}

void vdjx_arena::release(bool for_good) {
	if (mode == 1) {
		size_t at = 0;
		for (auto& pc : pieces) {
			(void) hipMemUnmap(base + at, pc.second);
			(void) hipMemRelease((hipMemGenericAllocationHandle_t) pc.first);
			at += pc.second;
		}
		pieces.clear();
		mapped = 0;
		used = 0;
		if (for_good) { (void) hipMemAddressFree(base, reserved); base = nullptr; reserved = 0; mode = 0; }      // (else the range stays reserved: the arena may be used again)
		return;
	}
	for (auto& ch : chunks) (void) hipFree(ch.p);
	chunks.clear();
	cur = 0;
	used = 0;
}

extern "C" int vdjx_sync(vdjx_ctx* c) {
	if (!c) { vdjx_set_error("vdjx_sync: ctx is NULL"); return VDJX_EINVAL; }
	HIP_TRY(hipStreamSynchronize(c->stream));
	return VDJX_OK;
}

// ----------------------------------------------------------------------------------------------
// profiling
// ----------------------------------------------------------------------------------------------
vdjx_prof_scope::vdjx_prof_scope(vdjx_ctx* ctx, const char* nm, hipStream_t stream) : c(ctx), name(nm), st(stream) {
	if (!c || !c->profiling) return;
	if (!st) st = c->stream;
	if (!c->prof_only.empty() && c->prof_only != nm) return;          // (vdjx_profile_only: one scope is bracketed, the others cost nothing)
	auto take = [&](hipEvent_t* e) {
		if (!c->ev_free.empty()) { *e = c->ev_free.back(); c->ev_free.pop_back(); return true; }
		return hipEventCreate(e) == hipSuccess;
	};
	if (!take(&a) || !take(&b)) { a = b = nullptr; return; }
	(void) hipEventRecord(a, st);
}

vdjx_prof_scope::~vdjx_prof_scope() {
	if (!c || !c->profiling || !a) return;
	(void) hipEventRecord(b, st);
	c->prof_pending.push_back({name, a, b});
}

// `force` = false (the end of a compute call): the events are read when somebody asks for the numbers, not here -- reading forty
// events costs the host 0.1 ms during which the device has nothing to do; a bound on the pending list keeps the event pool small
void vdjx_prof_collect(vdjx_ctx* c, bool force) {
	if (!force && c->prof_pending.size() < 16384) return;
	for (auto& p : c->prof_pending) {
		float ms = 0;
		if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
			if (!c->prof.count(p.name)) c->prof_names.push_back(p.name);
			auto& e = c->prof[p.name];
			e.ms += ms;
			e.launches++;
		}
		c->ev_free.push_back(p.a);
		c->ev_free.push_back(p.b);
	}
	c->prof_pending.clear();
}

extern "C" uint64_t vdjx_stat(vdjx_ctx* c, const char* name) {
	if (!c || !name) return 0;
	auto it = c->stats.find(name);
	return it == c->stats.end() ? 0 : it->second;
}

extern "C" int vdjx_profile_enable(vdjx_ctx* c, int on) {
	if (!c) return VDJX_EINVAL;
	vdjx_prof_collect(c);
	c->profiling = on != 0;
	return VDJX_OK;
}

// Only the scopes of this name are bracketed from now on (NULL or "": all of them again).  Two event records per scope are 80 per step
// of the whole path -- 0.16 ms of a 10 M-pair step, 0.29 ms of a 1 M-pair one: a caller that times a region and wants ONE kernel's
// duration from inside it (bench.py: the roofline kernel) pays for that one.
extern "C" int vdjx_profile_only(vdjx_ctx* c, const char* name) {
	if (!c) return VDJX_EINVAL;
	vdjx_prof_collect(c);
	c->prof_only = name ? name : "";
	return VDJX_OK;
}

extern "C" int vdjx_profile_reset(vdjx_ctx* c) {
	if (!c) return VDJX_EINVAL;
	vdjx_prof_collect(c);
	c->prof.clear();
	c->prof_names.clear();
	return VDJX_OK;
}

extern "C" int vdjx_profile_count(vdjx_ctx* c) {
	if (!c) return VDJX_EINVAL;
	vdjx_prof_collect(c);
	return (int) c->prof_names.size();
}

extern "C" int vdjx_profile_get(vdjx_ctx* c, int idx, const char** name, double* total_ms, uint64_t* launches) {
	if (!c || idx < 0 || idx >= (int) c->prof_names.size()) { vdjx_set_error("vdjx_profile_get: bad index"); return VDJX_EINVAL; }
	const std::string& nm = c->prof_names[idx];
	if (name) *name = nm.c_str();
	if (total_ms) *total_ms = c->prof[nm].ms;
	if (launches) *launches = c->prof[nm].launches;
	return VDJX_OK;
}

// ----------------------------------------------------------------------------------------------
// K1 pool_pack: ASCII records -> 2-bit bases + masks + quality bytes
//   replaces the record parsing of A2:380-394 (layout written by bam_read.c:206-244)
//   One 256-thread workgroup stages 256 records through LDS with 16-byte coalesced loads; thread t then packs record t
//   FOUR CHARACTERS AT A TIME: aligned 32-bit LDS words funnelled to the record's own alignment (v_alignbyte), the four base
//   codes / not-ACGT flags / Phred<20 flags of a word by byte-parallel arithmetic, the quality bytes handed back through LDS so
//   that the quality rows leave as whole 16-byte lanes on consecutive addresses.  (One thread walking its record byte by byte --
//   170 one-byte LDS reads per record -- ran at 0.24 of the HBM rate.)  HBM-bound: reads 2*rl+1 B, writes 32 B + qstride per record.
// ----------------------------------------------------------------------------------------------
#define PACK_RECS 256
#define SW_H 0x80808080u
// bit 7 of every byte of x that is zero
__device__ inline u32 sw_zero_bytes(u32 x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & SW_H; }
// bits 7, 15, 23, 31 -> bits 0..3 (one multiplication lines the four bits up at 21..24: the partial products do not meet)
__device__ inline u32 sw_gather4(u32 m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

// four bases (byte 0 first): codes as 8 bits, first base most significant (seq_to_kmer.c:6-29: A0 T1 C2 G3; anything else 0), and
// the flags "not ACGT" / "neither ACGT nor N" (bit 7 per byte).  Bits 1-2 of an ASCII base tell A, C, T and G apart (0, 1, 2, 3):
// one byte permute looks up the letter those two bits stand for, and a character is a base iff it IS that letter -- one comparison
// instead of four (1.03 -> 0.98 ms per step at 10 M pairs: 5.4 TB/s of its job bytes, the rest is the staging through LDS).
__device__ inline u32 sw_base_codes(u32 w, u32& not_acgt, u32& other) {
	const u32 x = (w >> 1) & 0x03030303u;                               // A0 C1 T2 G3
	const u32 letter = __builtin_amdgcn_perm(0x47544341u, 0x47544341u, x);     // byte i = "ACTG"[x_i]
	const u32 acgt = sw_zero_bytes(w ^ letter);
	not_acgt = acgt ^ SW_H;
	other = 0;
	if (not_acgt) other = not_acgt & ~sw_zero_bytes(w ^ 0x4E4E4E4Eu);   // (rare)
	u32 code = __builtin_amdgcn_perm(0x03010200u, 0x03010200u, x);      // A0 T1 C2 G3
	code &= (acgt >> 7) * 3u;
	return (code * 0x40100401u) >> 24;                                  // byte i's two bits -> bits 7-2i, 6-2i
}
// four quality characters: bit 7 per byte where (uint8)(q - 33) < 20 (phred33, A2:150-152; MIN_BASE_QUALITY, A2:76,252)
__device__ inline u32 sw_low_quality(u32 w) {
	const u32 z = ((w | SW_H) - 0x21212121u) ^ ((w ^ ~0x21212121u) & SW_H);     // byte-wise w - 33
	return ~(((z & 0x7F7F7F7Fu) + 0x6C6C6C6Cu) | z) & SW_H;
}
// every bit of x between zeros: bit i -> bit 2i
__device__ inline u64 sw_spread32(u32 v) {
	u64 x = v;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}

// the reverse complement of a packed short read (rl <= 64): groups of two bits in reverse order, complemented (A0 <-> T1, C2 <-> G3: bit 0
// of the code), not-ACGT bases at code 0; the masks reversed
__device__ inline void pack_rc(u64 bhi, u64 blo, u64 nm, u64 gate, int rl, u64& rh, u64& rlo, u64& rnm, u64& rgate) {
	rh = __brevll(blo); rlo = __brevll(bhi);
	rh = ((rh >> 1) & 0x5555555555555555ull) | ((rh & 0x5555555555555555ull) << 1);
	rlo = ((rlo >> 1) & 0x5555555555555555ull) | ((rlo & 0x5555555555555555ull) << 1);
	const u32 s = 128u - 2u * (u32) rl;                       // 0 .. 126 (uniform)
	if (s >= 64u) { rlo = rh >> (s - 64u); rh = 0; }
	else if (s) { rlo = (rlo >> s) | (rh << (64u - s)); rh >>= s; }
	const u32 nb = 2u * (u32) rl;
	const u64 m_lo = nb >= 64u ? ~0ull : (1ull << nb) - 1ull, m_hi = nb > 64u ? (nb >= 128u ? ~0ull : (1ull << (nb - 64u)) - 1ull) : 0ull;
	rlo ^= 0x5555555555555555ull & m_lo;
	rh ^= 0x5555555555555555ull & m_hi;
	if (nm) {                                                 // (rare) base i of this record sits at the bits 2(rl-1-i): N of the read at i' = rl-1-i
		const u64 p_lo = sw_spread32((u32) nm), p_hi = sw_spread32((u32) (nm >> 32));
		rlo &= ~(p_lo | (p_lo << 1));
		rh &= ~(p_hi | (p_hi << 1));
	}
	rnm = __brevll(nm) >> (64u - (u32) rl);
	rgate = __brevll(gate) >> (64u - (u32) rl);
}
__device__ inline u64 shfl_down1_u64(u64 v) {
	const u32 lo = (u32) __shfl_down((int) (u32) v, 1), hi = (u32) __shfl_down((int) (u32) (v >> 32), 1);
	return ((u64) hi << 32) | lo;
}

// FWD: `ascii` holds the reads as extracted only; packed records 2i (as is) and 2i+1 (the reverse complement add_to_buffer writes
// after every read, bam_read.c:232-243: bases complemented in reverse order, qualities reversed) come out of read i, the second one
// from the PACKED first one (bit reversal, no second pass over the characters)
// WQ: the quality rows are written (false: they are read from the resident ASCII records later, nothing to copy)
template <bool FWD, bool WQ>
__global__ __launch_bounds__(PACK_RECS) void k_pool_pack(const uint8_t* __restrict__ ascii, size_t n_rec, int rl, size_t rec0,
                                                         u64* __restrict__ bases, u64* __restrict__ nmask,
                                                         u64* __restrict__ lowq, uint8_t* __restrict__ quals, int qstride,
                                                         u32* __restrict__ bad_strand) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const int reclen = 2 * rl + 1;
	const u32 tid = threadIdx.x;
	const size_t first = (size_t) blockIdx.x * PACK_RECS;
	const u32 nhere = (u32) (n_rec - first < PACK_RECS ? n_rec - first : PACK_RECS);
	const size_t bytes = (size_t) nhere * (size_t) reclen;
	const uint8_t* src = ascii + first * (size_t) reclen;       // 16-byte aligned: 256*reclen is a multiple of 16
	const size_t nvec = bytes / 16;
	typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
	for (size_t v = tid; v < nvec; v += PACK_RECS) ((u32x4_t*) lds)[v] = __builtin_nontemporal_load(&((const u32x4_t*) src)[v]);      // (read once: not kept in the caches)
	for (size_t b = nvec * 16 + tid; b < bytes; b += PACK_RECS) lds[b] = src[b];
	const u32 in_words = ((u32) PACK_RECS * (u32) reclen + 3u) / 4u + 4u;
	const u32* inw = (const u32*) lds;
	const u32 QW = (u32) qstride / 4u, OS = QW + 1u;              // quality row in LDS: an odd stride keeps the banks apart
	u32* orow = (u32*) lds + ((in_words + 3u) & ~3u);
	__syncthreads();
	u64 s_bhi = 0, s_blo = 0, s_nm = 0, s_gate = 0;          // (this thread's packed record, for the symmetry check behind the branch)
	if (tid < nhere) {
		const u32 start = tid * (u32) reclen;
		if (lds[start] != '0') atomicAdd(bad_strand, 1u);   // A2:383-391; the reference only ever writes '0' (bam_read.c:219,231)
		const u32 nw = ((u32) rl + 3u) / 4u;
		// every LDS word of the record is asked for before the first is used (unrolled to the longest record, uniform guards): the
		// loop with one read per trip waited for LDS twenty-six times per record
		constexpr int NWMAX = (VDJX_SHORT_READ_LEN + 3) / 4;
		u32 rb[NWMAX + 1], rq[NWMAX + 1];
		const u32 sb = start + 1u, sq = start + 1u + (u32) rl;
#pragma unroll
		for (int j = 0; j <= NWMAX; j++) {
			rb[j] = (u32) j <= nw ? inw[(sb >> 2) + j] : 0u;
			rq[j] = (u32) j <= nw ? inw[(sq >> 2) + j] : 0u;
		}
		// ---- bases
		u64 bhi = 0, blo = 0, nm = 0, lq = 0;
		u32 oth = 0;
		u32* my = orow + tid * OS;
		u32 qrow[NWMAX];                                  // (!FWD: the quality row stays in registers)
#pragma unroll
		for (int j = 0; j < NWMAX; j++) qrow[j] = 0x21212121u;
#pragma unroll
		for (int j = 0; j < NWMAX; j++) {
			if ((u32) j < nw) {
				const u32 valid = (u32) rl - 4u * j < 4u ? (u32) rl - 4u * j : 4u;
				const u32 vm = valid < 4u ? (1u << (8u * valid)) - 1u : 0xFFFFFFFFu;
				u32 w = __builtin_amdgcn_alignbyte(rb[j + 1], rb[j], sb & 3u);
				w = (w & vm) | (0x41414141u & ~vm);
				u32 na, ot;
				const u32 c8 = sw_base_codes(w, na, ot) >> (8u - 2u * valid);
				bhi = (bhi << (2u * valid)) | (blo >> (64u - 2u * valid));
				blo = (blo << (2u * valid)) | c8;
				nm |= (u64) sw_gather4(na) << (4u * j);
				oth += (u32) __popc(ot);
				// ---- qualities
				u32 q = __builtin_amdgcn_alignbyte(rq[j + 1], rq[j], sq & 3u);
				q = (q & vm) | (0x21212121u & ~vm);
				lq |= (u64) (sw_gather4(sw_low_quality(q)) & ((1u << valid) - 1u)) << (4u * j);
				if (FWD) my[j] = q; else if (WQ) qrow[j] = q;
			}
		}
		if (FWD) for (u32 j = nw; j < QW; j++) my[j] = 0x21212121u;
		const size_t g = FWD ? rec0 + 2 * (first + tid) : rec0 + first + tid;
		((ulonglong2*) bases)[g] = make_ulonglong2(bhi, blo);
		nmask[g] = nm;
		lowq[g] = lq | nm;                                            // (see vdjx_pool::d_lowq: the gate mask)
		if (!FWD && WQ) {
			// the row straight from the registers: four 16-byte stores per record (the staging through LDS of the FWD variant costs
			// 17 KB per workgroup, i.e. half the workgroups per CU and half the bytes in flight)
			uint4* qd = (uint4*) (quals + g * (size_t) qstride);
#pragma unroll
			for (int v = 0; v < NWMAX / 4; v++) if (v < qstride / 16) qd[v] = make_uint4(qrow[4 * v], qrow[4 * v + 1], qrow[4 * v + 2], qrow[4 * v + 3]);
		}
		if (FWD) {
			// the reverse complement from the packed read (pack_rc)
			u64 rh, rlo, rnm, rgate;
			pack_rc(bhi, blo, nm, lq | nm, rl, rh, rlo, rnm, rgate);
			((ulonglong2*) bases)[g + 1] = make_ulonglong2(rh, rlo);
			nmask[g + 1] = rnm;
			lowq[g + 1] = rgate;
			oth *= 2u;
		}
		if (oth) atomicAdd(bad_strand + 1, oth);     // (rare: reported through vdjx_stat("pool_other_bases"))
		s_bhi = bhi; s_blo = blo; s_nm = nm; s_gate = lq | nm;
	}
	if (!FWD) {
		// is every odd record the reverse complement of the record before it, masks mirrored (vdjx_pool::sym: what add_to_buffer writes,
		// bam_read.c:206-244)?  The even lane holds record 2i and takes record 2i+1's packed words from its neighbour.  bad_strand[2]
		// counts the couples that are not (and the records without a partner).
		u64 rh, rlo, rnm, rgate;
		pack_rc(s_bhi, s_blo, s_nm, s_gate, rl, rh, rlo, rnm, rgate);
		const u64 p_hi = shfl_down1_u64(s_bhi), p_lo = shfl_down1_u64(s_blo), p_nm = shfl_down1_u64(s_nm), p_gate = shfl_down1_u64(s_gate);
		if (tid < nhere && !(tid & 1u)) {
			const bool ok = ((rec0 + first) & 1) == 0 && tid + 1u < nhere && rh == p_hi && rlo == p_lo && rnm == p_nm && rgate == p_gate;
			if (!ok) atomicAdd(bad_strand + 2, 1u);
		}
		return;
	}
	__syncthreads();
	// ---- quality rows out: 16 bytes per lane, consecutive lanes on consecutive addresses
	const u32 rows = FWD ? 2u * nhere : nhere;
	const u32 V = QW / 4u;                                                // 16-byte pieces per row
	uint4* qd = (uint4*) (quals + (FWD ? rec0 + 2 * first : rec0 + first) * (size_t) qstride);
	for (u32 idx = tid; idx < rows * V; idx += PACK_RECS) {
		const u32 row = idx / V, v = idx - row * V;
		const u32* my = orow + (FWD ? row >> 1 : row) * OS;
		u32 x[4];
		if (!FWD || !(row & 1u)) {
#pragma unroll
			for (int q = 0; q < 4; q++) x[q] = my[4u * v + q];
		} else {
			// reversed qualities: byte i of the row = byte rl-1-i of the read's row
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int i0 = (int) (16u * v + 4u * q);                     // output bytes i0 .. i0+3 <- input bytes rl-1-i0 .. rl-4-i0
				const int top = rl - 1 - i0;                                 // input position of output byte i0 (may be negative: padding)
				u32 w = 0x21212121u;
				if (top >= 0) {
					const int lo = top - 3;                                  // lowest input byte needed (may be negative)
					const u32 wl = lo >= 0 ? my[lo >> 2] : 0u, wh = my[top >> 2];
					u32 asc;                                                 // input bytes lo .. lo+3 in ascending order
					if (lo >= 0) asc = __builtin_amdgcn_alignbyte(wh, wl, (u32) lo & 3u);
					else asc = wh << (8u * (u32) (-lo));                     // (top < 3: bytes below position 0 do not exist)
					w = __builtin_bswap32(asc);
					if (lo < 0) { const u32 keep = (1u << (8u * (u32) (top + 1))) - 1u; w = (w & keep) | (0x21212121u & ~keep); }
				}
				x[q] = w;
			}
		}
		qd[idx] = make_uint4(x[0], x[1], x[2], x[3]);
	}
}

// ---- packed host format (vdjx_pool_load_packed): a read is VDJX_PACKED_BYTES(rl) bytes -- ceil(rl/4) bytes of 2-bit bases (A0 T1 C2
// G3, seq_to_kmer.c:6-29; the first base in the top bits of byte 0, four per byte, code 0 where the base is not ACGT), then rl quality
// bytes (Phred+33; bit 7 set = the base is not ACGT), zero-padded to a multiple of 16: 64 bytes for 50 bp where the extracted ASCII
// record has 101.  Record 2i = the read, record 2i+1 = its reverse complement with reversed qualities (bam_read.c:231-243), exactly
// what the forward load makes of the ASCII.  One thread per read, its bytes as 16-byte loads straight from the (device) buffer.
__host__ __device__ inline unsigned vdjx_packed_bytes(int rl) { return (unsigned) (((rl + 3) / 4 + rl + 15) / 16 * 16); }
template <int NV>         // 16-byte pieces per read: vdjx_packed_bytes(rl) / 16 = 1 .. 5 (4 for reads of 39 .. 51 bases)
__global__ __launch_bounds__(PACK_RECS) void k_pool_unpack(const uint8_t* __restrict__ packed, size_t n_reads, int rl, size_t rec0,
                                                           u64* __restrict__ bases, u64* __restrict__ nmask, u64* __restrict__ lowq,
                                                           uint8_t* __restrict__ quals, int qstride, u32* __restrict__ bad_strand) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const u32 tid = threadIdx.x;
	const size_t first = (size_t) blockIdx.x * PACK_RECS;
	const u32 nhere = (u32) (n_reads - first < PACK_RECS ? n_reads - first : PACK_RECS);
	const u32 QW = (u32) qstride / 4u, OS = QW + 1u;
	u32* orow = (u32*) lds;
	typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
	if (tid < nhere) {
		u32 w[NV * 4 + 1];
		const u32x4_t* src = (const u32x4_t*) (packed + (first + tid) * (size_t) (NV * 16));
#pragma unroll
		for (int v = 0; v < NV; v++) { const u32x4_t x = __builtin_nontemporal_load(&src[v]); w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w; }
		w[NV * 4] = 0;
		const u32 nb = ((u32) rl + 3u) / 4u;                              // base bytes
		// bases: the bytes are the codes already, first base first: as a big-endian number of 8 nb bits, down by the 2 (4 nb - rl) unused ones
		u64 hi = 0, lo = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const u32 be = __builtin_bswap32(w[j]);                       // bytes 4j .. 4j+3, the first most significant
			if ((u32) (4 * j) < nb) {
				const u32 have = nb - 4u * j < 4u ? nb - 4u * j : 4u;      // bytes of this word that are base bytes
				const u32 val = have < 4u ? be >> (8u * (4u - have)) : be;
				hi = (hi << (8u * have)) | (lo >> (64u - 8u * have));
				lo = (lo << (8u * have)) | val;
			}
		}
		const u32 drop = 2u * (4u * nb - (u32) rl);                      // 0, 2, 4 or 6
		if (drop) { lo = (lo >> drop) | (hi << (64u - drop)); hi >>= drop; }
		// qualities: rl bytes from byte nb on
		u64 nm = 0, lq = 0;
		u32* my = orow + tid * OS;
		const u32 nw = ((u32) rl + 3u) / 4u;
#pragma unroll
		for (int j = 0; j < (VDJX_SHORT_READ_LEN + 3) / 4; j++) {
			if ((u32) j < nw) {
				const u32 at = nb + 4u * j;                                // byte offset of this word of qualities
				const u32 valid = (u32) rl - 4u * j < 4u ? (u32) rl - 4u * j : 4u;
				const u32 vm = valid < 4u ? (1u << (8u * valid)) - 1u : 0xFFFFFFFFu;
				u32 q = __builtin_amdgcn_alignbyte(w[(at >> 2) + 1], w[at >> 2], at & 3u);
				q = (q & vm) | (0x21212121u & ~vm);
				nm |= (u64) (sw_gather4(q & SW_H) & ((1u << valid) - 1u)) << (4u * j);
				q &= 0x7F7F7F7Fu;
				lq |= (u64) (sw_gather4(sw_low_quality(q)) & ((1u << valid) - 1u)) << (4u * j);
				my[j] = q;
			}
		}
		for (u32 j = nw; j < QW; j++) my[j] = 0x21212121u;
		if (nm) {                                                        // (rare) not-ACGT bases carry code 0, whatever the packer wrote
			const u64 p_lo = sw_spread32((u32) __brevll(nm << (64u - (u32) rl))), p_hi = sw_spread32((u32) (__brevll(nm << (64u - (u32) rl)) >> 32));
			lo &= ~(p_lo | (p_lo << 1));
			hi &= ~(p_hi | (p_hi << 1));
		}
		const size_t g = rec0 + 2 * (first + tid);
		((ulonglong2*) bases)[g] = make_ulonglong2(hi, lo);
		nmask[g] = nm;
		lowq[g] = lq | nm;
		u64 rh, rlo, rnm, rgate;
		pack_rc(hi, lo, nm, lq | nm, rl, rh, rlo, rnm, rgate);
		((ulonglong2*) bases)[g + 1] = make_ulonglong2(rh, rlo);
		nmask[g + 1] = rnm;
		lowq[g + 1] = rgate;
	}
	(void) bad_strand;
	__syncthreads();
	// ---- quality rows out, as in k_pool_pack<FWD>: 16 bytes per lane, consecutive lanes on consecutive addresses
	const u32 rows = 2u * nhere;
	const u32 V = QW / 4u;
	uint4* qd = (uint4*) (quals + (rec0 + 2 * first) * (size_t) qstride);
	for (u32 idx = tid; idx < rows * V; idx += PACK_RECS) {
		const u32 row = idx / V, v = idx - row * V;
		const u32* my = orow + (row >> 1) * OS;
		u32 x[4];
		if (!(row & 1u)) {
#pragma unroll
			for (int q = 0; q < 4; q++) x[q] = my[4u * v + q];
		} else {
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int i0 = (int) (16u * v + 4u * q);
				const int top = rl - 1 - i0;
				u32 w = 0x21212121u;
				if (top >= 0) {
					const int lo_ = top - 3;
					const u32 wl = lo_ >= 0 ? my[lo_ >> 2] : 0u, wh = my[top >> 2];
					u32 asc;
					if (lo_ >= 0) asc = __builtin_amdgcn_alignbyte(wh, wl, (u32) lo_ & 3u);
					else asc = wh << (8u * (u32) (-lo_));
					w = __builtin_bswap32(asc);
					if (lo_ < 0) { const u32 keep = (1u << (8u * (u32) (top + 1))) - 1u; w = (w & keep) | (0x21212121u & ~keep); }
				}
				x[q] = w;
			}
		}
		qd[idx] = make_uint4(x[0], x[1], x[2], x[3]);
	}
}

// reads of more than 64 bases (left-aligned words, vdjx_pool): one thread per record, character by character from the LDS-staged
// records -- the plain form of the packing; long reads are the rare case and take the simple kernel
#define PACKL_RECS 128
template <bool FWD, bool WQ>
__global__ __launch_bounds__(PACKL_RECS) void k_pool_pack_long(const uint8_t* __restrict__ ascii, size_t n_rec, int rl, size_t rec0, int W, int M,
                                                                u64* __restrict__ bases, u64* __restrict__ nmask, u64* __restrict__ lowq,
                                                                uint8_t* __restrict__ quals, int qstride, u32* __restrict__ bad_strand) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const int reclen = 2 * rl + 1;
	const u32 tid = threadIdx.x;
	const size_t first = (size_t) blockIdx.x * PACKL_RECS;
	const u32 nhere = (u32) (n_rec - first < PACKL_RECS ? n_rec - first : PACKL_RECS);
	const size_t bytes = (size_t) nhere * (size_t) reclen;
	const uint8_t* src = ascii + first * (size_t) reclen;       // 16-byte aligned: 128*reclen is a multiple of 16
	const size_t nvec = bytes / 16;
	for (size_t v = tid; v < nvec; v += PACKL_RECS) ((uint4*) lds)[v] = ((const uint4*) src)[v];
	for (size_t b = nvec * 16 + tid; b < bytes; b += PACKL_RECS) lds[b] = src[b];
	__syncthreads();
	if (tid >= nhere) return;
	const uint8_t* r = lds + (size_t) tid * reclen;
	if (r[0] != '0') atomicAdd(bad_strand, 1u);
	u32 other = 0;
	if constexpr (!FWD && !WQ) {
		// resident records (vdjx_pool_load_device: the one long-read load that is not waiting for PCIe): four characters at a time, like
		// the short reads' kernel (2.5 -> 1.1 ms at 5 M pairs of 100 bases)
		const size_t g = rec0 + first + tid;
		const u32* inw = (const u32*) lds;
		const u32 sb = tid * (u32) reclen + 1u, sq = sb + (u32) rl;
		const int ngroups = (rl + 3) / 4;
		u64 acc = 0, nm = 0, lq = 0;
		u32 blo = inw[sb >> 2], qlo = inw[sq >> 2];
		for (int j = 0; j < ngroups; j++) {
			const u32 valid = (u32) rl - 4u * (u32) j < 4u ? (u32) rl - 4u * (u32) j : 4u;
			const u32 vm = valid < 4u ? (1u << (8u * valid)) - 1u : 0xFFFFFFFFu;
			const u32 bhi = inw[(sb >> 2) + j + 1], qhi = inw[(sq >> 2) + j + 1];
			u32 w = __builtin_amdgcn_alignbyte(bhi, blo, sb & 3u);
			u32 q = __builtin_amdgcn_alignbyte(qhi, qlo, sq & 3u);
			blo = bhi; qlo = qhi;
			w = (w & vm) | (0x41414141u & ~vm);
			q = (q & vm) | (0x21212121u & ~vm);
			u32 na, ot;
			acc = (acc << 8) | sw_base_codes(w, na, ot);
			nm |= (u64) sw_gather4(na) << ((4 * j) & 63);
			lq |= (u64) (sw_gather4(sw_low_quality(q)) & ((1u << valid) - 1u)) << ((4 * j) & 63);
			other += (u32) __popc(ot);
			if ((j & 7) == 7 || j == ngroups - 1) { bases[g * (size_t) W + (j >> 3)] = acc << (64 - 8 * ((j & 7) + 1)); acc = 0; }
			if ((j & 15) == 15 || j == ngroups - 1) { nmask[g * (size_t) M + (j >> 4)] = nm; lowq[g * (size_t) M + (j >> 4)] = lq | nm; nm = 0; lq = 0; }
		}
		for (int w = (rl + 31) / 32; w < W; w++) bases[g * (size_t) W + w] = 0;
		for (int w = (rl + 63) / 64; w < M; w++) { nmask[g * (size_t) M + w] = 0; lowq[g * (size_t) M + w] = 0; }
		if (other) atomicAdd(bad_strand + 1, other);
		return;
	}
	for (int rev = 0; rev < (FWD ? 2 : 1); rev++) {
		const size_t g = FWD ? rec0 + 2 * (first + tid) + (size_t) rev : rec0 + first + tid;
		u64 acc = 0, nm = 0, lq = 0;
		for (int i = 0; i < rl; i++) {
			const int si = rev ? rl - 1 - i : i;
			const u32 ch = r[1 + si];
			u32 code = (0xD8u >> (((ch >> 1) & 3u) * 2u)) & 3u;                  // seq_to_kmer.c:6-29: A0 T1 C2 G3
			if (rev) code ^= 1u;                                                // complement: A<->T, C<->G
			const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
			other += (!acgt && ch != 'N') ? 1u : 0u;
			acc = (acc << 2) | (acgt ? code : 0u);
			nm |= (u64) (!acgt) << (i & 63);
			lq |= (u64) ((u32) (uint8_t) (r[1 + rl + si] - 33) < 20u) << (i & 63);
			if ((i & 31) == 31 || i == rl - 1) { bases[g * (size_t) W + (i >> 5)] = acc << (2 * (31 - (i & 31))); acc = 0; }
			if ((i & 63) == 63 || i == rl - 1) { nmask[g * (size_t) M + (i >> 6)] = nm; lowq[g * (size_t) M + (i >> 6)] = lq | nm; nm = 0; lq = 0; }
		}
		for (int w = (rl + 31) / 32; w < W; w++) bases[g * (size_t) W + w] = 0;
		for (int w = (rl + 63) / 64; w < M; w++) { nmask[g * (size_t) M + w] = 0; lowq[g * (size_t) M + w] = 0; }
		if (WQ) {
			uint8_t* q = quals + g * (size_t) qstride;
			for (int i = 0; i < qstride; i++) q[i] = i < rl ? r[1 + rl + (rev ? rl - 1 - i : i)] : (uint8_t) 33;
		}
	}
	if (other) atomicAdd(bad_strand + 1, other);
}

static int pool_alloc(vdjx_ctx* c, hipStream_t pack_stream, size_t R, size_t n_primary, int rl, vdjx_pool** out, u32** d_bad, bool external_quals = false) {
	*out = nullptr;
	if (rl < 1 || rl > VDJX_MAX_READ_LEN) { vdjx_set_error("read length %d outside [1,%d]", rl, VDJX_MAX_READ_LEN); return VDJX_ELIMIT; }
	if (R >= (1ull << 31)) { vdjx_set_error("too many records for one GPU: %zu", R); return VDJX_ELIMIT; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_pool* p = new vdjx_pool();
	p->ctx = c; p->device = c->device; p->n_primary = n_primary; p->n_records = R; p->rl = rl;
	if (rl > VDJX_SHORT_READ_LEN) { p->W = VDJX_LONG_W; p->M = VDJX_LONG_M; p->ob = 8; }
	vdjx_clear_errors();
	p->qstride = (rl + 15) / 16 * 16;
	size_t Ra = (R ? R : 1);
	Ra = (Ra + 15) & ~(size_t) 15;               // keeps every array 256-byte aligned inside the block
	const size_t qbytes = external_quals ? 0 : (size_t) p->qstride;
	const size_t wb = (size_t) p->W * 8, mb = (size_t) p->M * 8;
	// (+64: the long-read kernels read up to two words past a record's bases)
	hipError_t e = c->blocks.acquire(Ra * (wb + 2 * mb + qbytes) + 256 + 64, &p->d_block, &p->block_cap);
	if (e != hipSuccess) {
		vdjx_set_error("pool alloc: %s", hipGetErrorString(e));
		vdjx_pool_free(p);
		return VDJX_EHIP;
	}
	p->d_nmask = (u64*) p->d_block;
	p->d_lowq = (u64*) (p->d_block + Ra * mb);
	p->d_quals = p->d_quals2 = (const uint8_t*) (p->d_block + Ra * 2 * mb);
	*d_bad = (u32*) (p->d_block + Ra * (2 * mb + qbytes));
	p->d_bases = (u64*) (p->d_block + Ra * (2 * mb + qbytes) + 256);
	// [0] bad strand bytes, [1] other bases: cleared on the stream the packing will run on.  (A synchronous hipMemset here joined
	// every stream of the device: a step's first call waited for the previous step's mapped pairs to finish their 4 ms trip over
	// PCIe on the copy stream -- 1.3 ms per step at 10 M pairs, found in the HIP API trace as one 2.6 ms hipMemset every other step.)
	{
		const hipError_t em = hipMemsetAsync(*d_bad, 0, 16, pack_stream);
		if (em != hipSuccess) { vdjx_set_error("pool alloc: %s", hipGetErrorString(em)); vdjx_pool_free(p); return VDJX_EHIP; }
	}
	*out = p;
	return VDJX_OK;
}

static void pack_launch(vdjx_ctx* c, hipStream_t st, vdjx_pool* p, const uint8_t* d_ascii, size_t n, size_t rec0, bool fwd, u32* d_bad, bool write_quals = true, bool packed_in = false) {
	if (!n) return;
	uint8_t* q = (uint8_t*) p->d_quals;        // (the packed rows, when there are any)
	if (packed_in) {                           // the packed host format (k_pool_unpack): n reads -> 2 n records
		const size_t lds = (size_t) PACK_RECS * (p->qstride / 4 + 1) * 4 + 16;
		const dim3 grid((unsigned) ((n + PACK_RECS - 1) / PACK_RECS));
#define UNPACK(NV) hipLaunchKernelGGL(k_pool_unpack<NV>, grid, dim3(PACK_RECS), lds, st, d_ascii, n, p->rl, rec0, p->d_bases, p->d_nmask, p->d_lowq, q, p->qstride, d_bad)
		switch (vdjx_packed_bytes(p->rl) / 16) {          // 16-byte pieces per read: 4 for reads of 39..51 bases
			case 1: UNPACK(1); break;
			case 2: UNPACK(2); break;
			case 3: UNPACK(3); break;
			case 4: UNPACK(4); break;
			default: UNPACK(5); break;
		}
#undef UNPACK
		return;
	}
	if (p->W > 2) {
		const size_t lds = (size_t) PACKL_RECS * (2 * p->rl + 1) + 16;
		const dim3 grid((unsigned) ((n + PACKL_RECS - 1) / PACKL_RECS));
#define PL(F, Q) hipLaunchKernelGGL((k_pool_pack_long<F, Q>), grid, dim3(PACKL_RECS), lds, st, d_ascii, n, p->rl, rec0, p->W, p->M, p->d_bases, p->d_nmask, p->d_lowq, q, p->qstride, d_bad)
		if (fwd) PL(true, true); else if (write_quals) PL(false, true); else PL(false, false);
#undef PL
		return;
	}
	const size_t in_words = ((size_t) PACK_RECS * (2 * p->rl + 1) + 3) / 4 + 4;
	const size_t lds = (((in_words + 3) & ~(size_t) 3) + (fwd ? (size_t) PACK_RECS * (p->qstride / 4 + 1) : 0)) * 4 + 16;
	const dim3 grid((unsigned) ((n + PACK_RECS - 1) / PACK_RECS));
	if (fwd) hipLaunchKernelGGL((k_pool_pack<true, true>), grid, dim3(PACK_RECS), lds, st, d_ascii, n, p->rl, rec0, p->d_bases, p->d_nmask, p->d_lowq, q, p->qstride, d_bad);
	else if (write_quals) hipLaunchKernelGGL((k_pool_pack<false, true>), grid, dim3(PACK_RECS), lds, st, d_ascii, n, p->rl, rec0, p->d_bases, p->d_nmask, p->d_lowq, q, p->qstride, d_bad);
	else hipLaunchKernelGGL((k_pool_pack<false, false>), grid, dim3(PACK_RECS), lds, st, d_ascii, n, p->rl, rec0, p->d_bases, p->d_nmask, p->d_lowq, q, p->qstride, d_bad);
}

static bool sym_wanted() { static const bool on = getenv("VDJX_NO_SYM") == nullptr; return on; }

static int pool_finish(vdjx_ctx* c, vdjx_pool* p, u32* d_bad, vdjx_pool** out, bool fwd = false) {
	u32* both = (u32*) c->h_pin;
	both[0] = both[1] = both[2] = 0;
	hipError_t e = hipMemcpyAsync(both, d_bad, 12, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e == hipSuccess) e = hipGetLastError();
	if (e != hipSuccess) { vdjx_set_error("pool pack: %s", hipGetErrorString(e)); vdjx_pool_free(p); return VDJX_EHIP; }
	const u32 bad = both[0];
	// IUPAC codes other than N are packed as N (the reference keeps them inside its k-mer strings and exits in seq_to_int,
	// seq_to_kmer.c:22-24, as soon as one reaches a node): the count lets a caller see the deviation
	c->stats["pool_other_bases"] = both[1];
	// couples (record, its reverse complement): made by the forward load, verified by the packing otherwise (short reads)
	p->sym = sym_wanted() && p->W == 2 && p->n_records % 2 == 0 && p->n_primary % 2 == 0 && (fwd || both[2] == 0);
	c->stats["pool_symmetric"] = p->sym ? 1 : 0;
	if (bad) {
		// build_pre_graph prints "Initial char in input invalid" and exits (A2:383-391); we return an error
		vdjx_set_error("pool: %u records do not start with the '0' strand byte", bad);
		vdjx_pool_free(p);
		return VDJX_EINVAL;
	}
	*out = p;
	return VDJX_OK;
}

static int pool_pack_device(vdjx_ctx* c, const uint8_t* d_primary, size_t n_primary,
                            const uint8_t* d_secondary, size_t n_secondary, int rl, vdjx_pool** out) {
	vdjx_pool* p;
	u32* d_bad;
	// the quality characters stay where they are: the records are resident, and only the few low-count k-mers ever look at a row
	int rc = pool_alloc(c, c->stream, n_primary + n_secondary, n_primary, rl, &p, &d_bad, true);
	if (rc) return rc;
	*out = nullptr;
	p->qstride = 2 * rl + 1;
	p->d_quals = d_primary + 1 + rl;
	p->d_quals2 = d_secondary + 1 + rl;
	p->q_split = n_primary;
	{
		vdjx_prof_scope ps(c, "k_pool_pack");
		pack_launch(c, c->stream, p, d_primary, n_primary, 0, false, d_bad, false);
		pack_launch(c, c->stream, p, d_secondary, n_secondary, n_primary, false, d_bad, false);
	}
	return pool_finish(c, p, d_bad, out);
}

extern "C" int vdjx_pool_load_device(vdjx_ctx* c, const uint8_t* d_primary, size_t n_primary,
                                     const uint8_t* d_secondary, size_t n_secondary, int rl, vdjx_pool** out) {
	if (!c || !out) { vdjx_set_error("vdjx_pool_load_device: NULL argument"); return VDJX_EINVAL; }
	if ((n_primary && !d_primary) || (n_secondary && !d_secondary)) { vdjx_set_error("vdjx_pool_load_device: NULL pool"); return VDJX_EINVAL; }
	if (((uintptr_t) d_primary & 15) || ((uintptr_t) d_secondary & 15)) { vdjx_set_error("device pools must be 16-byte aligned"); return VDJX_EINVAL; }
	return pool_pack_device(c, d_primary, n_primary, d_secondary, n_secondary, rl, out);
}

// Host pools cross PCIe in chunks on the copy stream into two staging buffers while the previous chunk is packed on the main
// stream: with page-locked pools (vdjx_host_alloc) the upload is a DMA that the packing hides behind.
#define LOAD_CHUNK_RECS (256u * 1024u)          // a multiple of PACK_RECS: chunk starts stay 16-byte aligned in the staging buffer
static int pool_load_host(vdjx_ctx* c, const uint8_t* primary, size_t n_primary, const uint8_t* secondary, size_t n_secondary, int rl, bool fwd,
                          vdjx_pool** out, bool async = false, bool packed_in = false) {
	if (!c || !out) { vdjx_set_error("vdjx_pool_load: NULL argument"); return VDJX_EINVAL; }
	if ((n_primary && !primary) || (n_secondary && !secondary)) { vdjx_set_error("vdjx_pool_load: NULL pool"); return VDJX_EINVAL; }
	if (packed_in && (rl < 1 || rl > VDJX_SHORT_READ_LEN)) { vdjx_set_error("vdjx_pool_load_packed: reads of up to %d bases (rl=%d): longer ones go through vdjx_pool_load_forward", VDJX_SHORT_READ_LEN, rl); return VDJX_ELIMIT; }
	const size_t mul = fwd ? 2 : 1;
	vdjx_pool* p;
	u32* d_bad;
	int rc = pool_alloc(c, async ? c->copy_stream : c->stream, mul * (n_primary + n_secondary), mul * n_primary, rl, &p, &d_bad);
	if (rc) return rc;
	*out = nullptr;
	const size_t reclen = packed_in ? (size_t) vdjx_packed_bytes(rl) : 2 * (size_t) rl + 1;      // bytes per input record
	const size_t stage_bytes = (size_t) LOAD_CHUNK_RECS * (2 * (size_t) rl + 1) + 16;           // (sized for the ASCII form: the larger one)
	if (stage_bytes > c->stage_cap) {              // (sized for the longest records seen so far)
		(void) hipStreamSynchronize(c->copy_stream);
		(void) hipStreamSynchronize(c->stream);
		for (int i = 0; i < 2; i++) {
			if (c->d_stage[i]) (void) hipFree(c->d_stage[i]);
			c->d_stage[i] = nullptr;
			hipError_t e = hipMalloc(&c->d_stage[i], stage_bytes);
			if (e == hipSuccess && !c->ev_copied[i]) e = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming);
			if (e == hipSuccess && !c->ev_packed[i]) e = hipEventCreateWithFlags(&c->ev_packed[i], hipEventDisableTiming);
			if (e != hipSuccess) { c->stage_cap = 0; vdjx_set_error("pool staging: %s", hipGetErrorString(e)); vdjx_pool_free(p); return VDJX_EHIP; }
		}
		c->stage_cap = stage_bytes;
	}
	if (async) {
		// the whole load on the copy stream (chunk after chunk: copy, pack), so that the main stream keeps computing on another pool;
		// vdjx_pool_wait joins it
		for (int which = 0; which < 2; which++) {
			const uint8_t* src = which ? secondary : primary;
			const size_t n = which ? n_secondary : n_primary;
			const size_t rec0 = which ? mul * n_primary : 0;
			int turn = 0;
			for (size_t at = 0; at < n; at += LOAD_CHUNK_RECS, turn ^= 1) {
				const size_t m = n - at < LOAD_CHUNK_RECS ? n - at : LOAD_CHUNK_RECS;
				const hipError_t e = hipMemcpyAsync(c->d_stage[turn], src + at * reclen, m * reclen, hipMemcpyHostToDevice, c->copy_stream);
				if (e != hipSuccess) { vdjx_set_error("pool upload: %s", hipGetErrorString(e)); (void) hipStreamSynchronize(c->copy_stream); vdjx_pool_free(p); return VDJX_EHIP; }
				pack_launch(c, c->copy_stream, p, (const uint8_t*) c->d_stage[turn], m, rec0 + mul * at, fwd, d_bad, true, packed_in);
			}
		}
		p->pending_bad = d_bad;
		*out = p;
		return VDJX_OK;
	}
	int turn = 0;
	bool used[2] = {false, false};
	vdjx_prof_scope ps(c, "k_pool_pack");
	for (int which = 0; which < 2; which++) {
		const uint8_t* src = which ? secondary : primary;
		const size_t n = which ? n_secondary : n_primary;
		const size_t rec0 = which ? mul * n_primary : 0;
		for (size_t at = 0; at < n; at += LOAD_CHUNK_RECS, turn ^= 1) {
			const size_t m = n - at < LOAD_CHUNK_RECS ? n - at : LOAD_CHUNK_RECS;
			hipError_t e = hipSuccess;
			if (used[turn]) e = hipStreamWaitEvent(c->copy_stream, c->ev_packed[turn], 0);           // the buffer's previous chunk is packed
			if (e == hipSuccess) e = hipMemcpyAsync(c->d_stage[turn], src + at * reclen, m * reclen, hipMemcpyHostToDevice, c->copy_stream);
			if (e == hipSuccess) e = hipEventRecord(c->ev_copied[turn], c->copy_stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_copied[turn], 0);
			if (e != hipSuccess) { vdjx_set_error("pool upload: %s", hipGetErrorString(e)); (void) hipStreamSynchronize(c->copy_stream); vdjx_pool_free(p); return VDJX_EHIP; }
			pack_launch(c, c->stream, p, (const uint8_t*) c->d_stage[turn], m, rec0 + mul * at, fwd, d_bad, true, packed_in);
			(void) hipEventRecord(c->ev_packed[turn], c->stream);
			used[turn] = true;
		}
	}
	return pool_finish(c, p, d_bad, out, fwd);
}

extern "C" int vdjx_pool_load(vdjx_ctx* c, const uint8_t* primary, size_t n_primary,
                              const uint8_t* secondary, size_t n_secondary, int rl, vdjx_pool** out) {
	return pool_load_host(c, primary, n_primary, secondary, n_secondary, rl, false, out);
}

extern "C" int vdjx_pool_load_forward(vdjx_ctx* c, const uint8_t* primary_reads, size_t n_primary_reads,
                                      const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out) {
	return pool_load_host(c, primary_reads, n_primary_reads, secondary_reads, n_secondary_reads, rl, true, out);
}

extern "C" int vdjx_pool_load_forward_begin(vdjx_ctx* c, const uint8_t* primary_reads, size_t n_primary_reads,
                                            const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out) {
	return pool_load_host(c, primary_reads, n_primary_reads, secondary_reads, n_secondary_reads, rl, true, out, true);
}

// the reads in the packed host format (k_pool_unpack): 64 bytes per 50 bp read instead of the 101 of the extracted ASCII record
extern "C" size_t vdjx_packed_read_bytes(int rl) { return rl >= 1 && rl <= VDJX_SHORT_READ_LEN ? (size_t) vdjx_packed_bytes(rl) : 0; }
extern "C" int vdjx_pool_load_packed(vdjx_ctx* c, const uint8_t* primary_reads, size_t n_primary_reads,
                                     const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out) {
	return pool_load_host(c, primary_reads, n_primary_reads, secondary_reads, n_secondary_reads, rl, true, out, false, true);
}
extern "C" int vdjx_pool_load_packed_begin(vdjx_ctx* c, const uint8_t* primary_reads, size_t n_primary_reads,
                                           const uint8_t* secondary_reads, size_t n_secondary_reads, int rl, vdjx_pool** out) {
	return pool_load_host(c, primary_reads, n_primary_reads, secondary_reads, n_secondary_reads, rl, true, out, true, true);
}
// host side of the format: n ASCII records of 2*rl+1 bytes ('0' + bases + Phred+33 characters, what add_to_buffer writes first for a
// read, bam_read.c:219-230) -> n packed reads.  Plain C, no GPU: what an extraction that wants to skip the ASCII stage would do per read.
extern "C" int vdjx_pack_reads(const uint8_t* ascii_reads, size_t n, int rl, uint8_t* out_packed) {
	if (rl < 1 || rl > VDJX_SHORT_READ_LEN) { vdjx_set_error("vdjx_pack_reads: rl=%d outside [1,%d]", rl, VDJX_SHORT_READ_LEN); return VDJX_ELIMIT; }
	if (n && (!ascii_reads || !out_packed)) { vdjx_set_error("vdjx_pack_reads: NULL argument"); return VDJX_EINVAL; }
	const size_t S = vdjx_packed_bytes(rl), reclen = 2 * (size_t) rl + 1, nb = ((size_t) rl + 3) / 4;
	for (size_t i = 0; i < n; i++) {
		const uint8_t* r = ascii_reads + i * reclen;
		uint8_t* o = out_packed + i * S;
		memset(o, 0, S);
		if (r[0] != '0') { vdjx_set_error("vdjx_pack_reads: record %zu does not start with the '0' strand byte", i); return VDJX_EINVAL; }
		for (int j = 0; j < rl; j++) {
			const uint8_t ch = r[1 + j];
			const int code = ch == 'A' ? 0 : ch == 'T' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : -1;      // seq_to_kmer.c:6-29
			if (code > 0) o[j >> 2] |= (uint8_t) (code << (2 * (3 - (j & 3))));
			o[nb + j] = (uint8_t) ((r[1 + rl + j] & 0x7F) | (code < 0 ? 0x80 : 0));
		}
	}
	return VDJX_OK;
}

extern "C" int vdjx_pool_wait(vdjx_pool* p) {
	if (!p) { vdjx_set_error("vdjx_pool_wait: NULL pool"); return VDJX_EINVAL; }
	if (!p->pending_bad) return VDJX_OK;
	if (!vdjx_ctx_alive(p->ctx)) { vdjx_set_error("vdjx_pool_wait: the pool's context is gone"); return VDJX_ESTATE; }
	HIP_TRY(hipSetDevice(p->device));
	u32 both[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(both, p->pending_bad, 8, hipMemcpyDeviceToHost, p->ctx->copy_stream));
	HIP_TRY(hipStreamSynchronize(p->ctx->copy_stream));
	p->pending_bad = nullptr;
	const u32 bad = both[0];
	p->ctx->stats["pool_other_bases"] = both[1];      // IUPAC codes packed as N (see pool_finish)
	p->sym = sym_wanted() && p->W == 2 && p->n_records % 2 == 0 && p->n_primary % 2 == 0;      // (vdjx_pool_load_forward_begin: the couples are made by the packing)
	p->ctx->stats["pool_symmetric"] = p->sym ? 1 : 0;
	if (bad) { vdjx_set_error("pool: %u records do not start with the '0' strand byte", bad); return VDJX_EINVAL; }
	return VDJX_OK;
}

extern "C" int vdjx_host_alloc(vdjx_ctx* c, size_t bytes, void** out) {
	if (!c || !out) { vdjx_set_error("vdjx_host_alloc: NULL argument"); return VDJX_EINVAL; }
	*out = nullptr;
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
	{
		std::lock_guard<std::mutex> g(c->host_blocks_mu);
		c->host_blocks.emplace_back((const char*) *out, bytes ? bytes : 1);
	}
	return VDJX_OK;
}

extern "C" void vdjx_host_free(vdjx_ctx* c, void* p) {
	if (!p) return;
	if (c) {
		(void) hipSetDevice(c->device);
		std::lock_guard<std::mutex> g(c->host_blocks_mu);
		for (size_t i = 0; i < c->host_blocks.size(); i++)
			if (c->host_blocks[i].first == (const char*) p) { c->host_blocks.erase(c->host_blocks.begin() + (long) i); break; }
	}
	(void) hipHostFree(p);
}

// [p, p + bytes) inside one of the context's own page-locked blocks?  (hipHostMalloc'ed memory has one address on both sides)
bool vdjx_host_block_holds(vdjx_ctx* c, const void* p, size_t bytes) {
	std::lock_guard<std::mutex> g(c->host_blocks_mu);
	for (const auto& b : c->host_blocks)
		if ((const char*) p >= b.first && (const char*) p + bytes <= b.first + b.second) return true;
	return false;
}

extern "C" int vdjx_host_take_rows(void* dst, const void* src, size_t stride, size_t first, size_t len, const uint32_t* idx, size_t n) {
	if (n && (!dst || !src || !idx)) { vdjx_set_error("vdjx_host_take_rows: NULL argument"); return VDJX_EINVAL; }
	if (first + len > stride) { vdjx_set_error("vdjx_host_take_rows: [first, first + len) leaves the row"); return VDJX_EINVAL; }
	for (size_t i = 0; i < n; i++) memcpy((char*) dst + i * len, (const char*) src + (size_t) idx[i] * stride + first, len);
	return VDJX_OK;
}

extern "C" size_t vdjx_pool_records(const vdjx_pool* p) { return p ? p->n_records : 0; }

extern "C" void vdjx_pool_free(vdjx_pool* p) {
	if (!p) return;
	(void) hipSetDevice(p->device);
	if (p->d_block) {
		if (vdjx_ctx_alive(p->ctx)) {
			if (p->ctx->ri_job) (void) vdjx_ri_join(p->ctx);    // (a begun index build may be reading these records)
			(void) hipStreamSynchronize(p->ctx->stream);     // nothing in flight may still read the block
			if (p->pending_bad) (void) hipStreamSynchronize(p->ctx->copy_stream);
			p->ctx->blocks.release(p->d_block, p->block_cap);
		} else
			free_dev(p->d_block);
	}
	delete p;
}
