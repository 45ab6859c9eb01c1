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

static int load_set(vdjx_ctx* c, const u32* codes, size_t n, u32** d_bits) {
	const size_t words = (size_t) 1 << 27;
	if (!*d_bits) HIP_TRY(hipMalloc(d_bits, words * 4));
	HIP_TRY(hipMemsetAsync(*d_bits, 0, words * 4, c->stream));
	if (n) {
		u32* d_codes = nullptr;
		HIP_TRY(hipMalloc(&d_codes, n * 4));
		hipError_t e = hipMemcpyAsync(d_codes, codes, n * 4, hipMemcpyHostToDevice, c->stream);
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_bitmap_set, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, d_codes, n, *d_bits);
			e = hipStreamSynchronize(c->stream);
		}
		(void) hipFree(d_codes);
		if (e != hipSuccess) { vdjx_set_error("anchor set load: %s", hipGetErrorString(e)); return VDJX_EHIP; }
	}
	return VDJX_OK;
}

extern "C" int vdjx_anchor_sets_load(vdjx_ctx* c, const uint32_t* v_codes, size_t nv, const uint32_t* j_codes, size_t nj) {
	if (!c || (nv && !v_codes) || (nj && !j_codes)) { vdjx_set_error("vdjx_anchor_sets_load: NULL argument"); return VDJX_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	vdjx_clear_errors();
	int rc = load_set(c, v_codes, nv, &c->d_vbits);
	if (rc) return rc;
	rc = load_set(c, j_codes, nj, &c->d_jbits);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
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
