// profiles/micro/ea_calib.hip -- calibration of the L2's memory-side read counters on gfx950 (test infrastructure, not the product).
// Known byte counts in the access patterns the hot path uses, one kernel each, so that rocprofv3's TCC_EA0_RDREQ / _32B / TCC_BUBBLE /
// _DRAM counts can be read against them (profiles/ea_calib.sh, profiles/README.md "what the traffic column means"):
//   k_stream16   every lane 16 B, consecutive lanes on consecutive addresses (the packed pool, the tuple streams)
//   k_gather16   every lane one random 16-byte slot of a table much larger than the caches (the survivor / read-index tables)
//   k_gather8 / k_gather4   random 8- and 4-byte words (chain words, successor lists, class arrays)
//   k_stream16 over 64 MiB, twice: the second pass can be served by the Infinity Cache (is it counted, and as what?)
// usage: ea_calib [table_MiB=4096]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(uint32_t* p, size_t n) {
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = (uint32_t) (i * 2654435761u);
}
__global__ void k_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
	uint32_t acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}
// the j-th access of the launch goes to slot (j * odd) mod 2^bits: a permutation, every slot at most once, neighbours far apart
__global__ void k_gather16(const uint4* __restrict__ p, uint64_t mask, size_t n_acc, uint32_t* sink) {
	uint32_t acc = 0;
	for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_acc; j += (size_t) gridDim.x * blockDim.x) { const uint4 v = p[(j * 0x9E3779B97F4A7C15ull >> 7) & mask]; acc ^= v.x ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather8(const uint2* __restrict__ p, uint64_t mask, size_t n_acc, uint32_t* sink) {
	uint32_t acc = 0;
	for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_acc; j += (size_t) gridDim.x * blockDim.x) { const uint2 v = p[(j * 0x9E3779B97F4A7C15ull >> 7) & mask]; acc ^= v.x ^ v.y; }
	if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather4(const uint32_t* __restrict__ p, uint64_t mask, size_t n_acc, uint32_t* sink) {
	uint32_t acc = 0;
	for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_acc; j += (size_t) gridDim.x * blockDim.x) acc ^= p[(j * 0x9E3779B97F4A7C15ull >> 7) & mask];
	if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream16_small(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {      // (its own name: the 64 MiB passes)
	uint32_t acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
	const size_t mib = argc > 1 ? (size_t) atoll(argv[1]) : 4096;
	size_t bytes = 1;
	while (bytes * 2 <= mib << 20) bytes *= 2;             // a power of two
	uint8_t* d;
	uint32_t* sink;
	CK(hipMalloc(&d, bytes));
	CK(hipMalloc(&sink, 4));
	k_fill<<<4096, 256>>>((uint32_t*) d, bytes / 4);
	CK(hipDeviceSynchronize());
	const size_t n_acc = (size_t) 1 << 26;                  // 64 Mi random accesses per gather launch
	k_stream16<<<8192, 256>>>((const uint4*) d, bytes / 16, sink);
	k_gather16<<<8192, 256>>>((const uint4*) d, bytes / 16 - 1, n_acc, sink);
	k_gather8<<<8192, 256>>>((const uint2*) d, bytes / 8 - 1, n_acc, sink);
	k_gather4<<<8192, 256>>>((const uint32_t*) d, bytes / 4 - 1, n_acc, sink);
	CK(hipDeviceSynchronize());
	const size_t small = (size_t) 64 << 20;
	for (int pass = 0; pass < 3; pass++) k_stream16_small<<<8192, 256>>>((const uint4*) d, small / 16, sink);
	CK(hipDeviceSynchronize());
	printf("{\"table_bytes\": %zu, \"k_stream16\": {\"bytes\": %zu}, \"k_gather16\": {\"accesses\": %zu, \"bytes\": %zu}, \"k_gather8\": {\"accesses\": %zu, \"bytes\": %zu}, "
	       "\"k_gather4\": {\"accesses\": %zu, \"bytes\": %zu}, \"k_stream16_small\": {\"bytes_per_pass\": %zu, \"passes\": 3}}\n",
	       bytes, bytes, n_acc, n_acc * 16, n_acc, n_acc * 8, n_acc, n_acc * 4, small);
	return 0;
}
