// Issue rate of 32-bit integer multiplies against adds / xors on gfx950: one wave per SIMD, a dependent chain of N operations.
//   hipcc -O3 --offload-arch=gfx950 -o mulrate mulrate.hip && ./mulrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int OP> __global__ void k(unsigned* out, unsigned a, unsigned b) {
	unsigned x = threadIdx.x + a, y = threadIdx.x * 3 + b;
#pragma unroll 64
	for (int i = 0; i < N; i++) {
		if (OP == 0) x = x * y + 1u;                       // v_mul_lo_u32 (+ add)
		if (OP == 1) x = (x ^ y) + 1u;                     // xor + add
		if (OP == 2) x = __umulhi(x, y) + 1u;              // v_mul_hi_u32
		if (OP == 3) x = __umul24(x, y) + 1u;              // v_mul_u32_u24
		if (OP == 4) { unsigned long long z = (unsigned long long) x * 0x9E3779B97F4A7C15ull; x = (unsigned) (z >> 32) ^ (unsigned) z; }    // 64-bit multiply by a constant
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int OP> float run(unsigned* d, int blocks) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u);
	hipEventRecord(e0);
	for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms / 20;
}
int main() {
	unsigned* d; hipMalloc(&d, 256 * 256 * 64 * 4);
	const int blocks = 256;                                // one workgroup of 4 waves per CU: one wave per SIMD
	const char* nm[] = {"mul_lo+add", "xor+add", "mul_hi+add", "mul24+add", "mul64 by constant"};
	float t[5] = {run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks), run<4>(d, blocks)};
	for (int i = 0; i < 5; i++) printf("%-20s %.3f ms for %d dependent steps: %.1f ns per step\n", nm[i], t[i], N, t[i] * 1e6 / N);
	return 0;
}
