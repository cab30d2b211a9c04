// Diagnostic: rate of 64-bit atomic adds to random rows of a table in HBM, two sums per row:
//  A: each lane adds to both sums of its row (two instructions, the lanes' rows all different)
//  B: lanes pair up: lane 2k adds sum 0 and lane 2k+1 adds sum 1 of row k's key (one instruction covers 32 rows)
//  C: one sum only (half the adds)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t h)
{
	h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
	return h;
}

template <int MODE>
__global__ void __launch_bounds__(256)
adds(unsigned long long *vals, uint32_t mask, uint32_t n, uint32_t rows)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t slot = (uint32_t)mix(i % rows) & mask;
	if (MODE == 0) {
		atomicAdd(&vals[(size_t)slot * 2], 1ull);
		atomicAdd(&vals[(size_t)slot * 2 + 1], 3ull);
	} else if (MODE == 2) {
		atomicAdd(&vals[(size_t)slot * 2], 1ull);
	} else {
		const uint32_t lane = threadIdx.x & 63;
#pragma unroll
		for (int half = 0; half < 2; half++) {
			const uint32_t s = __shfl(slot, (int)(half * 32 + (lane >> 1)), 64);
			atomicAdd(&vals[(size_t)s * 2 + (lane & 1)], (lane & 1) ? 3ull : 1ull);
		}
	}
}

int main(void)
{
	const uint32_t n = 1u << 22, cap = 1u << 19, rows = 190000;
	unsigned long long *vals;
	CHECK(hipMalloc(&vals, (size_t)cap * 16));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int mode = 0; mode < 3; mode++) {
		float best = 1e9f;
		for (int rep = 0; rep < 6; rep++) {
			CHECK(hipMemset(vals, 0, (size_t)cap * 16));
			CHECK(hipEventRecord(e0));
			if (mode == 0) hipLaunchKernelGGL(adds<0>, dim3(n / 256), dim3(256), 0, 0, vals, cap - 1, n, rows);
			if (mode == 1) hipLaunchKernelGGL(adds<1>, dim3(n / 256), dim3(256), 0, 0, vals, cap - 1, n, rows);
			if (mode == 2) hipLaunchKernelGGL(adds<2>, dim3(n / 256), dim3(256), 0, 0, vals, cap - 1, n, rows);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep && ms < best) best = ms;
		}
		unsigned long long h[2];
		uint64_t s0 = 0, s1 = 0;
		unsigned long long *hv = (unsigned long long *)malloc((size_t)cap * 16);
		CHECK(hipMemcpy(hv, vals, (size_t)cap * 16, hipMemcpyDeviceToHost));
		for (uint32_t k = 0; k < cap; k++) { s0 += hv[2 * k]; s1 += hv[2 * k + 1]; }
		free(hv); (void)h;
		printf("mode %d: %.4f ms for %u records (%.1f G lane-adds/s)  sums %llu %llu\n", mode, best, n,
			(mode == 2 ? 1.0 : 2.0) * n / (best * 1e-3) / 1e9, (unsigned long long)s0, (unsigned long long)s1);
	}
	return 0;
}
