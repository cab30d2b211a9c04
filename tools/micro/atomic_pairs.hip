// Diagnostic: rate of 64-bit atomic adds to random rows of a table in HBM, two sums per row:
//  A: each lane adds to both sums of its row (two instructions, the lanes' rows all different)
//  B: lanes pair up: lane 2k adds sum 0 and lane 2k+1 adds sum 1 of row k's key (one instruction covers 32 rows)
//  C: one sum only (half the adds)
//  D: what the table kernel does per record: an atomic load of the row's key word (another array), then B
//  E: key and sums in one 32-byte row, ONE returning add instruction: lane 4k adds 0 to the key word
//     (and gets the key back), lanes 4k+1 / 4k+2 add the sums
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

template <int MODE>
__global__ void __launch_bounds__(256)
probes(unsigned long long *keys, unsigned long long *vals, unsigned long long *rows4, uint32_t mask, uint32_t n, uint32_t nrows,
	unsigned long long *sink)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t slot = (uint32_t)mix(i % nrows) & mask;
	unsigned long long got = 0;
	if (MODE == 3) {
		got = __hip_atomic_load(&keys[(size_t)slot * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
		for (int half = 0; half < 2; half++) {
			const uint32_t s = __shfl(slot, (int)(half * 32 + (lane >> 1)), 64);
			atomicAdd(&vals[(size_t)s * 2 + (lane & 1)], (lane & 1) ? 3ull : 1ull);
		}
	} else {
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint32_t s = __shfl(slot, (int)(q * 16 + (lane >> 2)), 64);
			const uint32_t j = lane & 3;
			unsigned long long r = 0;
			if (j < 3) r = atomicAdd(&rows4[(size_t)s * 4 + j], (j == 0) ? 0ull : (j == 1) ? 1ull : 3ull);
			const unsigned long long k = __shfl(r, (int)(lane & ~3u), 64);
			if ((lane >> 2) + 16 * q == lane) got = k;     /* (some lane keeps it) */
			got ^= k;
		}
	}
	if (got == 0x1234567ull) sink[0] = got;
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
	{
		const uint32_t n = 1u << 22, cap = 1u << 19, nrows = 190000;
		unsigned long long *keys, *vals2, *rows4, *sink;
		CHECK(hipMalloc(&keys, (size_t)cap * 16));
		CHECK(hipMalloc(&vals2, (size_t)cap * 16));
		CHECK(hipMalloc(&rows4, (size_t)cap * 32));
		CHECK(hipMalloc(&sink, 8));
		CHECK(hipMemset(keys, 0, (size_t)cap * 16));
		for (int mode = 3; mode < 5; mode++) {
			float best = 1e9f;
			for (int rep = 0; rep < 6; rep++) {
				CHECK(hipMemset(vals2, 0, (size_t)cap * 16));
				CHECK(hipMemset(rows4, 0, (size_t)cap * 32));
				hipEvent_t e0, e1;
				CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
				CHECK(hipEventRecord(e0));
				if (mode == 3) hipLaunchKernelGGL(probes<3>, dim3(n / 256), dim3(256), 0, 0, keys, vals2, rows4, cap - 1, n, nrows, sink);
				else hipLaunchKernelGGL(probes<4>, dim3(n / 256), dim3(256), 0, 0, keys, vals2, rows4, cap - 1, n, nrows, sink);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (rep && ms < best) best = ms;
			}
			printf("mode %d: %.4f ms for %u records\n", mode, best, n);
		}
	}
	return 0;
}
