// Microbenchmark: cost of the fast kernel's column-store pattern.
// Each wave owns a tile of 64 records and writes NCOL columns (widths as in the
// IPv4 IPFIX template of the benchmark).  Variants:
//   0  record index = tile * 64 + lane                   (tile-aligned output order)
//   1  arrival order: the tile's records are 2-3 runs of 30 records that lie
//      (30 + 22)-record periods apart, like IPv4 runs between IPv6 runs
//   2  as 1, but a store covers one run only (three narrower stores)
// Usage: store_pattern <variant> [nrec]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static const int widths[14] = { 4, 4, 2, 2, 1, 2, 8, 8, 4, 4, 4, 4, 1, 8 };   // + nparsed + mask

struct params { uint8_t *col[14]; uint32_t ntiles; uint32_t variant; };

__global__ void __launch_bounds__(256)
k(params p)
{
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t tile = blockIdx.x * 4 + wave; tile < p.ntiles; tile += gridDim.x * 4) {
		uint32_t g = tile * 64 + lane;     // position among the IPv4 records
		uint32_t ridx = g;
		if (p.variant) {
			uint32_t run = g / 30, i = g % 30;  // IPv4 run number; runs alternate 7:3 with 22-record IPv6 runs
			ridx = run * 30 + (run * 3 / 7) * 22 + i;
		}
		uint64_t v = g * 0x9E3779B97F4A7C15ull;
#pragma unroll
		for (int c = 0; c < 14; c++) {
			const int w = (c == 0) ? 4 : (c == 1) ? 4 : (c == 2) ? 2 : (c == 3) ? 2 : (c == 4) ? 1 : (c == 5) ? 2
				: (c == 6) ? 8 : (c == 7) ? 8 : (c == 8) ? 4 : (c == 9) ? 4 : (c == 10) ? 4 : (c == 11) ? 4 : (c == 12) ? 1 : 8;
			uint8_t *col = p.col[c];
			if (w == 1) col[ridx] = (uint8_t)v;
			else if (w == 2) ((uint16_t *)col)[ridx] = (uint16_t)v;
			else if (w == 4) ((uint32_t *)col)[ridx] = (uint32_t)v;
			else ((uint64_t *)col)[ridx] = v;
		}
	}
}

int main(int argc, char **argv)
{
	int variant = (argc > 1) ? atoi(argv[1]) : 0;
	size_t nrec = (argc > 2) ? strtoull(argv[2], 0, 0) : (size_t)2936010;   // 70 % of 4 Mi
	size_t span = nrec * 2 + 4096;       // arrival-order indices reach about 1.43 * nrec
	params p;
	size_t bytes = 0;
	for (int c = 0; c < 14; c++) {
		CHECK(hipMalloc(&p.col[c], span * widths[c]));
		CHECK(hipMemset(p.col[c], 0, span * widths[c]));
		bytes += nrec * widths[c];
	}
	p.ntiles = (uint32_t)(nrec / 64);
	p.variant = (uint32_t)variant;
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	uint32_t blocks = (p.ntiles + 3) / 4;
	if (blocks > 4096) blocks = 4096;
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p);
	CHECK(hipEventRecord(a, 0));
	for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p);
	CHECK(hipEventRecord(b, 0));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	ms /= 10;
	printf("variant %d: %.4f ms per launch, %.1f MB written, %.0f GB/s\n", variant, ms, bytes / 1e6, bytes / ms / 1e6);
	return 0;
}
