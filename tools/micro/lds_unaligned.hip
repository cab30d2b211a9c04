// Check on the device: records staged in LDS with global_load_lds_dwordx4
// (lane-linear image), then read back at arbitrary byte offsets with 1/2/4/8/16
// byte LDS loads.  Prints the number of mismatches against the host's bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t *p, uint32_t nbytes, uint32_t stride, uint32_t off, uint8_t *out)
{
	extern __shared__ uint8_t l[];
	const uint32_t lane = threadIdx.x;
	const uint32_t nchunk = (nbytes + 15) / 16;
	for (uint32_t c0 = 0; c0 < nchunk; c0 += 64) {
		if (c0 + lane < nchunk)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 16 * (c0 + lane)),
				(__attribute__((address_space(3))) void *)(l + 16 * c0), 16, 0, 0);
	}
	__builtin_amdgcn_s_waitcnt(0);
	__syncthreads();
	const uint8_t *r = l + lane * stride + off;
	uint8_t a; uint16_t b; uint32_t c; uint64_t d; u32x4 e;
	a = r[0];
	__builtin_memcpy(&b, r + 1, 2);
	__builtin_memcpy(&c, r + 3, 4);
	__builtin_memcpy(&d, r + 7, 8);
	__builtin_memcpy(&e, r + 15, 16);
	uint8_t *o = out + lane * 31;
	o[0] = a;
	__builtin_memcpy(o + 1, &b, 2);
	__builtin_memcpy(o + 3, &c, 4);
	__builtin_memcpy(o + 7, &d, 8);
	__builtin_memcpy(o + 15, &e, 16);
}

int main()
{
	const uint32_t stride = 46, nbytes = 64 * 46 + 64;
	std::vector<uint8_t> h(nbytes + 64);
	for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7 + 3);
	uint8_t *dp, *dout;
	CHECK(hipMalloc(&dp, h.size()));
	CHECK(hipMalloc(&dout, 64 * 31));
	CHECK(hipMemcpy(dp, h.data(), h.size(), hipMemcpyHostToDevice));
	int bad = 0;
	for (uint32_t off = 0; off < 16; off++) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, dp, nbytes, stride, off, dout);
		std::vector<uint8_t> o(64 * 31);
		CHECK(hipMemcpy(o.data(), dout, o.size(), hipMemcpyDeviceToHost));
		for (int lane = 0; lane < 64; lane++)
			if (memcmp(&o[lane * 31], &h[lane * stride + off], 31)) bad++;
	}
	printf("unaligned LDS reads after global_load_lds: %d mismatching lanes of %d\n", bad, 16 * 64);
	return bad != 0;
}
