// recip_check.hip -- exhaustive comparison of short reciprocal sequences with the correctly rounded 1.0f / x.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/recip_check tools/recip_check.hip && tools/build/recip_check
//
// The exact Moller-Trumbore tests divide once per (triangle, ray) pair: inv_det = 1.0f / det, which the compiler expands to
// v_div_scale x2, v_rcp, five FMAs, v_div_fmas and v_div_fixup.  det has passed |det| >= EPSILON (1e-21 ... see D_EPSILON)
// when it is inverted, so the scaling that sequence carries for tiny and huge operands is never needed for tiny ones.  This
// program runs every one of the 2^32 float bit patterns through candidate sequences and counts the operands whose result
// differs from 1.0f / x by a bit, per exponent, so that the range a sequence is exact on can be read off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>

#define NSEQ 4
__device__ __forceinline__ float seq(int which, float x)
{
	float r = __builtin_amdgcn_rcpf(x);
	if (which == 0)
		return r;
	float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	if (which == 1)
		return r;
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	if (which == 2)
		return r;
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	return r;
}

__global__ void k_check(unsigned long long *bad) // bad[which][256 exponents]
{
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const unsigned bits = (unsigned)i;
		const float x = __uint_as_float(bits);
		if (!(fabsf(x) <= 3.4e38f)) // NaN, inf
			continue;
		if (x == 0.0f)
			continue;
		const float want = 1.0f / x;
#pragma unroll
		for (int w = 0; w < NSEQ; w++) {
			const float got = seq(w, x);
			if (__float_as_uint(got) != __float_as_uint(want))
				atomicAdd(&bad[w * 256 + ((bits >> 23) & 255u)], 1ull);
		}
	}
}

int main()
{
	unsigned long long *d, h[NSEQ * 256];
	hipMalloc(&d, sizeof h);
	hipMemset(d, 0, sizeof h);
	hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d);
	if (hipDeviceSynchronize() != hipSuccess)
		return 1;
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *names[NSEQ] = { "v_rcp_f32", "rcp + 1 Newton step (2 FMA)", "rcp + 2 steps (4 FMA)", "rcp + 3 steps (6 FMA)" };
	for (int w = 0; w < NSEQ; w++) {
		unsigned long long total = 0;
		int lo = -1, hi = -1;
		for (int e = 0; e < 256; e++)
			if (h[w * 256 + e]) {
				total += h[w * 256 + e];
				if (lo < 0)
					lo = e;
				hi = e;
			}
		printf("%-32s mismatches %llu", names[w], total);
		if (total)
			printf("  biased exponents %d..%d", lo, hi);
		// the exponents in between that are clean
		unsigned long long mid = 0;
		for (int e = 2; e <= 250; e++)
			mid += h[w * 256 + e];
		printf("  (exponents 2..250: %llu)\n", mid);
		if (total && total < 4000) {
			printf("   per exponent:");
			for (int e = 0; e < 256; e++)
				if (h[w * 256 + e])
					printf(" %d:%llu", e, h[w * 256 + e]);
			printf("\n");
		}
	}
	return 0;
}
