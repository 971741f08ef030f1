/*
 * ugrt_fmath.h -- normative scalar float functions of the ugrt ABI.
 *
 * The reference (sushruta/uniformgrid-raytracing) leans on three pieces of
 * platform arithmetic whose results are NOT fixed by IEEE-754:
 *   - acosf()            grid_kernel.cu:409,439,468  misc_kernel.cu:191
 *   - (int)/(unsigned)   grid_kernel.cu:208-216,415-418,443-447 (float -> integer,
 *                        also for NaN and out-of-range inputs)
 *   - floor() -> int     grid_kernel.cu:208-216, trace_kernel.cu:72
 * A grid ray tracer's cell ids depend on every last bit of these, so the ABI
 * pins them down here: the HIP kernels, the host code and the CPU oracle all
 * include THIS file, and every operation in it is a correctly rounded IEEE
 * fp32 add/mul/div/sqrt (compile with -ffp-contract=off everywhere).
 *
 * Plain C99 / C++ / HIP.  No dependencies.
 */
#ifndef UGRT_FMATH_H
#define UGRT_FMATH_H

#if defined(__HIPCC__)
#define UGRT_HD __host__ __device__ __forceinline__
#else
#define UGRT_HD static inline
#endif

#define UGRT_PI_F 3.14159274101257324f /* (float)M_PI, main.cu:186-187 */

/* (int)x as the reference's target hardware did it: round toward zero,
 * NaN -> 0, saturating.  (x86 would give INT_MIN for NaN; CUDA and CDNA give 0.) */
UGRT_HD int ugrt_f2i_portable(float x)
{
	if (!(x == x))
		return 0;
	if (x >= 2147483648.0f)
		return 2147483647;
	if (x <= -2147483648.0f)
		return (-2147483647 - 1);
	return (int)x;
}

/* (unsigned)x: round toward zero, NaN -> 0, negatives -> 0, saturating. */
UGRT_HD unsigned int ugrt_f2u_portable(float x)
{
	if (!(x == x))
		return 0u;
	if (x <= 0.0f)
		return 0u;
	if (x >= 4294967296.0f)
		return 4294967295u;
	return (unsigned int)x;
}

/* On the device these ARE one instruction each: v_cvt_i32_f32 / v_cvt_u32_f32 round toward zero, give 0 for a NaN
 * and saturate (the portable forms spell that out in ~15 instructions and three nested branches).  Named by inline
 * assembly because a C cast is undefined for the inputs that matter; checked against the portable forms over every
 * float on the device (ugrt_ctx_get_state "f2i_mismatches"). */
UGRT_HD int ugrt_f2i(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	int r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
	return r;
#else
	return ugrt_f2i_portable(x);
#endif
}
UGRT_HD unsigned int ugrt_f2u(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	unsigned int r;
	asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
	return r;
#else
	return ugrt_f2u_portable(x);
#endif
}

/* floorf without libm: exact for every float. */
UGRT_HD float ugrt_floorf(float x)
{
	float t;
	if (!(x == x))
		return x;
	if (x >= 8388608.0f || x <= -8388608.0f)
		return x; /* already integral */
	t = (float)(int)x;
	if (t > x)
		t = t - 1.0f;
	return t;
}

/*
 * UGRT_FLAG_STRICT_TEXTURE: position and weight of CUDA's linear texture filter for the normalized coordinate x on N
 * texels, as the CUDA C Programming Guide states it ("Texture Fetching", linear filtering): xB = N x - 0.5,
 * i = floor(xB), alpha = frac(xB) "stored in 9-bit fixed point format with 8 bits of fractional value".  The guide
 * does not say how alpha is rounded (nearest is taken here) nor how the unit forms xB, and no CUDA device is here
 * to ask: PARITY UNPINNED.  Clamp addressing: the last texel pair is used with alpha = 1 at the upper edge.
 */
UGRT_HD void ugrt_tex_linear8(float x, int N, int *i_out, float *alpha_out)
{
	float xB = x * (float)N - 0.5f, fl, q;
	int i;
	if (!(xB > 0.0f))
		xB = 0.0f;
	fl = ugrt_floorf(xB);
	i = ugrt_f2i(fl);
	if (i > N - 2) {
		i = N - 2;
		fl = (float)i;
	}
	q = ugrt_floorf((xB - fl) * 256.0f + 0.5f);
	if (q > 256.0f)
		q = 256.0f;
	*i_out = i;
	*alpha_out = q * (1.0f / 256.0f);
}

/* (int)floor(x) */
UGRT_HD int ugrt_floor2i_portable(float x)
{
	return ugrt_f2i_portable(ugrt_floorf(x));
}
UGRT_HD int ugrt_floor2i(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	/* v_floor_f32 is exact for every float (it differs from ugrt_floorf only in the sign of a zero, which the
	 * conversion drops) */
	return ugrt_f2i(__builtin_floorf(x));
#else
	return ugrt_floor2i_portable(x);
#endif
}

/*
 * acosf: rational approximation on z = x^2 (|x| <= 0.5) or z = (1 -+ x)/2,
 * the classic fdlibm/msun single-precision scheme, written out so that the
 * order of every operation is fixed.  |x| > 1 and NaN give NaN, as the
 * reference's acosf does.  Max error < 1 ulp-ish; what matters is that it is
 * the SAME function on the CPU and on the GPU.
 */
UGRT_HD float ugrt_acosf_R(float z)
{
	const float pS0 = 1.6666586697e-01f;
	const float pS1 = -4.2743422091e-02f;
	const float pS2 = -8.6563630030e-03f;
	const float qS1 = -7.0662963390e-01f;
	float p = z * (pS0 + z * (pS1 + z * pS2));
	float q = 1.0f + z * qS1;
	return p / q;
}

UGRT_HD float ugrt_acosf(float x)
{
	const float pio2_hi = 1.57079637050628662109375f;
	const float pio2_lo = -4.37113900018624283e-8f;
	const float pi_hi = 3.14159274101257324f;
	float z, s, r;
	if (!(x == x))
		return x;
	if (x > 1.0f || x < -1.0f)
		return (x - x) / (x - x); /* NaN */
	if (x == 1.0f)
		return 0.0f;
	if (x == -1.0f)
		return pi_hi;
	if (x <= 0.5f && x >= -0.5f) {
		z = x * x;
		r = ugrt_acosf_R(z);
		return pio2_hi - (x - (pio2_lo - x * r));
	}
	if (x < 0.0f) {
		z = (1.0f + x) * 0.5f;
		s = __builtin_sqrtf(z);
		r = ugrt_acosf_R(z);
		return pi_hi - 2.0f * (s + s * r);
	}
	z = (1.0f - x) * 0.5f;
	s = __builtin_sqrtf(z);
	r = ugrt_acosf_R(z);
	return 2.0f * (s + s * r);
}

#endif /* UGRT_FMATH_H */
