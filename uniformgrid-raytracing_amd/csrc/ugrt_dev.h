// ugrt_dev.h -- device-side scalar geometry of the render path.
//
// Every expression keeps the operand order and association of the reference's
// device code (cited per function), because cell ids, hit ids and the strict
// `<` tie-breaks depend on the last bit.  Built with -ffp-contract=off: each
// + - * / sqrt is one correctly rounded IEEE fp32 operation, no FMA.
#ifndef UGRT_DEV_H
#define UGRT_DEV_H

#include "ugrt_ctx.h"

#define D_EPSILON 1e-21f // main.cu.h:42

// main.cu.h:44-56
#define D_CROSS(dest, v1, v2)                        \
	do {                                         \
		dest[0] = v1[1] * v2[2] - v1[2] * v2[1]; \
		dest[1] = v1[2] * v2[0] - v1[0] * v2[2]; \
		dest[2] = v1[0] * v2[1] - v1[1] * v2[0]; \
	} while (0)
#define D_DOT(v1, v2) (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2])
#define D_NORMALIZE(A)                                                                         \
	do {                                                                                   \
		float l_ = 1.0f / __builtin_sqrtf(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);    \
		A[0] *= l_;                                                                    \
		A[1] *= l_;                                                                    \
		A[2] *= l_;                                                                    \
	} while (0)

// grid_kernel.cu:4-11 mulMatrixVector_D with vec[3] == 1 (x*1.0f is exact)
#define D_MULMV_ROW(m, base, r, x, y, z) \
	(m[(base) + (r)] * (x) + m[(base) + 4 + (r)] * (y) + m[(base) + 8 + (r)] * (z) + m[(base) + 12 + (r)] * 1.0f)

// grid_kernel.cu:13-36 getTransformedVertex: MV, divide, P, divide
__device__ __forceinline__ void d_transformed_vertex(const CamBlock &cam, float px, float py, float pz,
						     float *ndc)
{
	const float *m = cam.cc;
	float t0 = D_MULMV_ROW(m, 16, 0, px, py, pz);
	float t1 = D_MULMV_ROW(m, 16, 1, px, py, pz);
	float t2 = D_MULMV_ROW(m, 16, 2, px, py, pz);
	float t3 = D_MULMV_ROW(m, 16, 3, px, py, pz);
	float qx = t0 / t3, qy = t1 / t3, qz = t2 / t3;
	t0 = D_MULMV_ROW(m, 32, 0, qx, qy, qz);
	t1 = D_MULMV_ROW(m, 32, 1, qx, qy, qz);
	t2 = D_MULMV_ROW(m, 32, 2, qx, qy, qz);
	t3 = D_MULMV_ROW(m, 32, 3, qx, qy, qz);
	ndc[0] = t0 / t3;
	ndc[1] = t1 / t3;
	ndc[2] = t2 / t3;
}

// grid_kernel.cu:132-146
__device__ __forceinline__ float d_min3(float e1, float e2, float e3)
{
	return (e1 < e2) ? ((e1 < e3) ? e1 : e3) : ((e2 < e3) ? e2 : e3);
}
__device__ __forceinline__ float d_max3(float e1, float e2, float e3)
{
	return (e1 > e2) ? ((e1 > e3) ? e1 : e3) : ((e2 > e3) ? e2 : e3);
}
__device__ __forceinline__ int d_imin3(int e1, int e2, int e3)
{
	return (e1 < e2) ? ((e1 < e3) ? e1 : e3) : ((e2 < e3) ? e2 : e3);
}
__device__ __forceinline__ int d_imax3(int e1, int e2, int e3)
{
	return (e1 > e2) ? ((e1 > e3) ? e1 : e3) : ((e2 > e3) ? e2 : e3);
}
__device__ __forceinline__ int d_clampi(int v, int lo, int hi)
{
	v = v < lo ? lo : v;
	return v > hi ? hi : v;
}

// grid_kernel.cu:354-363 getMagnitude
__device__ __forceinline__ float d_magnitude(const float *vec)
{
	float rad = 0;
	rad += vec[0] * vec[0];
	rad += vec[1] * vec[1];
	rad += vec[2] * vec[2];
	return __builtin_sqrtf(rad);
}

// grid_kernel.cu:395-422 getEffective_x
__device__ __forceinline__ unsigned d_effective_x(const CamBlock &cam, const float *vec, float max, int nbx2)
{
	const float *cc = cam.cc;
	float upDotValue = vec[0] * cc[16 + 1] + vec[1] * cc[16 + 5] + vec[2] * cc[16 + 9];
	float tmp[3];
	tmp[0] = vec[0] - upDotValue * cc[16 + 1];
	tmp[1] = vec[1] - upDotValue * cc[16 + 5];
	tmp[2] = vec[2] - upDotValue * cc[16 + 9];
	float val = d_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	float forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] + tmp[2] * cc[16 + 10];
	float angle = ugrt_acosf(forwardDotValue);
	float rightDotValue = tmp[0] * cc[16 + 0] + tmp[1] * cc[16 + 4] + tmp[2] * cc[16 + 8];
	int q = ugrt_f2i((angle / max) * (float)nbx2);
	return (rightDotValue > 0) ? (unsigned)(nbx2 + q) : (unsigned)(nbx2 - q);
}

// grid_kernel.cu:452-479 getEffective_y; :468 multiplies where a sum was meant
__device__ __forceinline__ unsigned d_effective_y(const CamBlock &cam, const float *vec, float max, int nby2)
{
	const float *cc = cam.cc;
	float rightDotValue = vec[0] * cc[16 + 0] + vec[1] * cc[16 + 4] + vec[2] * cc[16 + 8];
	float tmp[3];
	tmp[0] = vec[0] - rightDotValue * cc[16 + 0];
	tmp[1] = vec[1] - rightDotValue * cc[16 + 4];
	tmp[2] = vec[2] - rightDotValue * cc[16 + 8];
	float val = d_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	float upDotValue = tmp[0] * cc[16 + 1] + tmp[1] * cc[16 + 5] + tmp[2] * cc[16 + 9];
	float forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] * tmp[2] * cc[16 + 10];
	float angle = ugrt_acosf(forwardDotValue);
	return (upDotValue > 0) ? ugrt_f2u((float)nby2 + (angle / max) * (float)nby2)
				: ugrt_f2u((float)nby2 - (angle / max) * (float)nby2);
}

// trace_kernel.cu:96-114: ray through the CORNER of pixel (col,row), x flipped.
// The 5x5 texture fetch is the exact float bilinear interpolation of the node
// table `tex` (DESIGN.md "ray set-up").
__device__ __forceinline__ void d_ray_dir(const CamBlock &cam, const float *__restrict__ tex, int col, int row,
					  float *ray_direction)
{
	float ftx = (float)col / (float)cam.W;
	float fty = (float)row / (float)cam.H;
	ftx = 1 - ftx;
	float xs = ftx * 4.0f, ys = fty * 4.0f;
	int i = ugrt_f2i(xs), j = ugrt_f2i(ys);
	i = i > 3 ? 3 : i;
	j = j > 3 ? 3 : j;
	float a = xs - (float)i, b = ys - (float)j;
	if (cam.strict_tex) { // (uniform: a scalar branch) the reference's own coordinates through the documented filter rule
		ugrt_tex_linear8(ftx * 0.8f + 0.1f, 5, &i, &a);
		ugrt_tex_linear8(fty * 0.8f + 0.1f, 5, &j, &b);
	}
	float w00 = (1.0f - a) * (1.0f - b);
	float w10 = a * (1.0f - b);
	float w01 = (1.0f - a) * b;
	float w11 = a * b;
	const float *n00 = tex + (j * 5 + i) * 4;
	const float *n01 = tex + ((j + 1) * 5 + i) * 4;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		float T = ((w00 * n00[k] + w10 * n00[4 + k]) + w01 * n01[k]) + w11 * n01[4 + k];
		ray_direction[k] = T - cam.cc[k];
	}
	D_NORMALIZE(ray_direction);
}

// inv_det = 1.0f / det of the Moller-Trumbore tests, for a det that has passed |det| >= D_EPSILON.
// The compiler's division is v_div_scale x2, v_rcp, five FMAs, v_div_fmas, v_div_fixup: the scaling serves operands
// whose reciprocal is (nearly) denormal or overflows.  v_rcp_f32 and ONE Newton step give the correctly rounded
// reciprocal - the same bits as the division - for every float of magnitude [2^-125, 2^124): checked over all 2^32
// bit patterns on the device (tools/recip_check.hip; ugrt_ctx_get_state "recip_mismatches" runs the same check on this
// function, tests/test_gpu_parity.py).  D_EPSILON is 2^-70; a det of 2^124 or more, an infinity or a NaN takes the
// division itself.  Seven vector instructions less per (triangle, ray) test, not a bit of difference.
__device__ __forceinline__ float d_recip_det(float det)
{
	float r = __builtin_amdgcn_rcpf(det);
	const float e = __builtin_fmaf(-det, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	if (__builtin_expect(!(__builtin_fabsf(det) < 0x1p124f), 0))
		r = 1.0f / det;
	return r;
}

// Common front half of trace_kernel.cu:4-33 / light_kernel.cu:13-40.
// tri = {tvec[3], edge1[3], edge2[3]}.  Returns false when the reference
// returns 0 before computing t; otherwise *t_out = DOT(edge2,qvec)*inv_det.
__device__ __forceinline__ bool d_mt_core(const float *tvec, const float *edge1, const float *edge2,
					  const float *dir, float *t_out)
{
	float pvec[3], qvec[3];
	D_CROSS(pvec, dir, edge2);
	float det = D_DOT(edge1, pvec);
	if (det > -D_EPSILON && det < D_EPSILON)
		return false;
	float inv_det = d_recip_det(det);
	float u = D_DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return false;
	D_CROSS(qvec, tvec, edge1);
	float v = D_DOT(dir, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f)
		return false;
	*t_out = D_DOT(edge2, qvec) * inv_det;
	return true;
}

// The same test for rays that SHARE their origin (a light's shadow rays): tvec is then the triangle's own, and so are
// qvec = tvec x edge1 and T = edge2 . qvec, which d_mt_core forms anew for every ray.  They are formed once per triangle
// (d_mt_shared: the same operations in the same order, so the same floats) and every ray's test starts from them:
// 14 operations less per (triangle, ray) pair, results bit for bit those of d_mt_core.
__device__ __forceinline__ void d_mt_shared(const float *tvec, const float *edge1, const float *edge2, float *qvec, float *T)
{
	D_CROSS(qvec, tvec, edge1);
	*T = D_DOT(edge2, qvec);
}
__device__ __forceinline__ bool d_mt_core_shared(const float *tvec, const float *edge1, const float *edge2, const float *qvec, float T,
						 const float *dir, float *t_out)
{
	float pvec[3];
	D_CROSS(pvec, dir, edge2);
	float det = D_DOT(edge1, pvec);
	if (det > -D_EPSILON && det < D_EPSILON)
		return false;
	float inv_det = d_recip_det(det);
	float u = D_DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return false;
	float v = D_DOT(dir, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f)
		return false;
	*t_out = T * inv_det;
	return true;
}

// how far d_mt_core gets (work counters only): 0 = |det| < eps, 1 = u outside, 2 = v or u+v outside, 3 = t computed
__device__ __forceinline__ int d_mt_stage(const float *tvec, const float *edge1, const float *edge2, const float *dir)
{
	float pvec[3], qvec[3];
	D_CROSS(pvec, dir, edge2);
	float det = D_DOT(edge1, pvec);
	if (det > -D_EPSILON && det < D_EPSILON)
		return 0;
	float inv_det = d_recip_det(det);
	float u = D_DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return 1;
	D_CROSS(qvec, tvec, edge1);
	float v = D_DOT(dir, qvec) * inv_det;
	return (v < 0.0f || u + v > 1.0f) ? 2 : 3;
}

// trace_kernel.cu:4-45 intersectTriUV: |t|, accepted when 0 < t < oldt
__device__ __forceinline__ float d_intersect_tri_uv(const float *tvec, const float *edge1, const float *edge2,
						    const float *dir, float oldt)
{
	float t;
	if (!d_mt_core(tvec, edge1, edge2, dir, &t))
		return 0.0f;
	if (t < 0)
		t *= -1;
	return (t < oldt && t > 0) ? t : 0.0f;
}

// light_kernel.cu:13-50 intersectTri: signed t, accepted when t < oldt
__device__ __forceinline__ float d_intersect_tri(const float *tvec, const float *edge1, const float *edge2,
						 const float *dir, float oldt)
{
	float t;
	if (!d_mt_core(tvec, edge1, edge2, dir, &t))
		return 0.0f;
	return (t < oldt) ? t : 0.0f;
}

// intersectTri for rays of one origin, from the triangle's shared part (d_mt_shared)
__device__ __forceinline__ float d_intersect_tri_shared(const float *tvec, const float *edge1, const float *edge2, const float *qvec,
							float T, const float *dir, float oldt)
{
	float t;
	if (!d_mt_core_shared(tvec, edge1, edge2, qvec, T, dir, &t))
		return 0.0f;
	return (t < oldt) ? t : 0.0f;
}

// trace_kernel.cu:159-173 / light_kernel.cu:131-146: stage one triangle as
// {origin - v0, v1 - v0, v2 - v0}
__device__ __forceinline__ void d_stage_triangle(const float *__restrict__ verts, const int *__restrict__ tris,
						 u32 face, float ox, float oy, float oz, float *out9)
{
	int f1 = 3 * tris[face * 3 + 0];
	int f2 = 3 * tris[face * 3 + 1];
	int f3 = 3 * tris[face * 3 + 2];
	float v0x = verts[f1 + 0], v0y = verts[f1 + 1], v0z = verts[f1 + 2];
	out9[3] = verts[f2 + 0] - v0x;
	out9[4] = verts[f2 + 1] - v0y;
	out9[5] = verts[f2 + 2] - v0z;
	out9[6] = verts[f3 + 0] - v0x;
	out9[7] = verts[f3 + 1] - v0y;
	out9[8] = verts[f3 + 2] - v0z;
	out9[0] = ox - v0x;
	out9[1] = oy - v0y;
	out9[2] = oz - v0z;
}

// shader_kernel.cu:46-86 lambert_color_pixel (drop == false) and :88-128
// lambert_color_drop_off_pixel (drop == true); Rv = 3x3 of cam.cc[16..]
template <bool DROP>
__device__ __forceinline__ void d_lambert(const CamBlock &cam, const float *point, const float *normal,
					  float *color, const float *material, float drop_off)
{
	const float *cc = cam.cc;
	float lpv[3], pv[3], nv[3], light_dir[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		lpv[k] = cc[16 + k] * cam.light[0] + cc[16 + 4 + k] * cam.light[1] + cc[16 + 8 + k] * cam.light[2];
		pv[k] = cc[16 + k] * point[0] + cc[16 + 4 + k] * point[1] + cc[16 + 8 + k] * point[2];
		nv[k] = cc[16 + k] * normal[0] + cc[16 + 4 + k] * normal[1] + cc[16 + 8 + k] * normal[2];
	}
	D_NORMALIZE(nv);
	light_dir[0] = pv[0] - lpv[0];
	light_dir[1] = pv[1] - lpv[1];
	light_dir[2] = pv[2] - lpv[2];
	D_NORMALIZE(light_dir);
#pragma unroll
	for (int k = 0; k < 3; k++) {
		if (DROP)
			color[k] += material[k] * 0.5f * drop_off;
		else
			color[k] += material[k] * 0.5f;
	}
	float dot_diffuse = D_DOT(light_dir, nv);
	if (dot_diffuse > 0)
		dot_diffuse *= 1;
	else
		dot_diffuse *= -1;
	if (dot_diffuse > 0) {
#pragma unroll
		for (int k = 0; k < 3; k++) {
			if (DROP)
				color[k] += material[3 + k] * 1.0f * dot_diffuse * drop_off;
			else
				color[k] += material[3 + k] * 1.0f * dot_diffuse;
		}
	}
}

__device__ __forceinline__ unsigned char d_to_u8(float c)
{
	return (unsigned char)(ugrt_f2u(c * 255) & 0xFFu);
}

#endif
