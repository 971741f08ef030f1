/*
 * ugrt_oracle.c -- CPU restatement of the render hot path of
 * sushruta/uniformgrid-raytracing (reference at /root/reference, read-only).
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The product
 * (uniformgrid-raytracing_amd/, include/ugrt.h) never links or imports it.
 *
 * What it is: plain scalar C that follows the reference's device and host code
 * one stage at a time, with the reference's launch geometry turned into loops
 * (block -> tile, thread -> pixel).  Every function cites the reference
 * file:line it restates.  Quirks of the reference that change pixels
 *  are reproduced on purpose.
 *
 * PARITY STATUS: "parity unpinned" for the kernel arithmetic.  The reference
 * ships no tests, golden vectors or fixtures, and its
 * CUDA 2.x sources (nvcc, cutil, CUDPP 1.1 library, GLUT, texture unit) cannot
 * be built in this image without writing stand-ins for that toolchain, which
 * the build rules forbid.  What IS pinned:
 *   - the OBJ/MTL loader: oracle/_ref/ref_objdump is the reference's own
 *     obj_parser sources compiled with g++ (oracle/Makefile);
 *   - three known answers recorded in SURVEY.md section 8(c) from the
 *     reference's kernels (tests/test_oracle_kat.py);
 *   - integer primitives (scan / stable sort / compact) by definition.
 * Platform arithmetic the reference leaves open (acosf, float->int of NaN,
 * the texture unit's bilinear filter, GL's gluPerspective/gluLookAt) is fixed
 * by include/ugrt_fmath.h and by the formulas documented in DESIGN.md.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
 * All float expressions evaluate in fp32 (x86-64 SSE, FLT_EVAL_METHOD 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ugrt_fmath.h"

#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t u32;

/* main.cu.h:44-56 -- the reference's vector macros, association kept. */
#define CROSS(dest, v1, v2)                          \
	do {                                         \
		dest[0] = v1[1] * v2[2] - v1[2] * v2[1]; \
		dest[1] = v1[2] * v2[0] - v1[0] * v2[2]; \
		dest[2] = v1[0] * v2[1] - v1[1] * v2[0]; \
	} while (0)
#define DOT(v1, v2) (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2])
#define NORMALIZE(A)                                                               \
	do {                                                                       \
		float l_ = 1.0f / __builtin_sqrtf(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]); \
		A[0] *= l_;                                                        \
		A[1] *= l_;                                                        \
		A[2] *= l_;                                                        \
	} while (0)
#define ORC_EPSILON 1e-21f /* main.cu.h:42 (a double literal demoted to float on sm_12) */

int orc_set_threads(int n)
{
#ifdef _OPENMP
	if (n > 0)
		omp_set_num_threads(n);
	return omp_get_max_threads();
#else
	(void)n;
	return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Camera (camera.h:135-253, per_frame_funcs.h:18-43,161-434)                 */
/* ------------------------------------------------------------------------- */

/* camera.h:137 gluPerspective(FOVY, W/H, near, far): GL is not available; the
 * Mesa-GLU formulation in double, stored as float. */
static void orc_perspective(float fovy, float aspect, float zn, float zf, float *P)
{
	double radians = (double)fovy / 2.0 * 3.14159265358979323846 / 180.0;
	double deltaZ = (double)zf - (double)zn;
	double sine = sin(radians);
	double cotangent = cos(radians) / sine;
	int i;
	for (i = 0; i < 16; i++)
		P[i] = 0.0f;
	P[0] = (float)(cotangent / (double)aspect);
	P[5] = (float)cotangent;
	P[10] = (float)(-((double)zf + (double)zn) / deltaZ);
	P[11] = -1.0f;
	P[14] = (float)(-2.0 * (double)zn * (double)zf / deltaZ);
	P[15] = 0.0f;
}

/* camera.h:141 gluLookAt: Mesa-GLU formulation in float. */
static void orc_lookat(const float *eye, const float *look, const float *up, float *MV)
{
	float fw[3], s[3], u[3], r;
	int i;
	fw[0] = look[0] - eye[0];
	fw[1] = look[1] - eye[1];
	fw[2] = look[2] - eye[2];
	r = __builtin_sqrtf(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
	if (r != 0.0f) {
		fw[0] /= r;
		fw[1] /= r;
		fw[2] /= r;
	}
	CROSS(s, fw, up);
	r = __builtin_sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
	if (r != 0.0f) {
		s[0] /= r;
		s[1] /= r;
		s[2] /= r;
	}
	CROSS(u, s, fw);
	for (i = 0; i < 16; i++)
		MV[i] = 0.0f;
	MV[0] = s[0];
	MV[4] = s[1];
	MV[8] = s[2];
	MV[1] = u[0];
	MV[5] = u[1];
	MV[9] = u[2];
	MV[2] = -fw[0];
	MV[6] = -fw[1];
	MV[10] = -fw[2];
	MV[15] = 1.0f;
	/* glTranslatef(-eye): m[12+r] = m[r]*x + m[4+r]*y + m[8+r]*z + m[12+r] */
	for (i = 0; i < 3; i++)
		MV[12 + i] = MV[i] * (-eye[0]) + MV[4 + i] * (-eye[1]) + MV[8 + i] * (-eye[2]) + 0.0f;
}

/* camera.h:216-239 Intersect3Planes */
static void orc_intersect3(float *p, const float *n1, const float *n2, const float *n3)
{
	float n2n3[3], n3n1[3], n1n2[3], den;
	n1n2[0] = (n1[1] * n2[2] - n2[1] * n1[2]);
	n1n2[1] = (n1[2] * n2[0] - n1[0] * n2[2]);
	n1n2[2] = (n1[0] * n2[1] - n2[0] * n1[1]);
	n2n3[0] = (n2[1] * n3[2] - n3[1] * n2[2]);
	n2n3[1] = (n2[2] * n3[0] - n2[0] * n3[2]);
	n2n3[2] = (n2[0] * n3[1] - n3[0] * n2[1]);
	n3n1[0] = (n3[1] * n1[2] - n1[1] * n3[2]);
	n3n1[1] = (n3[2] * n1[0] - n3[0] * n1[2]);
	n3n1[2] = (n3[0] * n1[1] - n1[0] * n3[1]);
	den = n1[0] * n2n3[0] + n1[1] * n2n3[1] + n1[2] * n2n3[2];
	p[0] = -(n1[3] * n2n3[0] + n2[3] * n3n1[0] + n3[3] * n1n2[0]) / den;
	p[1] = -(n1[3] * n2n3[1] + n2[3] * n3n1[1] + n3[3] * n1n2[1]) / den;
	p[2] = -(n1[3] * n2n3[2] + n2[3] * n3n1[2] + n3[3] * n1n2[2]) / den;
}

/*
 * Camera::adjustCameraAndPosition + getGLMatrices + getFrustumProperties
 * (camera.h:86-253).  planes is [6][6] (only [i][0..4] written), corners [8][3].
 */
void orc_camera(const float *eye, const float *look, const float *up, float zn, float zf,
		float fovy, float aspect, float *worldori, float *MV, float *P, float *MVP,
		float *planes, float *corners)
{
	int i, j, k;
	const int sel[6][2] = { { 0, -1 }, { 0, 1 }, { 1, 1 }, { 1, -1 }, { 2, 1 }, { 2, -1 } };
	orc_perspective(fovy, aspect, zn, zf, P);
	orc_lookat(eye, look, up, MV);
	worldori[0] = eye[0];
	worldori[1] = eye[1];
	worldori[2] = eye[2];
	worldori[3] = 1.0f;
	/* camera.h:150-162 getMVPMatrix */
	for (i = 0; i < 4; i++)
		for (k = 0; k < 4; k++) {
			MVP[i * 4 + k] = 0;
			for (j = 0; j < 4; j++)
				MVP[i * 4 + k] += (MV[i * 4 + j] * P[j * 4 + k]);
		}
	/* camera.h:167-213 getFrustumPlanes: row3 -+ row{0,1,2} */
	for (i = 0; i < 36; i++)
		planes[i] = 0.0f;
	for (i = 0; i < 6; i++) {
		float *pl = planes + i * 6;
		int r = sel[i][0];
		float sg = (float)sel[i][1];
		const float *m = MVP;
		if (sg > 0) {
			pl[0] = m[3] + m[0 + r];
			pl[1] = m[7] + m[4 + r];
			pl[2] = m[11] + m[8 + r];
			pl[3] = m[15] + m[12 + r];
		} else {
			pl[0] = m[3] - m[0 + r];
			pl[1] = m[7] - m[4 + r];
			pl[2] = m[11] - m[8 + r];
			pl[3] = m[15] - m[12 + r];
		}
		pl[4] = __builtin_sqrtf(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
	}
	for (i = 0; i < 6; i++) {
		float *pl = planes + i * 6;
		pl[0] /= pl[4];
		pl[1] /= pl[4];
		pl[2] /= pl[4];
		pl[3] /= pl[4];
	}
	/* camera.h:241-253 getFrustumCorners */
	orc_intersect3(corners + 0, planes + 0, planes + 12, planes + 24);
	orc_intersect3(corners + 3, planes + 6, planes + 12, planes + 24);
	orc_intersect3(corners + 6, planes + 6, planes + 18, planes + 24);
	orc_intersect3(corners + 9, planes + 0, planes + 18, planes + 24);
	orc_intersect3(corners + 12, planes + 0, planes + 12, planes + 30);
	orc_intersect3(corners + 15, planes + 6, planes + 12, planes + 30);
	orc_intersect3(corners + 18, planes + 6, planes + 18, planes + 30);
	orc_intersect3(corners + 21, planes + 0, planes + 18, planes + 30);
}

/* per_frame_funcs.h:18-39 fillCoordinatesData */
void orc_camcoords(const float *worldori, const float *corners, const float *MV, const float *P,
		   const float *MVP, float *cc)
{
	int i;
	for (i = 0; i < 4; i++)
		cc[i] = worldori[i];
	for (i = 0; i < 4; i++) {
		cc[i * 3 + 0 + 4] = corners[i * 3 + 0];
		cc[i * 3 + 1 + 4] = corners[i * 3 + 1];
		cc[i * 3 + 2 + 4] = corners[i * 3 + 2];
	}
	for (i = 0; i < 16; i++)
		cc[16 + i] = MV[i];
	for (i = 0; i < 16; i++)
		cc[32 + i] = P[i];
	for (i = 0; i < 16; i++)
		cc[48 + i] = MVP[i];
}

/*
 * per_frame_funcs.h:161-419 setDirectionTexture: the 5x5 node table.
 * node(j,i) = a_i + (j/4)(b_i - a_i), a_i = c0 + (i/4)(c1 - c0), b_i = c3 + (i/4)(c2 - c3)
 * with the reference's C arithmetic: float differences, double products and
 * sums, every stored value rounded to float.  tex is [5][5][4] (w = 0).
 */
static float orc_lerp_host(float a, float b, double w)
{
	return (float)((double)a + w * (double)(float)(b - a));
}

void orc_dirtex(const float *cc, float *tex)
{
	static const double wq[5] = { 0.0, 0.25, 0.5, 0.75, 1.0 };
	int i, j, k;
	for (j = 0; j < 5; j++)
		for (i = 0; i < 5; i++)
			for (k = 0; k < 3; k++) {
				float c0 = cc[4 + k], c1 = cc[7 + k], c2 = cc[10 + k], c3 = cc[13 + k];
				float a = (i == 0) ? c0 : (i == 4) ? c1 : orc_lerp_host(c0, c1, wq[i]);
				float b = (i == 0) ? c3 : (i == 4) ? c2 : orc_lerp_host(c3, c2, wq[i]);
				float v = (j == 0) ? a : (j == 4) ? b : orc_lerp_host(a, b, wq[j]);
				tex[(j * 5 + i) * 4 + k] = v;
			}
	for (j = 0; j < 25; j++)
		tex[j * 4 + 3] = 0.0f;
}

/* ------------------------------------------------------------------------- */
/* Shared device helpers                                                      */
/* ------------------------------------------------------------------------- */

/* grid_kernel.cu:4-11 mulMatrixVector_D */
static void orc_mulmv(float *result, const float *mat, const float *vec)
{
	result[0] = mat[0] * vec[0] + mat[4] * vec[1] + mat[8] * vec[2] + mat[12] * vec[3];
	result[1] = mat[1] * vec[0] + mat[5] * vec[1] + mat[9] * vec[2] + mat[13] * vec[3];
	result[2] = mat[2] * vec[0] + mat[6] * vec[1] + mat[10] * vec[2] + mat[14] * vec[3];
	result[3] = mat[3] * vec[0] + mat[7] * vec[1] + mat[11] * vec[2] + mat[15] * vec[3];
}

/* grid_kernel.cu:13-36 getTransformedVertex (vert_mv is unused downstream) */
static void orc_transformed_vertex(const float *cc, const float *vertlist, int faceID, float *vert_mvp)
{
	float point[4], tmp[4];
	point[0] = vertlist[faceID + 0];
	point[1] = vertlist[faceID + 1];
	point[2] = vertlist[faceID + 2];
	point[3] = 1.0f;
	orc_mulmv(tmp, &cc[16], point);
	point[0] = tmp[0] / tmp[3];
	point[1] = tmp[1] / tmp[3];
	point[2] = tmp[2] / tmp[3];
	point[3] = 1.0f;
	orc_mulmv(tmp, &cc[32], point);
	vert_mvp[0] = tmp[0] / tmp[3];
	vert_mvp[1] = tmp[1] / tmp[3];
	vert_mvp[2] = tmp[2] / tmp[3];
}

/* grid_kernel.cu:132-146 min_d / max_d (float, 3 args) */
static float orc_min3(float e1, float e2, float e3)
{
	return (e1 < e2) ? ((e1 < e3) ? e1 : e3) : ((e2 < e3) ? e2 : e3);
}
static float orc_max3(float e1, float e2, float e3)
{
	return (e1 > e2) ? ((e1 > e3) ? e1 : e3) : ((e2 > e3) ? e2 : e3);
}
static int orc_imin3(int e1, int e2, int e3)
{
	return (e1 < e2) ? ((e1 < e3) ? e1 : e3) : ((e2 < e3) ? e2 : e3);
}
static int orc_imax3(int e1, int e2, int e3)
{
	return (e1 > e2) ? ((e1 > e3) ? e1 : e3) : ((e2 > e3) ? e2 : e3);
}
static int orc_clampi(int v, int lo, int hi)
{
	if (v < lo)
		v = lo;
	if (v > hi)
		v = hi;
	return v;
}

/* grid_kernel.cu:354-363 getMagnitude */
static float orc_magnitude(const float *vec)
{
	float rad = 0;
	int i;
	for (i = 0; i < 3; i++)
		rad += vec[i] * vec[i];
	return __builtin_sqrtf(rad);
}

/* grid_kernel.cu:395-422 getEffective_x ; nbx2 = NUM_BLOCKS_X / 2 */
static unsigned orc_effective_x(const float *cc, const float *vec, float max, int nbx2)
{
	float upDotValue = vec[0] * cc[16 + 1] + vec[1] * cc[16 + 5] + vec[2] * cc[16 + 9];
	float tmp[3], val, forwardDotValue, angle, rightDotValue;
	unsigned blx;
	tmp[0] = vec[0] - upDotValue * cc[16 + 1];
	tmp[1] = vec[1] - upDotValue * cc[16 + 5];
	tmp[2] = vec[2] - upDotValue * cc[16 + 9];
	val = orc_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] + tmp[2] * cc[16 + 10];
	angle = ugrt_acosf(forwardDotValue);
	rightDotValue = tmp[0] * cc[16 + 0] + tmp[1] * cc[16 + 4] + tmp[2] * cc[16 + 8];
	if (rightDotValue > 0)
		blx = (unsigned)(nbx2 + ugrt_f2i((angle / max) * (float)nbx2));
	else
		blx = (unsigned)(nbx2 - ugrt_f2i((angle / max) * (float)nbx2));
	return blx;
}

/* grid_kernel.cu:452-479 getEffective_y, including the `*` typo at :468 */
static unsigned orc_effective_y(const float *cc, const float *vec, float max, int nby2)
{
	float tmp[3], val, upDotValue, forwardDotValue, angle;
	float rightDotValue = vec[0] * cc[16 + 0] + vec[1] * cc[16 + 4] + vec[2] * cc[16 + 8];
	unsigned bly;
	tmp[0] = vec[0] - rightDotValue * cc[16 + 0];
	tmp[1] = vec[1] - rightDotValue * cc[16 + 4];
	tmp[2] = vec[2] - rightDotValue * cc[16 + 8];
	val = orc_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	upDotValue = tmp[0] * cc[16 + 1] + tmp[1] * cc[16 + 5] + tmp[2] * cc[16 + 9];
	forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] * tmp[2] * cc[16 + 10];
	angle = ugrt_acosf(forwardDotValue);
	if (upDotValue > 0)
		bly = ugrt_f2u((float)nby2 + (angle / max) * (float)nby2);
	else
		bly = ugrt_f2u((float)nby2 - (angle / max) * (float)nby2);
	return bly;
}

/* ------------------------------------------------------------------------- */
/* Grid build (frustum_grid.h:210-532, grid_kernel.cu, misc_kernel.cu:4-60)   */
/* ------------------------------------------------------------------------- */

/*
 * DSKernel (grid_kernel.cu:164-243) for every triangle: clamped cell range on
 * the nbx x nby screen grid and min ndc z.  rng[f] = {gxmin,gxmax,gymin,gymax}.
 * [gy_lo, gy_hi) restricts binning to a band of tile rows (multi-GPU image
 * sharding, not in the reference): a triangle outside the band gets size 0;
 * the reference is gy_lo = 0, gy_hi = nby.  NUM_SLABS is 1 (main.cu.h:18).
 */
void orc_persp_ranges(const float *cc, const int *facelist, const float *vertexlist, int F, int nbx,
		      int nby, int gy_lo, int gy_hi, int *rng, u32 *sizeList, float *projCoordZ)
{
	int curface;
#pragma omp parallel for schedule(static)
	for (curface = 0; curface < F; curface++) {
		float v1[3], v2[3], v3[3];
		float xmin, xmax, ymin, ymax, zmin;
		int gxmin, gxmax, gymin, gymax;
		int face1 = 3 * facelist[curface * 3 + 0];
		int face2 = 3 * facelist[curface * 3 + 1];
		int face3 = 3 * facelist[curface * 3 + 2];
		orc_transformed_vertex(cc, vertexlist, face1, v1);
		orc_transformed_vertex(cc, vertexlist, face2, v2);
		orc_transformed_vertex(cc, vertexlist, face3, v3);
		xmin = orc_min3(v1[0], v2[0], v3[0]);
		ymin = orc_min3(v1[1], v2[1], v3[1]);
		zmin = orc_min3(v1[2], v2[2], v3[2]);
		xmax = orc_max3(v1[0], v2[0], v3[0]);
		ymax = orc_max3(v1[1], v2[1], v3[1]);
		gxmin = ugrt_floor2i(((xmin + 1.0f) / 2.0f) * (float)nbx);
		gymin = ugrt_floor2i(((ymin + 1.0f) / 2.0f) * (float)nby);
		gxmax = ugrt_floor2i(((xmax + 1.0f) / 2.0f) * (float)nbx);
		gymax = ugrt_floor2i(((ymax + 1.0f) / 2.0f) * (float)nby);
		gxmin = orc_clampi(gxmin, 0, nbx - 1);
		gymin = orc_clampi(gymin, 0, nby - 1);
		gxmax = orc_clampi(gxmax, 0, nbx - 1);
		gymax = orc_clampi(gymax, 0, nby - 1);
		if (projCoordZ)
			projCoordZ[curface] = zmin;
		if (gymax < gy_lo || gymin >= gy_hi) {
			/* outside this rank's band */
			rng[curface * 4 + 0] = 0;
			rng[curface * 4 + 1] = -1;
			rng[curface * 4 + 2] = 0;
			rng[curface * 4 + 3] = -1;
			sizeList[curface] = 0;
			continue;
		}
		if (gymin < gy_lo)
			gymin = gy_lo;
		if (gymax > gy_hi - 1)
			gymax = gy_hi - 1;
		rng[curface * 4 + 0] = gxmin;
		rng[curface * 4 + 1] = gxmax;
		rng[curface * 4 + 2] = gymin;
		rng[curface * 4 + 3] = gymax;
		sizeList[curface] = (u32)((gxmax - gxmin + 1) * (gymax - gymin + 1));
	}
}

/* DS_spherical_Kernel (grid_kernel.cu:481-659): light-space angular bins. */
void orc_sph_ranges(const float *cc, const int *facelist, const float *vertexlist, int F, int nbx,
		    int nby, float xM, float yM, int *rng, u32 *sizeList, float *projCoordZ)
{
	int curface;
#pragma omp parallel for schedule(static)
	for (curface = 0; curface < F; curface++) {
		int bl[3][2], k;
		float radius[3];
		int gxmin, gxmax, gymin, gymax, size_x, size_y;
		for (k = 0; k < 3; k++) {
			float point[3];
			int face = 3 * facelist[curface * 3 + k];
			point[0] = vertexlist[face + 0];
			point[1] = vertexlist[face + 1];
			point[2] = vertexlist[face + 2];
			point[0] -= cc[0];
			point[1] -= cc[1];
			point[2] -= cc[2];
			radius[k] = orc_magnitude(point);
			point[0] /= radius[k];
			point[1] /= radius[k];
			point[2] /= radius[k];
			bl[k][0] = (int)orc_effective_x(cc, point, xM, nbx / 2);
			bl[k][1] = (int)orc_effective_y(cc, point, yM, nby / 2);
		}
		gxmin = orc_imin3(bl[0][0], bl[1][0], bl[2][0]);
		gymin = orc_imin3(bl[0][1], bl[1][1], bl[2][1]);
		gxmax = orc_imax3(bl[0][0], bl[1][0], bl[2][0]);
		gymax = orc_imax3(bl[0][1], bl[1][1], bl[2][1]);
		gxmin = orc_clampi(gxmin, 0, nbx - 1);
		gymin = orc_clampi(gymin, 0, nby - 1);
		gxmax = orc_clampi(gxmax, 0, nbx - 1);
		gymax = orc_clampi(gymax, 0, nby - 1);
		/* grid_kernel.cu:634-644: both branches equal max - min + 1 */
		size_x = gxmax - gxmin + 1;
		size_y = gymax - gymin + 1;
		rng[curface * 4 + 0] = gxmin;
		rng[curface * 4 + 1] = gxmax;
		rng[curface * 4 + 2] = gymin;
		rng[curface * 4 + 3] = gymax;
		sizeList[curface] = (u32)(size_x * size_y);
		if (projCoordZ)
			projCoordZ[curface] = orc_min3(radius[0], radius[1], radius[2]);
	}
}

/* cudppScan FORWARD|INCLUSIVE add u32 (frustum_grid.h:249); returns total. */
u32 orc_inclusive_scan(const u32 *in, u32 *out, int n)
{
	u32 s = 0;
	int i;
	for (i = 0; i < n; i++) {
		s += in[i];
		out[i] = s;
	}
	return s;
}

/* DSFillkernel (grid_kernel.cu:245-332) / DS_spherical_Fillkernel (:661-854):
 * key = (gx*nby + gy)*NUM_SLABS + 0, value = triangle, x-major inside a triangle. */
void orc_fill_2d(const int *rng, const u32 *scanList, int F, int nby, u32 *keyList, u32 *valueList)
{
	int curface;
#pragma omp parallel for schedule(dynamic, 1024)
	for (curface = 0; curface < F; curface++) {
		u32 offset = curface ? scanList[curface - 1] : 0;
		int gxmin = rng[curface * 4 + 0], gxmax = rng[curface * 4 + 1];
		int gymin = rng[curface * 4 + 2], gymax = rng[curface * 4 + 3];
		int size_x = gxmax - gxmin + 1, size_y = gymax - gymin + 1, i, j;
		if (scanList[curface] == offset)
			continue;
		for (i = 0; i < size_x; i++)
			for (j = 0; j < size_y; j++) {
				keyList[offset + i * size_y + j] = (u32)((gxmin + i) * nby + (gymin + j));
				valueList[offset + i * size_y + j] = (u32)curface;
			}
	}
}

/*
 * Host z-range loop of buildGrid (frustum_grid.h:221-241: zMin = +2, zMax = -2) and of
 * buildSphericalGrid (:384-404: 9999.9 / -9999.9): zMin = smallest non-negative value, zMax = largest.
 */
void orc_zrange(const float *projCoordZ, int F, float zmin_init, float zmax_init, float *zMin_out, float *zMax_out)
{
	float zMin = zmin_init, zMax = zmax_init;
	int i;
	for (i = 0; i < F; i++) {
		if (zMin > projCoordZ[i] && projCoordZ[i] >= 0.0f)
			zMin = projCoordZ[i];
		if (zMax < projCoordZ[i])
			zMax = projCoordZ[i];
	}
	*zMin_out = zMin;
	*zMax_out = zMax;
}

/*
 * SlabKernel (grid_kernel.cu:334-352).  The reference leaves zList[f] untouched for a negative
 * depth (uninitialised memory); here such a triangle keeps the caller's value, which
 * the build defines as 0.
 */
void orc_slab_kernel(const float *projCoordZ, int F, int slabs, float zMin, float zMax, u32 *zList)
{
	int curface;
	for (curface = 0; curface < F; curface++) {
		float pCoord = projCoordZ[curface];
		if (pCoord >= 0.0f) {
			unsigned binID = ugrt_f2u((float)slabs * (float)((pCoord - zMin) / (zMax - zMin)));
			if (binID >= (unsigned)slabs)
				binID = (unsigned)slabs - 1u;
			zList[curface] = binID;
		}
	}
}

/* DSFillkernel / DS_spherical_Fillkernel with NUM_SLABS > 1 (grid_kernel.cu:310,322 / :819,846):
 * key = ((gx*nby + gy)*slabs + clamp(zSlabs[f])). */
void orc_fill_slabs(const int *rng, const u32 *scanList, const u32 *zList, int F, int nby, int slabs, u32 *keyList,
		    u32 *valueList)
{
	int curface;
#pragma omp parallel for schedule(dynamic, 1024)
	for (curface = 0; curface < F; curface++) {
		u32 offset = curface ? scanList[curface - 1] : 0;
		int gxmin = rng[curface * 4 + 0], gxmax = rng[curface * 4 + 1];
		int gymin = rng[curface * 4 + 2], gymax = rng[curface * 4 + 3];
		int size_x = gxmax - gxmin + 1, size_y = gymax - gymin + 1, i, j;
		int gzmin = orc_clampi((int)zList[curface], 0, slabs - 1);
		if (scanList[curface] == offset)
			continue;
		for (i = 0; i < size_x; i++)
			for (j = 0; j < size_y; j++) {
				keyList[offset + i * size_y + j] = (u32)(((gxmin + i) * nby + (gymin + j)) * slabs + gzmin);
				valueList[offset + i * size_y + j] = (u32)curface;
			}
	}
}

/* cudppSort key-value radix (frustum_grid.h:298, decision_data.h:177):
 * any STABLE sort by key is the same function; counting sort on keys < nkeys. */
int orc_stable_sort_pairs(u32 *keys, u32 *values, u32 n, u32 nkeys)
{
	u32 *cnt = (u32 *)calloc((size_t)nkeys + 1, sizeof(u32));
	u32 *k2 = (u32 *)malloc((size_t)(n ? n : 1) * sizeof(u32));
	u32 *v2 = (u32 *)malloc((size_t)(n ? n : 1) * sizeof(u32));
	u32 i, s = 0;
	if (!cnt || !k2 || !v2)
		return -1;
	for (i = 0; i < n; i++) {
		if (keys[i] >= nkeys) {
			free(cnt);
			free(k2);
			free(v2);
			return -2;
		}
		cnt[keys[i]]++;
	}
	for (i = 0; i < nkeys; i++) {
		u32 c = cnt[i];
		cnt[i] = s;
		s += c;
	}
	for (i = 0; i < n; i++) {
		u32 d = cnt[keys[i]]++;
		k2[d] = keys[i];
		v2[d] = values[i];
	}
	memcpy(keys, k2, (size_t)n * sizeof(u32));
	memcpy(values, v2, (size_t)n * sizeof(u32));
	free(cnt);
	free(k2);
	free(v2);
	return 0;
}

/*
 * do_scan_dump (misc_kernel.cu:4-24) -> cudppCompact (frustum_grid.h:334) ->
 * set_as_zero (:26) -> create_histogram (:36-60) -> cudppScan exclusive
 * (frustum_grid.h:361).  Returns cells_used.
 */
u32 orc_cell_boundaries(const u32 *key_list, u32 R, u32 C, u32 *span, u32 *offset)
{
	u32 *pos = (u32 *)malloc((size_t)(R ? R : 1) * sizeof(u32));
	u32 *flag = (u32 *)malloc((size_t)(R ? R : 1) * sizeof(u32));
	u32 *compacted = (u32 *)malloc((size_t)(R ? R : 1) * sizeof(u32));
	u32 i, used = 0, s = 0;
	for (i = 0; i < R; i++) {
		pos[i] = i;
		flag[i] = (i == 0) ? 1u : (key_list[i] != key_list[i - 1]);
	}
	for (i = 0; i < R; i++)
		if (flag[i])
			compacted[used++] = pos[i];
	for (i = 0; i < C; i++)
		span[i] = 0;
	for (i = 0; i < used; i++) {
		u32 sp = (i == used - 1) ? R - compacted[i] : compacted[i + 1] - compacted[i];
		span[key_list[compacted[i]]] = sp;
	}
	for (i = 0; i < C; i++) {
		offset[i] = s;
		s += span[i];
	}
	free(pos);
	free(flag);
	free(compacted);
	return used;
}

/* ------------------------------------------------------------------------- */
/* Primary rays (trace_kernel.cu:4-270)                                       */
/* ------------------------------------------------------------------------- */

/* trace_kernel.cu:4-45 intersectTriUV */
static float orc_intersect_tri_uv(const float *tvec, const float *edge1, const float *edge2,
				  const float *dir, float oldt)
{
	float u, v, t, pvec[3], qvec[3], det, inv_det, retValue = 0;
	CROSS(pvec, dir, edge2);
	det = DOT(edge1, pvec);
	if (det > -ORC_EPSILON && det < ORC_EPSILON)
		return 0;
	inv_det = 1.0f / det;
	u = DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return 0;
	CROSS(qvec, tvec, edge1);
	v = DOT(dir, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f)
		return 0;
	t = DOT(edge2, qvec) * inv_det;
	if (t < 0)
		t *= -1;
	if (t < oldt && t > 0)
		retValue = t;
	return retValue;
}

/*
 * Ray through pixel (col,row): trace_kernel.cu:96-114.  The reference reads a
 * 5x5 float4 texture with normalized coordinates ftx*0.8+0.1 and hardware
 * bilinear filtering.  The texture unit's fixed-point filter is not
 * reproducible; the ABI defines the sample as the exact float bilinear
 * interpolation below (texel coordinate 4*ftx; at W = 1024 the weights are the
 * same k/256 values the hardware filter uses).  See DESIGN.md "ray set-up".
 */
static int g_strict_texture = 0;
/* UGRT_FLAG_STRICT_TEXTURE for every ray set-up that follows (test infrastructure: a process-wide switch) */
void orc_set_strict_texture(int on)
{
	g_strict_texture = on != 0;
}

void orc_ray_dir(const float *cc, const float *tex, int col, int row, int W, int H, float *ray_direction)
{
	float ftx = (float)col / (float)W;
	float fty = (float)row / (float)H;
	float xs, ys, a, b, w00, w10, w01, w11;
	int i, j, k;
	ftx = 1 - ftx;
	xs = ftx * 4.0f;
	ys = fty * 4.0f;
	i = ugrt_f2i(xs);
	j = ugrt_f2i(ys);
	if (i > 3)
		i = 3;
	if (j > 3)
		j = 3;
	a = xs - (float)i;
	b = ys - (float)j;
	if (g_strict_texture) {
		ugrt_tex_linear8(ftx * 0.8f + 0.1f, 5, &i, &a);
		ugrt_tex_linear8(fty * 0.8f + 0.1f, 5, &j, &b);
	}
	w00 = (1.0f - a) * (1.0f - b);
	w10 = a * (1.0f - b);
	w01 = (1.0f - a) * b;
	w11 = a * b;
	for (k = 0; k < 3; k++) {
		float t00 = tex[((j)*5 + i) * 4 + k], t10 = tex[((j)*5 + i + 1) * 4 + k];
		float t01 = tex[((j + 1) * 5 + i) * 4 + k], t11 = tex[((j + 1) * 5 + i + 1) * 4 + k];
		float T = ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;
		ray_direction[k] = T - cc[k];
	}
	NORMALIZE(ray_direction);
}

/*
 * rckernel_alpha (trace_kernel.cu:84-270), NUM_SLABS = 1.  One tile of 8x8
 * pixels per cell; cell id = bx*nby + by (:138).  Tiles by in [gy_lo, gy_hi).
 * counters (optional, 2 x u64): [0] Moller-Trumbore tests, [1] refs staged.
 */
void orc_trace_primary(const float *cc, const float *tex, int W, int H, int nbx, int nby, int gy_lo,
		       int gy_hi, const u32 *value_list, const u32 *span, const u32 *offset,
		       const float *vertlist, const int *trilist, float *normal_out, float *t_out,
		       float *dir_out, int *shadowed_out, int *id_out, unsigned long long *counters)
{
	int tile;
	int ntiles = nbx * (gy_hi - gy_lo);
	unsigned long long tests = 0, refs = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tests, refs)
	for (tile = 0; tile < ntiles; tile++) {
		int bx = tile / (gy_hi - gy_lo);
		int by = gy_lo + tile % (gy_hi - gy_lo);
		int cell = bx * nby + by;
		u32 sp = span[cell], off = offset[cell];
		int lane;
		refs += sp;
		for (lane = 0; lane < 64; lane++) {
			int tx = lane & 7, ty = lane >> 3;
			int col = bx * 8 + tx, rowp = by * 8 + ty;
			int pixelID = rowp * W + col;
			float ray_direction[3], e1[3] = { 0, 0, 0 }, e2[3] = { 0, 0, 0 };
			float oldt = 99999999.9f;
			int rayDone = 0;
			u32 tri_intersected = 0, r;
			orc_ray_dir(cc, tex, col, rowp, W, H, ray_direction);
			for (r = 0; r < sp; r++) {
				u32 curface = value_list[off + r];
				int face1 = 3 * trilist[curface * 3 + 0];
				int face2 = 3 * trilist[curface * 3 + 1];
				int face3 = 3 * trilist[curface * 3 + 2];
				float vd[9], value;
				/* trace_kernel.cu:159-173 staging */
				vd[0] = vertlist[face1 + 0];
				vd[1] = vertlist[face1 + 1];
				vd[2] = vertlist[face1 + 2];
				vd[3] = vertlist[face2 + 0] - vd[0];
				vd[4] = vertlist[face2 + 1] - vd[1];
				vd[5] = vertlist[face2 + 2] - vd[2];
				vd[6] = vertlist[face3 + 0] - vd[0];
				vd[7] = vertlist[face3 + 1] - vd[1];
				vd[8] = vertlist[face3 + 2] - vd[2];
				vd[0] = cc[0] - vd[0];
				vd[1] = cc[1] - vd[1];
				vd[2] = cc[2] - vd[2];
				value = orc_intersect_tri_uv(&vd[0], &vd[3], &vd[6], ray_direction, oldt);
				tests++;
				if (value) {
					oldt = value;
					rayDone = 1;
					tri_intersected = off + r;
					e1[0] = vd[3];
					e1[1] = vd[4];
					e1[2] = vd[5];
					e2[0] = vd[6];
					e2[1] = vd[7];
					e2[2] = vd[8];
				}
			}
			/* trace_kernel.cu:56-82 isWithin, slab 0 of 1 */
			if (rayDone == 1) {
				float point[4], tmp[4];
				int z_value;
				point[0] = cc[0] + oldt * ray_direction[0];
				point[1] = cc[1] + oldt * ray_direction[1];
				point[2] = cc[2] + oldt * ray_direction[2];
				point[3] = 1.0f;
				orc_mulmv(tmp, &cc[48], point);
				tmp[2] /= tmp[3];
				z_value = ugrt_floor2i(tmp[2] * 1.0f);
				rayDone = (z_value == 0) ? 2 : 1;
			}
			if (rayDone == 2) {
				float nrm[3];
				NORMALIZE(e1);
				NORMALIZE(e2);
				CROSS(nrm, e1, e2);
				NORMALIZE(nrm);
				if (nrm[0] < 0)
					nrm[0] *= -1;
				if (nrm[1] < 0)
					nrm[1] *= -1;
				if (nrm[2] < 0)
					nrm[2] *= -1;
				t_out[pixelID] = oldt;
				shadowed_out[pixelID] = 0;
				id_out[pixelID] = (int)value_list[tri_intersected];
				normal_out[pixelID * 3 + 0] = nrm[0];
				normal_out[pixelID * 3 + 1] = nrm[1];
				normal_out[pixelID * 3 + 2] = nrm[2];
			} else {
				t_out[pixelID] = -1.0f;
				shadowed_out[pixelID] = 0;
				id_out[pixelID] = -2;
				normal_out[pixelID * 3 + 0] = -1.0f;
				normal_out[pixelID * 3 + 1] = -1.0f;
				normal_out[pixelID * 3 + 2] = -1.0f;
			}
			dir_out[pixelID * 3 + 0] = ray_direction[0];
			dir_out[pixelID * 3 + 1] = ray_direction[1];
			dir_out[pixelID * 3 + 2] = ray_direction[2];
		}
	}
	if (counters) {
		counters[0] = tests;
		counters[1] = refs;
	}
}

/*
 * rckernel_alpha with NUM_SLABS = slabs > 1 (trace_kernel.cu:84-270).  Per tile the slabs are walked front
 * to back; a hit is accepted (rayDone = 2) in the slab whose index equals floor(ndc_z * slabs) (isWithin,
 * :56-82); the tile stops when all 64 rays are accepted (:217-228).  Reproduced as written, including:
 *  - isWithin returns 0 for a ray that IS accepted (:59-62), so when the tile goes on (another ray is still
 *    open) the accepted ray falls back to "no hit" while keeping oldt, and ends as a miss unless a nearer
 *    triangle of a later slab is accepted;
 *  - triangles are binned by their slab relative to [zMin, zMax] (SlabKernel), hits are gated by the
 *    absolute ndc depth.
 */
void orc_trace_primary_slabs(const float *cc, const float *tex, int W, int H, int nbx, int nby, int gy_lo,
			     int gy_hi, int slabs, const u32 *value_list, const u32 *span, const u32 *offset,
			     const float *vertlist, const int *trilist, float *normal_out, float *t_out,
			     float *dir_out, int *shadowed_out, int *id_out)
{
	int tile;
	int ntiles = nbx * (gy_hi - gy_lo);
#pragma omp parallel for schedule(dynamic, 1)
	for (tile = 0; tile < ntiles; tile++) {
		int bx = tile / (gy_hi - gy_lo);
		int by = gy_lo + tile % (gy_hi - gy_lo);
		int cell = bx * nby + by;
		float ray_direction[64][3], e1[64][3], e2[64][3], oldt[64];
		int rayDone[64], lane, slab, beam_done = 0;
		u32 tri_intersected[64];
		for (lane = 0; lane < 64; lane++) {
			orc_ray_dir(cc, tex, bx * 8 + (lane & 7), by * 8 + (lane >> 3), W, H, ray_direction[lane]);
			oldt[lane] = 99999999.9f;
			rayDone[lane] = 0;
			tri_intersected[lane] = 0;
			e1[lane][0] = e1[lane][1] = e1[lane][2] = 0;
			e2[lane][0] = e2[lane][1] = e2[lane][2] = 0;
		}
		for (slab = 0; slab < slabs && !beam_done; slab++) {
			u32 sp = span[cell * slabs + slab], off = offset[cell * slabs + slab], r;
			int all_done = 1;
			for (lane = 0; lane < 64; lane++) {
				if (rayDone[lane] != 2) {
					for (r = 0; r < sp; r++) {
						u32 curface = value_list[off + r];
						int face1 = 3 * trilist[curface * 3 + 0];
						int face2 = 3 * trilist[curface * 3 + 1];
						int face3 = 3 * trilist[curface * 3 + 2];
						float vd[9], value;
						vd[0] = vertlist[face1 + 0];
						vd[1] = vertlist[face1 + 1];
						vd[2] = vertlist[face1 + 2];
						vd[3] = vertlist[face2 + 0] - vd[0];
						vd[4] = vertlist[face2 + 1] - vd[1];
						vd[5] = vertlist[face2 + 2] - vd[2];
						vd[6] = vertlist[face3 + 0] - vd[0];
						vd[7] = vertlist[face3 + 1] - vd[1];
						vd[8] = vertlist[face3 + 2] - vd[2];
						vd[0] = cc[0] - vd[0];
						vd[1] = cc[1] - vd[1];
						vd[2] = cc[2] - vd[2];
						value = orc_intersect_tri_uv(&vd[0], &vd[3], &vd[6], ray_direction[lane], oldt[lane]);
						if (value) {
							oldt[lane] = value;
							rayDone[lane] = 1;
							tri_intersected[lane] = off + r;
							e1[lane][0] = vd[3];
							e1[lane][1] = vd[4];
							e1[lane][2] = vd[5];
							e2[lane][0] = vd[6];
							e2[lane][1] = vd[7];
							e2[lane][2] = vd[8];
						}
					}
				}
				/* isWithin, trace_kernel.cu:56-82 */
				if (rayDone[lane] == 0 || rayDone[lane] == 2) {
					rayDone[lane] = 0;
				} else {
					float point[4], tmp[4];
					int z_value;
					point[0] = cc[0] + oldt[lane] * ray_direction[lane][0];
					point[1] = cc[1] + oldt[lane] * ray_direction[lane][1];
					point[2] = cc[2] + oldt[lane] * ray_direction[lane][2];
					point[3] = 1.0f;
					orc_mulmv(tmp, &cc[48], point);
					tmp[2] /= tmp[3];
					z_value = ugrt_floor2i(tmp[2] * (float)slabs);
					rayDone[lane] = (z_value == slab) ? 2 : 1;
				}
				if (rayDone[lane] != 2)
					all_done = 0;
			}
			beam_done = all_done;
		}
		for (lane = 0; lane < 64; lane++) {
			int pixelID = (by * 8 + (lane >> 3)) * W + bx * 8 + (lane & 7);
			if (rayDone[lane] == 2) {
				float nrm[3];
				NORMALIZE(e1[lane]);
				NORMALIZE(e2[lane]);
				CROSS(nrm, e1[lane], e2[lane]);
				NORMALIZE(nrm);
				if (nrm[0] < 0)
					nrm[0] *= -1;
				if (nrm[1] < 0)
					nrm[1] *= -1;
				if (nrm[2] < 0)
					nrm[2] *= -1;
				t_out[pixelID] = oldt[lane];
				shadowed_out[pixelID] = 0;
				id_out[pixelID] = (int)value_list[tri_intersected[lane]];
				normal_out[pixelID * 3 + 0] = nrm[0];
				normal_out[pixelID * 3 + 1] = nrm[1];
				normal_out[pixelID * 3 + 2] = nrm[2];
			} else {
				t_out[pixelID] = -1.0f;
				shadowed_out[pixelID] = 0;
				id_out[pixelID] = -2;
				normal_out[pixelID * 3 + 0] = -1.0f;
				normal_out[pixelID * 3 + 1] = -1.0f;
				normal_out[pixelID * 3 + 2] = -1.0f;
			}
			dir_out[pixelID * 3 + 0] = ray_direction[lane][0];
			dir_out[pixelID * 3 + 1] = ray_direction[lane][1];
			dir_out[pixelID * 3 + 2] = ray_direction[lane][2];
		}
	}
}

/* ------------------------------------------------------------------------- */
/* Shadow rays: mapping, re-ordering, trace                                   */
/* ------------------------------------------------------------------------- */

/*
 * mapSort_Effective_kernel (misc_kernel.cu:255-296).  cc = LIGHT camera block.
 * Pixels [p0, p0+n): d_map[i] = pixel id, d_map[n+i] = light cell or the
 * sentinel lnbx*lnby.  (The reference has p0 = 0, n = IMAGE_SIZE.)
 */
void orc_map_rays(const float *cc, const float *t_value_list, const float *ray_direction,
		  const float *cmPt, float xM, float yM, int lnbx, int lnby, int p0, int n, u32 *d_map)
{
	int i;
#pragma omp parallel for schedule(static)
	for (i = 0; i < n; i++) {
		int pixelId = p0 + i;
		float ptIntersection[3], lightRayDirection[3];
		float tVal = t_value_list[pixelId];
		int blx, bly, blockIndex;
		ptIntersection[0] = cmPt[0] + tVal * ray_direction[pixelId * 3 + 0];
		ptIntersection[1] = cmPt[1] + tVal * ray_direction[pixelId * 3 + 1];
		ptIntersection[2] = cmPt[2] + tVal * ray_direction[pixelId * 3 + 2];
		lightRayDirection[0] = ptIntersection[0] - cc[0];
		lightRayDirection[1] = ptIntersection[1] - cc[1];
		lightRayDirection[2] = ptIntersection[2] - cc[2];
		NORMALIZE(lightRayDirection);
		blx = (int)orc_effective_x(cc, lightRayDirection, xM, lnbx / 2);
		bly = (int)orc_effective_y(cc, lightRayDirection, yM, lnby / 2);
		if (blx >= 0 && blx < lnbx && bly >= 0 && bly < lnby)
			blockIndex = blx * lnby + bly;
		else
			blockIndex = lnbx * lnby;
		d_map[i] = (u32)pixelId;
		d_map[n + i] = (u32)blockIndex;
	}
}

/*
 * processData (per_frame_funcs.h:116-137) = DecisionData::sort_by_block_indices
 * .. stream_compact (decision_data.h:171-271) with blockScan /
 * preStreamCompaction / tag_thread (misc_kernel.cu:298-333).
 * d_map is sorted in place; prefixMap gets the start index of every <=64-ray
 * chunk; returns the number of chunks ("numCudaBlocks").
 */
u32 orc_process_rays(u32 *d_map, int n, u32 nkeys, u32 *prefixMap, u32 prefix_cap)
{
	u32 *valid = (u32 *)malloc((size_t)(n ? n : 1) * sizeof(u32));
	u32 *segscan = (u32 *)malloc((size_t)(n ? n : 1) * sizeof(u32));
	u32 count = 0;
	int i;
	orc_stable_sort_pairs(d_map + n, d_map, (u32)n, nkeys);
	/* blockScan: head flags + all-ones */
	for (i = 0; i < n; i++)
		valid[i] = (i == 0) ? 1u : (d_map[n + i] != d_map[n + i - 1]);
	/* cudppSegmentedScan inclusive add of ones: 1-based rank inside the run */
	for (i = 0; i < n; i++)
		segscan[i] = valid[i] ? 1u : segscan[i - 1] + 1u;
	/* preStreamCompaction (threshold 64) + tag_thread + cudppCompact */
	for (i = 0; i < n; i++)
		if (segscan[i] % 64u == 1u) {
			if (count < prefix_cap)
				prefixMap[count] = (u32)i;
			count++;
		}
	free(valid);
	free(segscan);
	return count;
}

/* light_kernel.cu:13-50 intersectTri (signed t) */
static float orc_intersect_tri(const float *tvec, const float *edge1, const float *edge2,
			       const float *dir, float oldt)
{
	float u, v, t, pvec[3], qvec[3], det, inv_det, retValue = 0;
	CROSS(pvec, dir, edge2);
	det = DOT(edge1, pvec);
	if (det > -ORC_EPSILON && det < ORC_EPSILON)
		return 0;
	inv_det = 1.0f / det;
	u = DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return 0;
	CROSS(qvec, tvec, edge1);
	v = DOT(dir, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f)
		return 0;
	t = DOT(edge2, qvec) * inv_det;
	if (t < oldt)
		retValue = t;
	return retValue;
}

/* light_kernel.cu:1-11 isSmaller */
static int orc_is_smaller(const float *a, const float *b, const float *ref)
{
	float epsilon = 1e-03f;
	float distance_a = __builtin_sqrtf((a[0] - ref[0]) * (a[0] - ref[0]) + (a[1] - ref[1]) * (a[1] - ref[1]) +
					   (a[2] - ref[2]) * (a[2] - ref[2]));
	float distance_b = __builtin_sqrtf((b[0] - ref[0]) * (b[0] - ref[0]) + (b[1] - ref[1]) * (b[1] - ref[1]) +
					   (b[2] - ref[2]) * (b[2] - ref[2]));
	return (distance_a + epsilon < distance_b) ? 1 : 0;
}

/*
 * mod_light_rckernel (light_kernel.cu:52-270), NUM_SLABS = 1.  cc = LIGHT camera.
 * Chunk k = sorted rays [prefixMap[k], prefixMap[k+1]) (the last one ends at n).
 * strict != 0 reproduces the reference's launch: block b in [0, launch_blocks)
 * handles chunk b-1, only while b < nchunks (:76-85), so the last chunk and
 * every chunk >= launch_blocks-1 are never traced.
 * strict == 0 traces every chunk.  The sentinel cell (>= C) has span 0 (Q11).
 * is_shadowed is only ever set to 1.  counters: [0] MT tests, [1] refs staged.
 */
void orc_trace_shadow_slabs(const float *cc, const u32 *curflist, const float *vertlist, const int *trilist,
			    const u32 *blockcnt, const u32 *blockcntscan, u32 C, int slabs, const float *t_value_list,
			    const float *ray_direction_list, int *is_shadowed, const u32 *d_map,
			    const u32 *prefixmap, const float *cmPt, u32 nchunks, u32 launch_blocks, int n,
			    int strict, unsigned long long *counters);

void orc_trace_shadow(const float *cc, const u32 *curflist, const float *vertlist, const int *trilist,
		      const u32 *blockcnt, const u32 *blockcntscan, u32 C, const float *t_value_list,
		      const float *ray_direction_list, int *is_shadowed, const u32 *d_map,
		      const u32 *prefixmap, const float *cmPt, u32 nchunks, u32 launch_blocks, int n,
		      int strict, unsigned long long *counters)
{
	orc_trace_shadow_slabs(cc, curflist, vertlist, trilist, blockcnt, blockcntscan, C, 1, t_value_list,
			       ray_direction_list, is_shadowed, d_map, prefixmap, cmPt, nchunks, launch_blocks, n, strict,
			       counters);
}

/* NUM_SLABS = slabs: the block walks the lists of all slabs of its light cell (light_kernel.cu:105-113,
 * blockcnt[cell*NUM_SLABS + p]); a ray is shadowed as soon as any of them holds an occluder. */
void orc_trace_shadow_slabs(const float *cc, const u32 *curflist, const float *vertlist, const int *trilist,
			    const u32 *blockcnt, const u32 *blockcntscan, u32 C, int slabs, const float *t_value_list,
			    const float *ray_direction_list, int *is_shadowed, const u32 *d_map,
			    const u32 *prefixmap, const float *cmPt, u32 nchunks, u32 launch_blocks, int n,
			    int strict, unsigned long long *counters)
{
	long long k, kend;
	unsigned long long tests = 0, refs = 0;
	if (strict) {
		u32 lim = nchunks < launch_blocks ? nchunks : launch_blocks;
		kend = (long long)lim - 1; /* blocks 1..lim-1 -> chunks 0..lim-2 */
	} else {
		kend = nchunks;
	}
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tests, refs)
	for (k = 0; k < kend; k++) {
		u32 start = prefixmap[k];
		u32 end = (k + 1 < (long long)nchunks) ? prefixmap[k + 1] : (u32)n;
		u32 cell = d_map[n + start];
		u32 q, r;
		int p;
		for (p = 0; p < slabs; p++) {
		u32 sp = (cell < C) ? blockcnt[cell * (u32)slabs + (u32)p] : 0;
		u32 off = (cell < C) ? blockcntscan[cell * (u32)slabs + (u32)p] : 0;
		refs += sp;
		for (q = start; q < end; q++) {
			int pseudoPixelId = (int)d_map[q];
			float rayDirection[3], ptIntersection[3];
			float tVal = t_value_list[pseudoPixelId];
			if (is_shadowed[pseudoPixelId] == 1 && p > 0)
				continue; /* rayDoneMap == 2 from an earlier slab */
			rayDirection[0] = cmPt[0] + tVal * ray_direction_list[pseudoPixelId * 3 + 0];
			rayDirection[1] = cmPt[1] + tVal * ray_direction_list[pseudoPixelId * 3 + 1];
			rayDirection[2] = cmPt[2] + tVal * ray_direction_list[pseudoPixelId * 3 + 2];
			ptIntersection[0] = rayDirection[0];
			ptIntersection[1] = rayDirection[1];
			ptIntersection[2] = rayDirection[2];
			rayDirection[0] -= cc[0];
			rayDirection[1] -= cc[1];
			rayDirection[2] -= cc[2];
			NORMALIZE(rayDirection);
			for (r = 0; r < sp; r++) {
				u32 curface = curflist[off + r];
				int face1 = 3 * trilist[curface * 3 + 0];
				int face2 = 3 * trilist[curface * 3 + 1];
				int face3 = 3 * trilist[curface * 3 + 2];
				float vd[9], value;
				vd[0] = vertlist[face1 + 0];
				vd[1] = vertlist[face1 + 1];
				vd[2] = vertlist[face1 + 2];
				vd[3] = vertlist[face2 + 0] - vd[0];
				vd[4] = vertlist[face2 + 1] - vd[1];
				vd[5] = vertlist[face2 + 2] - vd[2];
				vd[6] = vertlist[face3 + 0] - vd[0];
				vd[7] = vertlist[face3 + 1] - vd[1];
				vd[8] = vertlist[face3 + 2] - vd[2];
				vd[0] = cc[0] - vd[0];
				vd[1] = cc[1] - vd[1];
				vd[2] = cc[2] - vd[2];
				value = orc_intersect_tri(&vd[0], &vd[3], &vd[6], rayDirection, 999999.9f);
				tests++;
				if (value) {
					float pt[3];
					pt[0] = cc[0] + value * rayDirection[0];
					pt[1] = cc[1] + value * rayDirection[1];
					pt[2] = cc[2] + value * rayDirection[2];
					if (orc_is_smaller(pt, ptIntersection, cc)) {
						is_shadowed[pseudoPixelId] = 1;
						break; /* rayDoneMap = 2: later hits change nothing */
					}
				}
			}
		}
		} /* slabs */
	}
	if (counters) {
		counters[0] = tests;
		counters[1] = refs;
	}
}

/* ------------------------------------------------------------------------- */
/* Shading (shader_kernel.cu) and PPM (per_app_funcs.h:39-66)                 */
/* ------------------------------------------------------------------------- */

/* shader_kernel.cu:46-86 lambert_color_pixel / :88-128 ..._drop_off_pixel
 * (drop_off = 1 reproduces the former bit for bit: x*1.0f == x). */
static void orc_lambert(const float *cc, const float *light_position, const float *point,
			const float *normal, float *color, const float *material, float drop_off,
			int use_drop_off)
{
	float light_dir[3], light_position_view[3], point_view[3], normal_view[3], dot_diffuse;
	const float light_ambient[3] = { 0.5f, 0.5f, 0.5f };
	const float light_diffuse[3] = { 1.0f, 1.0f, 1.0f };
	int k;
	for (k = 0; k < 3; k++) {
		light_position_view[k] = cc[16 + k] * light_position[0] + cc[16 + 4 + k] * light_position[1] +
					 cc[16 + 8 + k] * light_position[2];
		point_view[k] = cc[16 + k] * point[0] + cc[16 + 4 + k] * point[1] + cc[16 + 8 + k] * point[2];
		normal_view[k] = cc[16 + k] * normal[0] + cc[16 + 4 + k] * normal[1] + cc[16 + 8 + k] * normal[2];
	}
	NORMALIZE(normal_view);
	light_dir[0] = point_view[0] - light_position_view[0];
	light_dir[1] = point_view[1] - light_position_view[1];
	light_dir[2] = point_view[2] - light_position_view[2];
	NORMALIZE(light_dir);
	for (k = 0; k < 3; k++) {
		if (use_drop_off)
			color[k] += material[k] * light_ambient[k] * drop_off;
		else
			color[k] += material[k] * light_ambient[k];
	}
	dot_diffuse = DOT(light_dir, normal_view);
	if (dot_diffuse > 0)
		dot_diffuse *= 1;
	else
		dot_diffuse *= -1;
	if (dot_diffuse > 0) {
		for (k = 0; k < 3; k++) {
			if (use_drop_off)
				color[k] += material[3 + k] * light_diffuse[k] * dot_diffuse * drop_off;
			else
				color[k] += material[3 + k] * light_diffuse[k] * dot_diffuse;
		}
	}
}

static unsigned char orc_to_u8(float c)
{
	return (unsigned char)(ugrt_f2u(c * 255) & 0xFFu);
}

/* shader_kernel.cu:259-273 get_along_x / get_along_y (spot_shade's angles) */
static float orc_along_x(const float *cc, const float *vec)
{
	float upDotValue = vec[0] * cc[16 + 1] + vec[1] * cc[16 + 5] + vec[2] * cc[16 + 9];
	float tmp[3], val, forwardDotValue, angle, rightDotValue;
	tmp[0] = vec[0] - upDotValue * cc[16 + 1];
	tmp[1] = vec[1] - upDotValue * cc[16 + 5];
	tmp[2] = vec[2] - upDotValue * cc[16 + 9];
	val = orc_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] + tmp[2] * cc[16 + 10];
	angle = ugrt_acosf(forwardDotValue);
	rightDotValue = tmp[0] * cc[16 + 0] + tmp[1] * cc[16 + 4] + tmp[2] * cc[16 + 8];
	if (!(rightDotValue > 0))
		angle = -1.0f * angle;
	return angle;
}
static float orc_along_y(const float *cc, const float *vec)
{
	float tmp[3], val, upDotValue, forwardDotValue, angle;
	float rightDotValue = vec[0] * cc[16 + 0] + vec[1] * cc[16 + 4] + vec[2] * cc[16 + 8];
	tmp[0] = vec[0] - rightDotValue * cc[16 + 0];
	tmp[1] = vec[1] - rightDotValue * cc[16 + 4];
	tmp[2] = vec[2] - rightDotValue * cc[16 + 8];
	val = orc_magnitude(tmp);
	tmp[0] /= val;
	tmp[1] /= val;
	tmp[2] /= val;
	upDotValue = tmp[0] * cc[16 + 1] + tmp[1] * cc[16 + 5] + tmp[2] * cc[16 + 9];
	forwardDotValue = tmp[0] * cc[16 + 2] + tmp[1] * cc[16 + 6] * tmp[2] * cc[16 + 10];
	angle = ugrt_acosf(forwardDotValue);
	if (!(upDotValue > 0))
		angle = -1.0f * angle;
	return angle;
}

/*
 * lambertian_shade (shader_kernel.cu:165-221) when spot == 0,
 * spot_shade (:275-345) when spot != 0 (dump may be NULL).
 * cc = whatever camera block is current (the LIGHT camera in display(), Q17).
 * Misses (id < 0) read mat_idx[-2] in the reference; here they shade black and
 * keep their id.  Pixels [p0, p0+n).
 */
void orc_shade(const float *cc, const float *light_position, unsigned char *d_img, const float *dd_normal,
	       const float *dd_t_value, const float *dd_dir, int *dd_intersect_id, const float *d_cam_pos,
	       const int *mat_idx, const float *mat_list, int mat_count, int p0, int n, int spot, float *dump)
{
	int i;
	const float qpi = (float)(3.14159265358979323846 / 4);
#pragma omp parallel for schedule(static)
	for (i = 0; i < n; i++) {
		int pixelID = p0 + i, k;
		float material[6], color[3] = { 0.0f, 0.0f, 0.0f }, drop_off = 1.0f;
		int tri_intersected = dd_intersect_id[pixelID];
		int idx = (tri_intersected >= 0) ? mat_idx[tri_intersected] : tri_intersected;
		if (spot) {
			float ptIntersection[3], lightRayDirection[3], x, y;
			float tVal = dd_t_value[pixelID];
			ptIntersection[0] = d_cam_pos[0] + tVal * dd_dir[pixelID * 3 + 0];
			ptIntersection[1] = d_cam_pos[1] + tVal * dd_dir[pixelID * 3 + 1];
			ptIntersection[2] = d_cam_pos[2] + tVal * dd_dir[pixelID * 3 + 2];
			lightRayDirection[0] = ptIntersection[0] - cc[0];
			lightRayDirection[1] = ptIntersection[1] - cc[1];
			lightRayDirection[2] = ptIntersection[2] - cc[2];
			NORMALIZE(lightRayDirection);
			x = orc_along_x(cc, lightRayDirection);
			y = orc_along_y(cc, lightRayDirection);
			if (dump) {
				dump[pixelID * 2 + 0] = x;
				dump[pixelID * 2 + 1] = y;
			}
			if (x < qpi && x > -qpi && y < qpi && y > -qpi)
				drop_off = 1.0f;
			else
				drop_off = 0.25f;
		}
		dd_intersect_id[pixelID] = idx;
		if (idx >= 0 && idx < mat_count) {
			float point[3];
			float t_value = dd_t_value[pixelID];
			for (k = 0; k < 3; k++) {
				material[k] = mat_list[idx * 6 + 3 + k];
				material[3 + k] = mat_list[idx * 6 + 3 + k];
			}
			if (spot || t_value > 0) {
				point[0] = d_cam_pos[0] + t_value * dd_dir[pixelID * 3 + 0];
				point[1] = d_cam_pos[1] + t_value * dd_dir[pixelID * 3 + 1];
				point[2] = d_cam_pos[2] + t_value * dd_dir[pixelID * 3 + 2];
				orc_lambert(cc, light_position, point, &dd_normal[pixelID * 3], color, material,
					    drop_off, spot);
				for (k = 0; k < 3; k++)
					if (color[k] > 1.0f)
						color[k] = 1.0f;
			}
		}
		d_img[pixelID * 3 + 0] = orc_to_u8(color[0]);
		d_img[pixelID * 3 + 1] = orc_to_u8(color[1]);
		d_img[pixelID * 3 + 2] = orc_to_u8(color[2]);
	}
}

/* shader_kernel.cu:347-359 shadow_kernel */
void orc_add_shadows(unsigned char *d_img, const int *is_shadowed, int p0, int n)
{
	int i;
	for (i = 0; i < n; i++) {
		int pixelID = p0 + i;
		if (is_shadowed[pixelID] == 1) {
			d_img[pixelID * 3 + 0] /= 3;
			d_img[pixelID * 3 + 1] /= 3;
			d_img[pixelID * 3 + 2] /= 3;
		}
	}
}

/* shader_kernel.cu:4-44 Noise / InterPolation / PerlinNoise (octaves = 1) */
static float orc_noise(int x)
{
	u32 ux = (u32)x;
	ux = (ux << 13) ^ ux;
	ux = (ux * (ux * ux * 15731u + 789221u) + 1376312589u) & 0x7fffffffu;
	return (float)(int)ux / 2147483648.0f;
}
static float orc_interp(float a, float b, float c)
{
	return a + (b - a) * c * c * (3 - 2 * c);
}
static float orc_perlin(float x, float y, int width, int seed, float periode)
{
	float a, b, freq = 1.0f / periode, zone_x, zone_y;
	int num = ugrt_f2i((float)width * freq);
	int step_x = ugrt_f2i(x * freq), step_y = ugrt_f2i(y * freq);
	int box, noisedata;
	zone_x = x * freq - (float)step_x;
	zone_y = y * freq - (float)step_y;
	box = step_x + step_y * num;
	noisedata = box + seed;
	a = orc_interp(orc_noise(noisedata), orc_noise(noisedata + 1), zone_x);
	b = orc_interp(orc_noise(noisedata + num), orc_noise(noisedata + 1 + num), zone_x);
	return orc_interp(a, b, zone_y) * 324.0f;
}

/* shader_kernel.cu:505-547 perlin_noise_shade; InterLinear(a,b,c) = a*(1-c)+b*c */
void orc_shade_perlin(unsigned char *d_img, const int *dd_intersect_id, int W, int p0, int n)
{
	int i;
	for (i = 0; i < n; i++) {
		int pixelID = p0 + i;
		float x = (float)(pixelID % W), y = (float)(pixelID / W);
		float v1 = orc_perlin(x, y, 12413, 63, 100.0f), v2 = orc_perlin(x, y, 12413, 63, 25.0f);
		float v3 = orc_perlin(x, y, 12413, 63, 12.5f), v4 = orc_perlin(x, y, 12413, 63, 6.25f);
		float v5 = orc_perlin(x, y, 12413, 63, 3.125f), v6 = orc_perlin(x, y, 12413, 63, 1.56f);
		float tmp = (float)(ugrt_f2i(v1) + ugrt_f2i(v2 * 0.25f) + ugrt_f2i(v3 * 0.125f) +
				    ugrt_f2i(v4 * 0.0625f) + ugrt_f2i(v5 * 0.03125f) + ugrt_f2i(v6 * 0.0156f));
		int r = ugrt_f2i(tmp * (1 - 0.0f) + 0.0f * 0.0f);
		int g = ugrt_f2i(0.0f * (1 - 0.0f) + tmp * 0.0f);
		int b = ugrt_f2i(0.0f * (1 - tmp) + 0.0f * tmp);
		if (r > 255)
			r = 255;
		if (g > 255)
			g = 255;
		if (b > 255)
			b = 255;
		if (dd_intersect_id[pixelID] >= 0) {
			d_img[pixelID * 3 + 0] = (unsigned char)r;
			d_img[pixelID * 3 + 1] = (unsigned char)g;
			d_img[pixelID * 3 + 2] = (unsigned char)b;
		} else {
			d_img[pixelID * 3 + 0] = 0;
			d_img[pixelID * 3 + 1] = 0;
			d_img[pixelID * 3 + 2] = 0;
		}
	}
}

/* per_app_funcs.h:39-66 writePPM */
int orc_write_ppm(const char *filename, int W, int H, const unsigned char *h_image)
{
	FILE *fp = fopen(filename, "w");
	int i;
	if (!fp)
		return 1;
	fprintf(fp, "P3\n");
	fprintf(fp, "%d %d\n", W, H);
	fprintf(fp, "%d\n", 255);
	for (i = 0; i < (3 * W * H); i++) {
		if (i % (3 * W) == 0)
			fprintf(fp, "\n");
		fprintf(fp, "%d ", (int)(float)h_image[i]);
	}
	fprintf(fp, "\n");
	fclose(fp);
	return 0;
}

/* transformation_kernel.cu:4-18 copy_data_transform; cr = cosf(rot), sr = sinf(rot)
 * are evaluated once on the host and passed in (DESIGN.md "animation"). */
void orc_animate(float *vertexlist, const float *orig_list, int size, int offset, float cr, float sr)
{
	int vert;
	for (vert = 0; vert < size; vert++) {
		float x = ((orig_list[vert * 3 + 0] - 12.0f) / 12.0f);
		float y = ((orig_list[vert * 3 + 1] - 11.0f) / 12.0f);
		float z = ((orig_list[vert * 3 + 2] - 4.5f) / 12.0f);
		vertexlist[(offset + vert) * 3 + 0] = (x * cr - y * sr) * 9.0f + 14.5f;
		vertexlist[(offset + vert) * 3 + 1] = (x * sr + y * cr) * 9.0f + 13.0f;
		vertexlist[(offset + vert) * 3 + 2] = z * 9.0f + 4.0f;
	}
}

/* scene.h:370-439 Model::some_material: positional token stream.
 * Returns the number of materials, or -1 if the file cannot be opened.
 * mat_list may be NULL to query the count. */
int orc_parse_materials(const char *file, float *mat_list, int cap)
{
	FILE *fp = fopen(file, "r");
	char cjunk[512];
	int num_materials = 0, mt, i;
	if (!fp)
		return -1;
	while (fscanf(fp, "%511s", cjunk) != EOF)
		if (strcmp(cjunk, "newmtl") == 0)
			num_materials++;
	fclose(fp);
	if (!mat_list)
		return num_materials;
	fp = fopen(file, "r");
	for (mt = 0; mt < num_materials && mt < cap; mt++) {
		for (i = 0; i < 3; i++)
			if (fscanf(fp, "%511s", cjunk) != 1)
				break;
		for (i = 0; i < 3; i++)
			if (fscanf(fp, "%f", &mat_list[mt * 6 + i]) != 1)
				break;
		if (fscanf(fp, "%511s", cjunk) != 1)
			break;
		for (i = 0; i < 3; i++)
			if (fscanf(fp, "%f", &mat_list[mt * 6 + 3 + i]) != 1)
				break;
		for (i = 0; i < 12; i++)
			if (fscanf(fp, "%511s", cjunk) != 1)
				break;
	}
	fclose(fp);
	return num_materials;
}

/* ------------------------------------------------------------------------- */
/* Uniform grid + 3D-DDA reflection bounce.  NOT IN THE REFERENCE             */
/* (README.md:1 claims it; uniform_grid.h:196-350 is a dead copy of the       */
/* perspective builder).  Spec: DESIGN.md "A13".  Parity unpinned; validated  */
/* against orc_brute_nearest below.                                           */
/* ------------------------------------------------------------------------- */

/* grid: lo[3], cellsize[3] inv[3] dims[3]  -> packed in float g[12]:
 * g[0..2] = lo, g[3..5] = cell size, g[6..8] = 1/cell size, dims separate. */
void orc_uniform_setup(const float *bbmin, const float *bbmax, const int *dims, float *g)
{
	int k;
	for (k = 0; k < 3; k++) {
		float ext = bbmax[k] - bbmin[k];
		float pad = ext * 1e-4f + 1e-4f;
		float lo = bbmin[k] - pad, hi = bbmax[k] + pad;
		float cs = (hi - lo) / (float)dims[k];
		g[k] = lo;
		g[3 + k] = cs;
		g[6 + k] = 1.0f / cs;
	}
}

static int orc_ucell(const float *g, const int *dims, int k, float p)
{
	int c = ugrt_floor2i((p - g[k]) * g[6 + k]);
	return orc_clampi(c, 0, dims[k] - 1);
}

/* per-triangle world bbox -> 3-D cell range; rng6[f] = {x0,x1,y0,y1,z0,z1} */
void orc_uniform_ranges(const float *g, const int *dims, const int *facelist, const float *vertexlist,
			int F, int *rng6, u32 *sizeList)
{
	int f;
#pragma omp parallel for schedule(static)
	for (f = 0; f < F; f++) {
		int k;
		u32 sz = 1;
		for (k = 0; k < 3; k++) {
			float a = vertexlist[3 * facelist[f * 3 + 0] + k];
			float b = vertexlist[3 * facelist[f * 3 + 1] + k];
			float c = vertexlist[3 * facelist[f * 3 + 2] + k];
			int lo = orc_ucell(g, dims, k, orc_min3(a, b, c));
			int hi = orc_ucell(g, dims, k, orc_max3(a, b, c));
			rng6[f * 6 + 2 * k] = lo;
			rng6[f * 6 + 2 * k + 1] = hi;
			sz *= (u32)(hi - lo + 1);
		}
		sizeList[f] = sz;
	}
}

/* key = (gx*Gy + gy)*Gz + gz, x-major then y then z inside a triangle */
void orc_fill_3d(const int *rng6, const u32 *scanList, int F, const int *dims, u32 *keyList, u32 *valueList)
{
	int f;
#pragma omp parallel for schedule(dynamic, 1024)
	for (f = 0; f < F; f++) {
		u32 offset = f ? scanList[f - 1] : 0;
		const int *r = rng6 + f * 6;
		int sy = r[3] - r[2] + 1, sz = r[5] - r[4] + 1, i, j, k;
		if (scanList[f] == offset)
			continue; /* no references: outside this rank's window of a sharded build */
		for (i = r[0]; i <= r[1]; i++)
			for (j = r[2]; j <= r[3]; j++)
				for (k = r[4]; k <= r[5]; k++) {
					u32 pos = offset + (u32)(((i - r[0]) * sy + (j - r[2])) * sz + (k - r[4]));
					keyList[pos] = (u32)((i * dims[1] + j) * dims[2] + k);
					valueList[pos] = (u32)f;
				}
	}
}

/* Moller-Trumbore, signed t, same operation order as intersectTriUV; returns
 * 1 and *t_out when the ray (o,d) hits with tmin < t < tbest. */
static int orc_mt_signed(const float *o, const float *d, const float *v0, const float *v1p, const float *v2p,
			 float tmin, float tbest, float *t_out)
{
	float tvec[3], e1[3], e2[3], pvec[3], qvec[3], det, inv_det, u, v, t;
	int k;
	for (k = 0; k < 3; k++) {
		e1[k] = v1p[k] - v0[k];
		e2[k] = v2p[k] - v0[k];
		tvec[k] = o[k] - v0[k];
	}
	CROSS(pvec, d, e2);
	det = DOT(e1, pvec);
	if (det > -ORC_EPSILON && det < ORC_EPSILON)
		return 0;
	inv_det = 1.0f / det;
	u = DOT(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f)
		return 0;
	CROSS(qvec, tvec, e1);
	v = DOT(d, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f)
		return 0;
	t = DOT(e2, qvec) * inv_det;
	if (t > tmin && t < tbest) {
		*t_out = t;
		return 1;
	}
	return 0;
}

/*
 * Secondary-ray generation for the reflection bounce.
 * For pixel p with a primary hit (t > 0, id >= 0) on a material with
 * reflect[mat] > 0: P = cam + t*dir; n = normalize(e1 x e2) of the hit
 * triangle, flipped to face the viewer; r = dir - 2(dir.n)n; o = P + eps*n.
 * ray[p] = {ox,oy,oz, dx,dy,dz}; active[p] = 1/0.
 */
void orc_reflect_rays(const float *cam, const float *t_list, const float *dir_list, const int *id_list,
		      const int *mat_idx, const float *reflect, int mat_count, const float *vertlist,
		      const int *trilist, float eps, int p0, int n, float *rays, int *active)
{
	int i;
#pragma omp parallel for schedule(static)
	for (i = 0; i < n; i++) {
		int p = p0 + i, k, id = id_list[p], m;
		float t = t_list[p], e1[3], e2[3], nn[3], P[3], d[3], dn;
		active[p] = 0;
		for (k = 0; k < 6; k++)
			rays[p * 6 + k] = 0.0f;
		if (!(t > 0) || id < 0)
			continue;
		m = mat_idx[id];
		if (m < 0 || m >= mat_count || !(reflect[m] > 0))
			continue;
		for (k = 0; k < 3; k++) {
			float v0 = vertlist[3 * trilist[id * 3 + 0] + k];
			e1[k] = vertlist[3 * trilist[id * 3 + 1] + k] - v0;
			e2[k] = vertlist[3 * trilist[id * 3 + 2] + k] - v0;
			d[k] = dir_list[p * 3 + k];
			P[k] = cam[k] + t * d[k];
		}
		CROSS(nn, e1, e2);
		NORMALIZE(nn);
		dn = DOT(d, nn);
		if (dn > 0) {
			nn[0] = -nn[0];
			nn[1] = -nn[1];
			nn[2] = -nn[2];
			dn = -dn;
		}
		for (k = 0; k < 3; k++) {
			rays[p * 6 + k] = P[k] + eps * nn[k];
			rays[p * 6 + 3 + k] = d[k] - (2.0f * dn) * nn[k];
		}
		active[p] = 1;
	}
}

/*
 * 3D-DDA (Amanatides & Woo) through the uniform grid.  For every active ray:
 * clip against the grid box, walk cells front to back, in each cell test the
 * cell's triangle list (ascending triangle id) and keep the nearest hit with
 * t > 0; stop at the first cell whose best hit lies inside the cell's exit
 * parameter (t <= tnext).  Output hit_t (-1 miss) and hit_id (-2 miss).
 * counters: [0] MT tests, [1] cells visited, [2] active rays.
 */
void orc_trace_dda(const float *g, const int *dims, const u32 *value_list, const u32 *span,
		   const u32 *offset, const float *vertlist, const int *trilist, const float *rays,
		   const int *active, int p0, int n, float *hit_t, int *hit_id, unsigned long long *counters)
{
	int i;
	unsigned long long tests = 0, cells = 0, nact = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tests, cells, nact)
	for (i = 0; i < n; i++) {
		int p = p0 + i, k, c[3], step[3], done = 0;
		float o[3], d[3], tmax[3], tdelta[3], tenter = 0.0f, texit = 3.0e38f, best_t = 3.0e38f;
		int best_id = -2;
		hit_t[p] = -1.0f;
		hit_id[p] = -2;
		if (!active[p])
			continue;
		nact++;
		for (k = 0; k < 3; k++) {
			o[k] = rays[p * 6 + k];
			d[k] = rays[p * 6 + 3 + k];
		}
		/* slab clip */
		for (k = 0; k < 3; k++) {
			float lo = g[k], hi = g[k] + g[3 + k] * (float)dims[k];
			if (d[k] != 0.0f) {
				float inv = 1.0f / d[k];
				float t0 = (lo - o[k]) * inv, t1 = (hi - o[k]) * inv;
				if (t0 > t1) {
					float s = t0;
					t0 = t1;
					t1 = s;
				}
				if (t0 > tenter)
					tenter = t0;
				if (t1 < texit)
					texit = t1;
			} else if (o[k] < lo || o[k] > hi) {
				texit = -1.0f;
			}
		}
		if (!(tenter <= texit))
			continue;
		for (k = 0; k < 3; k++) {
			float pe = o[k] + tenter * d[k];
			c[k] = orc_ucell(g, dims, k, pe);
			if (d[k] > 0.0f) {
				step[k] = 1;
				tmax[k] = ((g[k] + (float)(c[k] + 1) * g[3 + k]) - o[k]) / d[k];
				tdelta[k] = g[3 + k] / d[k];
			} else if (d[k] < 0.0f) {
				step[k] = -1;
				tmax[k] = ((g[k] + (float)c[k] * g[3 + k]) - o[k]) / d[k];
				tdelta[k] = -g[3 + k] / d[k];
			} else {
				step[k] = 0;
				tmax[k] = 3.0e38f;
				tdelta[k] = 3.0e38f;
			}
		}
		/* every step leaves a cell for good: dims[0]+dims[1]+dims[2] bounds the walk */
		for (k = dims[0] + dims[1] + dims[2] + 3; k > 0 && !done; k--) {
			u32 cell = (u32)((c[0] * dims[1] + c[1]) * dims[2] + c[2]);
			u32 sp = span[cell], off = offset[cell], r;
			int ax = (tmax[0] < tmax[1]) ? ((tmax[0] < tmax[2]) ? 0 : 2) : ((tmax[1] < tmax[2]) ? 1 : 2);
			float tnext = tmax[ax];
			cells++;
			for (r = 0; r < sp; r++) {
				u32 f = value_list[off + r];
				float t;
				tests++;
				if (orc_mt_signed(o, d, &vertlist[3 * trilist[f * 3 + 0]], &vertlist[3 * trilist[f * 3 + 1]],
						  &vertlist[3 * trilist[f * 3 + 2]], 0.0f, best_t, &t)) {
					best_t = t;
					best_id = (int)f;
				}
			}
			if (best_id >= 0 && best_t <= tnext) {
				done = 1;
				break;
			}
			c[ax] += step[ax];
			if (step[ax] == 0 || c[ax] < 0 || c[ax] >= dims[ax])
				break;
			tmax[ax] += tdelta[ax];
		}
		if (done) {
			hit_t[p] = best_t;
			hit_id[p] = best_id;
		}
	}
	if (counters) {
		counters[0] = tests;
		counters[1] = cells;
		counters[2] = nact;
	}
}

/* Brute force nearest hit over ALL triangles: the check for orc_trace_dda. */
void orc_brute_nearest(const float *vertlist, const int *trilist, int F, const float *rays,
		       const int *active, int p0, int n, float *hit_t, int *hit_id)
{
	int i;
#pragma omp parallel for schedule(dynamic, 64)
	for (i = 0; i < n; i++) {
		int p = p0 + i, f, best_id = -2;
		float best_t = 3.0e38f;
		hit_t[p] = -1.0f;
		hit_id[p] = -2;
		if (!active[p])
			continue;
		for (f = 0; f < F; f++) {
			float t;
			if (orc_mt_signed(&rays[p * 6], &rays[p * 6 + 3], &vertlist[3 * trilist[f * 3 + 0]],
					  &vertlist[3 * trilist[f * 3 + 1]], &vertlist[3 * trilist[f * 3 + 2]], 0.0f,
					  best_t, &t)) {
				best_t = t;
				best_id = f;
			}
		}
		if (best_id >= 0) {
			hit_t[p] = best_t;
			hit_id[p] = best_id;
		}
	}
}

/*
 * Shade with one reflection bounce: local = Lambert colour of
 * the primary hit as in lambertian_shade (float, clamped to 1); reflected =
 * Lambert colour of the secondary hit (black on a miss); out = (1-k)*local +
 * k*reflected, k = reflect[material]; pixels without an active secondary ray
 * are written exactly as lambertian_shade writes them.
 */
void orc_shade_reflect(const float *cc, const float *light_position, unsigned char *d_img,
		       const float *dd_normal, const float *dd_t_value, const float *dd_dir,
		       int *dd_intersect_id, const float *d_cam_pos, const int *mat_idx, const float *mat_list,
		       const float *reflect, int mat_count, const float *vertlist, const int *trilist,
		       const float *rays, const int *active, const float *hit_t, const int *hit_id, int p0, int n)
{
	int i;
#pragma omp parallel for schedule(static)
	for (i = 0; i < n; i++) {
		int pixelID = p0 + i, k;
		float material[6], color[3] = { 0.0f, 0.0f, 0.0f };
		int tri = dd_intersect_id[pixelID];
		int idx = (tri >= 0) ? mat_idx[tri] : tri;
		dd_intersect_id[pixelID] = idx;
		if (idx >= 0 && idx < mat_count) {
			float point[3];
			float t_value = dd_t_value[pixelID];
			for (k = 0; k < 3; k++) {
				material[k] = mat_list[idx * 6 + 3 + k];
				material[3 + k] = mat_list[idx * 6 + 3 + k];
			}
			if (t_value > 0) {
				point[0] = d_cam_pos[0] + t_value * dd_dir[pixelID * 3 + 0];
				point[1] = d_cam_pos[1] + t_value * dd_dir[pixelID * 3 + 1];
				point[2] = d_cam_pos[2] + t_value * dd_dir[pixelID * 3 + 2];
				orc_lambert(cc, light_position, point, &dd_normal[pixelID * 3], color, material, 1.0f, 0);
				for (k = 0; k < 3; k++)
					if (color[k] > 1.0f)
						color[k] = 1.0f;
			}
			if (active[pixelID]) {
				float kr = reflect[idx], rc[3] = { 0.0f, 0.0f, 0.0f };
				int hid = hit_id[pixelID];
				if (hid >= 0) {
					int hm = mat_idx[hid];
					if (hm >= 0 && hm < mat_count) {
						float hp[3], e1[3], e2[3], nn[3], hmat[6];
						float ht = hit_t[pixelID];
						for (k = 0; k < 3; k++) {
							float v0 = vertlist[3 * trilist[hid * 3 + 0] + k];
							e1[k] = vertlist[3 * trilist[hid * 3 + 1] + k] - v0;
							e2[k] = vertlist[3 * trilist[hid * 3 + 2] + k] - v0;
							hp[k] = rays[pixelID * 6 + k] + ht * rays[pixelID * 6 + 3 + k];
							hmat[k] = mat_list[hm * 6 + 3 + k];
							hmat[3 + k] = mat_list[hm * 6 + 3 + k];
						}
						NORMALIZE(e1);
						NORMALIZE(e2);
						CROSS(nn, e1, e2);
						NORMALIZE(nn);
						orc_lambert(cc, light_position, hp, nn, rc, hmat, 1.0f, 0);
						for (k = 0; k < 3; k++)
							if (rc[k] > 1.0f)
								rc[k] = 1.0f;
					}
				}
				for (k = 0; k < 3; k++)
					color[k] = (1.0f - kr) * color[k] + kr * rc[k];
			}
		}
		d_img[pixelID * 3 + 0] = orc_to_u8(color[0]);
		d_img[pixelID * 3 + 1] = orc_to_u8(color[1]);
		d_img[pixelID * 3 + 2] = orc_to_u8(color[2]);
	}
}
