// ugrt_dda.h -- pieces shared by the bounce kernels (ugrt_dda.hip: per-ray and beam kernels of rounds 1 and 2;
// ugrt_dda_walk.hip: the window kernel): grid geometry, the bundle cull, triangle records.
#ifndef UGRT_DDA_H
#define UGRT_DDA_H

#include "ugrt_packet.h"

struct DGrid {
	float lo[3], cs[3], inv[3];
	int dims[3];
};

// Segments of the window kernel's long ray groups (ugrt_dda_walk.hip, "split walks"); all pointers null: no splitting
#define WK_FBW 32      // windows of a group whose job counts are remembered (one byte each; later windows share the last)
#define WK_MAXSEG 4    // segments a group is cut into at most
struct WalkSplit {
	u32 *hdr;                // this launch: [0] rays, [1] rays per wave, [2] segments listed in `items`, [3] jobs of the launch before
	const uint2 *items;      // the cut groups' segments: x = group | segment << 24 | segments << 28, y = first window | end window << 16
	const unsigned char *cut; // per group: listed in `items` (its turn among the whole groups is skipped)
	unsigned char *fb;       // jobs of (group, window): written by this launch, read (and cleared) by the next one's k_dda_segments
	const u32 *chunk;        // per 64 list entries: span of pixels * 8 + chunk of the span (what the groups' history is kept under)
	u32 *walked_prev;        // per pixel: windows the ray walked in the launch before (which segments it is given to)
	u32 *walked;             // the same of this launch (the smallest over the segments that saw the ray end)
	u32 *done;               // per group: segments that have finished
	unsigned long long *key; // per pixel: closest hit over the segments, t bits << 32 | segment << 29 | behind << 28 | list position; (a later segment's hit behind its cell: walk again)
	u32 p0;                  // first pixel of the context's band: the per-pixel arrays below are indexed by pixel - p0
	u32 *tend;               // per pixel: exit parameter of the ray's last cell (all ones until a segment sees the ray leave)
	u32 *texam;              // per pixel: exit parameter of the last cell any segment has looked at for the ray (+inf: a segment stopped it)
};

// what the host passes to k_dda_segments beside WalkSplit (ugrt_dda_split_state)
struct WalkSplitHost {
	uint2 *items;
	unsigned char *cut;
	const u32 *hdr_prev;
	u32 *hdr_next;
	u32 load, force, maxg, maxseg;
};

// what a group's history is kept under: the list is made of chunks of 64 entries that belong to one span of pixels
// (k_dda_prepare), in an order that changes from launch to launch; `chunk` names every chunk's span and place in it
__device__ __forceinline__ u32 d_group_key(const u32 *__restrict__ chunk, u32 g, u32 RPW)
{
	const u32 slot0 = g * RPW;
	return chunk[slot0 >> 6] * (64u / RPW) + (slot0 & 63u) / RPW;
}

__device__ __forceinline__ int d_dcell(const DGrid &g, int k, float p)
{
	int c = ugrt_floor2i((p - g.lo[k]) * g.inv[k]);
	return d_clampi(c, 0, g.dims[k] - 1);
}

// Cull of one triangle against a BUNDLE of rays with different origins.  Moller-Trumbore's numerators do
// not change when the origin slides along its ray (A = d.(e2 x tvec), and d.(e2 x d) = 0), so every ray of
// the bundle is represented by the point o' = o + t_in * d where it enters the current cell: the points of
// a bundle then lie within a fraction of a cell of each other.  With the boxes o' in oc +- orad, d in
// dc +- dr:
//     A = d.(e2 x (oc - v0)) + (o' - oc).(d x e2),   |second term| <= sum_k orad_k max|(d x e2)_k|
// and likewise for B (e1 x d) and A + B - det (d x (e2 - e1)), the maxima taken over the direction box; the
// first terms are the interval dot products of the single-origin cull (d_cull_cr).  (Bounding the second term
// by |d| |e2| |o' - oc| instead leaves twice as many triangles for the exact tests.)  The margins cover the rounding of the exact test, whose operands are
// tvec = o - v0 with the ray's own origin: `reach` bounds |o - oc|.  A culled triangle fails the exact
// float test on every ray of the bundle, so results do not change by a bit.
struct BeamBox {
	float oc[3], orad[3], dc[3], dr[3];
	float reach; // >= |o - oc|_inf over the bundle
};

__device__ __forceinline__ float d_uniform(float v)
{
	return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

__device__ __forceinline__ BeamBox d_beam_box(const float *o, const float *d, float tin, bool in)
{
	BeamBox bx;
	const float inf = __builtin_huge_valf();
	float on2 = 0.0f, dm2 = 0.0f;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const float p = o[k] + tin * d[k];
		float lo = d_wave_fmin(in ? p : inf), hi = d_wave_fmax(in ? p : -inf);
		bx.oc[k] = 0.5f * (lo + hi);
		// half width + the distance of the computed o' from the exact point of the ray (a few ulps of |o'|)
		bx.orad[k] = 0.5f * (hi - lo) * 1.0001f + 7.63e-6f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-6f;
		on2 += bx.orad[k] * bx.orad[k];
		lo = d_wave_fmin(in ? d[k] : inf);
		hi = d_wave_fmax(in ? d[k] : -inf);
		bx.dc[k] = 0.5f * (lo + hi);
		bx.dr[k] = 0.5f * (hi - lo) * 1.0001f + 1e-6f;
		const float da = fabsf(bx.dc[k]) + bx.dr[k];
		dm2 += da * da;
	}
	const float tmax = d_wave_fmax(in ? tin : 0.0f);
	const float on = __builtin_sqrtf(on2) * 1.0001f, dmax = __builtin_sqrtf(dm2) * 1.0001f;
	bx.reach = (tmax * dmax + on) * 1.001f;
	// uniform values: keep them in scalar registers
#pragma unroll
	for (int k = 0; k < 3; k++) {
		bx.oc[k] = d_uniform(bx.oc[k]);
		bx.dc[k] = d_uniform(bx.dc[k]);
		bx.dr[k] = d_uniform(bx.dr[k]);
		bx.orad[k] = d_uniform(bx.orad[k]);
	}
	bx.reach = d_uniform(bx.reach);
	return bx;
}

// true = no ray of the bundle can pass the exact test on triangle {v0, e1, e2}
__device__ __forceinline__ bool d_cull_beam(const float *v0, const float *e1, const float *e2, const BeamBox &bx)
{
#pragma clang fp contract(fast)
	const float tc[3] = { bx.oc[0] - v0[0], bx.oc[1] - v0[1], bx.oc[2] - v0[2] };
	float nA[3], nB[3], nD[3], nC[3];
	D_CROSS(nA, e2, tc);
	D_CROSS(nB, tc, e1);
	D_CROSS(nD, e2, e1);
#pragma unroll
	for (int k = 0; k < 3; k++)
		nC[k] = nA[k] + nB[k] - nD[k];
	const float a = fmaxf(fmaxf(fabsf(tc[0]), fabsf(tc[1])), fabsf(tc[2])) + bx.reach;
	const float b = fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2]));
	const float c = fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]));
	const float K = 6.0f / 65536.0f;
	const float mA = fmaxf(K * a * c, 1e-25f), mB = fmaxf(K * a * b, 1e-25f), mD = fmaxf(K * b * c, 1e-25f);
	const float Dm = nD[0] * bx.dc[0] + nD[1] * bx.dc[1] + nD[2] * bx.dc[2];
	const float Dr = fabsf(nD[0]) * bx.dr[0] + fabsf(nD[1]) * bx.dr[1] + fabsf(nD[2]) * bx.dr[2];
	if (!(Dm + Dr < 1e15f && Dm - Dr > -1e15f))
		return false;
	// |w . (d x e)| over the boxes of w = o' - oc and d: sum_k orad_k * (|(dc x e)_k| + the spread of d)
#define D_ORIGIN_TERM(E)                                                                                        \
	(bx.orad[0] * (fabsf(bx.dc[1] * E[2] - bx.dc[2] * E[1]) + bx.dr[1] * fabsf(E[2]) + bx.dr[2] * fabsf(E[1])) + \
	 bx.orad[1] * (fabsf(bx.dc[2] * E[0] - bx.dc[0] * E[2]) + bx.dr[2] * fabsf(E[0]) + bx.dr[0] * fabsf(E[2])) + \
	 bx.orad[2] * (fabsf(bx.dc[0] * E[1] - bx.dc[1] * E[0]) + bx.dr[0] * fabsf(E[1]) + bx.dr[1] * fabsf(E[0])))
	const float e21[3] = { e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2] };
	const float Am = nA[0] * bx.dc[0] + nA[1] * bx.dc[1] + nA[2] * bx.dc[2];
	const float Ar = (fabsf(nA[0]) * bx.dr[0] + fabsf(nA[1]) * bx.dr[1] + fabsf(nA[2]) * bx.dr[2] + D_ORIGIN_TERM(e2)) * 1.0001f;
	const float Bm = nB[0] * bx.dc[0] + nB[1] * bx.dc[1] + nB[2] * bx.dc[2];
	const float Br = (fabsf(nB[0]) * bx.dr[0] + fabsf(nB[1]) * bx.dr[1] + fabsf(nB[2]) * bx.dr[2] + D_ORIGIN_TERM(e1)) * 1.0001f;
	const float Cm = nC[0] * bx.dc[0] + nC[1] * bx.dc[1] + nC[2] * bx.dc[2];
	const float Cr = (fabsf(nC[0]) * bx.dr[0] + fabsf(nC[1]) * bx.dr[1] + fabsf(nC[2]) * bx.dr[2] + D_ORIGIN_TERM(e21)) * 1.0001f;
#undef D_ORIGIN_TERM
	const float mC = mA + mB + mD;
	if (Dm - Dr > mD) // det > 0 for every ray of the bundle
		return (Am + Ar < -mA) || (Bm + Br < -mB) || (Cm - Cr > mC);
	if (Dm + Dr < -mD) // det < 0
		return (Am - Ar > mA) || (Bm - Br > mB) || (Cm + Cr < -mC);
	return false;
}

// {v0, e1, e2} of one triangle: the 48-B record, or the gather + the reference's two edge subtractions
template <bool REC>
__device__ __forceinline__ void d_load_record(const float4 *__restrict__ rec, const float *__restrict__ verts,
					      const int *__restrict__ tris, u32 face, float *r9)
{
	if (REC) {
		const float4 a = rec[face * 3 + 0], b = rec[face * 3 + 1], c = rec[face * 3 + 2];
		r9[0] = a.x;
		r9[1] = a.y;
		r9[2] = a.z;
		r9[3] = a.w;
		r9[4] = b.x;
		r9[5] = b.y;
		r9[6] = b.z;
		r9[7] = b.w;
		r9[8] = c.x;
	} else {
		const int f1 = 3 * tris[face * 3 + 0], f2 = 3 * tris[face * 3 + 1], f3 = 3 * tris[face * 3 + 2];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			r9[k] = verts[f1 + k];
			r9[3 + k] = verts[f2 + k] - r9[k];
			r9[6 + k] = verts[f3 + k] - r9[k];
		}
	}
}

#endif
