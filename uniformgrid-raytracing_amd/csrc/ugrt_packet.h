// ugrt_packet.h -- device helpers shared by the tracers (ugrt_trace.hip, ugrt_dda.hip): XCD-aware work
// placement, triangle records, conservative packet culls, cross-lane primitives.
#ifndef UGRT_PACKET_H
#define UGRT_PACKET_H

#include "ugrt_dev.h"

#define TRI_STRIDE 12     // floats per staged triangle (9 used, 48 B: ds_read_b128 x3)
#define WL_THREADS 256

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Work lists
// are ordered so that neighbours share data (same cell, same triangle batch, same beam), so the
// persistent waves take CONTIGUOUS slices per XCD: logical id = (b % 8) * (grid/8) + b / 8.
// Placement only affects speed, never results.
__device__ __forceinline__ u32 d_xcd_block()
{
	const u32 g = gridDim.x, b = blockIdx.x;
	return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
}

// ---------------------------------------------------------------------------
// Packet culling.  All 64 rays of a work item share one origin (the eye, or the light), so for
// a triangle {tvec, e1, e2} the three Moller-Trumbore numerators are LINEAR in the direction d:
//   det = d.(e2 x e1)   A = u*det = d.(e2 x tvec)   B = v*det = d.(tvec x e1)
// A hit needs A/det >= 0, B/det >= 0, (A+B)/det <= 1.  Interval arithmetic over the bounding box
// of the item's directions shows "no lane can pass" for most triangles of a cell list (the
// lists come from clamped bounding boxes); only survivors get the per-lane test.
// One LANE culls one TRIANGLE, so a batch of 64 is culled for the price of one per-lane test.
// The margins (2^-16 relative to the operand magnitudes, against rounding errors of ~2^-22)
// make the cull strictly conservative: a culled triangle fails the exact float test of
// intersectTriUV / intersectTri on every lane, so results do not change by a bit.
// ---------------------------------------------------------------------------
struct DirBox {
	float lo[3], hi[3];
};

// The decision every packet cull ends with, from the centre +- half width of det, A = u det, B = v det and C = A + B - det
// over the packet (margins m*).  Either sign of det gives the same three tests once the numerators are taken WITH that
// sign (for det < 0: A - Ar > mA is -A + Ar < -mA, ...), so the sign bit of Dm is xor-ed into them and no branch is left:
// the two-sided form cost ~45 instructions per box under nested divergent branches, this one ~20.  Same decisions (a
// negation is exact); a NaN anywhere compares false = not culled.
__device__ __forceinline__ bool d_cull_decide(float Dm, float Dr, float Am, float Ar, float Bm, float Br, float Cm, float Cr,
					      float mA, float mB, float mD, float mC)
{
	const u32 sg = __float_as_uint(Dm) & 0x80000000u;
	const float sA = __uint_as_float(__float_as_uint(Am) ^ sg), sB = __uint_as_float(__float_as_uint(Bm) ^ sg),
		    sC = __uint_as_float(__float_as_uint(Cm) ^ sg);
	const float aD = fabsf(Dm);
	const bool one_sign = aD - Dr > mD;   // det > 0 (or < 0) for every direction of the box
	const bool sane = aD + Dr < 1e15f;    // (the margins were sized for operands below that)
	const bool out = (sA + Ar < -mA) | (sB + Br < -mB) | (sC - Cr > mC); // u < 0, v < 0, u + v > 1
	return one_sign & sane & out;
}

__device__ __forceinline__ void d_interval_dot(const float *n, const DirBox &bx, float *fmin, float *fmax)
{
#pragma clang fp contract(fast)
	float mn = 0.0f, mx = 0.0f;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		float p = n[k] * bx.lo[k], q = n[k] * bx.hi[k];
		mn += fminf(p, q);
		mx += fmaxf(p, q);
	}
	*fmin = mn;
	*fmax = mx;
}

// true = no direction inside the box can hit the triangle
__device__ __forceinline__ bool d_cull(const float *tv, const float *e1, const float *e2, const DirBox &bx)
{
#ifdef UGRT_DEBUG_CULL_ALL
	return (tv[0] + tv[1] + tv[2] + e1[0] + e1[1] + e1[2] + e2[0] + e2[1] + e2[2]) != 12345.678f;
// box of the directions of the lanes with `valid`; other lanes do not contribute (uniform: scalar registers)
__device__ __forceinline__ DirBox d_dir_box(const float *d, bool valid)
{
	DirBox bx;
	const float inf = __builtin_huge_valf();
#pragma unroll
	for (int k = 0; k < 3; k++) {
		bx.lo[k] = d_wave_fmin(valid ? d[k] : inf);
		bx.hi[k] = d_wave_fmax(valid ? d[k] : -inf);
	}
	return bx;
}

// Exchanges for reductions over the lanes of an 8x8-pixel tile's QUADRANT (lane bits 0, 1, 3, 4) and then across the
// quadrants (bits 2, 5), all wave-uniform control flow, every lane active.  Bits 0, 1 and 3 are DPP controls (the
// operation folds into the instruction); bit 4 and bit 5 use gfx950's v_permlane16_swap / v_permlane32_swap: with both
// operands the same value, one result holds the even rows (lower half) everywhere and the other the odd rows (upper
// half), so op(r0, r1) is the exchange with lane ^ 16 (^ 32).  Bit 2 after bits 0, 1: the quads are uniform by then,
// and row_half_mirror pairs the two quads of every 8 lanes.  (__shfl_xor is a ds_bpermute_b32 + wait per step.)
#define D_DPP_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define D_DPP_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define D_DPP_XOR8 0x128       // row_ror:8
#define D_DPP_HALF_MIRROR 0x141
template <typename OP>
__device__ __forceinline__ int d_quadrant_reduce(int v) // over lane bits 0, 1, 3, 4
{
	v = OP::op(v, d_dpp_i<D_DPP_XOR1>(0, v));
	v = OP::op(v, d_dpp_i<D_DPP_XOR2>(0, v));
	v = OP::op(v, d_dpp_i<D_DPP_XOR8>(0, v));
	const auto r = __builtin_amdgcn_permlane16_swap((u32)v, (u32)v, false, false);
	return OP::op((int)r[0], (int)r[1]);
}
template <typename OP>
__device__ __forceinline__ int d_across_quadrants(int v) // of quadrant-uniform values: over lane bits 2 and 5
{
	v = OP::op(v, d_dpp_i<D_DPP_HALF_MIRROR>(0, v));
	const auto r = __builtin_amdgcn_permlane32_swap((u32)v, (u32)v, false, false);
	return OP::op((int)r[0], (int)r[1]);
}

#endif
#pragma clang fp contract(fast)
	float nA[3], nB[3], nD[3], nC[3];
	D_CROSS(nA, e2, tv);
	D_CROSS(nB, tv, e1);
	D_CROSS(nD, e2, e1);
	const float a = fmaxf(fmaxf(fabsf(tv[0]), fabsf(tv[1])), fabsf(tv[2]));
	const float b = fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2]));
	const float c = fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]));
	const float K = 6.0f / 65536.0f;
	const float mA = fmaxf(K * a * c, 1e-25f), mB = fmaxf(K * a * b, 1e-25f), mD = fmaxf(K * b * c, 1e-25f);
	float Dmin, Dmax, lo, hi;
	d_interval_dot(nD, bx, &Dmin, &Dmax);
	if (!(Dmax < 1e15f && Dmin > -1e15f))
		return false;
#pragma unroll
	for (int k = 0; k < 3; k++)
		nC[k] = nA[k] + nB[k] - nD[k];
	if (Dmin > mD) { // det > 0 on every lane
		d_interval_dot(nA, bx, &lo, &hi);
		if (hi < -mA)
			return true; // u < 0
		d_interval_dot(nB, bx, &lo, &hi);
		if (hi < -mB)
			return true; // v < 0
		d_interval_dot(nC, bx, &lo, &hi);
		return lo > mA + mB + mD; // u + v > 1
	}
	if (Dmax < -mD) { // det < 0 on every lane
		d_interval_dot(nA, bx, &lo, &hi);
		if (lo > mA)
			return true;
		d_interval_dot(nB, bx, &lo, &hi);
		if (lo > mB)
			return true;
		d_interval_dot(nC, bx, &lo, &hi);
		return hi < -(mA + mB + mD);
	}
	return false;
}

// one triangle as {origin - v0, v1 - v0, v2 - v0}: from the 48-B record, or gathered as the
// reference does (trace_kernel.cu:151-175) when the caller's arrays are not the ones last built
template <bool REC>
__device__ __forceinline__ void d_load_triangle(const float4 *__restrict__ rec, const float *__restrict__ verts,
						const int *__restrict__ tris, u32 face, float ox, float oy, float oz,
						float *t9)
{
	if (REC) {
		const float4 a = rec[face * 3 + 0], b = rec[face * 3 + 1], c = rec[face * 3 + 2];
		t9[0] = ox - a.x;
		t9[1] = oy - a.y;
		t9[2] = oz - a.z;
		t9[3] = a.w;
		t9[4] = b.x;
		t9[5] = b.y;
		t9[6] = b.z;
		t9[7] = b.w;
		t9[8] = c.x;
	} else {
		d_stage_triangle(verts, tris, face, ox, oy, oz, t9);
	}
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ u32 d_rank_in_mask(unsigned long long mask)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// The same test with the triangle's part hoisted (it is reused for several boxes) and the box given as
// centre +- half width: f(d) = n.d ranges over n.c -+ sum_k |n_k| r_k.
struct CullTri {
	float nA[3], nB[3], nC[3], nD[3];
	float mA, mB, mD, mT;
};
struct CBox {
	float c[3], r[3];
};
// a box of directions as centre and half width (the half width a little wide: far more than the rounding of c and r)
__device__ __forceinline__ CBox d_cbox(const DirBox &b)
{
	CBox o;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		o.c[k] = 0.5f * (b.lo[k] + b.hi[k]);
		o.r[k] = 0.5f * (b.hi[k] - b.lo[k]) * 1.0001f + 1e-6f;
	}
	return o;
}

// (The cull is outside the numeric contract: it only has to be conservative, and its margins are 2^6
// times the rounding error, so its dot products may contract to FMAs; the exact tests never do.)
__device__ __forceinline__ CullTri d_cull_prep(const float *tv, const float *e1, const float *e2)
{
#pragma clang fp contract(fast)
	CullTri t;
	D_CROSS(t.nA, e2, tv);
	D_CROSS(t.nB, tv, e1);
	D_CROSS(t.nD, e2, e1);
#pragma unroll
	for (int k = 0; k < 3; k++)
		t.nC[k] = t.nA[k] + t.nB[k] - t.nD[k];
	const float a = fmaxf(fmaxf(fabsf(tv[0]), fabsf(tv[1])), fabsf(tv[2]));
	const float b = fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2]));
	const float c = fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]));
	const float K = 6.0f / 65536.0f;
	t.mA = fmaxf(K * a * c, 1e-25f);
	t.mB = fmaxf(K * a * b, 1e-25f);
	t.mD = fmaxf(K * b * c, 1e-25f);
	t.mT = K * a * b * c;
	return t;
}

// Lower bound of the |t| that intersectTriUV computes for ANY direction of the box (0 = no bound).
// t = T * (1 / det) with T = e2.(tvec x e1), the same for every ray, and det = d.(e2 x e1).  The exact path
// rounds T within ~2^-21 a b c of its true value (mT is 2^6 times that) and its det stays inside the interval
// the cull tests the sign of (|Dm| + Dr + mD, margins as there), so
//   |t| >= (|T| - mT) / (|Dm| + Dr + mD) * (1 - 2^-22),
// and the factor below leaves 2^-13 for the reciprocal's ulp and the two roundings.  A triangle whose bound
// exceeds the closest hit of every ray of a packet cannot be accepted by any of them (strict t < oldt).
__device__ __forceinline__ float d_cull_tlow(const CullTri &t, const float *e2, const CBox &bx)
{
#pragma clang fp contract(fast)
	const float Dm = t.nD[0] * bx.c[0] + t.nD[1] * bx.c[1] + t.nD[2] * bx.c[2];
	const float Dr = fabsf(t.nD[0]) * bx.r[0] + fabsf(t.nD[1]) * bx.r[1] + fabsf(t.nD[2]) * bx.r[2];
	const float T = fabsf(e2[0] * t.nB[0] + e2[1] * t.nB[1] + e2[2] * t.nB[2]) - t.mT;
	const float bound = fabsf(Dm) + Dr + t.mD;
	const float lo = T * __builtin_amdgcn_rcpf(bound) * 0.9998779296875f;
	return lo > 0.0f ? lo : 0.0f; // (a NaN ends here as 0 too)
}

__device__ __forceinline__ bool d_cull_cr(const CullTri &t, const CBox &bx)
{
#pragma clang fp contract(fast)
	const float Dm = t.nD[0] * bx.c[0] + t.nD[1] * bx.c[1] + t.nD[2] * bx.c[2];
	const float Dr = fabsf(t.nD[0]) * bx.r[0] + fabsf(t.nD[1]) * bx.r[1] + fabsf(t.nD[2]) * bx.r[2];
	const float Am = t.nA[0] * bx.c[0] + t.nA[1] * bx.c[1] + t.nA[2] * bx.c[2];
	const float Ar = fabsf(t.nA[0]) * bx.r[0] + fabsf(t.nA[1]) * bx.r[1] + fabsf(t.nA[2]) * bx.r[2];
	const float Bm = t.nB[0] * bx.c[0] + t.nB[1] * bx.c[1] + t.nB[2] * bx.c[2];
	const float Br = fabsf(t.nB[0]) * bx.r[0] + fabsf(t.nB[1]) * bx.r[1] + fabsf(t.nB[2]) * bx.r[2];
	const float Cm = t.nC[0] * bx.c[0] + t.nC[1] * bx.c[1] + t.nC[2] * bx.c[2];
	const float Cr = fabsf(t.nC[0]) * bx.r[0] + fabsf(t.nC[1]) * bx.r[1] + fabsf(t.nC[2]) * bx.r[2];
	return d_cull_decide(Dm, Dr, Am, Ar, Bm, Br, Cm, Cr, t.mA, t.mB, t.mD, t.mA + t.mB + t.mD);
}

// The same test against four boxes that share one half width (the quadrants of a tile: their radii differ by a few per
// cent, the largest is taken for all -- a larger box only culls less): the radius terms are formed once instead of four
// times (~100 instead of ~180 instructions per batch with survivors).  Bit q of the result = box q cannot be hit.
__device__ __forceinline__ u32 d_cull_cr4(const CullTri &t, const CBox *bx, const float *r)
{
#pragma clang fp contract(fast)
	const float Dr = fabsf(t.nD[0]) * r[0] + fabsf(t.nD[1]) * r[1] + fabsf(t.nD[2]) * r[2];
	const float Ar = fabsf(t.nA[0]) * r[0] + fabsf(t.nA[1]) * r[1] + fabsf(t.nA[2]) * r[2];
	const float Br = fabsf(t.nB[0]) * r[0] + fabsf(t.nB[1]) * r[1] + fabsf(t.nB[2]) * r[2];
	const float Cr = fabsf(t.nC[0]) * r[0] + fabsf(t.nC[1]) * r[1] + fabsf(t.nC[2]) * r[2];
	const float mC = t.mA + t.mB + t.mD;
	u32 culled = 0u;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const float *c = bx[q].c;
		const float Dm = t.nD[0] * c[0] + t.nD[1] * c[1] + t.nD[2] * c[2];
		const float Am = t.nA[0] * c[0] + t.nA[1] * c[1] + t.nA[2] * c[2];
		const float Bm = t.nB[0] * c[0] + t.nB[1] * c[1] + t.nB[2] * c[2];
		const float Cm = t.nC[0] * c[0] + t.nC[1] * c[1] + t.nC[2] * c[2];
		culled |= d_cull_decide(Dm, Dr, Am, Ar, Bm, Br, Cm, Cr, t.mA, t.mB, t.mD, mC) ? (1u << q) : 0u;
	}
	return culled;
}

__device__ __forceinline__ float d_readlane(float v, int l)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}


// persistent launches: 256 CUs x 8 single-wave workgroups per SIMD-quad
static inline int launch_blocks_for(u32 upper, int waves = -1)
{
	u32 g = waves > 0 ? (u32)waves : 256u * 32u;
	if (upper < g)
		g = upper ? upper : 1u;
	return (int)g;
}

// ---------------------------------------------------------------------------
// Wave reductions on the DPP path (row_shr 1/2/4/8, row_bcast 15/31: ~4 VALU instructions per step).
// __shfl_xor compiles to ds_bpermute_b32 on gfx950, i.e. six dependent trips through the LDS crossbar
// per reduction.  `v` of lanes that must not contribute is the identity (+-inf, ~0).  The result is
// uniform (read from lane 63).
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int d_dpp_i(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
struct DOpMin {
	template <typename T>
	__device__ __forceinline__ static T op(T a, T b) { return b < a ? b : a; }
};
struct DOpMax {
	template <typename T>
	__device__ __forceinline__ static T op(T a, T b) { return b > a ? b : a; }
};
__device__ __forceinline__ int d_as_int(int v) { return v; }
__device__ __forceinline__ int d_as_int(float v) { return __float_as_int(v); }
__device__ __forceinline__ void d_from_int(int i, int *v) { *v = i; }
__device__ __forceinline__ void d_from_int(int i, float *v) { *v = __int_as_float(i); }

template <typename OP, typename T>
__device__ __forceinline__ T d_wave_reduce(T v, T identity)
{
	const int id = d_as_int(identity);
	T o;
	d_from_int(d_dpp_i<0x111>(id, d_as_int(v)), &o); // row_shr:1
	v = OP::op(v, o);
	d_from_int(d_dpp_i<0x112>(id, d_as_int(v)), &o); // row_shr:2
	v = OP::op(v, o);
	d_from_int(d_dpp_i<0x114>(id, d_as_int(v)), &o); // row_shr:4
	v = OP::op(v, o);
	d_from_int(d_dpp_i<0x118>(id, d_as_int(v)), &o); // row_shr:8: lane 15 of every row holds the row's result
	v = OP::op(v, o);
	d_from_int(d_dpp_i<0x142, 0xA>(id, d_as_int(v)), &o); // row_bcast:15 into rows 1 and 3
	v = OP::op(v, o);
	d_from_int(d_dpp_i<0x143, 0xC>(id, d_as_int(v)), &o); // row_bcast:31 into rows 2 and 3
	v = OP::op(v, o);
	T r;
	d_from_int(__builtin_amdgcn_readlane(d_as_int(v), 63), &r);
	return r;
}
__device__ __forceinline__ int d_wave_imin(int v) { return d_wave_reduce<DOpMin>(v, 0x7FFFFFFF); }
__device__ __forceinline__ int d_wave_imax(int v) { return d_wave_reduce<DOpMax>(v, (int)0x80000000); }
// floats go through their order-preserving integer image: integer min/max fold into the DPP instruction
// (v_min_i32_dpp: one instruction per step), the float forms need a move and a canonicalisation beside it
__device__ __forceinline__ int d_ordered(float f)
{
	const int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float d_unordered(int b) { return __int_as_float(b ^ ((b >> 31) & 0x7FFFFFFF)); }
__device__ __forceinline__ float d_wave_fmin(float v) { return d_unordered(d_wave_imin(d_ordered(v))); }
__device__ __forceinline__ float d_wave_fmax(float v) { return d_unordered(d_wave_imax(d_ordered(v))); }

// box of the directions of the lanes with `valid`; other lanes do not contribute (uniform: scalar registers)
__device__ __forceinline__ DirBox d_dir_box(const float *d, bool valid)
{
	DirBox bx;
	const float inf = __builtin_huge_valf();
#pragma unroll
	for (int k = 0; k < 3; k++) {
		bx.lo[k] = d_wave_fmin(valid ? d[k] : inf);
		bx.hi[k] = d_wave_fmax(valid ? d[k] : -inf);
	}
	return bx;
}

// Exchanges for reductions over the lanes of an 8x8-pixel tile's QUADRANT (lane bits 0, 1, 3, 4) and then across the
// quadrants (bits 2, 5), all wave-uniform control flow, every lane active.  Bits 0, 1 and 3 are DPP controls (the
// operation folds into the instruction); bit 4 and bit 5 use gfx950's v_permlane16_swap / v_permlane32_swap: with both
// operands the same value, one result holds the even rows (lower half) everywhere and the other the odd rows (upper
// half), so op(r0, r1) is the exchange with lane ^ 16 (^ 32).  Bit 2 after bits 0, 1: the quads are uniform by then,
// and row_half_mirror pairs the two quads of every 8 lanes.  (__shfl_xor is a ds_bpermute_b32 + wait per step.)
#define D_DPP_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define D_DPP_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define D_DPP_XOR8 0x128       // row_ror:8
#define D_DPP_HALF_MIRROR 0x141
template <typename OP>
__device__ __forceinline__ int d_quadrant_reduce(int v) // over lane bits 0, 1, 3, 4
{
	v = OP::op(v, d_dpp_i<D_DPP_XOR1>(0, v));
	v = OP::op(v, d_dpp_i<D_DPP_XOR2>(0, v));
	v = OP::op(v, d_dpp_i<D_DPP_XOR8>(0, v));
	const auto r = __builtin_amdgcn_permlane16_swap((u32)v, (u32)v, false, false);
	return OP::op((int)r[0], (int)r[1]);
}
template <typename OP>
__device__ __forceinline__ int d_across_quadrants(int v) // of quadrant-uniform values: over lane bits 2 and 5
{
	v = OP::op(v, d_dpp_i<D_DPP_HALF_MIRROR>(0, v));
	const auto r = __builtin_amdgcn_permlane32_swap((u32)v, (u32)v, false, false);
	return OP::op((int)r[0], (int)r[1]);
}

#endif
