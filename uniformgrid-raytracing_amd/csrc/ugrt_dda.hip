// ugrt_dda.hip -- reflection bounce: 3D-DDA through the uniform grid (not in the reference).
//
// Spec: clip the ray against the grid box, walk the cells front to back
// (Amanatides & Woo), test every triangle of a cell's list in ascending id with the reference's
// Moller-Trumbore (signed t, 0 < t < best), stop at the first cell whose best hit lies before the cell's
// exit.  Two kernels compute it:
//   k_trace_dda_walk (ugrt_dda_walk.hip, the default) -- the window kernel of round 3;
//   k_trace_dda_ray (option "dda_kernel" = 1) -- round 1's kernel: every ray walks and tests alone, long
//     lists are tested by the whole wave for one owner at a time.  Kept as the plain GPU form of the specification
//     that the window kernel is compared with (tests/test_gpu_parity.py::test_bounce_kernels_agree).
// (Round 2's beam kernel, which shared the work inside a cell among the rays that stood in it, was the window kernel's
// predecessor; it went in round 4, when nothing used it any more.)
#include "ugrt_dda.h"


#define DDA_AHEAD 4   // cells planned (and their headers fetched) per round trip

// 64-bit wave minimum on the DPP path (two 32-bit moves per step); the result is uniform
__device__ __forceinline__ unsigned long long d_wave_min_u64(unsigned long long v)
{
	u32 lo = (u32)v, hi = (u32)(v >> 32);
#define D_MIN64_STEP(CTRL, RM)                                                                   \
	{                                                                                        \
		const u32 olo = (u32)d_dpp_i<CTRL, RM>(-1, (int)lo), ohi = (u32)d_dpp_i<CTRL, RM>(-1, (int)hi); \
		const bool less = ohi < hi || (ohi == hi && olo < lo);                           \
		lo = less ? olo : lo;                                                            \
		hi = less ? ohi : hi;                                                            \
	}
	D_MIN64_STEP(0x111, 0xF)
	D_MIN64_STEP(0x112, 0xF)
	D_MIN64_STEP(0x114, 0xF)
	D_MIN64_STEP(0x118, 0xF)
	D_MIN64_STEP(0x142, 0xA)
	D_MIN64_STEP(0x143, 0xC)
#undef D_MIN64_STEP
	lo = (u32)__builtin_amdgcn_readlane((int)lo, 63);
	hi = (u32)__builtin_amdgcn_readlane((int)hi, 63);
	return ((unsigned long long)hi << 32) | lo;
}

// Default results for every pixel of the band + the list of active secondary rays, in 8x8 TILE order: 64
// consecutive entries are neighbours on the screen, i.e. rays that start next to each other and point the
// same way (the order is irrelevant for the results).  Each wave owns DDA_PREP_SPAN / 64 consecutive tiles
// and reserves its slots with ONE atomic (a counter bumped once per 64 pixels serialises ~30 k same-address
// atomics, 0.08 ms at 1080p).  A span's entries are padded to whole chunks of 64 (entries ~0: no ray): the spans
// land in the list in the order their atomics arrive, which changes from launch to launch, but a ray group (16, 32
// or 64 consecutive entries) then always holds the same pixels, and `chunk` says which span and which of its chunks
// every 64 entries are -- what the window kernel keeps the groups' history under (null: not wanted).
#define DDA_PREP_SPAN 512
__global__ __launch_bounds__(256) void k_dda_prepare(const int *__restrict__ active, int p0, int npix, int W,
						      float *__restrict__ hit_t, int *__restrict__ hit_id,
						      u32 *__restrict__ list, u32 *__restrict__ count,
						      u32 *__restrict__ ticket, u32 pix_blocks, const u32 *__restrict__ span, u32 C,
						      u32 *__restrict__ bitmap, u32 *__restrict__ chunk, u32 *__restrict__ count_next)
{
	const int lane = threadIdx.x & 63;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		*count_next = 0u; // (no kernel of this launch uses it; the next launch counts into it)
	if (blockIdx.x >= pix_blocks) {
		// the workgroups behind the pixels' write the window kernel's occupancy bitmap: bit c = span[c] != 0, one
		// 64-bit word per wave and 64 cells
		for (u32 base = ((blockIdx.x - pix_blocks) * 4u + (threadIdx.x >> 6)) * 64u; base < C; base += (gridDim.x - pix_blocks) * 256u) {
			const u32 c = base + (u32)lane;
			const unsigned long long m = __ballot(c < C && span[c] != 0u);
			if (lane == 0) {
				bitmap[base >> 5] = (u32)m;
				bitmap[(base >> 5) + 1u] = (u32)(m >> 32);
			}
		}
		return;
	}
	const int first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * DDA_PREP_SPAN;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		*ticket = 0; // the tracer's waves draw their ray groups from it
	if (first >= npix)
		return;
	const int nbx = W >> 3;
	u32 total = 0;
	unsigned long long flags = 0ull; // bit k: this lane's pixel of round k is active
	int pix[DDA_PREP_SPAN / 64];
#pragma unroll
	for (int k = 0; k < DDA_PREP_SPAN / 64; k++) {
		const int tile = (first >> 6) + k; // tiles of the band, x-major
		// inside a tile 16 consecutive lanes are a 4x4-pixel block (a group of 16 or 32 list entries is a
		// compact bundle of rays, not a strip)
		const int lc = (lane & 3) | ((lane >> 2) & 4), lr = ((lane >> 2) & 3) | ((lane >> 3) & 4);
		const int col = (tile % nbx) * 8 + lc, row = (tile / nbx) * 8 + lr;
		const int i = first + k * 64;
		pix[k] = p0 + row * W + col;
		bool a = false;
		if (i < npix) {
			a = active[pix[k]] != 0;
			hit_t[pix[k]] = -1.0f;
			hit_id[pix[k]] = -2;
		}
		flags |= (unsigned long long)a << k;
		total += (u32)__popcll(__ballot(a));
	}
	if (total == 0)
		return;
	const u32 padded = (total + 63u) & ~63u;
	u32 base = 0;
	if (lane == 0)
		base = atomicAdd(count, padded);
	base = __shfl(base, 0);
	if (chunk && (u32)lane < padded / 64u)
		chunk[(base >> 6) + (u32)lane] = (u32)(first / DDA_PREP_SPAN) * (DDA_PREP_SPAN / 64) + (u32)lane;
	if ((u32)lane < padded - total)
		list[base + total + (u32)lane] = 0xFFFFFFFFu;
#pragma unroll
	for (int k = 0; k < DDA_PREP_SPAN / 64; k++) {
		const bool a = (flags >> k) & 1ull;
		const unsigned long long mask = __ballot(a);
		if (a)
			list[base + d_rank_in_mask(mask)] = (u32)pix[k];
		base += (u32)__popcll(mask);
	}
}

// One lane per secondary ray, Amanatides & Woo stepping.  A ray's work is a chain of dependent
// loads (cell header -> triangle id -> record), so the kernel is bound by its LONGEST ray; cells
// with many triangles (the debris cloud) are therefore tested by the whole wave: the owning
// lane's ray is broadcast, 64 triangles are tested at once (lane = triangle) and the nearest
// accepted hit is found with a 64-bit wave min on (t bits << 32 | r).  Sequentially the cell loop
// keeps the first r with the smallest accepted t (strict <), which is exactly that minimum.
template <bool COUNT, bool REC>
__global__ __launch_bounds__(64) void k_trace_dda_ray(DGrid g, const u32 *__restrict__ value_list,
						   const u32 *__restrict__ span, const u32 *__restrict__ offset,
						   const float *__restrict__ verts, const int *__restrict__ tris,
						   const float4 *__restrict__ rec,
						   const float *__restrict__ rays, const u32 *__restrict__ list,
						   const u32 *__restrict__ count_p, float *__restrict__ hit_t,
						   int *__restrict__ hit_id, unsigned long long *__restrict__ counters,
						   u32 DDA_RPW, u32 DDA_COOP)
{
	const int lane = threadIdx.x;
	const u32 count = *count_p;
	// DDA_RPW rays per wave: the walk of a ray is a serial chain, and the rays that cross the debris
	// cloud carry most of the tests, so few rays per wave spreads those chains over the chip while
	// all 64 lanes still serve the cooperative rounds
	for (u32 grp = blockIdx.x; grp * DDA_RPW < count; grp += gridDim.x) {
	const u32 slot = grp * DDA_RPW + (u32)lane;
	bool inb = (u32)lane < DDA_RPW && slot < count;
	const int p = inb ? (int)list[slot] : 0;
	inb = inb && p != -1; // (padding: k_dda_prepare)
	float res_t = -1.0f;
	int res_id = -2;
	u32 n_cells = 0, n_tests = 0;
	float o[3] = { 0, 0, 0 }, d[3] = { 0, 0, 0 }, tmax[3] = { 0, 0, 0 }, tdelta[3] = { 0, 0, 0 };
	int c[3] = { 0, 0, 0 }, step[3] = { 0, 0, 0 };
	float best_t = 3.0e38f;
	int best_id = -2;
	const bool is_active = inb;
	bool walking = false;
	if (is_active) {
		float tenter = 0.0f, texit = 3.0e38f;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			o[k] = rays[p * 6 + k];
			d[k] = rays[p * 6 + 3 + k];
		}
#pragma unroll
		for (int k = 0; k < 3; k++) {
			float lo = g.lo[k], hi = g.lo[k] + g.cs[k] * (float)g.dims[k];
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
		if (tenter <= texit) {
			walking = true;
#pragma unroll
			for (int k = 0; k < 3; k++) {
				float pe = o[k] + tenter * d[k];
				c[k] = d_dcell(g, k, pe);
				if (d[k] > 0.0f) {
					step[k] = 1;
					tmax[k] = ((g.lo[k] + (float)(c[k] + 1) * g.cs[k]) - o[k]) / d[k];
					tdelta[k] = g.cs[k] / d[k];
				} else if (d[k] < 0.0f) {
					step[k] = -1;
					tmax[k] = ((g.lo[k] + (float)c[k] * g.cs[k]) - o[k]) / d[k];
					tdelta[k] = -g.cs[k] / d[k];
				} else {
					step[k] = 0;
					tmax[k] = 3.0e38f;
					tdelta[k] = 3.0e38f;
				}
			}
		}
	}
	// every step leaves a cell for good, so dims[0]+dims[1]+dims[2] bounds the walk
	int guard = g.dims[0] + g.dims[1] + g.dims[2] + 3;
	while (__ballot(walking) != 0ull) {
		// The walk itself does not depend on what the cells hold, so the next DDA_AHEAD cells are
		// planned first and their headers fetched together: one memory round trip per DDA_AHEAD steps.
		u32 pcell[DDA_AHEAD], psp[DDA_AHEAD], poff[DDA_AHEAD];
		float ptnext[DDA_AHEAD];
		bool pvalid[DDA_AHEAD], pend[DDA_AHEAD];
		bool planning = walking;
#pragma unroll
		for (int q = 0; q < DDA_AHEAD; q++) {
			pvalid[q] = planning;
			pend[q] = false;
			pcell[q] = 0;
			ptnext[q] = 0.0f;
			if (planning) {
				pcell[q] = (u32)((c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2]);
				int ax = (tmax[0] < tmax[1]) ? ((tmax[0] < tmax[2]) ? 0 : 2) : ((tmax[1] < tmax[2]) ? 1 : 2);
				ptnext[q] = ax == 0 ? tmax[0] : (ax == 1 ? tmax[1] : tmax[2]);
				// step along ax (written out: no dynamically indexed registers)
				bool outside;
				if (ax == 0) {
					c[0] += step[0];
					outside = step[0] == 0 || c[0] < 0 || c[0] >= g.dims[0];
					tmax[0] += tdelta[0];
				} else if (ax == 1) {
					c[1] += step[1];
					outside = step[1] == 0 || c[1] < 0 || c[1] >= g.dims[1];
					tmax[1] += tdelta[1];
				} else {
					c[2] += step[2];
					outside = step[2] == 0 || c[2] < 0 || c[2] >= g.dims[2];
					tmax[2] += tdelta[2];
				}
				if (outside || --guard <= 0) {
					pend[q] = true; // the walk ends after this cell unless it ends there with a hit
					planning = false;
				}
			}
		}
#pragma unroll
		for (int q = 0; q < DDA_AHEAD; q++) {
			psp[q] = pvalid[q] ? span[pcell[q]] : 0u;
			poff[q] = pvalid[q] ? offset[pcell[q]] : 0u;
		}
#pragma unroll
		for (int q = 0; q < DDA_AHEAD; q++) {
			const bool here = walking && pvalid[q];
			const u32 sp = here ? psp[q] : 0u, off = poff[q];
			if (COUNT && here) {
				n_cells++;
				n_tests += sp;
			}
			// small lists: the owning lane tests them itself, in list order
			if (here && sp < DDA_COOP) {
				for (u32 r = 0; r < sp; r++) {
					u32 f = value_list[off + r];
					float t9[9], t;
					d_load_triangle<REC>(rec, verts, tris, f, o[0], o[1], o[2], t9);
					if (d_mt_core(&t9[0], &t9[3], &t9[6], d, &t) && t > 0.0f && t < best_t) {
						best_t = t;
						best_id = (int)f;
					}
				}
			}
			// long lists: one owner at a time, 64 triangles per round
			unsigned long long heavy = __ballot(here && sp >= DDA_COOP);
			while (heavy != 0ull) {
				const int l = (int)__builtin_ctzll(heavy);
				heavy &= heavy - 1ull;
				const float ox = __shfl(o[0], l), oy = __shfl(o[1], l), oz = __shfl(o[2], l);
				const float dl[3] = { __shfl(d[0], l), __shfl(d[1], l), __shfl(d[2], l) };
				const float bt = __shfl(best_t, l);
				const u32 spl = (u32)__shfl((int)sp, l), offl = (u32)__shfl((int)off, l);
				unsigned long long kbest = ~0ull;
				for (u32 base = 0; base < spl; base += 64) {
					const u32 r = base + (u32)lane;
					unsigned long long key = ~0ull;
					if (r < spl) {
						float t9[9], t;
						d_load_triangle<REC>(rec, verts, tris, value_list[offl + r], ox, oy, oz, t9);
						if (d_mt_core(&t9[0], &t9[3], &t9[6], dl, &t) && t > 0.0f && t < bt)
							key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)r;
					}
					key = d_wave_min_u64(key);
					kbest = key < kbest ? key : kbest;
				}
				if (lane == l && kbest != ~0ull) {
					best_t = __uint_as_float((u32)(kbest >> 32));
					best_id = (int)value_list[off + (u32)(kbest & 0xFFFFFFFFull)];
				}
			}
			if (here) {
				if (best_id >= 0 && best_t <= ptnext[q]) {
					res_t = best_t;
					res_id = best_id;
					walking = false;
				} else if (pend[q]) {
					walking = false;
				}
			}
		}
	}
	if (inb) {
		hit_t[p] = res_t;
		hit_id[p] = res_id;
	}
	if (COUNT && inb) {
		// work counters of the algorithmic-byte formula: candidates tested, cells visited, active rays
		if (n_tests)
			atomicAdd(&counters[0], (unsigned long long)n_tests);
		if (n_cells)
			atomicAdd(&counters[1], (unsigned long long)n_cells);
		atomicAdd(&counters[2], 1ull);
	}
	} // groups
}


// counters of a counting launch (UGRT_FLAG_COUNT_WORK): [0..2] tests, cells, rays (the algorithmic-byte formula), then
// the window kernel's statistics (ugrt_dda_walk.hip: WS_*), DS_END words in all
enum { DS_ITER = 3, DS_END = 3 + UGRT_DDA_STATS };

// ugrt_dda_walk.hip
int ugrt_dda_walk_launch(ugrt_ctx *ctx, const DGrid &g, const u32 *d_value_list, const u32 *d_span, const u32 *d_offset,
			 u32 *bitmap, const float *d_vertlist, const int *d_trilist, const float4 *rec, const float *d_rays,
			 const u32 *list, const u32 *dcount, float *d_hit_t, int *d_hit_id, unsigned long long *counters,
			 bool counting, u32 RPW, u32 CULL_MIN, u32 CULL_WORK, int blocks, const WalkSplit &sp, const WalkSplitHost &sph);
int ugrt_dda_split_state(ugrt_ctx *ctx, u32 RPW, u32 total_refs, WalkSplit *sp, WalkSplitHost *sph);
int ugrt_dda_sort_keys_launch(ugrt_ctx *ctx, const DGrid &g, const float *d_rays, const u32 *list, const u32 *dcount, u32 cap,
			      u32 *keys);

extern "C" int ugrt_trace_dda(ugrt_ctx *ctx, const unsigned *d_value_list, const unsigned *d_span,
			      const unsigned *d_offset, const float *d_vertlist, const int *d_trilist,
			      const float *d_rays, const int *d_active, float *d_hit_t, int *d_hit_id)
{
	if (!ctx || !d_value_list || !d_span || !d_offset || !d_vertlist || !d_trilist || !d_rays || !d_active ||
	    !d_hit_t || !d_hit_id)
		return ugrt_fail(UGRT_EINVAL, "trace_dda: null argument");
	Grid &G = ctx->grid[UGRT_GRID_UNIFORM];
	if (!G.valid)
		return ugrt_fail(UGRT_EINVAL, "trace_dda: build the uniform grid first (it defines the cell geometry)");
	UGRT_HIP(hipSetDevice(ctx->device));
	DGrid g;
	for (int k = 0; k < 3; k++) {
		g.lo[k] = G.ug[k];
		g.cs[k] = G.ug[3 + k];
		g.inv[k] = G.ug[6 + k];
		g.dims[k] = G.dims[k];
	}
	int rc;
	// (k_dda_prepare pads every span of DDA_PREP_SPAN pixels to whole chunks of 64 list entries: the list can be that
	// much longer than the band has pixels when the band is not a whole number of spans)
	const size_t list_cap = ((size_t)ctx->npix + DDA_PREP_SPAN - 1) / DDA_PREP_SPAN * DDA_PREP_SPAN;
	if ((rc = ugrt_buf_reserve(ctx, ctx->wscan, list_cap * 4)))
		return rc;
	// (two ray counters in turn: a launch's prepare kernel clears the other one for the launch that follows -- no fill)
	// (the turn is taken where the prepare kernel is launched: a call that fails before leaves both as they were)
	const u32 turn = ctx->dda_turn ^ 1u;
	u32 *list = (u32 *)ctx->wscan.p, *dcount = ctx->d_small + (turn ? UGRT_DSMALL_DDA_RAYS_B : UGRT_DSMALL_DDA_RAYS);
	u32 *dcount_next = ctx->d_small + (turn ? UGRT_DSMALL_DDA_RAYS : UGRT_DSMALL_DDA_RAYS_B);
	const bool use_rec = ctx->rec_valid && ctx->rec_verts == d_vertlist && ctx->rec_tris == d_trilist;
	const float4 *rec = use_rec ? (const float4 *)ctx->trirec.p : (const float4 *)nullptr;
	const bool counting = (ctx->cfg.flags & UGRT_FLAG_COUNT_WORK) != 0;
	unsigned long long *dc = (unsigned long long *)(ctx->d_small + UGRT_DSMALL_DDA);
	if (!counting)
		ugrt_prof_begin(ctx, UGRT_ST_WORKLIST);
	// (the window kernel's occupancy bitmap is written by extra workgroups of the same launch)
	const bool walk = ctx->opt[UGRT_OPT_DDA_KERNEL] <= 0;
	const u32 ncell_all = (u32)g.dims[0] * (u32)g.dims[1] * (u32)g.dims[2];
	if (walk && (rc = ugrt_buf_reserve(ctx, ctx->ubitmap, ((size_t)ncell_all + 63) / 64 * 8 + 8)))
		return rc;
	const u32 pix_blocks = (u32)((ctx->npix + 4 * DDA_PREP_SPAN - 1) / (4 * DDA_PREP_SPAN));
	const u32 bm_blocks = walk ? ((ncell_all + 255u) / 256u < 1024u ? (ncell_all + 255u) / 256u : 1024u) : 0u;
	// launch shape (ugrt_ctx_set_option; no effect on results): which kernel, rays per wave, list length from
	// which a lone ray's cell is tested by the whole wave, list length from which a shared cell is culled first
	// 0 window, 1 per-ray (the window kernel packs the steps left per axis into 10 bits each: ugrt_ctx_create admits at
	// most 1024 cells per axis)
	const int kernel = ctx->opt[UGRT_OPT_DDA_KERNEL] > 0 ? ctx->opt[UGRT_OPT_DDA_KERNEL] : 0;
	u32 DDA_RPW = ctx->opt[UGRT_OPT_DDA_RPW] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_RPW] : 32u;
	const u32 DDA_COOP = ctx->opt[UGRT_OPT_DDA_COOP] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_COOP] : 8u;
	const u32 CULL_MIN = ctx->opt[UGRT_OPT_DDA_CULL_MIN] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_CULL_MIN] : 8u;
	if (DDA_RPW > 64u)
		DDA_RPW = 64u;
	// split walks of the window kernel (ugrt_dda_walk.hip): list positions go into 28 bits of their merge key, and the
	// context's own grid tells how many there are
	WalkSplit sp = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr };
	WalkSplitHost sph = {};
	if (kernel == 0 && !counting && ctx->opt[UGRT_OPT_DDA_SORT] != 1 &&
	    (rc = ugrt_dda_split_state(ctx, DDA_RPW, d_span == (const unsigned *)G.span.p && d_offset == (const unsigned *)G.offset.p ? G.R : 0xFFFFFFFFu,
				       &sp, &sph)))
		return rc;
	hipLaunchKernelGGL(k_dda_prepare, dim3(pix_blocks + bm_blocks), dim3(256), 0, ctx->stream, d_active, ctx->p0, ctx->npix,
			   ctx->cfg.width, d_hit_t, d_hit_id, list, dcount, ctx->d_small + UGRT_DSMALL_TICKET, pix_blocks, d_span,
			   ncell_all, (u32 *)ctx->ubitmap.p, (u32 *)sp.chunk, dcount_next);
	ctx->dda_turn = turn;
	if (!counting) {
		ugrt_prof_end(ctx, UGRT_ST_WORKLIST);
		ugrt_prof_begin(ctx, UGRT_ST_TRACE_DDA);
	}
	UGRT_HIP(hipGetLastError());
	// the launch is persistent (groups of rays are drawn from a ticket); "dda_blocks" caps its waves, which a context
	// that runs beside another stream's kernels uses to leave registers and LDS of every CU to them
	int blocks = launch_blocks_for((u32)ctx->npix / DDA_RPW + 1u);
	if (ctx->opt[UGRT_OPT_DDA_BLOCKS] > 0 && blocks > ctx->opt[UGRT_OPT_DDA_BLOCKS])
		blocks = ctx->opt[UGRT_OPT_DDA_BLOCKS];
	// option dda_sort (SURVEY 8f.2 as written): the list sorted by (entry cell, octant) with the frame's pair sort
	if (ctx->opt[UGRT_OPT_DDA_SORT] == 1) {
		const u32 cap = (u32)list_cap;
		if ((rc = ugrt_buf_reserve(ctx, ctx->dsort, (size_t)cap * 12)))
			return rc;
		u32 *k0 = (u32 *)ctx->dsort.p, *k1 = k0 + cap, *l1 = k1 + cap;
		if ((rc = ugrt_dda_sort_keys_launch(ctx, g, d_rays, list, dcount, cap, k0)))
			return rc;
		int bits = 3, cb = 1;
		while ((1ull << cb) < (unsigned long long)g.dims[0] * g.dims[1] * g.dims[2])
			cb++;
		bits += cb;
		if (bits > 24)
			return ugrt_fail(UGRT_EINVAL, "trace_dda: dda_sort needs a grid of at most 2^21 cells");
		if ((rc = ugrt_prim_sort_pairs(ctx, k0, k1, list, l1, cap, bits, dcount)))
			return rc;
		list = l1;
	}
	if (kernel == 0) {
		// window kernel (ugrt_dda_walk.hip)
		if (counting)
			UGRT_HIP(hipMemsetAsync(dc, 0, DS_END * sizeof(unsigned long long), ctx->stream));
		if ((rc = ugrt_dda_walk_launch(ctx, g, d_value_list, d_span, d_offset, (u32 *)ctx->ubitmap.p, d_vertlist, d_trilist, rec,
					       d_rays, (const u32 *)list, (const u32 *)dcount, d_hit_t, d_hit_id,
					       counting ? dc : (unsigned long long *)nullptr, counting, DDA_RPW, CULL_MIN,
					       ctx->opt[UGRT_OPT_DDA_CULL_WORK] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_CULL_WORK] : 10u * DDA_RPW, blocks, sp, sph)))
			return rc;
		if (counting) {
			unsigned long long h[DS_END];
			UGRT_HIP(hipMemcpyAsync(h, dc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
			UGRT_HIP(hipStreamSynchronize(ctx->stream));
			ctx->stats[3] = h[0];
			ctx->stats[4] = h[1];
			ctx->stats[5] = h[2];
			for (int i = 0; i < UGRT_DDA_STATS; i++)
				ctx->dda_stats[i] = DS_ITER + i < DS_END ? h[DS_ITER + i] : 0ull;
			return UGRT_OK;
		}
		ugrt_prof_end(ctx, UGRT_ST_TRACE_DDA);
		return UGRT_OK;
	}
#define UGRT_LAUNCH_DDA(CNTV, RECV, DC)                                                                               \
	hipLaunchKernelGGL((k_trace_dda_ray<CNTV, RECV>), dim3(blocks), dim3(64), 0, ctx->stream, g, d_value_list, d_span,    \
			   d_offset, d_vertlist, d_trilist, rec, d_rays, (const u32 *)list, (const u32 *)dcount, d_hit_t,     \
			   d_hit_id, DC, DDA_RPW, DDA_COOP)
	if (counting) {
		// counting variant (never the timed one): same traversal + atomics per ray / per wave
		UGRT_HIP(hipMemsetAsync(dc, 0, DS_END * sizeof(unsigned long long), ctx->stream));
		if (use_rec)
			UGRT_LAUNCH_DDA(true, true, dc);
		else
			UGRT_LAUNCH_DDA(true, false, dc);
		UGRT_HIP(hipGetLastError());
		unsigned long long h[DS_END];
		UGRT_HIP(hipMemcpyAsync(h, dc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
		UGRT_HIP(hipStreamSynchronize(ctx->stream));
		ctx->stats[3] = h[0];
		ctx->stats[4] = h[1];
		ctx->stats[5] = h[2];
		for (int i = 0; i < UGRT_DDA_STATS; i++)
			ctx->dda_stats[i] = DS_ITER + i < DS_END ? h[DS_ITER + i] : 0ull;
		return UGRT_OK;
	}
	if (use_rec)
		UGRT_LAUNCH_DDA(false, true, (unsigned long long *)nullptr);
	else
		UGRT_LAUNCH_DDA(false, false, (unsigned long long *)nullptr);
#undef UGRT_LAUNCH_DDA
	ugrt_prof_end(ctx, UGRT_ST_TRACE_DDA);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// work sharing of the window kernel's last counting launch (UGRT_FLAG_COUNT_WORK), in the order of ugrt_dda_walk.hip's
// WS_* enum: windows, jobs, rays in jobs, cull batches, triangles culled against a bundle, exact-test rounds,
// (survivor, ray) pairs of those rounds, empty windows, [8..23] waves by log2(shader cycles / 4096), [24] sum and [25]
// maximum of the waves' cycles, then the cycles per phase
extern "C" int ugrt_stats_dda(ugrt_ctx *ctx, unsigned long long *stats, int n)
{
	if (!ctx || !stats || n < 0)
		return ugrt_fail(UGRT_EINVAL, "stats_dda: bad argument");
	for (int i = 0; i < n; i++)
		stats[i] = i < UGRT_DDA_STATS ? ctx->dda_stats[i] : 0ull;
	return UGRT_OK;
}
