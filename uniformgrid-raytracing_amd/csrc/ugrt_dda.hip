// ugrt_dda.hip -- reflection bounce: 3D-DDA through the uniform grid (not in the reference).
//
// Spec: clip the ray against the grid box, walk the cells front to back
// (Amanatides & Woo), test every triangle of a cell's list in ascending id with the reference's
// Moller-Trumbore (signed t, 0 < t < best), stop at the first cell whose best hit lies before the cell's
// exit.  Two kernels compute it:
//   k_trace_dda_beam (default) -- the rays of a wave are neighbours on the screen (the list is built in 8x8
//     tile order), so they walk the same cells at the same time.  The walk stays per lane and exact, but the
//     work inside a cell is shared: the lanes that stand in the same cell form a group, the cell's triangles
//     are loaded ONCE per group (lane = triangle), culled against the group's ray bundle with the interval
//     test of the primary/shadow tracers extended to rays with different origins, and only the survivors
//     are tested by the group's rays (lane = ray, triangle broadcast from registers).
//   k_trace_dda_ray (option "dda_kernel" = 1) -- round 1's kernel: every ray walks and tests alone, long
//     lists are tested by the whole wave for one owner at a time.  Kept as the before/after reference.
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


// ---------------------------------------------------------------------------
// beam kernel
// ---------------------------------------------------------------------------

#define BEAM_AHEAD 8 // cells planned (and their headers fetched) per round trip of the beam kernel
#define DDA_MAXLAG 7 // a ray may wait this many steps for the rays behind it (phase alignment, below)

// beam-kernel statistics (COUNT variant only): counters[3..]
enum { DS_ITER = 3, DS_GROUPS, DS_GROUP_LANES, DS_CULL_BATCHES, DS_CULL_TESTS, DS_EXACT_ROUNDS, DS_EXACT_LANES, DS_SOLO_ROUNDS,
       DS_HIST_CYCLES /* 16 buckets: waves by log2(cycles / 4096) */, DS_SUM_CYCLES = DS_HIST_CYCLES + 16, DS_MAX_CYCLES,
       DS_PHASE /* 8: cycles in plan+headers, job list, operand arrival, box, cull, exact rounds, lone rays, rest */, DS_PHASE_HEAVY = DS_PHASE + 8 /* the same for waves of >= 2^20 cycles, then their count and their rounds, groups, iterations */, DS_END = DS_PHASE_HEAVY + 12 };

// phase stamps of the COUNT variant: the cycles since the previous stamp go to phase PH
#define DDA_STAMP(PH)                                              \
	do {                                                       \
		if (COUNT) {                                       \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
			ph[PH] += now_ - tstamp;                   \
			tstamp = now_;                             \
		}                                                  \
	} while (0)

template <bool COUNT, bool REC>
__global__ __launch_bounds__(64, 3) void k_trace_dda_beam(DGrid g, const u32 *__restrict__ value_list,
							const u32 *__restrict__ span, const u32 *__restrict__ offset,
							const float *__restrict__ verts, const int *__restrict__ tris,
							const float4 *__restrict__ rec, const float *__restrict__ rays,
							const u32 *__restrict__ list, const u32 *__restrict__ count_p,
							float *__restrict__ hit_t, int *__restrict__ hit_id,
							unsigned long long *__restrict__ counters, u32 DDA_RPW, u32 DDA_COOP,
							u32 CULL_MIN, u32 *__restrict__ ticket)
{
	__shared__ u32 s_cell[BEAM_AHEAD][64], s_sp[BEAM_AHEAD][64], s_off[BEAM_AHEAD][64], s_flag[BEAM_AHEAD][64];
	__shared__ float s_tin[BEAM_AHEAD][64], s_tnext[BEAM_AHEAD][64];
	__shared__ u32 s_jend[BEAM_AHEAD];
	__shared__ unsigned short s_job[BEAM_AHEAD * 64]; // step << 8 | a lane that stands in the job's cell
	const int lane = threadIdx.x;
	const u32 count = *count_p;
	// The first group of a wave is its block index; further groups are drawn from a ticket, so a wave that is
	// done takes the next group whatever the others do.  (No ticket for the first group: thousands of waves
	// that start together would queue up on that one address, ~12 ns each.)
	for (u32 grp = blockIdx.x; (unsigned long long)grp * DDA_RPW < count;) {
		const unsigned long long clk0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
		unsigned long long tstamp = clk0, ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		const u32 slot = grp * DDA_RPW + (u32)lane;
		bool inb = (u32)lane < DDA_RPW && slot < count;
		const int p = inb ? (int)list[slot] : 0;
		inb = inb && p != -1; // (padding: k_dda_prepare)
		float res_t = -1.0f;
		int res_id = -2;
		u32 n_cells = 0, n_tests = 0;
		u32 st_iter = 0, st_groups = 0, st_glanes = 0, st_cb = 0, st_ct = 0, st_er = 0, st_el = 0, st_solo = 0;
		float o[3] = { 0, 0, 0 }, d[3] = { 0, 0, 0 }, tmax[3] = { 0, 0, 0 }, tdelta[3] = { 0, 0, 0 };
		int c[3] = { 0, 0, 0 }, step[3] = { 0, 0, 0 };
		float best_t = 3.0e38f, tcur = 0.0f;
		int best_id = -2;
		bool walking = false;
		// set-up: exactly the arithmetic of the per-ray kernel and of the specification
		if (inb) {
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
				tcur = tenter;
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
		// Phase alignment.  Every step moves a ray to a neighbour cell along its own direction signs, so
		// w = sx*cx + sy*cy + sz*cz grows by exactly one per step, and two rays of one octant can only meet
		// in a cell at equal w.  Rays that start up to DDA_MAXLAG steps ahead of the rearmost ray of their
		// cluster therefore wait that many steps: from then on neighbours stand in the same cell in the same
		// iteration, which is what the sharing below needs.  (Waiting changes no result.)
		int lag = 0;
		{
			const int w0 = step[0] * c[0] + step[1] * c[1] + step[2] * c[2];
			bool open = walking;
			for (int pass = 0; pass < 4 && __ballot(open) != 0ull; pass++) {
				const int wmin = d_wave_imin(open ? w0 : 0x7FFFFFFF);
				if (open && w0 - wmin <= DDA_MAXLAG) {
					lag = w0 - wmin;
					open = false;
				}
			}
		}
		// every step leaves a cell for good, so dims[0]+dims[1]+dims[2] bounds the walk
		int guard = g.dims[0] + g.dims[1] + g.dims[2] + 3;
		u32 njobs = 0;
		while (__ballot(walking) != 0ull) {
			if (COUNT)
				st_iter++;
			DDA_STAMP(7);
			// 1. plan the next BEAM_AHEAD cells and fetch their headers together (one round trip per BEAM_AHEAD
			//    steps).  The plan lives in LDS, every lane its own column.
			{
				u32 pcell[BEAM_AHEAD], pflag[BEAM_AHEAD];
				bool planning = walking;
#pragma unroll
				for (int q = 0; q < BEAM_AHEAD; q++) {
					pflag[q] = 0u;
					pcell[q] = 0u;
					s_tin[q][lane] = tcur;
					if (planning && lag > 0) {
						lag--;
					} else if (planning) {
						pflag[q] = 1u;
						pcell[q] = (u32)((c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2]);
						int ax = (tmax[0] < tmax[1]) ? ((tmax[0] < tmax[2]) ? 0 : 2) : ((tmax[1] < tmax[2]) ? 1 : 2);
						tcur = ax == 0 ? tmax[0] : (ax == 1 ? tmax[1] : tmax[2]);
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
							pflag[q] = 3u; // the walk ends after this cell unless it ends there with a hit
							planning = false;
						}
					}
					s_tnext[q][lane] = tcur;
					s_cell[q][lane] = pcell[q];
				}
				u32 psp[BEAM_AHEAD], poff[BEAM_AHEAD];
#pragma unroll
				for (int q = 0; q < BEAM_AHEAD; q++) {
					psp[q] = pflag[q] ? span[pcell[q]] : 0u;
					poff[q] = pflag[q] ? offset[pcell[q]] : 0u;
				}
				// 2. jobs: the distinct non-empty cells of every step (a cell can only be met by rays that are in
				//    phase, i.e. in the same step).  Lone rays with short lists test alone (flag 4).
				if (COUNT) {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					DDA_STAMP(0);
				}
				u32 nj = 0;
#pragma unroll
				for (int q = 0; q < BEAM_AHEAD; q++) {
					s_sp[q][lane] = psp[q];
					s_off[q][lane] = poff[q];
					unsigned long long todo = __ballot(psp[q] != 0u);
					while (todo != 0ull) {
						const int l = (int)__builtin_ctzll(todo);
						const u32 X = (u32)__builtin_amdgcn_readlane((int)pcell[q], l);
						const unsigned long long grpm = __ballot(psp[q] != 0u && pcell[q] == X);
						todo &= ~grpm;
						const u32 S = (u32)__builtin_amdgcn_readlane((int)psp[q], l);
						if ((grpm & (grpm - 1ull)) == 0ull && S < DDA_COOP) {
							if (lane == l)
								pflag[q] |= 4u;
							continue;
						}
						if (lane == 0)
							s_job[nj] = (unsigned short)((q << 8) | l);
						nj++;
					}
					s_flag[q][lane] = pflag[q];
					if (lane == 0)
						s_jend[q] = nj;
				}
				njobs = nj;
				DDA_STAMP(1);
			}
			// 3. the jobs in step order; the triangle ids of job j+2 and the records of job j+1 are in flight while
			//    job j is tested
			u32 idA = 0u, fN = 0u; // this lane's triangle of job j+2 and of job j+1
			float rN[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define DDA_JOB_LEN(J) (s_sp[s_job[J] >> 8][s_job[J] & 63u])
#define DDA_JOB_BASE(J) (s_off[s_job[J] >> 8][s_job[J] & 63u])
#define DDA_JOB_CELL(J) (s_cell[s_job[J] >> 8][s_job[J] & 63u])
			// (every lane loads, beyond the end of a list the last triangle again, and past the last job the last
			// job again: the number of loads in flight does not depend on the data, so the compiler can count them)
#define DDA_JOB_ID(J) value_list[DDA_JOB_BASE(J) + min((u32)lane, DDA_JOB_LEN(J) - 1u)]
			if (njobs > 0u) {
				fN = DDA_JOB_ID(0u);
				idA = DDA_JOB_ID(min(1u, njobs - 1u));
				d_load_record<REC>(rec, verts, tris, fN, rN);
			}
			u32 j = 0;
#pragma unroll 1
			for (int q = 0; q < BEAM_AHEAD; q++) {
				const u32 flag = s_flag[q][lane];
				const bool here = walking && (flag & 1u);
				const u32 sp = here ? s_sp[q][lane] : 0u, off = s_off[q][lane], cell = s_cell[q][lane];
				const float tin = s_tin[q][lane], tnext = s_tnext[q][lane];
				if (COUNT && here) {
					n_cells++;
					n_tests += sp;
				}
				const u32 jend = s_jend[q];
#pragma unroll 1
				for (; j < jend; j++) {
					// rotate the pipeline: this job's registers, next job's records, the ids of the one after
					u32 f0 = fN;
					float r0[9];
#pragma unroll
					for (int k = 0; k < 9; k++)
						r0[k] = rN[k];
					// this job's operands must have arrived here, before the next loads are issued
					asm volatile("" : "+v"(f0), "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "+v"(r0[4]), "+v"(r0[5]),
						     "+v"(r0[6]), "+v"(r0[7]), "+v"(r0[8]), "+v"(idA));
					DDA_STAMP(2);
					fN = idA;
					d_load_record<REC>(rec, verts, tris, fN, rN);
					idA = DDA_JOB_ID(min(j + 2u, njobs - 1u));
					const u32 X = DDA_JOB_CELL(j), base = DDA_JOB_BASE(j), S = DDA_JOB_LEN(j);
					const bool in = sp != 0u && cell == X;
					const unsigned long long grpm = __ballot(in);
					if (grpm == 0ull)
						continue; // every ray of the group ended in an earlier step of this block
					const bool use_cull = S >= CULL_MIN;
					BeamBox bx;
					if (use_cull)
						bx = d_beam_box(o, d, tin, in);
					DDA_STAMP(3);
					if (COUNT) {
						st_groups++;
						st_glanes += (u32)__popcll(grpm);
					}
					for (u32 b = 0; b < S; b += 64u) {
						const bool have = b + (u32)lane < S;
						u32 f = f0;
						float r9[9];
#pragma unroll
						for (int k = 0; k < 9; k++)
							r9[k] = r0[k];
						if (b != 0u && have) { // lists beyond 64 triangles: the later batches are fetched here
							f = value_list[base + b + (u32)lane];
							d_load_record<REC>(rec, verts, tris, f, r9);
						}
						bool keep = have;
						if (use_cull && have)
							keep = !d_cull_beam(&r9[0], &r9[3], &r9[6], bx);
						unsigned long long m = __ballot(keep);
						DDA_STAMP(4);
						if (COUNT && use_cull) {
							st_cb++;
							st_ct += (u32)__popcll(__ballot(have));
						}
						// survivors in list order (ascending id: the strict '<' keeps the first of equal t)
						while (m != 0ull) {
							const int s = (int)__builtin_ctzll(m);
							m &= m - 1ull;
							float bt[9];
#pragma unroll
							for (int k = 0; k < 9; k++)
								bt[k] = d_readlane(r9[k], s);
							const u32 bf = (u32)__builtin_amdgcn_readlane((int)f, s);
							if (COUNT) {
								st_er++;
								st_el += (u32)__popcll(grpm);
							}
							if (in) {
								const float tv[3] = { o[0] - bt[0], o[1] - bt[1], o[2] - bt[2] };
								float t;
								if (d_mt_core(tv, &bt[3], &bt[6], d, &t) && t > 0.0f && t < best_t) {
									best_t = t;
									best_id = (int)bf;
								}
							}
						}
						DDA_STAMP(5);
					}
				}
				DDA_STAMP(7);
				if (__ballot(here && (flag & 4u)) != 0ull) {
					const u32 ns = (here && (flag & 4u)) ? sp : 0u;
					for (u32 r = 0; r < ns; r++) {
						const u32 f = value_list[off + r];
						float t9[9], t;
						d_load_triangle<REC>(rec, verts, tris, f, o[0], o[1], o[2], t9);
						if (d_mt_core(&t9[0], &t9[3], &t9[6], d, &t) && t > 0.0f && t < best_t) {
							best_t = t;
							best_id = (int)f;
						}
						if (COUNT)
							st_solo++;
					}
				}
				DDA_STAMP(6);
				if (here) {
					if (best_id >= 0 && best_t <= tnext) {
						res_t = best_t;
						res_id = best_id;
						walking = false;
					} else if (flag & 2u) {
						walking = false;
					}
				}
			}
		}
		if (inb) {
			hit_t[p] = res_t;
			hit_id[p] = res_id;
		}
		if (COUNT) {
			if (inb) {
				if (n_tests)
					atomicAdd(&counters[0], (unsigned long long)n_tests);
				if (n_cells)
					atomicAdd(&counters[1], (unsigned long long)n_cells);
				atomicAdd(&counters[2], 1ull);
				if (st_solo)
					atomicAdd(&counters[DS_SOLO_ROUNDS], (unsigned long long)st_solo);
			}
			if (lane == 0) { // wave-uniform counts
				atomicAdd(&counters[DS_ITER], (unsigned long long)st_iter);
				atomicAdd(&counters[DS_GROUPS], (unsigned long long)st_groups);
				atomicAdd(&counters[DS_GROUP_LANES], (unsigned long long)st_glanes);
				atomicAdd(&counters[DS_CULL_BATCHES], (unsigned long long)st_cb);
				atomicAdd(&counters[DS_CULL_TESTS], (unsigned long long)st_ct);
				atomicAdd(&counters[DS_EXACT_ROUNDS], (unsigned long long)st_er);
				atomicAdd(&counters[DS_EXACT_LANES], (unsigned long long)st_el);
				const unsigned long long cyc = __builtin_amdgcn_s_memtime() - clk0;
				int bucket = 0;
				while (bucket < 15 && (cyc >> (12 + bucket)) > 1ull)
					bucket++;
				atomicAdd(&counters[DS_HIST_CYCLES + bucket], 1ull);
				atomicAdd(&counters[DS_SUM_CYCLES], cyc);
				for (int k = 0; k < 8; k++)
					atomicAdd(&counters[DS_PHASE + k], ph[k]);
				if (cyc >= (1ull << 20)) {
					for (int k = 0; k < 8; k++)
						atomicAdd(&counters[DS_PHASE_HEAVY + k], ph[k]);
					atomicAdd(&counters[DS_PHASE_HEAVY + 8], 1ull);
					atomicAdd(&counters[DS_PHASE_HEAVY + 9], (unsigned long long)st_er);
					atomicAdd(&counters[DS_PHASE_HEAVY + 10], (unsigned long long)st_groups);
					atomicAdd(&counters[DS_PHASE_HEAVY + 11], (unsigned long long)st_iter);
				}
				atomicMax(&counters[DS_MAX_CYCLES], cyc);
			}
		}
		if (lane == 0)
			grp = gridDim.x + atomicAdd(ticket, 1u);
		grp = (u32)__builtin_amdgcn_readfirstlane((int)grp);
	} // groups
}


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
	// 0 window, 1 per-ray, 2 beam (the window kernel packs the steps left per axis into 10 bits each: ugrt_ctx_create
	// admits at most 1024 cells per axis)
	const int kernel = ctx->opt[UGRT_OPT_DDA_KERNEL] > 0 ? ctx->opt[UGRT_OPT_DDA_KERNEL] : 0;
	const bool beam = kernel != 1;
	u32 DDA_RPW = ctx->opt[UGRT_OPT_DDA_RPW] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_RPW] : (kernel == 2 ? 64u : 32u);
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
	do {                                                                                                          \
		if (beam)                                                                                             \
			hipLaunchKernelGGL((k_trace_dda_beam<CNTV, RECV>), dim3(blocks), dim3(64), 0, ctx->stream, g,  \
					   d_value_list, d_span, d_offset, d_vertlist, d_trilist, rec, d_rays,         \
					   (const u32 *)list, (const u32 *)dcount, d_hit_t, d_hit_id, DC, DDA_RPW,     \
					   DDA_COOP, CULL_MIN, ctx->d_small + UGRT_DSMALL_TICKET);                                                      \
		else                                                                                                  \
			hipLaunchKernelGGL((k_trace_dda_ray<CNTV, RECV>), dim3(blocks), dim3(64), 0, ctx->stream, g,   \
					   d_value_list, d_span, d_offset, d_vertlist, d_trilist, rec, d_rays,         \
					   (const u32 *)list, (const u32 *)dcount, d_hit_t, d_hit_id, DC, DDA_RPW,     \
					   DDA_COOP);                                                                  \
	} while (0)
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

// work sharing of the beam kernel's last counting launch (UGRT_FLAG_COUNT_WORK): [0] wave iterations (blocks of
// BEAM_AHEAD steps), [1] cell groups processed, [2] rays in those groups, [3] cull batches, [4] triangles culled
// against a bundle, [5] exact-test rounds (one broadcast triangle), [6] rays that ran those rounds,
// [7] exact tests of lone rays, [8..23] waves by log2(shader cycles / 4096), [24] sum and [25] maximum of the
// waves' cycles
extern "C" int ugrt_stats_dda(ugrt_ctx *ctx, unsigned long long *stats, int n)
{
	if (!ctx || !stats || n < 0)
		return ugrt_fail(UGRT_EINVAL, "stats_dda: bad argument");
	for (int i = 0; i < n; i++)
		stats[i] = i < UGRT_DDA_STATS ? ctx->dda_stats[i] : 0ull;
	return UGRT_OK;
}
