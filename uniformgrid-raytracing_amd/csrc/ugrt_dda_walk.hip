// ugrt_dda_walk.hip -- reflection bounce, window kernel (round 3; DESIGN.md A13, section 5 "Window DDA").
//
// Same specification and the same arithmetic as k_trace_dda_ray / k_trace_dda_beam (ugrt_dda.hip): clip, walk the
// cells front to back (Amanatides & Woo), per cell every triangle in ascending id with the reference's
// Moller-Trumbore (signed t, 0 < t < best), stop at the first cell whose best hit lies before the cell's exit.
// What the round-2 beam kernel spent its time on, by its own phase stamps (profiles/r02_dda_phases.json), and
// what this kernel does about it:
//   * 83 % of the cells a ray visits are EMPTY, and 60 % of a wave's blocks of 8 steps hold no triangle for any of
//     its 64 rays, yet every step fetched span[] and offset[] of its cell (1024 divergent 4-byte gathers per wave
//     and block: 31 % of all wave cycles, bound by the CU's address path).  Here the walk consults a BITMAP of the
//     occupied cells (1 bit per cell, 128 KB for 128x128x64: cache resident, neighbouring rays share its lines); a
//     window of 8 steps without an occupied cell costs the plan and one short round trip, and the headers are
//     fetched per JOB (one lane per distinct cell), not per ray and step.
//   * A ray's stop test `best_t <= exit of the cell` only has to be evaluated where something can change: tnext
//     grows monotonically along the walk and best_t changes only in occupied cells, so "the ray stopped in one of
//     the empty cells before cell k" is the single comparison best_t <= t_in(k).  Empty steps therefore cost no
//     LDS traffic at all (the beam kernel re-read six plan words per ray and step: its "rest" phase, 15 %).
//   * Exact tests ran one broadcast triangle per round with ~18-24 of 64 lanes busy.  Here lanes are
//     (survivor, ray) PAIRS: a job with n <= 32 rays tests floor(64 / n) survivors per round, rays and survivors
//     come from LDS, closest hits are merged per ray by ds_min_u64 on (t bits << 32 | list position) -- the strict
//     `<` of the sequential loop in list order.  Lone rays are jobs of one ray (64 triangles per round).
// Results are bit-identical to the other two kernels (tests/test_gpu_parity.py checks all three against the CPU restatement).
#include "ugrt_dda.h"

#define WK_AHEAD 8    // steps planned per window
#define WK_MAXLAG 7   // phase alignment: a ray may wait this many steps for the rays behind it
#define WK_JOBCAP 64  // jobs listed per pass: job j lives in lane j's registers (a step has at most 64 jobs)
#define WK_NONE 0xFFFFFFFFu

// statistics of the counting variant: counters[0..2] = tests, cells, rays (the algorithmic-byte formula), then
enum { WS_WINDOWS = 3, WS_JOBS, WS_JOB_RAYS, WS_CULL_BATCHES, WS_CULL_TESTS, WS_ROUNDS, WS_ROUND_PAIRS, WS_EMPTY_WINDOWS,
       WS_HIST /* 16: waves by log2(cycles / 4096) */, WS_SUM_CYCLES = WS_HIST + 16, WS_MAX_CYCLES,
       WS_PHASE /* 8: plan+bitmap, job list+headers, operand arrival, box, cull, exact rounds, settle, rest */,
       WS_PHASE_HEAVY = WS_PHASE + 8 /* the same for waves of >= 2^19 cycles, then their count, rounds, jobs, windows */,
       WS_END = WS_PHASE_HEAVY + 12 };
static_assert(WS_END <= 3 + UGRT_DDA_STATS, "window kernel statistics");

// key of the optional ray sort (SURVEY 8f.2 as written: entry cell, then direction octant)
__global__ __launch_bounds__(256) void k_dda_sort_keys(DGrid g, const float *__restrict__ rays, const u32 *__restrict__ list,
						       const u32 *__restrict__ count_p, u32 cap, u32 *__restrict__ keys)
{
	const u32 i = blockIdx.x * 256u + threadIdx.x;
	if (i >= cap)
		return;
	if (i >= *count_p) {
		keys[i] = 0xFFFFFFFFu >> 8; // padding sorts last (the sort runs over the capacity)
		return;
	}
	const u32 p = list[i];
	if (p == 0xFFFFFFFFu) { // (the list's own padding: k_dda_prepare)
		keys[i] = 0xFFFFFFFFu >> 8;
		return;
	}
	float o[3], d[3], tenter = 0.0f;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		o[k] = rays[p * 6 + k];
		d[k] = rays[p * 6 + 3 + k];
	}
#pragma unroll
	for (int k = 0; k < 3; k++)
		if (d[k] != 0.0f) {
			const float inv = 1.0f / d[k];
			const float lo = g.lo[k], hi = g.lo[k] + g.cs[k] * (float)g.dims[k];
			const float t0 = (lo - o[k]) * inv, t1 = (hi - o[k]) * inv;
			const float tn = t0 < t1 ? t0 : t1;
			tenter = tn > tenter ? tn : tenter;
		}
	u32 cell = 0;
#pragma unroll
	for (int k = 0; k < 3; k++)
		cell = cell * (u32)g.dims[k] + (u32)d_dcell(g, k, o[k] + tenter * d[k]);
	const u32 oct = (d[0] < 0.0f ? 1u : 0u) | (d[1] < 0.0f ? 2u : 0u) | (d[2] < 0.0f ? 4u : 0u);
	keys[i] = (cell << 3) | oct;
}

// Lane selects by a 64-bit lane mask held in a scalar register pair (a bit set = take the second value): one
// v_cndmask each.  The plan's step is written with these and with masks from __ballot: a step is three compares, a
// handful of scalar mask operations and ~25 selects/adds.  (As C++ `bool`s the compiler turned every mask into a
// 0/1 lane value and back: ~60 vector instructions per step, and the plan was 30 % of the kernel's.)
typedef unsigned long long m64;
__device__ __forceinline__ float d_msel(float a, float b, m64 m)
{
	float r;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
	return r;
}
__device__ __forceinline__ u32 d_msel(u32 a, u32 b, m64 m)
{
	u32 r;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
	return r;
}
__device__ __forceinline__ u32 d_msel0(u32 b, m64 m) // m ? b : 0
{
	u32 r;
	asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(b), "s"(m));
	return r;
}
__device__ __forceinline__ u32 d_mbit(m64 m) // m ? 1 : 0
{
	u32 r;
	asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
	return r;
}
// v_writelane_b32: lane `l` of r := v (both wave-uniform); the other lanes keep theirs
__device__ __forceinline__ u32 d_writelane(u32 r, u32 v, u32 l)
{
	// (one scalar operand beside m0: the constant bus; m0 is put back as it was found)
	u32 keep;
	asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(r), "=&s"(keep) : "s"(v), "s"(l));
	return r;
}
// bit offset of the chosen axis' field in the packed step counters: axis 0 -> 0, 1 -> 10, 2 -> 20
__device__ __forceinline__ u32 d_axis_shift(m64 a0, m64 a1)
{
	u32 r, q;
	asm("v_cndmask_b32_e64 %0, 20, 10, %1" : "=v"(q) : "s"(a1));
	asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(q), "s"(a0));
	return r;
}

#define WK_STAMP(PH)                                                          \
	do {                                                                  \
		if (COUNT) {                                                  \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
			ph[PH] += now_ - tstamp;                              \
			tstamp = now_;                                        \
		}                                                             \
	} while (0)

// SPLIT false: the kernel without split walks (no history is read or written): every `sp.` below is then a null the
// compiler folds away -- 40 scalar registers, the redo / segment tests of every window and the spills that came with them
template <bool COUNT, bool REC, bool SPLIT>
__global__ __launch_bounds__(64, 3) void k_trace_dda_walk(DGrid g, const u32 *__restrict__ value_list,
							const u32 *__restrict__ span, const u32 *__restrict__ offset,
							const u32 *__restrict__ bitmap, const float *__restrict__ verts,
							const int *__restrict__ tris, const float4 *__restrict__ rec,
							const float *__restrict__ rays, const u32 *__restrict__ list,
							const u32 *__restrict__ count_p, float *__restrict__ hit_t,
							int *__restrict__ hit_id, unsigned long long *__restrict__ counters,
							u32 RPW, u32 CULL_MIN, u32 CULL_WORK, u32 *__restrict__ ticket, WalkSplit sp_in)
{
	const WalkSplit sp = SPLIT ? sp_in : WalkSplit{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr };
	__shared__ u32 s_cell[WK_AHEAD][64];    // cell of (step, ray); written for occupied cells only
	__shared__ float s_tnext[WK_AHEAD][64]; // exit parameter of (step, ray); the entry of step q is the exit of q - 1
	__shared__ __attribute__((aligned(16))) float s_surv[64 * TRI_STRIDE]; // survivors of a batch: 9 floats + list position
	__shared__ __attribute__((aligned(16))) float s_ray[64 * 8];           // the group's rays {o, d}
	__shared__ unsigned long long s_best[64];                            // per ray: closest hit of the running job
	__shared__ unsigned char s_rank[64]; // k-th ray of the running job
	const int lane = threadIdx.x;
	const u32 count = *count_p;
	// work items: the ray groups, the long ones of the last launch cut into segments of windows (sp.items; "Split
	// walks" below); without a list, item i is group i whole
	const u32 ncut = sp.items ? sp.hdr[2] : 0u; // the cut groups' segments come first, then every group in turn
	const u32 nitems = ncut + (u32)(((unsigned long long)count + RPW - 1u) / RPW);
	m64 redo = 0ull; // rays of a cut group whose merged result has to be walked again in one piece
	for (u32 it = blockIdx.x; it < nitems;) {
		const unsigned long long clk0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
		unsigned long long tstamp = clk0, ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		u32 grp = it - ncut, seg = 0u, nseg = 1u, wbeg = 0u, wend = 0xFFFFu;
		if (it < ncut) {
			const uint2 d = sp.items[it];
			grp = d.x & 0xFFFFFFu;
			if (redo == 0ull) {
				seg = (d.x >> 24) & 15u, nseg = d.x >> 28, wbeg = d.y & 0xFFFFu, wend = d.y >> 16;
			} else if (lane == 0) {
				// (ugrt_stats_dda_split, and what the next launch judges the history by.  Counted here, where the ray goes on:
				// next to the merge the same two lines cost 150 bytes of spilled registers per lane)
				atomicAdd(&sp.hdr[4], 1u);
				atomicAdd(&sp.hdr[5], (u32)__popcll(redo));
			}
		} else if (sp.cut && sp.cut[grp]) { // (a cut group's turn as a whole group: nothing to do)
			if (lane == 0)
				it = gridDim.x + atomicAdd(ticket, 1u);
			it = (u32)__builtin_amdgcn_readfirstlane((int)it);
			continue;
		}
		const u32 slot = grp * RPW + (u32)lane;
		bool inb = (u32)lane < RPW && slot < count;
		const int p = inb ? (int)list[slot] : 0;
		inb = inb && p != -1; // (the list is padded to whole chunks of 64 entries per span of pixels: k_dda_prepare)
		const u32 pb = (u32)p - sp.p0; // the ray's index in the split walks' per-pixel arrays
		u32 n_cells = 0, n_tests = 0;
		u32 st_win = 0, st_empty = 0, st_jobs = 0, st_jrays = 0, st_cb = 0, st_ct = 0, st_rounds = 0, st_pairs = 0;
		// the walk's state: exit parameters per axis and their increments (the specification's tmax / tdelta), the
		// current cell as its linear index with the signed index step per axis, and the steps left per axis before the
		// ray leaves the grid (0 for an axis the ray does not move along: choosing it ends the walk, as in the
		// specification's `step == 0 || c out of range`)
		float tmax[3] = { 0, 0, 0 }, tdelta[3] = { 0, 0, 0 };
		u32 cidx = 0u, remp = 0u; // remp: the three axes' steps left, 10 bits each
		int cstep[3] = { 0, 0, 0 };
		int w0 = 0; // phase of the ray's first cell (alignment below)
		float best_t = 3.0e38f, tcur = 0.0f;
		u32 best_ref = WK_NONE; // list position of the closest hit so far (value_list[best_ref] is its triangle)
		bool walking = false, hitstop = false;
		bool best_behind = false;               // (cut groups) the closest hit lies before the entry of the cell it was found in
		bool left = false;                      // (cut groups) the ray has left the grid (tcur is then the exit of its last cell) or misses it
		u32 widx = 0u;                          // window number
		u32 nwin = 0u;                          // windows this ray has walked
		// set-up: exactly the arithmetic of the per-ray kernel and of the specification
		{
			float o[3] = { 0, 0, 0 }, d[3] = { 0, 0, 0 };
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
					const int stride[3] = { g.dims[1] * g.dims[2], g.dims[2], 1 };
#pragma unroll
					for (int k = 0; k < 3; k++) {
						float pe = o[k] + tenter * d[k];
						const int ck = d_dcell(g, k, pe);
						cidx += (u32)(ck * stride[k]);
						if (d[k] > 0.0f) {
							tmax[k] = ((g.lo[k] + (float)(ck + 1) * g.cs[k]) - o[k]) / d[k];
							tdelta[k] = g.cs[k] / d[k];
							cstep[k] = stride[k];
							remp |= (u32)(g.dims[k] - 1 - ck) << (10 * k);
							w0 += ck;
						} else if (d[k] < 0.0f) {
							tmax[k] = ((g.lo[k] + (float)ck * g.cs[k]) - o[k]) / d[k];
							tdelta[k] = -g.cs[k] / d[k];
							cstep[k] = -stride[k];
							remp |= (u32)ck << (10 * k);
							w0 -= ck;
						} else {
							tmax[k] = 3.0e38f;
							tdelta[k] = 3.0e38f;
						}
					}
				}
			}
			// the rays live in LDS from here on: the walk does not need them, the jobs read them by ray index
			*reinterpret_cast<float4 *>(&s_ray[lane * 8]) = make_float4(o[0], o[1], o[2], d[0]);
			*reinterpret_cast<float2 *>(&s_ray[lane * 8 + 4]) = make_float2(d[1], d[2]);
			s_best[lane] = ~0ull;
		}
		left = inb && !walking; // (cut groups: a ray that misses the grid has "left" it)
		// Phase alignment (as in the beam kernel): w = sx*cx + sy*cy + sz*cz grows by one per step, rays of one octant
		// can only meet in a cell at equal w, so rays up to WK_MAXLAG steps ahead of the rearmost ray of their
		// cluster wait that many steps.  (Waiting changes no result.)
		int lag = 0;
		{
			bool open = walking;
			for (int pass = 0; pass < 4 && __ballot(open) != 0ull; pass++) {
				const int wmin = d_wave_imin(open ? w0 : 0x7FFFFFFF);
				if (open && w0 - wmin <= WK_MAXLAG) {
					lag = w0 - wmin;
					open = false;
				}
			}
		}
		// a later segment of a cut group is given the rays that walked as far as its first window in the launch before
		// (after the alignment, which has to be the one of the whole group: a ray is in the same cells in window w in
		// every segment)
		u32 wprev = 0xFFFFFFFFu; // windows the ray walked in the launch before: a cut group looks at no cell of the ray beyond them
		if (nseg > 1u) {
			wprev = inb ? sp.walked_prev[pb] : 0u;
			if (seg != 0u)
				walking = walking && wprev > wbeg;
		} else if (redo != 0ull) {
			// (only now: the alignment above has to be the whole group's, or the windows would not be the segments' windows)
			inb = inb && ((redo >> lane) & 1ull);
			walking = walking && inb;
		}
		if (nseg == 1u && redo != 0ull && inb) {
			// a ray of a cut group whose merged result could not be vouched for goes on from where the segments stopped
			// looking: it carries their closest hit along and looks at no cell of the windows they covered (wprev of
			// them) -- or starts afresh, when that hit was one found behind its cell by a later segment
			const unsigned long long k = sp.key[pb];
			sp.key[pb] = ~0ull;
			const bool afresh = k != ~0ull && ((k >> 28) & 1ull);
			if (k != ~0ull && !afresh) {
				best_t = __uint_as_float((u32)(k >> 32));
				best_ref = (u32)k & 0x0FFFFFFFu;
			}
			wprev = afresh ? 0u : sp.walked[pb];
		}
		// (The specification also bounds the walk by dims[0]+dims[1]+dims[2]+3 steps.  Every step that stays inside
		// uses up one of the sum(dims) - 3 steps the three axes have left, so that bound is never reached and is not
		// carried along here.)
		__syncthreads();
		while (__ballot(walking) != 0ull) {
			if (widx == wend)
				break; // the segment ends here; its rays that still walk are another segment's from this window on
			// windows before the segment, and windows the segments have covered for every ray of a group that is gone over
			// again: the walk alone, no cell is looked at
			const bool ff = widx < wbeg || (redo != 0ull && __ballot(walking && widx >= wprev) == 0ull);
			u32 wjobs = 0u;
			if (nseg > 1u && widx >= wprev)
				walking = false; // (as far as the ray went the last time: whether that was far enough is settled with the merge)
			nwin += walking ? 1u : 0u;
			if (COUNT)
				st_win++;
			WK_STAMP(7);
			// 1. plan WK_AHEAD steps (registers only) and look their cells up in the occupancy bitmap
			const float tstart = tcur;
			u32 pcell[WK_AHEAD], vmask, nem = 0u;
			float ptn[WK_AHEAD];
			bool ended;
			{
				// masks: P = rays still planning, E = rays that leave the grid in this window
				m64 P = __ballot(walking), E = 0ull;
				const float tstop = best_ref != WK_NONE ? best_t : __builtin_huge_valf(); // the hit carried along
				const u32 vfirst = (u32)lag; // a ray ahead of its cluster waits `lag` steps first (first window only)
				u32 vcount = 0u;
				const bool lagging = __ballot(lag > 0) != 0ull;
				// axis = (tx < ty) ? ((tx < tz) ? x : z) : ((ty < tz) ? y : z), as in the specification
#pragma unroll
				for (int q = 0; q < WK_AHEAD; q++) {
					m64 A = P;
					if (lagging) { // (wave-uniform: only the first window of a group has waiting rays)
						const m64 L = __ballot(lag > 0);
						lag -= (int)d_mbit(P & L);
						A = P & ~L;
					}
					pcell[q] = cidx;
					const m64 xy = __ballot(tmax[0] < tmax[1]), xz = __ballot(tmax[0] < tmax[2]), yz = __ballot(tmax[1] < tmax[2]);
					const m64 a0 = xy & xz, a1 = ~xy & yz;
					const float tn = d_msel(d_msel(tmax[2], tmax[1], a1), tmax[0], a0);
					const float tnew = tn + d_msel(d_msel(tdelta[2], tdelta[1], a1), tdelta[0], a0);
					const u32 cs = d_msel(d_msel((u32)cstep[2], (u32)cstep[1], a1), (u32)cstep[0], a0);
					const u32 sh = d_axis_shift(a0, a1);
					const m64 out = __ballot((remp & (0x3FFu << sh)) == 0u); // no step left along the chosen axis: the ray leaves
					tmax[0] = d_msel(tmax[0], tnew, A & a0);
					tmax[1] = d_msel(tmax[1], tnew, A & a1);
					tmax[2] = d_msel(tmax[2], tnew, A & ~(a0 | a1));
					remp -= d_msel0(1u << sh, A); // (wraps when the ray leaves: never read again)
					cidx += d_msel0(cs, A & ~out);
					tcur = d_msel(tcur, tn, A);
					vcount += d_mbit(A);
					// the walk ends after this cell when the ray leaves the grid, and at the latest here when the hit carried
					// along lies before this cell's exit
					const m64 stop = __ballot(tstop <= tn);
					E |= A & out;
					P &= ~(A & (out | stop));
					ptn[q] = tcur;
				}
				ended = (E >> lane) & 1ull;
				vmask = ((1u << vcount) - 1u) << vfirst; // the steps this ray took: a run of `vcount` from `vfirst`
				if (!ff) {
					u32 bw[WK_AHEAD];
#pragma unroll
					for (int q = 0; q < WK_AHEAD; q++)
						bw[q] = bitmap[pcell[q] >> 5];
#pragma unroll
					for (int q = 0; q < WK_AHEAD; q++)
						nem |= (((vmask >> q) & (bw[q] >> (pcell[q] & 31u))) & 1u) << q;
					if (redo != 0ull && widx < wprev)
						nem = 0u; // (this ray's cells of the window have been looked at by a segment)
				}
			}
			if (COUNT) {
				// (every slot's exit is needed to find the step a ray stopped in)
#pragma unroll
				for (int q = 0; q < WK_AHEAD; q++)
					s_tnext[q][lane] = ptn[q];
			}
			u32 qmask = 0u; // steps with an occupied cell on any ray
#pragma unroll
			for (int q = 0; q < WK_AHEAD; q++)
				qmask |= (__ballot((nem >> q) & 1u) != 0ull ? 1u : 0u) << q;
			WK_STAMP(0);
			if (qmask != 0u) {
				if (!COUNT) {
#pragma unroll
					for (int q = 0; q < WK_AHEAD; q++)
						s_tnext[q][lane] = ptn[q];
				}
#pragma unroll
				for (int q = 0; q < WK_AHEAD; q++)
					if ((nem >> q) & 1u)
						s_cell[q][lane] = pcell[q];
				__syncthreads();
				// 2. jobs = the distinct occupied cells of every step (rays that are in phase meet in the same step), listed
				//    in step order, at most WK_JOBCAP per pass
				u32 qnext = 0u;
				while (qnext < (u32)WK_AHEAD) {
					// job j of the pass is held by lane j: its cell, its step, and (below) its list; the job loop reads them
					// with v_readlane into scalar registers (as LDS arrays every job began with four dependent LDS round trips)
					u32 njobs = 0u, jc = 0u, jq = 0u;
					for (; qnext < (u32)WK_AHEAD; qnext++) {
						if (!((qmask >> qnext) & 1u))
							continue;
						const bool part = walking && ((nem >> qnext) & 1u);
						unsigned long long todo = __ballot(part);
						if (njobs != 0u && njobs + (u32)__popcll(todo) > (u32)WK_JOBCAP)
							break; // (the step's distinct cells are at most its rays: it goes to the next pass whole)
						const u32 cell = part ? s_cell[qnext][lane] : 0u;
						while (todo != 0ull) {
							const int l = (int)__builtin_ctzll(todo);
							const u32 X = (u32)__builtin_amdgcn_readlane((int)cell, l);
							todo &= ~__ballot(part && cell == X);
							jc = d_writelane(jc, X, njobs);
							jq = d_writelane(jq, qnext, njobs);
							njobs++;
						}
					}
					// the headers of the jobs' cells: one lane per job
					u32 jb = 0u, jl = 1u;
					if ((u32)lane < njobs) {
						jb = offset[jc];
						jl = span[jc];
					}
					WK_STAMP(1);
					// 3. the jobs in step order; the triangle ids of job j+2 and the records of job j+1 are in flight while
					//    job j is tested (every lane loads, beyond the end of a list the last triangle again, and past the
					//    last job the last job again: the number of loads in flight does not depend on the data)
#define WK_JOB_ID(J)                                                               \
	value_list[(u32)__builtin_amdgcn_readlane((int)jb, (int)(J)) +                 \
		   min((u32)lane, (u32)__builtin_amdgcn_readlane((int)jl, (int)(J)) - 1u)]
					u32 idA = 0u, fN = 0u;
					float rN[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
					if (njobs > 0u) {
						fN = WK_JOB_ID(0u);
						idA = WK_JOB_ID(min(1u, njobs - 1u));
						d_load_record<REC>(rec, verts, tris, fN, rN);
					}
#pragma unroll 1
					for (u32 j = 0; j < njobs; j++) {
						u32 f0 = fN;
						float r0[9];
#pragma unroll
						for (int k = 0; k < 9; k++)
							r0[k] = rN[k];
						// this job's operands must have arrived here, before the next loads are issued
						asm volatile("" : "+v"(f0), "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "+v"(r0[4]), "+v"(r0[5]),
							     "+v"(r0[6]), "+v"(r0[7]), "+v"(r0[8]), "+v"(idA));
						WK_STAMP(2);
						fN = idA;
						d_load_record<REC>(rec, verts, tris, fN, rN);
						idA = WK_JOB_ID(min(j + 2u, njobs - 1u));
						const u32 X = (u32)__builtin_amdgcn_readlane((int)jc, (int)j), q = (u32)__builtin_amdgcn_readlane((int)jq, (int)j);
						const u32 base = (u32)__builtin_amdgcn_readlane((int)jb, (int)j), S = (u32)__builtin_amdgcn_readlane((int)jl, (int)j);
						bool in = walking && ((nem >> q) & 1u) && s_cell[q][lane] == X;
						float tin = 0.0f;
						if (in) {
							// the ray may have stopped in one of the empty cells since its last occupied one: tnext grows along
							// the walk and best_t has not changed since, so that is the one comparison with this cell's entry
							tin = q != 0u ? s_tnext[q - 1u][lane] : tstart;
							if (best_ref != WK_NONE && best_t <= tin) {
								if (COUNT) { // the step it stopped in: the first whose exit is not before the hit
									u32 qs = 0u;
									while (!((vmask >> qs) & 1u) || !(best_t <= s_tnext[qs][lane]))
										qs++;
									n_cells += (u32)__popc(vmask & ((2u << qs) - 1u));
								}
								hitstop = true;
								walking = false;
								in = false;
							}
						}
						const unsigned long long grpm = __ballot(in);
						if (grpm == 0ull)
							continue; // every ray of the job ended in an earlier step of this window
						const u32 n = (u32)__popcll(grpm);
						wjobs++;
						if (COUNT) {
							st_jobs++;
							st_jrays += n;
							if (in)
								n_tests += S;
						}
						// Cull first?  The bundle box and the cull of a batch cost about as much as three exact rounds, an exact
						// round tests up to 64 (survivor, ray) pairs, and the cull removes about 60 % of a list: it pays from a few
						// hundred (triangle, ray) pairs on (default: 10 per ray of the wave: 320 at 32 rays per wave, 640 at 64).
						const bool use_cull = S >= CULL_MIN && S * n >= CULL_WORK;
						BeamBox bx;
						if (use_cull) { // the bundle box is formed from the rays' own lanes
							const float4 a = *reinterpret_cast<const float4 *>(&s_ray[lane * 8]);
							const float2 b = *reinterpret_cast<const float2 *>(&s_ray[lane * 8 + 4]);
							const float o[3] = { a.x, a.y, a.z }, d[3] = { a.w, b.x, b.y };
							bx = d_beam_box(o, d, tin, in);
						}
						WK_STAMP(3);
						// lanes as (survivor, ray) pairs: pair slot `lane` = survivor lane / n, ray number lane % n of the job
						// (more than 32 rays: one survivor per round)
						const u32 mper = 64u / n;
						if (in)
							s_rank[d_rank_in_mask(grpm)] = (unsigned char)lane;
						__syncthreads();
						const u32 my_si = (u32)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)n));
						const u32 prl = s_rank[(u32)lane - my_si * n];
						float po[3], pd[3];
						{
							const float4 a = *reinterpret_cast<const float4 *>(&s_ray[prl * 8u]);
							const float2 b = *reinterpret_cast<const float2 *>(&s_ray[prl * 8u + 4u]);
							po[0] = a.x, po[1] = a.y, po[2] = a.z, pd[0] = a.w, pd[1] = b.x, pd[2] = b.y;
						}
						for (u32 b = 0; b < S; b += 64u) {
							const bool have = b + (u32)lane < S;
							float r9[9];
#pragma unroll
							for (int k = 0; k < 9; k++)
								r9[k] = r0[k];
							if (b != 0u && have) { // lists beyond 64 triangles: the later batches are fetched here
								const u32 f = value_list[base + b + (u32)lane];
								d_load_record<REC>(rec, verts, tris, f, r9);
							}
							bool keep = have;
							if (use_cull && have)
								keep = !d_cull_beam(&r9[0], &r9[3], &r9[6], bx);
							const unsigned long long m = __ballot(keep);
							WK_STAMP(4);
							if (COUNT && use_cull) {
								st_cb++;
								st_ct += (u32)__popcll(__ballot(have));
							}
							if (m == 0ull)
								continue;
							const u32 ks = (u32)__popcll(m);
							if (keep) { // survivors in list order
								float4 *dst = reinterpret_cast<float4 *>(&s_surv[d_rank_in_mask(m) * TRI_STRIDE]);
								dst[0] = make_float4(r9[0], r9[1], r9[2], r9[3]);
								dst[1] = make_float4(r9[4], r9[5], r9[6], r9[7]);
								dst[2] = make_float4(r9[8], __uint_as_float(base + b + (u32)lane), 0.0f, 0.0f);
							}
							__syncthreads();
							for (u32 r = 0; r < ks; r += mper) {
								const bool act = my_si < mper && r + my_si < ks;
								if (act) {
									const float4 *src = reinterpret_cast<const float4 *>(&s_surv[(r + my_si) * TRI_STRIDE]);
									const float4 a = src[0], e = src[1], h = src[2];
									const float tv[3] = { po[0] - a.x, po[1] - a.y, po[2] - a.z };
									const float e1[3] = { a.w, e.x, e.y }, e2[3] = { e.z, e.w, h.x };
									float t;
									// closest hit per ray: (t, list position) ordered as the sequential loop's strict `<` in list order
									if (d_mt_core(tv, e1, e2, pd, &t) && t > 0.0f)
										atomicMin(&s_best[prl], ((unsigned long long)__float_as_uint(t) << 32) |
														 (unsigned long long)__float_as_uint(h.y));
								}
								if (COUNT) {
									st_rounds++;
									st_pairs += (u32)__popcll(__ballot(act));
								}
							}
							__syncthreads(); // the survivors are read before the next batch overwrites them
							WK_STAMP(5);
						}
						// the job is the step of its rays: merge its closest hit (strict <: an equal hit of an earlier cell
						// stays) and test the stop rule against the cell's exit
						if (in) {
							const unsigned long long k = s_best[lane];
							if (k != ~0ull) {
								s_best[lane] = ~0ull;
								const float jt = __uint_as_float((u32)(k >> 32));
								if (jt < best_t) {
									best_t = jt;
									best_ref = (u32)k;
									best_behind = jt < tin;
								}
							}
							if (best_ref != WK_NONE && best_t <= s_tnext[q][lane]) {
								if (COUNT)
									n_cells += (u32)__popc(vmask & ((2u << q) - 1u));
								hitstop = true;
								walking = false;
							}
						}
						WK_STAMP(6);
					}
#undef WK_JOB_ID
				}
			} else if (COUNT) {
				st_empty++;
			}
			// 4. the end of the window: the stop rule over the empty cells behind the ray's last occupied one, and the
			//    end of the walk
			if (walking) {
				if (best_ref != WK_NONE && best_t <= tcur) {
					if (COUNT) {
						u32 qs = 0u;
						while (!((vmask >> qs) & 1u) || !(best_t <= s_tnext[qs][lane]))
							qs++;
						n_cells += (u32)__popc(vmask & ((2u << qs) - 1u));
					}
					hitstop = true;
					walking = false;
				} else {
					if (COUNT)
						n_cells += (u32)__popc(vmask);
					if (ended) {
						walking = false;
						left = true;
					}
				}
			}
			// what the next launch cuts the long groups by: the jobs of this window
			if (sp.fb && wjobs != 0u && lane == 0)
				sp.fb[(size_t)d_group_key(sp.chunk, grp, RPW) * WK_FBW + (widx < (u32)WK_FBW ? widx : (u32)WK_FBW - 1u)] =
					(unsigned char)(wjobs < 255u ? wjobs : 255u);
			widx++;
			if (COUNT)
				__syncthreads(); // (s_tnext is rewritten by the next window's plan)
		}
		if (nseg == 1u) {
			if (inb) {
				hit_t[p] = hitstop ? best_t : -1.0f;
				hit_id[p] = hitstop ? (int)value_list[best_ref] : -2;
				if (sp.walked) {
					sp.walked[pb] = nwin;
					sp.walked_prev[pb] = 0xFFFFFFFFu; // (this array is the next launch's `walked`: its segments take minima)
				}
			}
			redo = 0ull;
		} else {
			// Split walks.  A segment walks its rays from the start (windows before its own without looking at a cell) and
			// tests the cells of its windows as if nothing had been hit before; the segments of a group run on different
			// waves at the same time, each with the rays that walked as far as its windows in the launch before, and none
			// looks further than a ray went then.  Per ray the sequential walk's result is the smallest (t, segment,
			// position) over the segments' closest hits: a hit found further along the walk with a smaller t lies in an
			// earlier cell, whose list holds its triangle too.  It is valid if its t is not beyond the exit of the last
			// cell that was looked at (the segments' windows join up) or the ray has left the grid.  Otherwise - the ray
			// goes further than the last time - the wave that does the merge takes the ray up from there (`redo`, at the
			// top of the loop), and where the argument rests on the lists alone - the winner was found by a later segment
			// BEHIND the entry of its cell - it walks the ray afresh.  The last segment to finish writes the group's
			// results.
			// (a ray that still walks where the segment ends, and went no further the last time, ends here: no later
			// segment has it)
			if (walking && widx >= wprev)
				walking = false;
			if (inb && (nwin != 0u || seg == 0u)) {
				if (best_ref != WK_NONE)
					atomicMin(&sp.key[pb], ((unsigned long long)__float_as_uint(best_t) << 32) | ((unsigned long long)seg << 29) |
								      ((unsigned long long)((best_behind && seg != 0u) ? 1u : 0u) << 28) | (unsigned long long)best_ref);
				if (left)
					atomicMin(&sp.tend[pb], __float_as_uint(tcur));
				atomicMax(&sp.texam[pb], hitstop ? 0x7F800000u : __float_as_uint(tcur));
				if (!walking) // (the first segment the ray ends in; later ones walk past its hit.  A ray that still walks where
					      // the segment ends goes on in the next one)
					atomicMin(&sp.walked[pb], nwin);
			}
			__threadfence();
			u32 fin = 0u;
			if (lane == 0)
				fin = atomicAdd(&sp.done[grp], 1u);
			fin = (u32)__builtin_amdgcn_readfirstlane((int)fin);
			redo = 0ull;
			if (fin == nseg - 1u) {
				__threadfence();
				bool again = false;
				if (inb) {
					const unsigned long long k = __hip_atomic_load(&sp.key[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const u32 teb = __hip_atomic_load(&sp.tend[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const float tx = __uint_as_float(__hip_atomic_load(&sp.texam[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					const float kt = __uint_as_float((u32)(k >> 32));
					const bool left = teb != 0xFFFFFFFFu;
					const bool hit = k != ~0ull && kt <= (left ? __uint_as_float(teb) : tx);
					// (the merge state as the next launch expects to find it)
					again = (!hit && !left) || (hit && ((k >> 28) & 1ull));
					// (the merge state as the next launch expects to find it; a ray that is gone over again picks its key up first)
					if (!again)
						sp.key[pb] = ~0ull;
					sp.tend[pb] = 0xFFFFFFFFu;
					sp.texam[pb] = 0u;
					sp.walked_prev[pb] = 0xFFFFFFFFu;
					hit_t[p] = hit ? kt : -1.0f;
					hit_id[p] = hit ? (int)value_list[(u32)k & 0x0FFFFFFFu] : -2;
				}
				redo = __ballot(again);
				if (lane == 0)
					sp.done[grp] = 0u;
			}
		}
		if (COUNT) {
			if (inb) {
				if (n_tests)
					atomicAdd(&counters[0], (unsigned long long)n_tests);
				if (n_cells)
					atomicAdd(&counters[1], (unsigned long long)n_cells);
				atomicAdd(&counters[2], 1ull);
			}
			if (lane == 0) { // wave-uniform counts
				atomicAdd(&counters[WS_WINDOWS], (unsigned long long)st_win);
				atomicAdd(&counters[WS_EMPTY_WINDOWS], (unsigned long long)st_empty);
				atomicAdd(&counters[WS_JOBS], (unsigned long long)st_jobs);
				atomicAdd(&counters[WS_JOB_RAYS], (unsigned long long)st_jrays);
				atomicAdd(&counters[WS_CULL_BATCHES], (unsigned long long)st_cb);
				atomicAdd(&counters[WS_CULL_TESTS], (unsigned long long)st_ct);
				atomicAdd(&counters[WS_ROUNDS], (unsigned long long)st_rounds);
				atomicAdd(&counters[WS_ROUND_PAIRS], (unsigned long long)st_pairs);
				const unsigned long long cyc = __builtin_amdgcn_s_memtime() - clk0;
				int bucket = 0;
				while (bucket < 15 && (cyc >> (12 + bucket)) > 1ull)
					bucket++;
				atomicAdd(&counters[WS_HIST + bucket], 1ull);
				atomicAdd(&counters[WS_SUM_CYCLES], cyc);
				for (int k = 0; k < 8; k++)
					atomicAdd(&counters[WS_PHASE + k], ph[k]);
				if (cyc >= (1ull << 19)) {
					for (int k = 0; k < 8; k++)
						atomicAdd(&counters[WS_PHASE_HEAVY + k], ph[k]);
					atomicAdd(&counters[WS_PHASE_HEAVY + 8], 1ull);
					atomicAdd(&counters[WS_PHASE_HEAVY + 9], (unsigned long long)st_rounds);
					atomicAdd(&counters[WS_PHASE_HEAVY + 10], (unsigned long long)st_jobs);
					atomicAdd(&counters[WS_PHASE_HEAVY + 11], (unsigned long long)st_win);
				}
				atomicMax(&counters[WS_MAX_CYCLES], cyc);
			}
		}
		__syncthreads(); // the next group's rays overwrite s_ray
		if (redo != 0ull)
			continue; // (the same group once more, whole, for the rays of `redo`)
		if (lane == 0)
			it = gridDim.x + atomicAdd(ticket, 1u);
		it = (u32)__builtin_amdgcn_readfirstlane((int)it);
	} // work items
}

// The cut groups of the launch that follows (one thread per group): a group whose jobs in the context's LAST bounce (the
// history in sp.fb, kept under spans of pixels: it need not have been the same rays) came to more than LOAD percent of
// the average group's, and to more than one of the launch's WAVES' share of all jobs, is cut into up to WK_MAXSEG
// segments of windows with about equal jobs.  FORCE >= 2 cuts every group into that many segments of three windows
// (tests).  Also clears the history for the launch that follows.  hdr_prev / hdr_next: the header words of the launch
// before / after this one.
#define SEG_THREADS 256
__global__ __launch_bounds__(SEG_THREADS) void k_dda_segments(const u32 *__restrict__ count_p, u32 RPW,
							      WalkSplit sp, uint2 *__restrict__ items, unsigned char *__restrict__ cut,
							      const u32 *__restrict__ hdr_prev, u32 *__restrict__ hdr_next, u32 LOAD, u32 FORCE,
							      u32 maxg, u32 WAVES, u32 MAXSEG)
{
	__shared__ u32 s_red[SEG_THREADS / 64];
	const u32 t = threadIdx.x;
	const u32 count = *count_p;
	u32 nb = (u32)(((unsigned long long)count + RPW - 1u) / RPW);
	nb = nb < maxg ? nb : maxg;
	if (blockIdx.x == 0u && t == 0u) {
		sp.hdr[0] = count;
		sp.hdr[1] = RPW;
		for (int i = 2; i <= 6; i++)
			hdr_next[i] = 0u; // (the counters of the launch after this one)
	}
	// Is the history worth cutting by?  When more than half of the groups the launch before cut had rays that went further
	// than the segments looked (a scene that moves fast), nothing is cut for a while: 1, 2, 4 ... 16 launches.
	// [7] launches still to sit out, [8] the length of the current pause
	bool pause = hdr_prev[7] != 0u;
	{
		u32 left = pause ? hdr_prev[7] - 1u : 0u, len = hdr_prev[8];
		if (!pause) {
			if (hdr_prev[6] != 0u && hdr_prev[4] * 2u > hdr_prev[6]) {
				len = len ? (2u * len < 16u ? 2u * len : 16u) : 1u;
				left = len - 1u;
				pause = true;
			} else if (hdr_prev[6] != 0u) {
				len = 0u; // (a launch that cut, and well)
			}
		}
		if (blockIdx.x == 0u && t == 0u) {
			sp.hdr[7] = left;
			sp.hdr[8] = len;
		}
	}
	// the average group's jobs are the launch's before the last (the sum this kernel forms is ready after it); the
	// history is kept under pixels and spans of pixels, so it serves a list of other rays as far as it goes
	const bool valid = hdr_prev[1] == RPW && hdr_prev[3] != 0u && !pause;
	// ... and more than a wave's share of all jobs: a launch of many more groups than waves has no tail to cut, only
	// the segments' extra windows to pay
	u32 limit = nb ? (u32)(((unsigned long long)hdr_prev[3] * LOAD) / ((unsigned long long)nb * 100ull)) + 1u : 1u;
	const u32 share = hdr_prev[3] / (WAVES ? WAVES : 1u);
	limit = limit > share ? limit : share;
	u32 jobs_all = 0u;
	// (whole waves take part in every turn: the places in the list are dealt per wave)
	for (u32 g0 = blockIdx.x * SEG_THREADS; g0 < nb; g0 += gridDim.x * SEG_THREADS) {
		const u32 g = g0 + t;
		u32 jobs = 0u, m = 0u;
		bool early = false;
		u32 fw[WK_FBW];
		unsigned char *f = nullptr;
		if (g < nb) {
			f = sp.fb + (size_t)d_group_key(sp.chunk, g, RPW) * WK_FBW;
			const uint4 *f4 = reinterpret_cast<const uint4 *>(f);
#pragma unroll
			for (int q = 0; q < WK_FBW / 16; q++) {
				const uint4 x = f4[q];
				const u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
				for (int k = 0; k < 16; k++)
					fw[q * 16 + k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
			}
#pragma unroll
			for (int w = 0; w < WK_FBW; w++)
				jobs += fw[w];
			m = FORCE >= 2u ? FORCE : (valid ? (jobs + limit - 1u) / limit : 1u);
			m = m < 1u ? 1u : (m > MAXSEG ? MAXSEG : m);
			// a group with more than a third of the jobs that would get it cut is listed too, whole: the long groups start
			// first, whatever their place in the list of rays
			early = m == 1u && valid && jobs * 3u > limit;
			cut[g] = (m > 1u || early) ? 1 : 0;
		}
		// places in the list: one atomic per wave (a few hundred groups are cut: one each would be ~6 us on one word)
		{
			const u32 mine = m > 1u ? m : (early ? 1u : 0u);
			u32 incl = mine;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = (u32)__shfl_up((int)incl, d);
				if ((t & 63u) >= (u32)d)
					incl += o;
			}
			const u32 wave_total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
			u32 base = 0u;
			if (wave_total != 0u) {
				if ((t & 63u) == 0u) {
					base = atomicAdd(&sp.hdr[2], wave_total);
					atomicAdd(&sp.hdr[6], (u32)__popcll(__ballot(mine > 1u))); // (groups cut)
				}
				base = (u32)__builtin_amdgcn_readfirstlane((int)base);
			}
			if (early)
				items[base + incl - mine] = make_uint2(g | (1u << 28), 0xFFFFu << 16);
			if (m > 1u) {
				u32 at = base + incl - mine;
				// segment s ends before the first window by which (s + 1) / m of the jobs have been seen
				u32 w = 0u, acc = 0u, first = 0u;
				for (u32 sgm = 0; sgm < m; sgm++) {
					u32 end = 0xFFFFu;
					if (sgm + 1u < m) {
						if (FORCE >= 2u) {
							end = 3u * (sgm + 1u);
						} else {
							const u32 want = (u32)(((unsigned long long)jobs * (sgm + 1u)) / m);
							while (w < (u32)WK_FBW - 1u && acc + fw[w] <= want) {
								acc += fw[w];
								w++;
							}
							end = w > first ? w : first + 1u; // (at least one window)
							w = end < (u32)WK_FBW - 1u ? end : (u32)WK_FBW - 1u;
						}
					}
					items[at++] = make_uint2(g | (sgm << 24) | (m << 28), first | (end << 16));
					first = end;
				}
			}
		}
		if (g < nb) {
			uint4 *z = reinterpret_cast<uint4 *>(f);
#pragma unroll
			for (int q = 0; q < WK_FBW / 16; q++)
				z[q] = make_uint4(0u, 0u, 0u, 0u);
		}
		jobs_all += jobs;
	}
	u32 sum = jobs_all;
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		sum += (u32)__shfl_xor((int)sum, m);
	if ((t & 63u) == 0u)
		s_red[t >> 6] = sum;
	__syncthreads();
	if (t == 0u) {
		u32 all = 0u;
		for (int w = 0; w < SEG_THREADS / 64; w++)
			all += s_red[w];
		if (all)
			atomicAdd(&sp.hdr[3], all);
	}
}

// split walks (option dda_split: 0 off, 1 = by the last launch's job counts, 2..WK_MAXSEG = every group, for tests): the
// state of this launch; the merge key holds a list position in 28 bits, the history is kept per group of >= 16 rays
int ugrt_dda_split_state(ugrt_ctx *ctx, u32 RPW, u32 total_refs, WalkSplit *sp, WalkSplitHost *sph)
{
	const int mode = ctx->opt[UGRT_OPT_DDA_SPLIT] >= 0 ? ctx->opt[UGRT_OPT_DDA_SPLIT] : 1;
	if (mode == 0 || RPW < 16u || (64u % RPW) != 0u || total_refs >= (1u << 28))
		return UGRT_OK;
	// (groups and chunks of the padded list: a whole number of 512-pixel spans, ugrt_dda.hip)
	const size_t npix = (size_t)ctx->npix, slots = (npix + 511) / 512 * 512, maxg = slots / 64 * (64 / RPW) + 64 / RPW, nchunk = slots / 64 + 1;
	// three header blocks in turn (this launch's, the one before, the one after), two `walked` arrays in turn
	const size_t o_items = 3 * 64, o_fb = o_items + maxg * WK_MAXSEG * sizeof(uint2), o_done = o_fb + maxg * WK_FBW + 16,
		     o_cut = o_done + maxg * 4 + 16, o_chunk = (o_cut + maxg + 15) / 16 * 16, o_key = (o_chunk + nchunk * 4 + 15) / 16 * 16,
		     o_tend = o_key + npix * 8, o_texam = o_tend + npix * 4, o_walked = o_texam + npix * 4, bytes = o_walked + 2 * npix * 4;
	const void *before = ctx->dsplit.p;
	int rc = ugrt_buf_reserve(ctx, ctx->dsplit, bytes);
	if (rc)
		return rc;
	if (ctx->dsplit.p != before || ctx->dsplit_rpw != RPW) { // no history yet (or one laid out for other groups)
		UGRT_HIP(hipMemsetAsync(ctx->dsplit.p, 0, ctx->dsplit.cap, ctx->stream));
		// the merge state of a ray at rest: no hit (all ones), not seen leaving (all ones), nothing looked at (0); no ray
		// has a history ("walked everywhere")
		UGRT_HIP(hipMemsetAsync((char *)ctx->dsplit.p + o_key, 0xFF, npix * 12, ctx->stream));
		UGRT_HIP(hipMemsetAsync((char *)ctx->dsplit.p + o_walked, 0xFF, 2 * npix * 4, ctx->stream));
		ctx->dsplit_rpw = RPW;
		ctx->dsplit_turn = 0;
	}
	const u32 turn = ctx->dsplit_turn++;
	char *b = (char *)ctx->dsplit.p;
	sp->hdr = (u32 *)(b + 64 * (turn % 3u));
	sp->items = (const uint2 *)(b + o_items);
	sp->cut = (const unsigned char *)(b + o_cut);
	sp->fb = (unsigned char *)(b + o_fb);
	sp->chunk = (const u32 *)(b + o_chunk);
	sp->done = (u32 *)(b + o_done);
	sp->key = (unsigned long long *)(b + o_key);
	sp->p0 = (u32)ctx->p0;
	sp->tend = (u32 *)(b + o_tend);
	sp->texam = (u32 *)(b + o_texam);
	sp->walked = (u32 *)(b + o_walked) + (size_t)(turn & 1u) * npix;
	sp->walked_prev = (u32 *)(b + o_walked) + (size_t)((turn + 1u) & 1u) * npix;
	sph->items = (uint2 *)(b + o_items);
	sph->cut = (unsigned char *)(b + o_cut);
	sph->hdr_prev = (const u32 *)(b + 64 * ((turn + 2u) % 3u));
	sph->hdr_next = (u32 *)(b + 64 * ((turn + 1u) % 3u));
	sph->load = ctx->opt[UGRT_OPT_DDA_SPLIT_LOAD] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_SPLIT_LOAD] : 400u;
	sph->force = mode >= 2 ? (u32)mode : 0u;
	sph->maxseg = mode >= 2 ? (u32)WK_MAXSEG
			       : (ctx->opt[UGRT_OPT_DDA_SPLIT_SEGMENTS] > 0 ? (u32)ctx->opt[UGRT_OPT_DDA_SPLIT_SEGMENTS] : (u32)WK_MAXSEG);
	sph->maxg = (u32)maxg;
	return UGRT_OK;
}

// launched by ugrt_trace_dda (ugrt_dda.hip) behind k_dda_prepare, which also wrote `bitmap` ((ncell + 63) / 64 * 2 words)
int ugrt_dda_walk_launch(ugrt_ctx *ctx, const DGrid &g, const u32 *d_value_list, const u32 *d_span, const u32 *d_offset,
			 u32 *bitmap, const float *d_vertlist, const int *d_trilist, const float4 *rec, const float *d_rays,
			 const u32 *list, const u32 *dcount, float *d_hit_t, int *d_hit_id, unsigned long long *counters,
			 bool counting, u32 RPW, u32 CULL_MIN, u32 CULL_WORK, int blocks, const WalkSplit &sp, const WalkSplitHost &sph)
{
	u32 *ticket = ctx->d_small + UGRT_DSMALL_TICKET;
	if (sp.items) {
		const u32 sblocks = (sph.maxg + SEG_THREADS - 1) / SEG_THREADS;
		hipLaunchKernelGGL(k_dda_segments, dim3(sblocks < 64u ? sblocks : 64u), dim3(SEG_THREADS), 0, ctx->stream, dcount,
				   RPW, sp, sph.items, sph.cut, sph.hdr_prev, sph.hdr_next, sph.load, sph.force, sph.maxg, (u32)blocks, sph.maxseg);
		UGRT_HIP(hipGetLastError());
	}
	// (a context without split walks -- option dda_split 0, or a launch that cannot keep a history -- runs the lean kernel)
	const bool split = sp.items || sp.fb || sp.walked || sp.cut;
#define WK_LAUNCH(CNTV, RECV)                                                                                                  \
	do {                                                                                                                   \
		if (split)                                                                                                     \
			hipLaunchKernelGGL((k_trace_dda_walk<CNTV, RECV, true>), dim3(blocks), dim3(64), 0, ctx->stream, g, d_value_list, \
					   d_span, d_offset, (const u32 *)bitmap, d_vertlist, d_trilist, rec, d_rays, list, dcount,  \
					   d_hit_t, d_hit_id, counters, RPW, CULL_MIN, CULL_WORK, ticket, sp);                       \
		else                                                                                                           \
			hipLaunchKernelGGL((k_trace_dda_walk<CNTV, RECV, false>), dim3(blocks), dim3(64), 0, ctx->stream, g, d_value_list, \
					   d_span, d_offset, (const u32 *)bitmap, d_vertlist, d_trilist, rec, d_rays, list, dcount,  \
					   d_hit_t, d_hit_id, counters, RPW, CULL_MIN, CULL_WORK, ticket, sp);                       \
	} while (0)
	if (counting) {
		if (rec)
			WK_LAUNCH(true, true);
		else
			WK_LAUNCH(true, false);
	} else {
		if (rec)
			WK_LAUNCH(false, true);
		else
			WK_LAUNCH(false, false);
	}
#undef WK_LAUNCH
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// the optional ray sort: keys of the listed rays (entry cell << 3 | octant); `cap` = entries of the list buffer
int ugrt_dda_sort_keys_launch(ugrt_ctx *ctx, const DGrid &g, const float *d_rays, const u32 *list, const u32 *dcount, u32 cap,
			      u32 *keys)
{
	hipLaunchKernelGGL(k_dda_sort_keys, dim3((cap + 255u) / 256u), dim3(256), 0, ctx->stream, g, d_rays, list, dcount, cap, keys);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// what the split walks of the context's last bounce did (synchronises the stream): [0] segments the cut groups were
// listed as, [1] jobs of the launch before (what the groups were measured against), [2] cut groups that had rays
// walked again in one piece, [3] those rays
extern "C" int ugrt_stats_dda_split(ugrt_ctx *ctx, unsigned out[4])
{
	if (!ctx || !out)
		return ugrt_fail(UGRT_EINVAL, "stats_dda_split: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	out[0] = out[1] = out[2] = out[3] = 0u;
	if (!ctx->dsplit.p || ctx->dsplit_turn == 0u)
		return UGRT_OK;
	u32 h[8];
	UGRT_HIP(hipMemcpyAsync(h, (const char *)ctx->dsplit.p + 64 * ((ctx->dsplit_turn - 1u) % 3u), sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	out[0] = h[2], out[1] = h[3], out[2] = h[4], out[3] = h[5];
	return UGRT_OK;
}
