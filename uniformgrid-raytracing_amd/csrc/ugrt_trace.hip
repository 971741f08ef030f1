// ugrt_trace.hip -- the three tracers: primary (perspective grid), shadow
// (spherical light grid) and reflection (uniform grid, 3D-DDA).
//
// Mapping to CDNA4: the reference's 8x8-thread CUDA block per grid tile
// (trace_kernel.cu:84, light_kernel.cu:52) is exactly one 64-lane wavefront,
// so "block per cell" becomes "wave per work item" with no multi-wave
// barriers.  A work item is (cell, one segment of that cell's triangles): the
// reference gives a whole cell to one block, and border cells that collect
// every clamped off-screen triangle then serialise the frame;
// here long cells are cut into segments that run on different waves and are
// merged with a 64-bit atomicMin on (t bits << 32 | ref index), which keeps the
// reference's tie-break (strict `<`, first ref in the sorted list wins,
// trace_kernel.cu:38).  Triangles are staged through LDS 64 at a time by the
// wave itself (trace_kernel.cu:151-175) and read back as LDS broadcasts.
// Waves are persistent: the launch is sized to the chip and each wave strides
// over the work list, whose length stays on the device.
#include <cstdlib>

#include "ugrt_packet.h"
#include "ugrt_rs_hist.h"
#include "ugrt_scan.h"


struct WItem {
	u32 cell;  // primary: screen cell; shadow: chunk index
	u32 begin; // first ref
	u32 count; // refs in this item (at most one segment)
	u32 multi; // primary: bit 0 = the cell is split across items; bits 1..15 its row, bits 16..31 its column of cells
};

// ---------------------------------------------------------------------------
// work lists
// ---------------------------------------------------------------------------
// primary: one entry per cell of the band, x-major like the cell ids.  The list is written by the scan of its
// counts (ugrt_scan.h): the counts are formed where the scan loads them, the items where it has their positions
struct WlPrimaryLoad {
	const u32 *span;
	u32 nby, gy_lo, rows, SEG;
	__device__ __forceinline__ void operator()(u32 base, u32 n, u32 (&v)[SC_ITEMS]) const
	{
#pragma unroll
		for (int k = 0; k < SC_ITEMS; k++) {
			const u32 i = base + (u32)k;
			u32 c = 0;
			if (i < n) {
				const u32 sp = span[(i / rows) * nby + gy_lo + (i % rows)];
				c = sp ? (sp + SEG - 1) / SEG : 1u;
			}
			v[k] = c;
		}
	}
};
struct WlPrimaryStore {
	static constexpr bool active = true;
	const u32 *span, *offset;
	u32 nby, gy_lo, rows, SEG;
	WItem *items;
	__device__ __forceinline__ void operator()(u32 base, u32 n, const u32 (&v)[SC_ITEMS], const u32 (&incl)[SC_ITEMS]) const
	{
		for (int k = 0; k < SC_ITEMS; k++) {
			const u32 i = base + (u32)k;
			if (i >= n)
				break;
			const u32 cx = i / rows, cy = gy_lo + (i % rows);
			const u32 cell = cx * nby + cy;
			const u32 sp = span[cell], off = offset[cell], cnt = v[k];
			const u32 first = incl[k] - cnt;
			for (u32 s = 0; s < cnt; s++) {
				WItem w;
				w.cell = cell;
				w.begin = off + s * SEG;
				const u32 left = sp - s * SEG;
				w.count = sp ? (left < SEG ? left : SEG) : 0u;
				w.multi = (cnt > 1 ? 1u : 0u) | (cy << 1) | (cx << 16); // (the tracer is spared two integer divisions per item)
				items[first + s] = w;
			}
		}
	}
};

// ---------------------------------------------------------------------------
// primary rays: rckernel_alpha, trace_kernel.cu:84-270 (NUM_SLABS = 1)
// ---------------------------------------------------------------------------
struct PrimaryOut {
	float *normal;
	float *t_value;
	float *ray_dir;
	int *shadowed;
	int *intersect_id;
};

// trace_kernel.cu:56-82 isWithin + :230-267 epilogue for one pixel.
// `ref` = index into value_list of the nearest accepted triangle, ~0u = none.
template <bool REC>
__device__ __forceinline__ void d_finish_pixel(const CamBlock &cam, const PrimaryOut &o, int pixelID,
					       const float *dir, float oldt, u32 ref,
					       const u32 *__restrict__ value_list, const float *__restrict__ verts,
					       const int *__restrict__ tris, const float4 *__restrict__ rec)
{
	bool ok = false;
	if (ref != 0xFFFFFFFFu) {
		float px = cam.cc[0] + oldt * dir[0];
		float py = cam.cc[1] + oldt * dir[1];
		float pz = cam.cc[2] + oldt * dir[2];
		const float *m = cam.cc;
		float hz = D_MULMV_ROW(m, 48, 2, px, py, pz);
		float hw = D_MULMV_ROW(m, 48, 3, px, py, pz);
		hz /= hw;
		ok = ugrt_floor2i(hz * 1.0f) == 0;
	}
	if (ok) {
		u32 face = value_list[ref];
		float tri[9];
		// the record holds the same two edge subtractions (one 48-B gather instead of three indices + nine floats)
		d_load_triangle<REC>(rec, verts, tris, face, 0.0f, 0.0f, 0.0f, tri);
		float *e1 = &tri[3], *e2 = &tri[6], nrm[3];
		D_NORMALIZE(e1);
		D_NORMALIZE(e2);
		D_CROSS(nrm, e1, e2);
		D_NORMALIZE(nrm);
		nrm[0] = nrm[0] < 0 ? nrm[0] * -1 : nrm[0];
		nrm[1] = nrm[1] < 0 ? nrm[1] * -1 : nrm[1];
		nrm[2] = nrm[2] < 0 ? nrm[2] * -1 : nrm[2];
		o.t_value[pixelID] = oldt;
		o.intersect_id[pixelID] = (int)face;
		o.normal[pixelID * 3 + 0] = nrm[0];
		o.normal[pixelID * 3 + 1] = nrm[1];
		o.normal[pixelID * 3 + 2] = nrm[2];
	} else {
		o.t_value[pixelID] = -1.0f;
		o.intersect_id[pixelID] = -2;
		o.normal[pixelID * 3 + 0] = -1.0f;
		o.normal[pixelID * 3 + 1] = -1.0f;
		o.normal[pixelID * 3 + 2] = -1.0f;
	}
	o.shadowed[pixelID] = 0;
	o.ray_dir[pixelID * 3 + 0] = dir[0];
	o.ray_dir[pixelID * 3 + 1] = dir[1];
	o.ray_dir[pixelID * 3 + 2] = dir[2];
}

#define SURV_CAP 128 // survivors buffered in LDS before the per-lane tests run (flush at >= 64)
#define JOB_CAP (4 * SURV_CAP)
#ifndef JOB_CHUNK
#define JOB_CHUNK 64u // jobs between two looks at the rays' closest hits
#endif

// One wave per item = (8x8-pixel tile, segment of its cell list).  lane = triangle: cull against the
// tile's direction box, then against the boxes of its four 4x4-pixel QUADRANTS; survivors go to LDS once
// and every (survivor, quadrant it may touch) pair becomes a JOB.  lane = (job of the round, ray of that
// job's quadrant): a round runs four jobs of ANY quadrants, so a small triangle costs a quarter of the lanes
// and a tile whose triangles crowd into one quadrant still fills the wave (with one list per quadrant the
// rounds of a flush were those of its longest list: 1.53 M rounds for 2.57 M jobs on the bench frame).
// A ray is then tested by different lanes in different rounds, so its closest hit is merged in LDS as
// min over (t bits << 32 | position in the cell list): the smallest t, and of equal t's the first of the
// list -- what the reference's strict "<" in list order keeps.
// qfar[q] = the largest of the closest-hit distances of quadrant q's 16 rays (lane bits 2 and 5 select the
// quadrant), NaN (bits ~0) while one of them has no hit: positive floats and that sentinel order as unsigned ints
__device__ __forceinline__ void d_quadrant_far(const unsigned long long *s_best, int lane, float *qfar)
{
	// (positive floats and the sentinel ~0 order as unsigned ints; the reduction is made on int images: bit 31 flipped)
	const u32 tb = (u32)d_quadrant_reduce<DOpMax>((int)((u32)(s_best[lane] >> 32) ^ 0x80000000u)) ^ 0x80000000u;
#pragma unroll
	for (int q = 0; q < 4; q++)
		qfar[q] = __uint_as_float((u32)__builtin_amdgcn_readlane((int)tb, ((q & 1) << 2) | ((q & 2) << 4)));
}

// work counters of the COUNT variant (ugrt_stats_primary)
enum { PS_ITEMS, PS_BATCHES, PS_BATCHES_KEPT, PS_REFS, PS_TILE_SURVIVORS, PS_SURVIVORS, PS_JOBS, PS_FLUSHES, PS_ROUNDS,
       PS_ROUNDS_DIV, PS_ROUNDS_V, PS_ROUNDS_T, PS_LANE_TESTS, PS_HITS, PS_PRUNABLE, PS_PRUNED, PS_END };
static_assert(PS_END <= UGRT_PRIMARY_STATS, "primary work counters");

template <bool REC, bool COUNT>
__global__ __launch_bounds__(64, 4) void k_trace_primary(CamBlock cam, const float *__restrict__ tex,
						       const WItem *__restrict__ items,
						       const u32 *__restrict__ nitems_p,
						       const u32 *__restrict__ value_list,
						       const float *__restrict__ verts, const int *__restrict__ tris,
						       const float4 *__restrict__ rec, PrimaryOut out,
						       u64 *__restrict__ best, int p0, unsigned long long *__restrict__ counters,
						       u32 ORDER, u32 CHUNK, u32 SLICES)
{
	__shared__ __attribute__((aligned(16))) float lds[SURV_CAP * TRI_STRIDE];
	__shared__ unsigned short jobs[JOB_CAP]; // survivor slot | lane offset of the quadrant << 7
	__shared__ unsigned short jobs2[JOB_CAP]; // the same jobs, nearest triangles first (ORDER)
	__shared__ unsigned short ready[64];     // the jobs of the current 64 that are still worth their tests
	__shared__ float s_dir[64 * 3];
	__shared__ unsigned long long s_best[64];
	const int lane = threadIdx.x;
	// a job's 16 rays: lane bits 0,1 (column) and 3,4 (row) inside the quadrant; the quadrant adds bits 2 and 5
	const int group = lane >> 4, rbase = (lane & 3) | (((lane >> 2) & 3) << 3);
	const u32 nitems = *nitems_p;
	const float ex = cam.cc[0], ey = cam.cc[1], ez = cam.cc[2];
	unsigned long long ps[PS_END] = { 0 };
	// SLICES 1: persistent waves, a contiguous slice of the list per XCD; otherwise one wave per item, runs of
	// 2^((SLICES >> 1) - 1) items per XCD in turn (workgroup b runs on XCD b % 8; a power of two: every wave maps its
	// index, and a division by a launch parameter is ~35 instructions)
	u32 first = blockIdx.x;
	if (SLICES == 1u) {
		first = d_xcd_block();
	} else if (SLICES > 1u) {
		const u32 rl = (SLICES >> 1) - 1u, j = blockIdx.x >> 3;
		u32 g = j >> rl; // round of eight runs
		if (SLICES & 1u) {
			// Centre out: the rounds are taken from the middle of the list outwards (mid, mid + 1, mid - 1, ...).  The list is
			// in screen order, column by column, and the cells that cost most (the long lists a camera looks at) sit around the
			// middle of the view: in list order their items start half-way through the launch and ARE its tail (the longest
			// last four times the mean: profiles/r03_primary_timeline.txt); started first they are over when the cheap ones run out.
			const u32 G = (nitems + (8u << rl) - 1u) >> (rl + 3u);
			if (g >= G)
				return;
			const u32 mid = (G - 1u) >> 1;
			g = (g & 1u) ? mid + ((g + 1u) >> 1) : mid - (g >> 1);
		}
		first = (((g << 3) + (blockIdx.x & 7u)) << rl) + (j & ((1u << rl) - 1u));
	}
	for (u32 it = first; it < nitems; it += gridDim.x) {
		const WItem w = items[it];
		if (COUNT) {
			ps[PS_ITEMS]++;
			ps[PS_REFS] += w.count;
		}
		// Two dependent gathers stand before every batch of 64 references (id, then record); issued where
		// they are needed, a wave spends three quarters of a batch waiting for them.  So they run one batch
		// ahead: the records of the next batch and the ids of the one after are in flight while the current
		// one is culled, and the item's first batch while its rays are set up.  (Indices are clamped to the
		// item's last reference instead of being masked: loads under a divergent branch make the compiler
		// wait for everything outstanding at the join.)
		const u32 last = w.count ? w.count - 1u : 0u;
		u32 id_next = 0;
		float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
		float rcx = 0.f;
		if (REC && w.count)
			id_next = value_list[w.begin + min((u32)lane, last)];
		const int bx = (int)(w.multi >> 16), by = (int)((w.multi >> 1) & 0x7FFFu);
		const int col = bx * 8 + (lane & 7), row = by * 8 + (lane >> 3);
		const int pixelID = row * cam.W + col;
		float dir[3];
		d_ray_dir(cam, tex, col, row, dir);
		if (REC && w.count) {
			ra = rec[id_next * 3u + 0u];
			rb = rec[id_next * 3u + 1u];
			rcx = reinterpret_cast<const float *>(rec)[id_next * 12u + 8u];
			id_next = value_list[w.begin + min(64u + (u32)lane, last)];
		}
		// direction boxes: reduce inside the quadrants (lane bits 0,1,3,4), then across them (bits 2,5);
		// centre and half width are formed per lane and read from one lane of each quadrant, so that
		// the boxes live in scalar registers
		CBox qb[4], tb;
		float qrmax[3]; // the largest of the four quadrants' half widths, per component (d_cull_cr4)
#pragma unroll
		for (int k = 0; k < 3; k++) {
			// (through the order-preserving int images of the floats: integer min / max fold into the DPP instruction)
			const int ilo = d_quadrant_reduce<DOpMin>(d_ordered(dir[k])), ihi = d_quadrant_reduce<DOpMax>(d_ordered(dir[k]));
			float lo = d_unordered(ilo), hi = d_unordered(ihi);
			const float qc = 0.5f * (lo + hi);
			const float qr = 0.5f * (hi - lo) * 1.0001f + 1e-6f; // far more than the rounding of c and r
			qrmax[k] = d_readlane(d_unordered(d_across_quadrants<DOpMax>(d_ordered(qr))), 0);
			lo = d_unordered(d_across_quadrants<DOpMin>(ilo));
			hi = d_unordered(d_across_quadrants<DOpMax>(ihi));
			tb.c[k] = d_readlane(0.5f * (lo + hi), 0);
			tb.r[k] = d_readlane(0.5f * (hi - lo) * 1.0001f + 1e-6f, 0);
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int src = ((q & 1) << 2) | ((q & 2) << 4);
				qb[q].c[k] = d_readlane(qc, src);
				qb[q].r[k] = d_readlane(qr, src);
			}
		}
		s_dir[lane * 3 + 0] = dir[0];
		s_dir[lane * 3 + 1] = dir[1];
		s_dir[lane * 3 + 2] = dir[2];
		s_best[lane] = ~0ull;
		u32 nsurv = 0, njobs = 0;
		// per quadrant: the farthest of its 16 rays' closest hits so far (NaN while a ray has none)
		float qfar[4] = { __uint_as_float(~0u), __uint_as_float(~0u), __uint_as_float(~0u), __uint_as_float(~0u) };
		for (u32 b = 0; b < w.count || nsurv; b += 64) {
			if (b < w.count) {
				const u32 cnt = (w.count - b) < 64u ? (w.count - b) : 64u;
				bool keep = false;
				float t9[9];
				CullTri ct;
				if (REC) {
					// (the same subtractions d_load_triangle makes)
					t9[0] = ex - ra.x, t9[1] = ey - ra.y, t9[2] = ez - ra.z;
					t9[3] = ra.w, t9[4] = rb.x, t9[5] = rb.y, t9[6] = rb.z, t9[7] = rb.w, t9[8] = rcx;
					ra = rec[id_next * 3u + 0u];
					rb = rec[id_next * 3u + 1u];
					rcx = reinterpret_cast<const float *>(rec)[id_next * 12u + 8u];
					id_next = value_list[w.begin + min(b + 128u + (u32)lane, last)];
					if ((u32)lane < cnt) {
						ct = d_cull_prep(&t9[0], &t9[3], &t9[6]);
						keep = !d_cull_cr(ct, tb);
					}
				} else if ((u32)lane < cnt) {
					d_load_triangle<REC>(rec, verts, tris, value_list[w.begin + b + lane], ex, ey, ez, t9);
					ct = d_cull_prep(&t9[0], &t9[3], &t9[6]);
					keep = !d_cull_cr(ct, tb);
				}
				if (COUNT) {
					ps[PS_BATCHES]++;
					ps[PS_TILE_SURVIVORS] += (u32)__popcll(__ballot(keep));
				}
				// The survivors of the tile cull are staged as they are; their quadrant culls, depth bounds and jobs are
				// made at the flush, lane = staged triangle: there the wave is full, here a batch has ~24 of them on 64 lanes
				// (the ~300 instructions of that part ran 110 k times for the bench frame, they now run 45 k times).
				const unsigned long long mask = __ballot(keep);
				if (mask != 0ull) {
					if (COUNT)
						ps[PS_BATCHES_KEPT]++;
					const u32 slot = nsurv + d_rank_in_mask(mask);
					if (keep) {
						float4 *dst = reinterpret_cast<float4 *>(&lds[slot * TRI_STRIDE]);
						dst[0] = make_float4(t9[0], t9[1], t9[2], t9[3]);
						dst[1] = make_float4(t9[4], t9[5], t9[6], t9[7]);
						dst[2] = make_float4(t9[8], __uint_as_float(w.begin + b + lane), 0.0f, 0.0f);
					}
					nsurv += (u32)__popcll(mask);
				}
				if (nsurv < 64u && b + 64 < w.count)
					continue; // keep collecting
			}
			// The jobs are taken 64 at a time.  lane = job: a job whose triangle lies behind the closest hits of all
			// 16 rays of its quadrant is dropped (the rays of the bench scene cross eleven surfaces each; lists are
			// in id order, so most triangles come after a nearer one).  lane = (job, ray): the exact per-ray test
			// of the reference, four of the remaining jobs a round.
			__syncthreads();
			// quadrant culls, depth bounds, jobs: lane = staged triangle
			for (u32 s0 = 0; s0 < nsurv; s0 += 64u) {
				const u32 slot = s0 + (u32)lane;
				u32 km = 0;
				if (slot < nsurv) {
					const float4 *src = reinterpret_cast<const float4 *>(&lds[slot * TRI_STRIDE]);
					const float4 a = src[0], c = src[1];
					const float e2x = lds[slot * TRI_STRIDE + 8u];
					const float tv[3] = { a.x, a.y, a.z }, e1[3] = { a.w, c.x, c.y }, e2[3] = { c.z, c.w, e2x };
					const CullTri ct = d_cull_prep(tv, e1, e2);
					const float tlow = d_cull_tlow(ct, e2, tb);
					const u32 out4 = d_cull_cr4(ct, qb, qrmax);
#pragma unroll
					for (int q = 0; q < 4; q++)
						km |= (((out4 >> q) & 1u) || tlow > qfar[q]) ? 0u : (1u << q);
					lds[slot * TRI_STRIDE + 10u] = tlow;
				}
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const unsigned long long mq = __ballot((km >> q) & 1u);
					const u32 qoff = (u32)(((q & 1) << 2) | ((q & 2) << 4));
					if ((km >> q) & 1u)
						jobs[njobs + d_rank_in_mask(mq)] = (unsigned short)(slot | (qoff << 7));
					njobs += (u32)__popcll(mq);
				}
			}
			__syncthreads();
			if (COUNT) {
				ps[PS_FLUSHES]++;
				ps[PS_SURVIVORS] += nsurv;
				ps[PS_JOBS] += njobs;
			}
			// Front to back.  The cell lists are in id order, so a flush's jobs meet their rays in no particular depth
			// order: 3.3 M of the bench frame's 4.2 M jobs lie behind the hits their quadrant ends the flush with, and
			// the depth bound dropped 0.9 M of them.  The jobs are therefore put in the order of their triangles'
			// lower bounds t_low before they run (a counting sort into 8 buckets of the bounds' float images, by
			// ballots: ~3 exact rounds' worth of instructions per flush), and the rays' closest hits are looked at every
			// CHUNK jobs.  The order only decides what is tested: the closest hits are merged by minimum.
			const unsigned short *jl = jobs;
			if (ORDER && njobs > CHUNK) {
				int kmin = 0x7FFFFFFF, kmax = 0;
				for (u32 j0 = 0; j0 < njobs; j0 += 64u)
					if (j0 + (u32)lane < njobs) {
						const int key = __float_as_int(lds[(jobs[j0 + (u32)lane] & 127u) * TRI_STRIDE + 10u]); // t_low >= 0
						kmin = key < kmin ? key : kmin;
						kmax = key > kmax ? key : kmax;
					}
				kmin = d_wave_imin(kmin);
				kmax = d_wave_imax(kmax);
				const u32 range = (u32)(kmax - kmin);
				const u32 bits = range ? 32u - (u32)__builtin_clz(range) : 0u;
				const u32 sh = bits > 3u ? bits - 3u : 0u; // (key - kmin) >> sh < 8
				u32 base[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
				for (u32 j0 = 0; j0 < njobs; j0 += 64u) {
					const bool valid = j0 + (u32)lane < njobs;
					const u32 bk = valid ? ((u32)(__float_as_int(lds[(jobs[j0 + (u32)lane] & 127u) * TRI_STRIDE + 10u]) - kmin) >> sh) : 8u;
#pragma unroll
					for (u32 k = 0; k < 7u; k++)
						base[k + 1] += (u32)__popcll(__ballot(bk == k));
				}
#pragma unroll
				for (u32 k = 1; k < 8u; k++)
					base[k] += base[k - 1];
				for (u32 j0 = 0; j0 < njobs; j0 += 64u) {
					const bool valid = j0 + (u32)lane < njobs;
					const u32 job = valid ? jobs[j0 + (u32)lane] : 0u;
					const u32 bk = valid ? ((u32)(__float_as_int(lds[(job & 127u) * TRI_STRIDE + 10u]) - kmin) >> sh) : 8u;
					u32 pos = 0;
#pragma unroll
					for (u32 k = 0; k < 8u; k++) {
						const unsigned long long m = __ballot(bk == k);
						if (bk == k)
							pos = base[k] + d_rank_in_mask(m);
						base[k] += (u32)__popcll(m);
					}
					if (valid)
						jobs2[pos] = (unsigned short)job;
				}
				__syncthreads();
				jl = jobs2;
			}
			for (u32 j0 = 0; j0 < njobs; j0 += CHUNK) {
				d_quadrant_far(s_best, lane, qfar);
				bool live = false;
				u32 myjob = 0;
				if ((u32)lane < CHUNK && j0 + (u32)lane < njobs) {
					myjob = jl[j0 + (u32)lane];
					const float tlow = lds[(myjob & 127u) * TRI_STRIDE + 10u];
					const u32 qo = myjob >> 7;
					const float far = qo == 0u ? qfar[0] : (qo == 4u ? qfar[1] : (qo == 32u ? qfar[2] : qfar[3]));
					live = !(tlow > far);
				}
				const unsigned long long lm = __ballot(live);
				const u32 nready = (u32)__popcll(lm);
				if (COUNT)
					ps[PS_PRUNED] += (u32)__popcll(__ballot((u32)lane < CHUNK && j0 + (u32)lane < njobs)) - nready;
				if (live)
					ready[d_rank_in_mask(lm)] = (unsigned short)myjob;
				__syncthreads();
				for (u32 r0 = 0; r0 < nready; r0 += 4u) {
					int stage = -1;
					if (r0 + (u32)group < nready) {
						const u32 job = ready[r0 + (u32)group];
						const int home = rbase | (int)(job >> 7);
						const float4 *src = reinterpret_cast<const float4 *>(&lds[(job & 127u) * TRI_STRIDE]);
						const float4 a = src[0], c = src[1], e = src[2];
						const float tv[3] = { a.x, a.y, a.z }, e1[3] = { a.w, c.x, c.y }, e2[3] = { c.z, c.w, e.x };
						const float rd[3] = { s_dir[home * 3 + 0], s_dir[home * 3 + 1], s_dir[home * 3 + 2] };
						const float v = d_intersect_tri_uv(tv, e1, e2, rd, 99999999.9f);
						if (v != 0.0f)
							atomicMin(&s_best[home], ((unsigned long long)__float_as_uint(v) << 32) |
											 (unsigned long long)__float_as_uint(e.y));
						if (COUNT)
							stage = v != 0.0f ? 4 : d_mt_stage(tv, e1, e2, rd);
					}
					if (COUNT) {
						ps[PS_ROUNDS]++;
						ps[PS_ROUNDS_DIV] += __ballot(stage >= 1) != 0ull;
						ps[PS_ROUNDS_V] += __ballot(stage >= 2) != 0ull;
						ps[PS_ROUNDS_T] += __ballot(stage >= 3) != 0ull;
						ps[PS_LANE_TESTS] += (u32)__popcll(__ballot(stage >= 0));
						ps[PS_HITS] += (u32)__popcll(__ballot(stage == 4));
					}
				}
				__syncthreads();
			}
			d_quadrant_far(s_best, lane, qfar); // for the culls of the batches to come
			if (COUNT) {
				// what a front-to-back order of the flush's jobs could have dropped: jobs whose triangle lies behind the
				// closest hits their quadrant ends up with (PS_PRUNED: the jobs the depth bound did drop in list order)
				for (u32 j0 = 0; j0 < njobs; j0 += 64u) {
					bool behind = false;
					if (j0 + (u32)lane < njobs) {
						const u32 job = jobs[j0 + (u32)lane];
						const float tlow = lds[(job & 127u) * TRI_STRIDE + 10u];
						const u32 qo = job >> 7;
						const float far = qo == 0u ? qfar[0] : (qo == 4u ? qfar[1] : (qo == 32u ? qfar[2] : qfar[3]));
						behind = tlow > far;
					}
					ps[PS_PRUNABLE] += (u32)__popcll(__ballot(behind));
				}
			}
			__syncthreads();
			nsurv = 0;
			njobs = 0;
		}
		const unsigned long long mine = s_best[lane];
		if (!(w.multi & 1u)) {
			const bool hit = mine != ~0ull;
			d_finish_pixel<REC>(cam, out, pixelID, dir, hit ? __uint_as_float((u32)(mine >> 32)) : 99999999.9f,
					    hit ? (u32)mine : 0xFFFFFFFFu, value_list, verts, tris, rec);
		} else if (mine != ~0ull) {
			atomicMin(reinterpret_cast<unsigned long long *>(&best[pixelID - p0]), mine);
		}
	}
	if (COUNT && lane == 0)
		for (int i = 0; i < PS_END; i++)
			if (ps[i])
				atomicAdd(&counters[i], ps[i]);
}

template <bool REC>
__global__ __launch_bounds__(64) void k_trace_primary_slabs(CamBlock cam, const float *__restrict__ tex, int slabs,
							     int gy_lo, int rows, u32 ntiles,
							     const u32 *__restrict__ span, const u32 *__restrict__ offset,
							     const u32 *__restrict__ value_list,
							     const float *__restrict__ verts, const int *__restrict__ tris,
							     const float4 *__restrict__ rec, PrimaryOut out)
{
	__shared__ __attribute__((aligned(16))) float lds[64 * TRI_STRIDE];
	const int lane = threadIdx.x;
	const float ex = cam.cc[0], ey = cam.cc[1], ez = cam.cc[2];
	for (u32 it = d_xcd_block(); it < ntiles; it += gridDim.x) {
		const int bx = (int)(it / (u32)rows), by = gy_lo + (int)(it % (u32)rows);
		const u32 cell = (u32)bx * (u32)cam.nby + (u32)by;
		const int col = bx * 8 + (lane & 7), row = by * 8 + (lane >> 3);
		const int pixelID = row * cam.W + col;
		float dir[3];
		d_ray_dir(cam, tex, col, row, dir);
		CBox tb;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const float lo = d_wave_fmin(dir[k]), hi = d_wave_fmax(dir[k]);
			tb.c[k] = 0.5f * (lo + hi);
			tb.r[k] = 0.5f * (hi - lo) * 1.0001f + 1e-6f;
		}
		float oldt = 99999999.9f;
		u32 ref = 0xFFFFFFFFu;
		int rayDone = 0;
		for (int slab = 0; slab < slabs; slab++) {
			const u32 sp = span[cell * (u32)slabs + (u32)slab], off = offset[cell * (u32)slabs + (u32)slab];
			for (u32 b = 0; b < sp; b += 64u) {
				const u32 cnt = (sp - b) < 64u ? (sp - b) : 64u;
				bool keep = false;
				float t9[9];
				if ((u32)lane < cnt) {
					d_load_triangle<REC>(rec, verts, tris, value_list[off + b + lane], ex, ey, ez, t9);
					const CullTri ct = d_cull_prep(&t9[0], &t9[3], &t9[6]);
					keep = !d_cull_cr(ct, tb);
				}
				const unsigned long long mask = __ballot(keep);
				const u32 nsurv = (u32)__popcll(mask);
				__syncthreads();
				if (keep) {
					float4 *dst = reinterpret_cast<float4 *>(&lds[d_rank_in_mask(mask) * TRI_STRIDE]);
					dst[0] = make_float4(t9[0], t9[1], t9[2], t9[3]);
					dst[1] = make_float4(t9[4], t9[5], t9[6], t9[7]);
					dst[2] = make_float4(t9[8], __uint_as_float(off + b + lane), 0.0f, 0.0f);
				}
				__syncthreads();
				if (rayDone != 2) {
					for (u32 k = 0; k < nsurv; k++) {
						const float4 *src = reinterpret_cast<const float4 *>(&lds[k * TRI_STRIDE]);
						const float4 a = src[0], c = src[1], e = src[2];
						const float tv[3] = { a.x, a.y, a.z }, e1[3] = { a.w, c.x, c.y }, e2[3] = { c.z, c.w, e.x };
						const float v = d_intersect_tri_uv(tv, e1, e2, dir, oldt);
						if (v != 0.0f) {
							oldt = v;
							rayDone = 1;
							ref = __float_as_uint(e.y);
						}
					}
				}
			}
			// isWithin, trace_kernel.cu:56-82
			if (rayDone == 0 || rayDone == 2) {
				rayDone = 0;
			} else {
				const float px = ex + oldt * dir[0], py = ey + oldt * dir[1], pz = ez + oldt * dir[2];
				const float *m = cam.cc;
				float hz = D_MULMV_ROW(m, 48, 2, px, py, pz);
				const float hw = D_MULMV_ROW(m, 48, 3, px, py, pz);
				hz /= hw;
				rayDone = ugrt_floor2i(hz * (float)slabs) == slab ? 2 : 1;
			}
			if (__ballot(rayDone != 2) == 0ull)
				break; // the beam is done, trace_kernel.cu:217-228
		}
		if (rayDone == 2) {
			const u32 face = value_list[ref];
			float tri[9];
			d_load_triangle<REC>(rec, verts, tris, face, 0.0f, 0.0f, 0.0f, tri);
			float *e1 = &tri[3], *e2 = &tri[6], nrm[3];
			D_NORMALIZE(e1);
			D_NORMALIZE(e2);
			D_CROSS(nrm, e1, e2);
			D_NORMALIZE(nrm);
			nrm[0] = nrm[0] < 0 ? nrm[0] * -1 : nrm[0];
			nrm[1] = nrm[1] < 0 ? nrm[1] * -1 : nrm[1];
			nrm[2] = nrm[2] < 0 ? nrm[2] * -1 : nrm[2];
			out.t_value[pixelID] = oldt;
			out.intersect_id[pixelID] = (int)face;
			out.normal[pixelID * 3 + 0] = nrm[0];
			out.normal[pixelID * 3 + 1] = nrm[1];
			out.normal[pixelID * 3 + 2] = nrm[2];
		} else {
			out.t_value[pixelID] = -1.0f;
			out.intersect_id[pixelID] = -2;
			out.normal[pixelID * 3 + 0] = -1.0f;
			out.normal[pixelID * 3 + 1] = -1.0f;
			out.normal[pixelID * 3 + 2] = -1.0f;
		}
		out.shadowed[pixelID] = 0;
		out.ray_dir[pixelID * 3 + 0] = dir[0];
		out.ray_dir[pixelID * 3 + 1] = dir[1];
		out.ray_dir[pixelID * 3 + 2] = dir[2];
	}
}

// pixels of split cells: take the merged (t, ref), finish, re-arm the slot
template <bool REC>
__global__ __launch_bounds__(256) void k_resolve_primary(CamBlock cam, const float *__restrict__ tex,
							  const u32 *__restrict__ span,
							  const u32 *__restrict__ value_list,
							  const float *__restrict__ verts, const int *__restrict__ tris,
							  const float4 *__restrict__ rec, PrimaryOut out,
							  u64 *__restrict__ best, int p0, int npix, u32 SEG)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= npix)
		return;
	int pixelID = p0 + i;
	int col = pixelID % cam.W, row = pixelID / cam.W;
	u32 cell = (u32)(col >> 3) * (u32)cam.nby + (u32)(row >> 3);
	if (span[cell] <= SEG)
		return;
	u64 b = best[i];
	best[i] = ~0ull;
	float dir[3];
	d_ray_dir(cam, tex, col, row, dir);
	u32 ref = (u32)(b & 0xFFFFFFFFull);
	float oldt = (b == ~0ull) ? 99999999.9f : __uint_as_float((u32)(b >> 32));
	if (b == ~0ull)
		ref = 0xFFFFFFFFu;
	d_finish_pixel<REC>(cam, out, pixelID, dir, oldt, ref, value_list, verts, tris, rec);
}

// total refs behind a span/offset pair: known for the context's own grids,
// read back (one 8-byte copy) for arrays that came from elsewhere
static int refs_of(ugrt_ctx *ctx, const u32 *d_span, const u32 *d_offset, u32 C, u32 *R)
{
	for (int g = 0; g < 3; g++)
		if (ctx->grid[g].valid && d_span == (const u32 *)ctx->grid[g].span.p &&
		    d_offset == (const u32 *)ctx->grid[g].offset.p) {
			*R = ctx->grid[g].R;
			return UGRT_OK;
		}
	UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_REFS, d_span + (C - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
	UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_REFS + 1, d_offset + (C - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	*R = ctx->h_pinned[UGRT_PIN_REFS] + ctx->h_pinned[UGRT_PIN_REFS + 1];
	return UGRT_OK;
}

// FrustumTracer::trace, frustum_tracer.h:35-58
extern "C" int ugrt_trace_primary(ugrt_ctx *ctx, const unsigned *d_value_list, const unsigned *d_span,
				  const unsigned *d_offset, float *d_normal, float *d_t_value, float *d_ray_dir,
				  int *d_shadowed, int *d_intersect_id, const float *d_vertlist, const int *d_trilist)
{
	if (!ctx || !d_value_list || !d_span || !d_offset || !d_normal || !d_t_value || !d_ray_dir || !d_shadowed ||
	    !d_intersect_id || !d_vertlist || !d_trilist)
		return ugrt_fail(UGRT_EINVAL, "trace_primary: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	const int rows = ctx->cfg.row_end - ctx->cfg.row_begin;
	const u32 ncell = (u32)ctx->nbx * (u32)rows;
	const u32 C = (u32)ctx->nbx * (u32)ctx->nby;
	const float *tex = ugrt_ctx_tex(ctx); // (stores the current camera's direction table first, if it is new)
	if (ctx->cfg.slabs > 1) { // NUM_SLABS > 1: the slab walk of trace_kernel.cu:132-229, one wave per tile
		PrimaryOut o = { d_normal, d_t_value, d_ray_dir, d_shadowed, d_intersect_id };
		const bool rec_ok = ctx->rec_valid && ctx->rec_verts == d_vertlist && ctx->rec_tris == d_trilist;
		ugrt_prof_begin(ctx, UGRT_ST_TRACE_PRIMARY);
		if (rec_ok)
			hipLaunchKernelGGL(k_trace_primary_slabs<true>, dim3(launch_blocks_for(ncell)), dim3(64), 0, st, ctx->cam,
					   tex, ctx->cfg.slabs, ctx->cfg.row_begin, rows, ncell, d_span,
					   d_offset, d_value_list, d_vertlist, d_trilist, (const float4 *)ctx->trirec.p, o);
		else
			hipLaunchKernelGGL(k_trace_primary_slabs<false>, dim3(launch_blocks_for(ncell)), dim3(64), 0, st, ctx->cam,
					   tex, ctx->cfg.slabs, ctx->cfg.row_begin, rows, ncell, d_span,
					   d_offset, d_value_list, d_vertlist, d_trilist, (const float4 *)nullptr, o);
		ugrt_prof_end(ctx, UGRT_ST_TRACE_PRIMARY);
		UGRT_HIP(hipGetLastError());
		return UGRT_OK;
	}
	u32 R = 0;
	int rc = refs_of(ctx, d_span, d_offset, C, &R);
	if (rc)
		return rc;
	// triangles per work item: long cells are cut into segments that run on different waves and are
	// merged with atomicMin + a resolve pass; cells up to SEG triangles finish inside their one wave.
	// Every cell of a closed scene carries the few hundred eye-plane-straddling triangles (Q9), so the
	// cut-off sits above that baseline.
	u32 SEG = ctx->opt[UGRT_OPT_PRIMARY_SEG] > 0 ? (u32)ctx->opt[UGRT_OPT_PRIMARY_SEG] : 1024u;
	SEG = SEG < 64u ? 64u : (SEG + 63u) / 64u * 64u;
	const size_t cap = (size_t)ncell + R / SEG + 1;
	if ((rc = ugrt_buf_reserve(ctx, ctx->wscan, (size_t)ncell * 4)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->witems, cap * sizeof(WItem))))
		return rc;
	u32 *incl = (u32 *)ctx->wscan.p;
	WItem *items = (WItem *)ctx->witems.p;
	ugrt_prof_begin(ctx, UGRT_ST_WORKLIST);
	{
		const WlPrimaryLoad load = { d_span, (u32)ctx->nby, (u32)ctx->cfg.row_begin, (u32)rows, SEG };
		const WlPrimaryStore store = { d_span, d_offset, (u32)ctx->nby, (u32)ctx->cfg.row_begin, (u32)rows, SEG, items };
		if ((rc = ugrt_scan_launch<true>(ctx, load, incl, ncell, ScanTailNone(), store)))
			return rc;
	}
	ugrt_prof_end(ctx, UGRT_ST_WORKLIST);
	UGRT_HIP(hipGetLastError());
	PrimaryOut out = { d_normal, d_t_value, d_ray_dir, d_shadowed, d_intersect_id };
	// One wave per work item, in list order: the dispatcher then evens out items of unequal cost by itself.  (Round 2
	// gave 16384 persistent waves two items each, in contiguous slices of the list per XCD for the sake of its L2: 0.305
	// ms alone on the 1 M-triangle frame, with a tail of long second items and of the XCD that holds the heavy screen
	// region - profiles/r03_primary_timeline.txt.  One item per wave in those slices 0.349, two per wave dealt item by
	// item over the XCDs 0.326, one per wave so dealt 0.254; profiles/r03_primary_waves.txt.)  "primary_waves" restores the
	// persistent form with that many waves.
	const bool p_slices = ctx->opt[UGRT_OPT_PRIMARY_WAVES] > 0;
	// (what an XCD gets are runs of `p_run` neighbouring items - neighbours share triangles and the XCD's L2 -, run r
	// going to XCD r % 8; 0 = item i to XCD i % 8.  128: the 1 M-triangle frame is indifferent up to 256 (0.254-0.258
	// ms, 0.278 at 1024), the 79 k-triangle hall, whose items cost the same, likes them long (0.126 at 0-32, 0.119 at
	// 128, 0.115 at 2048, 0.113 in the slices))
	u32 p_run = ctx->opt[UGRT_OPT_PRIMARY_XCD_RUN] >= 0 ? (u32)ctx->opt[UGRT_OPT_PRIMARY_XCD_RUN] : 128u;
	u32 p_run_log2 = 0;
	while (p_run >> (p_run_log2 + 1u))
		p_run_log2++;
	p_run = p_run ? 1u << p_run_log2 : 0u; // (a power of two: rounded down)
	size_t one_each = cap;
	if (p_run)
		one_each = (cap + 8u * p_run - 1) / (8u * p_run) * (8u * p_run); // (a whole number of rounds of runs: the mapping is a permutation)
	const int pwaves = p_slices ? launch_blocks_for((u32)cap, ctx->opt[UGRT_OPT_PRIMARY_WAVES]) : (int)(one_each < 0x7FFFFFFFu ? one_each : 0x7FFFFFFFu);
	ugrt_prof_begin(ctx, UGRT_ST_TRACE_PRIMARY);
	const bool use_rec = ctx->rec_valid && ctx->rec_verts == d_vertlist && ctx->rec_tris == d_trilist;
	const bool counting = (ctx->cfg.flags & UGRT_FLAG_COUNT_WORK) != 0;
	unsigned long long *pc = (unsigned long long *)(ctx->d_small + UGRT_DSMALL_PRIMARY);
	// launch shape of the flushes (no effect on results): jobs nearest first, closest hits looked at every p_chunk jobs
	const u32 p_order = ctx->opt[UGRT_OPT_PRIMARY_ORDER] == 0 ? 0u : 1u;
	u32 p_chunk = ctx->opt[UGRT_OPT_PRIMARY_CHUNK] > 0 ? (u32)ctx->opt[UGRT_OPT_PRIMARY_CHUNK] : (p_order ? 32u : 64u);
	p_chunk = p_chunk > 64u ? 64u : (p_chunk < 4u ? 4u : p_chunk);
#define LAUNCH_PRIMARY(REC_, COUNT_)                                                                                  \
	hipLaunchKernelGGL((k_trace_primary<REC_, COUNT_>), dim3(pwaves), dim3(64), 0, st, ctx->cam, tex,              \
			   (const WItem *)items, (const u32 *)(incl + (ncell - 1)), d_value_list, d_vertlist, d_trilist, \
			   (const float4 *)(REC_ ? ctx->trirec.p : nullptr), out, (u64 *)ctx->best.p, ctx->p0, pc, p_order, p_chunk, \
			   p_slices ? 1u : (p_run ? ((p_run_log2 + 1u) << 1) | (ctx->opt[UGRT_OPT_PRIMARY_CENTRE] != 0 ? 1u : 0u) : 0u))
	if (counting) {
		UGRT_HIP(hipMemsetAsync(pc, 0, UGRT_PRIMARY_STATS * 8, st));
		if (use_rec)
			LAUNCH_PRIMARY(true, true);
		else
			LAUNCH_PRIMARY(false, true);
		UGRT_HIP(hipMemcpyAsync(ctx->primary_stats, pc, UGRT_PRIMARY_STATS * 8, hipMemcpyDeviceToHost, st));
	} else if (use_rec) {
		LAUNCH_PRIMARY(true, false);
	} else {
		LAUNCH_PRIMARY(false, false);
	}
#undef LAUNCH_PRIMARY
	ugrt_prof_end(ctx, UGRT_ST_TRACE_PRIMARY);
	UGRT_HIP(hipGetLastError());
	ugrt_prof_begin(ctx, UGRT_ST_WORKLIST);
	if (use_rec)
		hipLaunchKernelGGL(k_resolve_primary<true>, dim3((ctx->npix + 255) / 256), dim3(256), 0, st, ctx->cam,
				   tex, d_span, d_value_list, d_vertlist, d_trilist,
				   (const float4 *)ctx->trirec.p, out, (u64 *)ctx->best.p, ctx->p0, ctx->npix, SEG);
	else
		hipLaunchKernelGGL(k_resolve_primary<false>, dim3((ctx->npix + 255) / 256), dim3(256), 0, st, ctx->cam,
				   tex, d_span, d_value_list, d_vertlist, d_trilist,
				   (const float4 *)nullptr, out, (u64 *)ctx->best.p, ctx->p0, ctx->npix, SEG);
	ugrt_prof_end(ctx, UGRT_ST_WORKLIST);
	UGRT_HIP(hipGetLastError());
	ctx->stats[0] = cap; // upper bound of primary work items
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// shadow rays: mod_light_rckernel, light_kernel.cu:52-270 (cam = LIGHT camera)
// ---------------------------------------------------------------------------
// Which sorted rays the reference traces: chunks [0, traced) = sorted rays [0, M) with
// M = prefix[traced] (or n when every chunk is traced).  Inside that set the grouping of rays is
// free, and so is the order in which a ray meets its cell's triangles: a ray's flag is 1 iff ANY
// triangle of its light cell passes the occlusion test (light_kernel.cu:186-203).  The reference
// gives a block 64 consecutive rays in PIXEL order (spread over the whole cell) and re-stages the
// cell's whole list for every such chunk.  Here, privately to this tracer (d_map / prefix / the
// light grid's arrays are not modified):
//   1. the traced rays of a cell are re-grouped by a Morton code of their direction from the
//      light: a group of 64 rays is a narrow beam, summarised by its direction box (32 B);
//   2. CULL pass, lane = triangle: a wave keeps 64 triangles of a cell in registers (with the
//      ray-independent halves of the interval test) and streams the cell's beam boxes past them;
//      the (beam, triangle) pairs that cannot be ruled out are appended to a list.  Triangles are
//      read once per cell instead of once per chunk;
//   3. the pair list is sorted by beam (rocPRIM), and
//   4. EXACT pass, lane = ray: each beam runs the reference's per-ray test on its own short list.
// The cull is conservative with margins far above fp32 rounding, so the flags do not change.
#ifndef GCHUNK
#define GCHUNK 32u  // beams a cull work item streams past its 64 triangles
#endif

struct GBox { // one beam (re-grouped rays of one light cell)
	float cx, cy, cz; // centre of the direction box
	float rx, ry, rz; // half widths (slightly widened)
	u32 ray_start;    // into the re-grouped ray list
	u32 ray_count;
};

__device__ __forceinline__ u32 d_spread10(u32 v)
{
	v &= 0x3FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}

__device__ __forceinline__ u32 d_dir_morton(const float *unit)
{
	u32 q[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		float f = (unit[k] * 0.5f + 0.5f) * 1023.0f;
		f = f > 0.0f ? f : 0.0f; // also drops NaN
		q[k] = f < 1023.0f ? (u32)f : 1023u;
	}
	return d_spread10(q[0]) | (d_spread10(q[1]) << 1) | (d_spread10(q[2]) << 2);
}

// bits 0..15 of v spread to the even bits
__device__ __forceinline__ u32 d_spread16(u32 v)
{
	v &= 0xFFFFu;
	v = (v | (v << 8)) & 0x00FF00FFu;
	v = (v | (v << 4)) & 0x0F0F0F0Fu;
	v = (v | (v << 2)) & 0x33333333u;
	v = (v | (v << 1)) & 0x55555555u;
	return v;
}

// Directions are points of a sphere: an octahedral map sends them to the unit square, whose Z curve
// needs two thirds of the bits of the cube's for the same angular resolution.  nu + nv code bits.
__device__ __forceinline__ u32 d_dir_oct(const float *unit, u32 nu, u32 nv)
{
	const float inv = 1.0f / (fabsf(unit[0]) + fabsf(unit[1]) + fabsf(unit[2]) + 1e-30f);
	float x = unit[0] * inv, y = unit[1] * inv;
	if (unit[2] < 0.0f) {
		const float ox = (1.0f - fabsf(y)) * (x >= 0.0f ? 1.0f : -1.0f);
		const float oy = (1.0f - fabsf(x)) * (y >= 0.0f ? 1.0f : -1.0f);
		x = ox;
		y = oy;
	}
	float fu = (x * 0.5f + 0.5f) * (float)(1u << nu), fv = (y * 0.5f + 0.5f) * (float)(1u << nv);
	fu = fu > 0.0f ? fu : 0.0f; // also drops NaN
	fv = fv > 0.0f ? fv : 0.0f;
	const u32 qu = fu < (float)((1u << nu) - 1u) ? (u32)fu : (1u << nu) - 1u;
	const u32 qv = fv < (float)((1u << nv) - 1u) ? (u32)fv : (1u << nv) - 1u;
	// nu >= nv >= nu - 1: u takes the even bits, so its extra bit is the top bit nu + nv - 1
	return d_spread16(qu) | (d_spread16(qv) << 1);
}

// KEY64: (cell << 30) | 30-bit Z code of the direction in the cube, in a 64-bit key; otherwise
// (cell << mbits) | mbits-bit octahedral code, in a 32-bit key (two radix passes fewer, half the key bytes)
template <bool KEY64>
__global__ __launch_bounds__(WL_THREADS) void k_shadow_keys(CamBlock cam, const float *__restrict__ t_value_list,
							     const float *__restrict__ ray_direction_list,
							     const u32 *__restrict__ d_map, const u32 *__restrict__ prefix,
							     u32 nchunks, u32 traced, u32 n, u32 C,
							     const u32 *__restrict__ span, const float *__restrict__ cmPt,
							     u32 mbits, void *__restrict__ keys, u32 *__restrict__ vals,
							     u32 *__restrict__ zero, u32 nzero,
							     const u32 *__restrict__ nchunks_dev, u32 launch_cap, u32 prefix_cap,
							     u32 *__restrict__ zero2, u32 nzero2, u32 *__restrict__ zero3, u32 nzero3, RsFirst hs)
{
	// (grid-stride: a bounded number of workgroups, which also count the first digit of the sort of these keys --
	// ugrt_rs_hist.h; 64-bit keys go to the library's sort and are not counted)
	__shared__ u32 s_rsh[RS_BINS * RS_PRIV];
	for (u32 z = blockIdx.x * WL_THREADS + threadIdx.x; z < nzero; z += gridDim.x * WL_THREADS)
		zero[z] = 0; // run starts/ends per light cell, written after the sort
	// (the pass's work counters + pair cursor, and the cull pass's output cursors: cleared here instead of by two fills)
	for (u32 z = blockIdx.x * WL_THREADS + threadIdx.x; z < nzero2; z += gridDim.x * WL_THREADS)
		zero2[z] = 0;
	for (u32 z = blockIdx.x * WL_THREADS + threadIdx.x; z < nzero3; z += gridDim.x * WL_THREADS)
		zero3[z] = 0;
	d_rs_zero(s_rsh, hs);
	__syncthreads();
	for (u32 i0 = blockIdx.x * WL_THREADS; i0 < n; i0 += gridDim.x * WL_THREADS) {
	const u32 i = i0 + threadIdx.x;
	u32 key32 = 0;
	if (i < n) {
	if (nchunks_dev) { // the chunk count never went to the host (UGRT_CHUNKS_ON_DEVICE): same rule, here
		nchunks = *nchunks_dev;
		// more chunks than the caller's prefix map holds: the host path refuses that (ugrt_sort_rays_chunks
		// reports it); here nothing past the written entries is read and nothing is traced
		if (nchunks > prefix_cap)
			nchunks = 0;
		const u32 lim = nchunks < launch_cap ? nchunks : launch_cap;
		traced = launch_cap == 0xFFFFFFFFu ? nchunks : (lim ? lim - 1u : 0u);
	}
	const u32 M = traced < nchunks ? prefix[traced] : n;
	const u32 pixel = d_map[i];
	u32 cell = d_map[n + i];
	u32 code = 0;
	if (i >= M) {
		cell = C + 1; // not traced by the reference's launch
	} else if (cell >= C || span[cell] == 0) {
		cell = C; // sentinel cell or empty list: nothing can shadow this ray
	} else {
		float tVal = t_value_list[pixel];
		float rd[3];
		rd[0] = (cmPt[0] + tVal * ray_direction_list[pixel * 3 + 0]) - cam.cc[0];
		rd[1] = (cmPt[1] + tVal * ray_direction_list[pixel * 3 + 1]) - cam.cc[1];
		rd[2] = (cmPt[2] + tVal * ray_direction_list[pixel * 3 + 2]) - cam.cc[2];
		D_NORMALIZE(rd);
		code = KEY64 ? d_dir_morton(rd) : d_dir_oct(rd, (mbits + 1u) / 2u, mbits / 2u);
	}
	if (KEY64)
		((u64 *)keys)[i] = ((u64)cell << 30) | (u64)code;
	else
		((u32 *)keys)[i] = key32 = (cell << mbits) | code;
	vals[i] = pixel;
	} // i < n
	if (!KEY64 && hs.hist)
		d_rs_count(s_rsh, key32 & 0xFFu, i < n);
	} // grid-stride
	__syncthreads();
	d_rs_flush(s_rsh, hs);
}

template <typename K>
__global__ __launch_bounds__(WL_THREADS) void k_shadow_runs(const K *__restrict__ keys, u32 n, u32 shift,
							     u32 *__restrict__ rstart, u32 *__restrict__ rend)
{
	u32 i = blockIdx.x * WL_THREADS + threadIdx.x;
	if (i >= n)
		return;
	u32 c = (u32)(keys[i] >> shift);
	if (i == 0 || (u32)(keys[i - 1] >> shift) != c)
		rstart[c] = i;
	if (i == n - 1 || (u32)(keys[i + 1] >> shift) != c)
		rend[c] = i + 1;
}

// per light cell: number of beams (ITEMS = false), or number of cull items = triangle batches x beam chunks -- formed
// where the scans of these counts load them (ugrt_scan.h)
struct ShadowCountLoad {
	const u32 *span, *rstart, *rend;
	u32 beam;
	unsigned long long *tests;
	bool ITEMS;
	__device__ __forceinline__ void operator()(u32 base, u32 C, u32 (&v)[SC_ITEMS]) const
	{
#pragma unroll
		for (int k = 0; k < SC_ITEMS; k++) {
			const u32 c = base + (u32)k;
			u32 x = 0;
			if (c < C) {
				const u32 g = (rend[c] - rstart[c] + beam - 1u) / beam, sp = span[c];
				x = ITEMS ? (sp + 63u) / 64u * ((g + GCHUNK - 1) / GCHUNK) : g;
				if (!ITEMS && g && sp)
					atomicAdd(tests, (unsigned long long)sp * (unsigned long long)g); // cells that matter are few
			}
			v[k] = x;
		}
	}
};

// smallest c with incl[c] > x (incl = inclusive scan over C cells, x < incl[C-1])
__device__ __forceinline__ u32 d_find_cell(const u32 *__restrict__ incl, u32 C, u32 x)
{
	u32 lo = 0, hi = C - 1;
	while (lo < hi) {
		u32 mid = (lo + hi) >> 1;
		if (incl[mid] > x)
			hi = mid;
		else
			lo = mid + 1;
	}
	return lo;
}

// The same for a wave-uniform x with all 64 lanes probing at once: 64-ary instead of binary, three dependent
// loads for 16 k cells instead of fourteen.
__device__ __forceinline__ u32 d_find_cell_wave(const u32 *__restrict__ incl, u32 C, u32 x, int lane)
{
	u32 lo = 0, n = C; // the answer lies in [lo, lo + n) and incl[lo + n - 1] > x
	while (n > 1u) {
		const u32 stride = (n + 63u) / 64u;
		u32 idx = lo + ((u32)lane + 1u) * stride - 1u;
		idx = idx < lo + n - 1u ? idx : lo + n - 1u;
		const unsigned long long above = __ballot(incl[idx] > x);
		const u32 f = (u32)__builtin_ctzll(above);
		const u32 nlo = lo + f * stride;
		const u32 left = lo + n - nlo;
		lo = nlo;
		n = stride < left ? stride : left;
	}
	return lo;
}

// the rays of a beam, as the reference rebuilds them (light_kernel.cu:166-184)
struct ShadowRay {
	float rd[3];
	float distance_b;
	int pixel;
};

__device__ __forceinline__ ShadowRay d_shadow_ray(const CamBlock &cam, const float *__restrict__ t_value_list,
						  const float *__restrict__ ray_direction_list, const float *cm, int pixel)
{
	ShadowRay r;
	const float lx = cam.cc[0], ly = cam.cc[1], lz = cam.cc[2];
	float tVal = t_value_list[pixel];
	float pI[3];
	pI[0] = cm[0] + tVal * ray_direction_list[pixel * 3 + 0];
	pI[1] = cm[1] + tVal * ray_direction_list[pixel * 3 + 1];
	pI[2] = cm[2] + tVal * ray_direction_list[pixel * 3 + 2];
	r.rd[0] = pI[0] - lx;
	r.rd[1] = pI[1] - ly;
	r.rd[2] = pI[2] - lz;
	// isSmaller's distance_b (light_kernel.cu:6) depends on the ray only
	r.distance_b = __builtin_sqrtf((pI[0] - lx) * (pI[0] - lx) + (pI[1] - ly) * (pI[1] - ly) +
				       (pI[2] - lz) * (pI[2] - lz));
	D_NORMALIZE(r.rd);
	r.pixel = pixel;
	return r;
}

// one workgroup of four waves per beam (`beam` re-grouped rays of one light cell): direction box of its rays.  The
// waves take the beam's 64-ray runs in turn and keep per-lane minima and maxima; the lanes are folded once at the end
// (one wave and a wave reduction per run took 49 us beside other frames once the beams were 2048 rays long).
#define BOX_WAVES 4
struct CullItem;
// the cull pass's item table, written by the same launch (defined with the items below)
__device__ void d_cull_items_fill(const u32 *__restrict__ iincl, const u32 *__restrict__ gincl, const u32 *__restrict__ span,
				  const u32 *__restrict__ offset, u32 C, CullItem *__restrict__ items, u32 first, u32 stride);
__global__ __launch_bounds__(64 * BOX_WAVES) void k_shadow_boxes(CamBlock cam, const u32 *__restrict__ gincl, u32 C,
						     const u32 *__restrict__ rstart, const u32 *__restrict__ rend,
						     const u32 *__restrict__ ray_pixels, const float *__restrict__ t_value_list,
						     const float *__restrict__ ray_direction_list,
						     const float *__restrict__ cmPt, GBox *__restrict__ boxes, u32 beam,
						     u32 *__restrict__ zero, u32 nzero, float4 *__restrict__ sray,
						     const u32 *__restrict__ iincl, const u32 *__restrict__ span,
						     const u32 *__restrict__ offset, CullItem *__restrict__ citems)
{
	__shared__ float s_box[BOX_WAVES][6];
	for (u32 z = blockIdx.x * (64u * BOX_WAVES) + threadIdx.x; z < nzero; z += gridDim.x * (64u * BOX_WAVES))
		zero[z] = 0; // candidate run starts/ends per beam, written after the pair sort
	// (the cull pass's items depend on the same two scans as the boxes: listed here instead of by a launch of their own)
	d_cull_items_fill(iincl, gincl, span, offset, C, citems, blockIdx.x * (64u * BOX_WAVES) + threadIdx.x, gridDim.x * (64u * BOX_WAVES));
	const u32 total = gincl[C - 1];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const float cm[3] = { cmPt[0], cmPt[1], cmPt[2] };
	const float inf = __builtin_huge_valf();
	for (u32 g = blockIdx.x; g < total; g += gridDim.x) {
		const u32 c = d_find_cell(gincl, C, g);
		const u32 ngrp = (rend[c] - rstart[c] + beam - 1u) / beam;
		const u32 j = g - (gincl[c] - ngrp);
		const u32 start = rstart[c] + beam * j;
		const u32 left = rend[c] - start;
		const u32 cnt = left < beam ? left : beam;
		float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
		for (u32 b = 64u * (u32)wave; b < cnt; b += 64u * BOX_WAVES) {
			if (b + (u32)lane < cnt) {
				ShadowRay r =
					d_shadow_ray(cam, t_value_list, ray_direction_list, cm, (int)ray_pixels[start + b + lane]);
				// the rebuilt ray, in beam order: the exact pass reads 64 of them as one 1-KB run instead of
				// gathering t and direction per pixel for every (segment, sub-group) item again
				sray[start + b + lane] = make_float4(r.rd[0], r.rd[1], r.rd[2], r.distance_b);
#pragma unroll
				for (int k = 0; k < 3; k++) {
					lo[k] = fminf(lo[k], r.rd[k]);
					hi[k] = fmaxf(hi[k], r.rd[k]);
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const float l = d_wave_fmin(lo[k]), h = d_wave_fmax(hi[k]);
			if (lane == 0) {
				s_box[wave][k] = l;
				s_box[wave][3 + k] = h;
			}
		}
		__syncthreads();
		if (threadIdx.x == 0) {
#pragma unroll
			for (int k = 0; k < 3; k++) {
				lo[k] = s_box[0][k];
				hi[k] = s_box[0][3 + k];
				for (int w = 1; w < BOX_WAVES; w++) {
					lo[k] = fminf(lo[k], s_box[w][k]);
					hi[k] = fmaxf(hi[k], s_box[w][3 + k]);
				}
			}
			GBox o;
			o.cx = 0.5f * (lo[0] + hi[0]);
			o.cy = 0.5f * (lo[1] + hi[1]);
			o.cz = 0.5f * (lo[2] + hi[2]);
			// half widths, widened by far more than the rounding of centre and width
			o.rx = 0.5f * (hi[0] - lo[0]) + 1e-6f;
			o.ry = 0.5f * (hi[1] - lo[1]) + 1e-6f;
			o.rz = 0.5f * (hi[2] - lo[2]) + 1e-6f;
			o.ray_start = start;
			o.ray_count = cnt;
			boxes[g] = o;
		}
		__syncthreads();
	}
}

#define PAIR_BUF 512u

__device__ __forceinline__ void d_flush_pairs(const u32 *buf_beam, const u32 *buf_tri, u32 nbuf, int lane,
					      u32 *__restrict__ pair_count, u32 pair_cap, u32 *__restrict__ pair_beam,
					      u32 *__restrict__ pair_tri)
{
	__syncthreads(); // single-wave block: orders the LDS writes before the reads below
	u32 base = 0;
	if (lane == 0)
		base = atomicAdd(pair_count, nbuf);
	base = __shfl(base, 0);
	for (u32 i = (u32)lane; i < nbuf; i += 64u)
		if (base + i < pair_cap) {
			pair_beam[base + i] = buf_beam[i];
			pair_tri[base + i] = buf_tri[i];
		}
	__syncthreads();
}

// The cull pass appends its pairs through PAIR_SEGS cursors instead of one: every wave ends with an append, and 10 k
// atomics on ONE word take 12 ns each -- 0.12 ms, the whole kernel (per-wave cycle stamps: a third of a wave's time
// went by in the appends).  Wave w appends to segment w % PAIR_SEGS of the staging arrays (cursors 256 B apart);
// k_pair_compact then moves the segments together and leaves the total where the single cursor used to be.  A segment
// that ran over reports a total that no buffer of this size could hold, so the caller's overflow handling applies.
#define PAIR_SEGS 64u
#define PAIR_SEG_STRIDE 64u // words between two cursors

// asynchronous shadow pass: the counts of the cull pass against the capacities the later launches were sized for
// (pg == nullptr: the waiting form, which reads the counts back instead)
struct PairCheck {
	const u32 *gcount;
	u32 cap, gbound;
	u32 *pg, *status, *report;
};

__global__ __launch_bounds__(256) void k_pair_compact(const u32 *__restrict__ segcnt, u32 segcap,
						       const u32 *__restrict__ sbeam, const u32 *__restrict__ stri,
						       u32 *__restrict__ pair_beam, u32 *__restrict__ pair_tri,
						       u32 *__restrict__ pair_count, PairCheck chk, RsFirst hs)
{
	__shared__ u32 s_cnt[PAIR_SEGS], s_base[PAIR_SEGS];
	__shared__ u32 s_over;
	__shared__ u32 s_rsh[RS_BINS * RS_PRIV]; // first digit of the pair sort that follows (ugrt_rs_hist.h)
	if (threadIdx.x == 0)
		s_over = 0u;
	d_rs_zero(s_rsh, hs);
	__syncthreads();
	if (threadIdx.x < PAIR_SEGS) {
		const u32 c = segcnt[threadIdx.x * PAIR_SEG_STRIDE];
		if (c > segcap)
			atomicMax(&s_over, c);
		s_cnt[threadIdx.x] = c < segcap ? c : segcap;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u32 acc = 0;
		for (u32 k = 0; k < PAIR_SEGS; k++) {
			s_base[k] = acc;
			acc += s_cnt[k];
		}
		if (blockIdx.x == 0) {
			const unsigned long long worst = (unsigned long long)s_over * PAIR_SEGS;
			u32 P = s_over ? (worst > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (u32)worst) : acc;
			pair_count[0] = P;
			if (chk.pg) {
				const u32 G = *chk.gcount;
				chk.report[0] = P;
				chk.report[1] = G;
				if (P > chk.cap || G > chk.gbound) {
					atomicOr(chk.status, UGRT_STATUS_PAIR_OVERFLOW);
					P = 0; // nothing is traced: the frame is reported as incomplete
				}
				chk.pg[0] = P;
				chk.pg[1] = G;
			}
		}
	}
	__syncthreads();
	const u32 seg = blockIdx.x % PAIR_SEGS, part = blockIdx.x / PAIR_SEGS, parts = gridDim.x / PAIR_SEGS;
	const u32 n = s_cnt[seg], base = s_base[seg];
	const size_t src = (size_t)seg * segcap;
	for (u32 i0 = part * 256u; i0 < n; i0 += parts * 256u) {
		const u32 i = i0 + threadIdx.x;
		u32 key = 0;
		if (i < n) {
			key = sbeam[src + i];
			pair_beam[base + i] = key;
			pair_tri[base + i] = stri[src + i];
		}
		if (hs.hist)
			d_rs_count(s_rsh, key & 0xFFu, i < n);
	}
	__syncthreads();
	d_rs_flush(s_rsh, hs);
}

// A cull item = (light cell, batch of 64 of its triangles, chunk of GCHUNK of its beams), as the kernel needs it:
// where the batch's ids start, how many there are, the first beam box and the number of boxes.
struct CullItem {
	u32 ref_base, cnt, gfirst, gcount;
};

__device__ __forceinline__ CullItem d_cull_item(const u32 *__restrict__ iincl, const u32 *__restrict__ gincl,
						 const u32 *__restrict__ span, const u32 *__restrict__ offset, u32 C, u32 it)
{
	const u32 c = d_find_cell(iincl, C, it);
	const u32 sp = span[c];
	const u32 nb = (sp + 63u) / 64u;
	const u32 ngrp = gincl[c] - (c ? gincl[c - 1] : 0u);
	const u32 gbase = gincl[c] - ngrp;
	const u32 nq = (ngrp + GCHUNK - 1) / GCHUNK;
	const u32 local = it - (iincl[c] - nb * nq);
	const u32 j = local / nq, q = local % nq;
	const u32 first = 64u * j, g0 = q * GCHUNK;
	CullItem d;
	d.ref_base = offset[c] + first;
	d.cnt = (sp - first) < 64u ? (sp - first) : 64u;
	d.gfirst = gbase + g0;
	d.gcount = ((g0 + GCHUNK) < ngrp ? (g0 + GCHUNK) : ngrp) - g0;
	return d;
}

// The items as a table (the first CULL_TABLE of them; the bench frame has 27 k): the kernel reads an item with one
// scalar load instead of a search and four dependent loads at the head of every item
#define CULL_TABLE (1u << 20)
__device__ void d_cull_items_fill(const u32 *__restrict__ iincl, const u32 *__restrict__ gincl, const u32 *__restrict__ span,
				  const u32 *__restrict__ offset, u32 C, CullItem *__restrict__ items, u32 first, u32 stride)
{
	const u32 total = iincl[C - 1] < CULL_TABLE ? iincl[C - 1] : CULL_TABLE;
	for (u32 it = first; it < total; it += stride)
		items[it] = d_cull_item(iincl, gincl, span, offset, C, it);
}

// CULL pass, lane = triangle.  Same test as d_cull with the box as centre +- half width: f(d) = n.d ranges over
// n.c -+ sum_k |n_k| r_k.  A wave takes its items in a fixed stride, so it knows the ones to come: an item's
// descriptor is requested three items ahead, the ids of its triangles two, their records one -- the head of an item
// was a chain of five dependent loads (25 k cycles, half of the kernel, for 15 beam iterations on average:
// per-wave cycle stamps).
template <bool REC>
__global__ __launch_bounds__(64) void k_shadow_cull(CamBlock cam, const u32 *__restrict__ iincl, const u32 *__restrict__ gincl,
						    u32 C, const u32 *__restrict__ span, const u32 *__restrict__ offset,
						    const u32 *__restrict__ value_list, const float4 *__restrict__ rec,
						    const float *__restrict__ verts, const int *__restrict__ tris,
						    const GBox *__restrict__ boxes, u32 *__restrict__ pair_count, u32 pair_cap,
						    u32 *__restrict__ pair_beam, u32 *__restrict__ pair_tri, u32 sbits,
						    const CullItem *__restrict__ table)
{
#pragma clang fp contract(fast) // cull arithmetic only (conservative by margin); no exact test in this kernel
	// candidate pairs are staged in LDS and flushed PAIR_BUF at a time: one atomic on the shared
	// output cursor per ~450 pairs instead of one per beam iteration
	__shared__ u32 buf_beam[PAIR_BUF], buf_tri[PAIR_BUF];
	u32 nbuf = 0; // wave-uniform
	{ // this wave's segment of the staging arrays (pair_cap = capacity of ONE segment)
		const u32 seg = blockIdx.x % PAIR_SEGS;
		pair_count += seg * PAIR_SEG_STRIDE;
		pair_beam += (size_t)seg * pair_cap;
		pair_tri += (size_t)seg * pair_cap;
	}
	const u32 total = iincl[C - 1];
	const int lane = threadIdx.x;
	const float lx = cam.cc[0], ly = cam.cc[1], lz = cam.cc[2];
	const u32 stride = gridDim.x;
	u32 it = d_xcd_block();
	// An item costs between 1 and GCHUNK beam iterations and the cost changes slowly along the list (cell by
	// cell), so neighbouring waves -- the eight of a SIMD -- would all hold cheap or all hold expensive items,
	// and the SIMDs with the expensive ones set the kernel's time (slowest wave 1.85 x the mean).  The waves'
	// first items are therefore dealt out through a multiplicative permutation of the wave index.
	if ((stride & (stride - 1u)) == 0u)
		it = (it * 40503u) & (stride - 1u);
	if (it >= total)
		return;
	// (items past the end read as empty; their loads go to the first id and its record)
	auto item_at = [&](u32 i) -> CullItem {
		CullItem d = { 0u, 0u, 0u, 0u };
		if (i < total)
			d = i < CULL_TABLE ? table[i] : d_cull_item(iincl, gincl, span, offset, C, i);
		return d;
	};
	auto id_of = [&](const CullItem &d) -> u32 {
		const u32 l = (u32)lane < d.cnt ? (u32)lane : (d.cnt ? d.cnt - 1u : 0u);
		return value_list[d.ref_base + l];
	};
	CullItem d_cur = item_at(it), d_1 = item_at(it + stride), d_2 = item_at(it + 2u * stride);
	u32 id_cur = id_of(d_cur), id_1 = id_of(d_1);
	float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
	float rcx = 0.f;
	if (REC) {
		ra = rec[id_cur * 3u + 0u];
		rb = rec[id_cur * 3u + 1u];
		rcx = reinterpret_cast<const float *>(rec)[id_cur * 12u + 8u];
	}
	for (;;) {
		const bool have = (u32)lane < d_cur.cnt;
		const u32 face = id_cur;
		float t9[9];
		if (REC) {
			t9[0] = lx - ra.x, t9[1] = ly - ra.y, t9[2] = lz - ra.z;
			t9[3] = ra.w, t9[4] = rb.x, t9[5] = rb.y, t9[6] = rb.z, t9[7] = rb.w, t9[8] = rcx;
			// the loads of the items to come
			ra = rec[id_1 * 3u + 0u];
			rb = rec[id_1 * 3u + 1u];
			rcx = reinterpret_cast<const float *>(rec)[id_1 * 12u + 8u];
		} else {
			d_load_triangle<REC>(rec, verts, tris, face, lx, ly, lz, t9);
		}
		id_cur = id_1;
		id_1 = id_of(d_2);
		const CullItem d_3 = item_at(it + 3u * stride);
		float nA[3], nB[3], nD[3], nC[3];
		float mA, mB, mD;
		u32 code;
		{
			const float *tv = &t9[0], *e1 = &t9[3], *e2 = &t9[6];
			D_CROSS(nA, e2, tv);
			D_CROSS(nB, tv, e1);
			D_CROSS(nD, e2, e1);
#pragma unroll
			for (int k = 0; k < 3; k++)
				nC[k] = nA[k] + nB[k] - nD[k];
			const float a = fmaxf(fmaxf(fabsf(tv[0]), fabsf(tv[1])), fabsf(tv[2]));
			const float b = fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2]));
			const float cc = fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]));
			const float K = 6.0f / 65536.0f;
			mA = fmaxf(K * a * cc, 1e-25f);
			mB = fmaxf(K * a * b, 1e-25f);
			mD = fmaxf(K * b * cc, 1e-25f);
			// apparent size from the light, large first: the exact pass stops a ray at its first occluder
			const float sa = __builtin_sqrtf(D_DOT(nD, nD)) / (D_DOT(tv, tv) + 1e-30f);
			float lc = (__log2f(sa + 1e-30f) + 30.0f) * 5.0f;
			lc = lc < 0.0f ? 0.0f : (lc > 255.0f ? 255.0f : lc);
			code = (255u - (u32)lc) >> (8u - sbits);
		}
		const float mC = mA + mB + mD;
		const u32 g0 = d_cur.gfirst, g1 = d_cur.gfirst + d_cur.gcount;
		GBox nxt = boxes[g0]; // wave-uniform address: scalar loads
		for (u32 g = g0; g < g1; g++) {
			const GBox bx = nxt;
			// the next beam's box is requested before this one is used: its latency hides behind the test
			nxt = boxes[(g + 1 < g1) ? g + 1 : g];
			bool keep = false;
			if (have) {
				const float Dm = nD[0] * bx.cx + nD[1] * bx.cy + nD[2] * bx.cz;
				const float Dr = fabsf(nD[0]) * bx.rx + fabsf(nD[1]) * bx.ry + fabsf(nD[2]) * bx.rz;
				const float Am = nA[0] * bx.cx + nA[1] * bx.cy + nA[2] * bx.cz;
				const float Ar = fabsf(nA[0]) * bx.rx + fabsf(nA[1]) * bx.ry + fabsf(nA[2]) * bx.rz;
				const float Bm = nB[0] * bx.cx + nB[1] * bx.cy + nB[2] * bx.cz;
				const float Br = fabsf(nB[0]) * bx.rx + fabsf(nB[1]) * bx.ry + fabsf(nB[2]) * bx.rz;
				const float Cm = nC[0] * bx.cx + nC[1] * bx.cy + nC[2] * bx.cz;
				const float Cr = fabsf(nC[0]) * bx.rx + fabsf(nC[1]) * bx.ry + fabsf(nC[2]) * bx.rz;
				keep = !d_cull_decide(Dm, Dr, Am, Ar, Bm, Br, Cm, Cr, mA, mB, mD, mC);
			}
			const unsigned long long mask = __ballot(keep);
			if (mask != 0ull) {
				if (keep) {
					const u32 pos = nbuf + d_rank_in_mask(mask);
					buf_beam[pos] = (g << sbits) | code;
					buf_tri[pos] = face;
				}
				nbuf += (u32)__popcll(mask);
				if (nbuf > PAIR_BUF - 64u) {
					d_flush_pairs(buf_beam, buf_tri, nbuf, lane, pair_count, pair_cap, pair_beam, pair_tri);
					nbuf = 0;
				}
			}
		}
		it += stride;
		if (it >= total)
			break;
		d_cur = d_1;
		d_1 = d_2;
		d_2 = d_3;
	}
	if (nbuf)
		d_flush_pairs(buf_beam, buf_tri, nbuf, lane, pair_count, pair_cap, pair_beam, pair_tri);
}

// (pg: {candidate pairs, beams} on the device when the host does not know them - the asynchronous form; the launch
// is then sized by an estimate and strides)
__global__ __launch_bounds__(WL_THREADS) void k_pair_runs(const u32 *__restrict__ beam, u32 P, u32 sbits,
							   u32 *__restrict__ pstart, u32 *__restrict__ pend,
							   const u32 *__restrict__ pg)
{
	if (pg)
		P = pg[0];
	for (u32 i = blockIdx.x * WL_THREADS + threadIdx.x; i < P; i += gridDim.x * WL_THREADS) {
		u32 b = beam[i] >> sbits;
		if (i == 0 || (beam[i - 1] >> sbits) != b)
			pstart[b] = i;
		if (i == P - 1 || (beam[i + 1] >> sbits) != b)
			pend[b] = i + 1;
	}
}


// an item with segment number XSEG_LAST takes all the remaining candidates of its beam (the segment is
// the 8-bit sort key of the item list; XSEG_LAST + 1 marks the padding behind the last item)
#define XSEG_LAST 254u
// items per beam, formed where the scan of the counts loads them (ugrt_scan.h; it runs over the capacity Gcap)
struct PairItemLoad {
	const u32 *pstart, *pend;
	const GBox *boxes;
	u32 G;
	unsigned long long *staged;
	u32 XSEG;
	const u32 *pg;
	__device__ __forceinline__ void operator()(u32 base, u32 Gcap, u32 (&v)[SC_ITEMS]) const
	{
		const u32 Gn = pg ? (pg[0] ? pg[1] : 0u) : G; // (no pairs: no items)
		unsigned long long mine = 0;
#pragma unroll
		for (int k = 0; k < SC_ITEMS; k++) {
			const u32 g = base + (u32)k;
			u32 x = 0;
			if (g < Gn && g < Gcap) {
				const u32 cand = pend[g] - pstart[g], nsub = (boxes[g].ray_count + 63u) / 64u;
				const u32 nseg = (cand + XSEG - 1) / XSEG;
				x = (nseg < XSEG_LAST + 1u ? nseg : XSEG_LAST + 1u) * nsub;
				mine += (unsigned long long)cand * nsub;
			}
			v[k] = x;
		}
		// candidates staged by the exact pass (work accounting): one atomic per wave
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1)
			mine += __shfl_xor(mine, m);
		if ((threadIdx.x & 63) == 0 && mine)
			atomicAdd(staged, mine);
	}
};

// The exact-pass items, listed once (the tracer then starts with two loads instead of a 12-step search)
// and ordered by SEGMENT first: all beams' first segments run before any second segment, so by the time a
// later segment of a beam is picked up its rays have mostly been flagged by the earlier ones and the item
// ends at its first ballot.  key = segment, value = beam << 7 | sub-group.
__global__ __launch_bounds__(WL_THREADS) void k_pair_items(const u32 *__restrict__ xincl, u32 G, u32 cap,
							    const u32 *__restrict__ pstart, const u32 *__restrict__ pend,
							    const GBox *__restrict__ boxes, u32 XSEG,
							    u32 *__restrict__ item_seg, u32 *__restrict__ item_sub,
							    u32 *__restrict__ status, RsFirst hs)
{
	// (grid-stride; the workgroups also count the items' 8-bit keys for the sort that follows -- ugrt_rs_hist.h)
	__shared__ u32 s_rsh[RS_BINS * RS_PRIV];
	d_rs_zero(s_rsh, hs);
	__syncthreads();
	const u32 nitems = xincl[G - 1];
	if (status && blockIdx.x == 0 && threadIdx.x == 0 && nitems > cap)
		atomicOr(status, UGRT_STATUS_ITEM_OVERFLOW); // asynchronous form: the list was sized by an estimate
	for (u32 i0 = blockIdx.x * WL_THREADS; i0 < cap; i0 += gridDim.x * WL_THREADS) {
		const u32 it = i0 + threadIdx.x;
		const bool ok = it < cap;
		u32 seg = XSEG_LAST + 1u, sub = 0u; // the list is sorted at its capacity: padding goes last
		if (ok && it < nitems) {
			const u32 g = d_find_cell(xincl, G, it);
			u32 nseg = (pend[g] - pstart[g] + XSEG - 1) / XSEG;
			nseg = nseg < XSEG_LAST + 1u ? nseg : XSEG_LAST + 1u;
			const u32 nsub = (boxes[g].ray_count + 63u) / 64u;
			const u32 local = it - (xincl[g] - nseg * nsub);
			seg = local / nsub;
			sub = (g << 7) | (local % nsub);
		}
		if (ok) {
			item_seg[it] = seg;
			item_sub[it] = sub;
		}
		if (hs.hist)
			d_rs_count(s_rsh, seg & 0xFFu, ok);
	}
	__syncthreads();
	d_rs_flush(s_rsh, hs);
}

// EXACT pass: item -> (beam, segment of its candidate list, 64-ray sub-group); lane = ray, the reference's test
#ifdef UGRT_SHADOW_TIMELINE
// Instrumented builds only (make EXTRA=-DUGRT_SHADOW_TIMELINE; tools/shadow_timeline.py): every wave of the exact pass
// leaves its start and end time (s_memrealtime, 100 MHz) and the number of items it worked on.
__device__ unsigned long long *g_shadow_tl;
#endif
template <bool REC>
__global__ __launch_bounds__(64) void k_trace_shadow(CamBlock cam, const u32 *__restrict__ xincl, u32 G,
						      const u32 *__restrict__ item_seg, const u32 *__restrict__ item_sub,
						      const GBox *__restrict__ boxes, const u32 *__restrict__ pstart,
						      const u32 *__restrict__ pend, const u32 *__restrict__ pair_tri,
						      const float *__restrict__ verts, const int *__restrict__ tris,
						      const float4 *__restrict__ rec, const float *__restrict__ t_value_list,
						      const float *__restrict__ ray_direction_list,
						      int *__restrict__ is_shadowed, const u32 *__restrict__ ray_pixels,
						      const float *__restrict__ cmPt, u32 XSEG, u32 *__restrict__ sub_done,
						      u32 nsubmax, const float4 *__restrict__ sray, u32 item_cap,
						      u32 *__restrict__ report, const u32 *__restrict__ status,
						      const unsigned long long *__restrict__ work, u32 SLICES, u32 W0, u32 SIEVE, u32 NSIEVE)
{
	__shared__ __attribute__((aligned(16))) float lds[64 * 16]; // per survivor: tvec, e1, e2, then the part all rays share: qvec, T
	const int lane = threadIdx.x;
#ifdef UGRT_SHADOW_TIMELINE
	const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
	u32 tl_n = 0;
#endif
	// every kernel that raises a status bit or counts work has finished: complete the pass's report
	if (blockIdx.x == 0 && lane == 0) {
		report[2] = *status;
		report[3] = 0u;
		reinterpret_cast<unsigned long long *>(report + 4)[0] = work[0];
		reinterpret_cast<unsigned long long *>(report + 4)[1] = work[1];
	}
	// (the list is written, and sorted, at its capacity: behind the last item come entries of segment XSEG_LAST + 1.  A
	// wave tells by its entry that there is nothing to do - not by the number of items, a load every one of the 400 k
	// single-item waves would have to wait for first.  Asynchronous form: a list cut at its estimated capacity is
	// flagged by k_pair_items.)
	const u32 total = item_cap;
	const float lx = cam.cc[0], ly = cam.cc[1], lz = cam.cc[2];
	// SLICES 1: persistent waves, a contiguous slice of the list per XCD; otherwise one wave per item, runs of
	// 2^((SLICES >> 1) - 1) items per XCD in turn (as the primary tracer)
	u32 first = blockIdx.x;
	if (SLICES == 1u) {
		first = d_xcd_block();
	} else if (SLICES > 1u) {
		const u32 rl = (SLICES >> 1) - 1u, j = blockIdx.x >> 3;
		first = ((((j >> rl) << 3) + (blockIdx.x & 7u)) << rl) + (j & ((1u << rl) - 1u));
	}
	// Which items a wave takes.  The list is sorted by segment, the first segments of all sub-groups come first and are
	// where the work is: of the 325 k items of the bench frame 42 k do anything, 33 k of them among the first 40 k; the
	// other 283 k find their sub-group flagged - 0.9 us each, but the chip starts fewer than four waves per ns, so they
	// were 77 of the pass's 178 us, and the long items behind them (up to 80 us: lit rays meet every candidate) started
	// late and were its tail (per-wave time stamps, profiles/r04_shadow_exact_timeline.txt).  So only the first W0 items
	// (>= the number of sub-groups: rays / 64 + beams) get a wave each; behind them a SIEVE wave looks at SIEVE items at
	// once, lane = item, and works off the few that have anything to do.  Its items lie NSIEVE apart: the working ones
	// cluster (neighbouring sub-groups of a beam with lit rays) and must not meet in one wave.
	// (persistent form: every wave takes single items, in a stride)
	const bool persistent = SLICES == 1u;
	for (u32 base = first; base < (persistent ? total : W0 + NSIEVE); base += gridDim.x) {
		const bool sieve = !persistent && base >= W0; // (base = W0 + v: the sieve wave's first item)
		const u32 n_v = sieve ? SIEVE : 1u, stride_v = sieve ? NSIEVE : 0u;
		u32 sgm_v = XSEG_LAST + 1u, gs_v = 0u;
		const u32 my_it = base + (u32)lane * stride_v;
		if ((u32)lane < n_v && my_it < total) {
			sgm_v = item_seg[my_it];
			gs_v = item_sub[my_it];
		}
		// Three quarters of the items find every ray of their sub-group flagged by an earlier segment (the
		// point of the segment-major order).  The sub-group says so in one word, and the flags are looked
		// at before the rays are rebuilt.  (a padding entry reads beam 0's word)
		const u32 flagged_v = __hip_atomic_load(sub_done + (size_t)(gs_v >> 7) * nsubmax + (gs_v & 127u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		unsigned long long todo = __ballot(sgm_v <= XSEG_LAST && flagged_v == 0u);
	while (todo != 0ull) {
#ifdef UGRT_SHADOW_TIMELINE
		tl_n++;
#endif
		const int tl = (int)__builtin_ctzll(todo);
		todo &= todo - 1ull;
		// The head of an item is a chain of loads that depend on each other, and the pass is as long as its
		// chains: what does not depend on a load is requested with it.  {entry} -> {the sub-group's "all flagged"
		// word} -> {beam, candidate run} -> {pixel, rebuilt ray, ids of the first candidates} -> {flag of the pixel, the
		// candidates' records}: five round trips where the straightforward order made ten.  Indices are clamped, not
		// masked: loads under a divergent branch are waited for at the join.
		const u32 sgm = (u32)__builtin_amdgcn_readlane((int)sgm_v, tl), gs = (u32)__builtin_amdgcn_readlane((int)gs_v, tl);
		const u32 g = gs >> 7, sub = gs & 127u; // the sub-groups of a beam share its candidate list (a padding entry reads beam 0's word)
		u32 *my_done = sub_done + (size_t)g * nsubmax + sub;
		const GBox bx = boxes[g];
		const u32 ps = pstart[g], pe = pend[g];
		const u32 p0 = ps + sgm * XSEG;
		const u32 p1 = (sgm != XSEG_LAST && (p0 + XSEG) < pe) ? (p0 + XSEG) : pe;
		const u32 rl0 = 64u * sub + (u32)lane;
		const bool have_ray = rl0 < bx.ray_count;
		const u32 ri = bx.ray_start + (have_ray ? rl0 : bx.ray_count - 1u); // (a listed sub-group has a ray)
		const int pixel = (int)ray_pixels[ri];
		const float4 q = sray[ri]; // = d_shadow_ray(pixel), stored by k_shadow_boxes
		const u32 id_first = pair_tri[min(p0 + (u32)lane, p1 - 1u)]; // (a listed segment has a candidate)
		u32 id_next = pair_tri[min(p0 + 64u + (u32)lane, p1 - 1u)];   // (beyond the run: its last candidate again)
		// a ray already flagged by another segment of its beam needs no more tests
		float qx = q.x, qy = q.y, qz = q.z, qw = q.w;
		// (all four requests go out together; the compiler would move the ones the early exit below does not need behind it)
		asm volatile("" : "+v"(id_next), "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qw));
		const int flag = is_shadowed[pixel];
		float t9n[9];
		d_load_triangle<REC>(rec, verts, tris, id_first, lx, ly, lz, t9n);
		// (the records are requested beside the flag, not behind the test of it)
		asm volatile("" : "+v"(t9n[0]), "+v"(t9n[3]), "+v"(t9n[4]), "+v"(t9n[8]));
		bool done = !have_ray | (flag == 1); // rayDoneMap == 2
		if (__ballot(!done) == 0ull) {
			if (lane == 0)
				__hip_atomic_store(my_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			continue;
		}
		ShadowRay r;
		r.rd[0] = qx;
		r.rd[1] = qy;
		r.rd[2] = qz;
		r.distance_b = qw;
		r.pixel = pixel;
		// the candidates were found for the whole beam; this wave's 64 (still undecided) rays are a
		// narrower packet, so each staged candidate is culled once more against their own box
		DirBox box = d_dir_box(r.rd, !done);
		unsigned long long boxed = __ballot(!done); // the rays the box was formed for
		bool hit = false;
		for (u32 b = p0; b < p1; b += 64) {
			const u32 ncand = (p1 - b) < 64u ? (p1 - b) : 64u;
			// rays flagged by the batches before need no more tests: once an eighth of the box's rays are gone the box is
			// formed again for the rest (six wave reductions against ~35 instructions for every candidate it then culls)
			{
				const unsigned long long open = __ballot(!done);
				if (8u * (u32)__popcll(open) <= 7u * (u32)__popcll(boxed)) {
					box = d_dir_box(r.rd, !done);
					boxed = open;
				}
			}
			// (this batch's records were requested a batch ago, the ids of the next one with them: now that batch's records
			// are requested, and the ids of the one after)
			float t9[9];
#pragma unroll
			for (int k = 0; k < 9; k++)
				t9[k] = t9n[k];
			d_load_triangle<REC>(rec, verts, tris, id_next, lx, ly, lz, t9n);
			id_next = pair_tri[min(b + 128u + (u32)lane, p1 - 1u)];
			const bool keep = (u32)lane < ncand && !d_cull(&t9[0], &t9[3], &t9[6], box);
			const unsigned long long mask = __ballot(keep);
			const u32 cnt = (u32)__popcll(mask);
			__syncthreads();
			if (keep) {
				// (all rays start at the light: tvec x e1 and e2 . (tvec x e1) are the triangle's, formed here once)
				float qv[3], T;
				d_mt_shared(&t9[0], &t9[3], &t9[6], qv, &T);
				float4 *dst = reinterpret_cast<float4 *>(&lds[d_rank_in_mask(mask) * 16u]);
				dst[0] = make_float4(t9[0], t9[1], t9[2], t9[3]);
				dst[1] = make_float4(t9[4], t9[5], t9[6], t9[7]);
				dst[2] = make_float4(t9[8], qv[0], qv[1], qv[2]);
				dst[3] = make_float4(T, 0.0f, 0.0f, 0.0f);
			}
			__syncthreads();
			if (!done) {
				for (u32 k = 0; k < cnt; k++) {
					const float4 *src = reinterpret_cast<const float4 *>(&lds[k * 16u]);
					const float4 a = src[0], c = src[1], e = src[2];
					const float T = lds[k * 16u + 12u];
					const float tv[3] = { a.x, a.y, a.z }, e1[3] = { a.w, c.x, c.y }, e2[3] = { c.z, c.w, e.x }, qv[3] = { e.y, e.z, e.w };
					const float value = d_intersect_tri_shared(tv, e1, e2, qv, T, r.rd, 999999.9f);
					if (value != 0.0f) {
						// light_kernel.cu:193-202 with isSmaller (:1-11)
						float pt[3];
						pt[0] = lx + value * r.rd[0];
						pt[1] = ly + value * r.rd[1];
						pt[2] = lz + value * r.rd[2];
						float distance_a = __builtin_sqrtf((pt[0] - lx) * (pt[0] - lx) + (pt[1] - ly) * (pt[1] - ly) +
										   (pt[2] - lz) * (pt[2] - lz));
						if (distance_a + 1e-03f < r.distance_b) {
							hit = true;
							done = true;
							break;
						}
					}
				}
			}
			// the whole beam is decided: skip the remaining candidates
			if (__ballot(!done) == 0ull)
				break;
		}
		if (hit)
			is_shadowed[r.pixel] = 1;
		if (__ballot(!done) == 0ull && lane == 0)
			__hip_atomic_store(my_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
		if (!persistent)
			break;
	}
#ifdef UGRT_SHADOW_TIMELINE
	if (g_shadow_tl && lane == 0) {
		g_shadow_tl[2 * (size_t)blockIdx.x] = tl0;
		g_shadow_tl[2 * (size_t)blockIdx.x + 1] = (__builtin_amdgcn_s_memrealtime() << 8) | (tl_n & 255u);
	}
#endif
}

// Every float bit pattern through d_recip_det against the division it stands for (ugrt_dev.h); *mismatches = operands whose
// reciprocal differs by a bit.  ugrt_ctx_get_state "recip_mismatches".
__global__ __launch_bounds__(256) void k_recip_selftest(unsigned long long *bad)
{
	unsigned long long mine = 0;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const float x = __uint_as_float((u32)i);
		if (x > -D_EPSILON && x < D_EPSILON) // the tests return before they divide
			continue;
		const float want = 1.0f / x, got = d_recip_det(x);
		const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
		mine += same ? 0u : 1u;
	}
	if (mine)
		atomicAdd(bad, mine);
}

int ugrt_recip_selftest(ugrt_ctx *ctx, unsigned long long *mismatches)
{
	UGRT_HIP(hipSetDevice(ctx->device));
	unsigned long long *d = nullptr;
	UGRT_HIP(hipMalloc((void **)&d, sizeof *d));
	hipError_t e = hipMemsetAsync(d, 0, sizeof *d, ctx->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_recip_selftest, dim3(4096), dim3(256), 0, ctx->stream, d);
		e = hipMemcpyAsync(mismatches, d, sizeof *d, hipMemcpyDeviceToHost, ctx->stream);
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	UGRT_HIP(e);
	return UGRT_OK;
}

// The cross-lane reductions of ugrt_packet.h (DPP controls, v_permlane16_swap / v_permlane32_swap) against the same
// reductions by __shfl_xor, on pseudo-random values; *mismatches = lanes that differ.  ugrt_ctx_get_state "lane_reduce_mismatches".
__global__ __launch_bounds__(64) void k_lane_reduce_selftest(unsigned long long *bad)
{
	const int lane = threadIdx.x;
	u32 mine = 0;
	for (u32 round = 0; round < 64u; round++) {
		u32 x = (blockIdx.x * 64u + (u32)lane) * 2654435761u + round * 40503u;
		x ^= x >> 15;
		x *= 2246822519u;
		x ^= x >> 13;
		const int v = (int)x;
		// inside the quadrants: lane bits 0, 1, 3, 4
		int lo = v, hi = v;
		for (int m = 1; m <= 16; m <<= 1) {
			if (m == 4)
				continue;
			const int ol = __shfl_xor(lo, m), oh = __shfl_xor(hi, m);
			lo = ol < lo ? ol : lo;
			hi = oh > hi ? oh : hi;
		}
		const int qlo = d_quadrant_reduce<DOpMin>(v), qhi = d_quadrant_reduce<DOpMax>(v);
		mine += (qlo != lo) + (qhi != hi);
		// across them: bits 2 and 5
		int alo = lo, ahi = hi;
		for (int m = 4; m <= 32; m <<= 3) {
			const int ol = __shfl_xor(alo, m), oh = __shfl_xor(ahi, m);
			alo = ol < alo ? ol : alo;
			ahi = oh > ahi ? oh : ahi;
		}
		mine += (d_across_quadrants<DOpMin>(qlo) != alo) + (d_across_quadrants<DOpMax>(qhi) != ahi);
		// the whole wave, floats (a quarter of the lanes do not contribute)
		const float f = __int_as_float((v & 0x3FFFFFFF) | 0x20000000) * ((v & 4) ? -1.0f : 1.0f);
		const bool in = (v & 3) != 0;
		float wlo = in ? f : __builtin_huge_valf(), whi = in ? f : -__builtin_huge_valf();
		for (int m = 32; m >= 1; m >>= 1) {
			wlo = fminf(wlo, __shfl_xor(wlo, m));
			whi = fmaxf(whi, __shfl_xor(whi, m));
		}
		mine += (d_wave_fmin(in ? f : __builtin_huge_valf()) != wlo) + (d_wave_fmax(in ? f : -__builtin_huge_valf()) != whi);
	}
	if (mine)
		atomicAdd(bad, (unsigned long long)mine);
}

int ugrt_lane_reduce_selftest(ugrt_ctx *ctx, unsigned long long *mismatches)
{
	UGRT_HIP(hipSetDevice(ctx->device));
	unsigned long long *d = nullptr;
	UGRT_HIP(hipMalloc((void **)&d, sizeof *d));
	hipError_t e = hipMemsetAsync(d, 0, sizeof *d, ctx->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_lane_reduce_selftest, dim3(1024), dim3(64), 0, ctx->stream, d);
		e = hipMemcpyAsync(mismatches, d, sizeof *d, hipMemcpyDeviceToHost, ctx->stream);
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	UGRT_HIP(e);
	return UGRT_OK;
}

// Every float bit pattern through the device forms of ugrt_f2i / ugrt_f2u / ugrt_floor2i (one or two instructions) against
// the portable forms of include/ugrt_fmath.h; *mismatches = operands that differ in any of the three.
// ugrt_ctx_get_state "f2i_mismatches".
__global__ __launch_bounds__(256) void k_f2i_selftest(unsigned long long *bad)
{
	unsigned long long mine = 0;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const float x = __uint_as_float((u32)i);
		const bool same = ugrt_f2i(x) == ugrt_f2i_portable(x) && ugrt_f2u(x) == ugrt_f2u_portable(x) &&
				  ugrt_floor2i(x) == ugrt_floor2i_portable(x);
		mine += same ? 0u : 1u;
	}
	if (mine)
		atomicAdd(bad, mine);
}

int ugrt_f2i_selftest(ugrt_ctx *ctx, unsigned long long *mismatches)
{
	UGRT_HIP(hipSetDevice(ctx->device));
	unsigned long long *d = nullptr;
	UGRT_HIP(hipMalloc((void **)&d, sizeof *d));
	hipError_t e = hipMemsetAsync(d, 0, sizeof *d, ctx->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_f2i_selftest, dim3(4096), dim3(256), 0, ctx->stream, d);
		e = hipMemcpyAsync(mismatches, d, sizeof *d, hipMemcpyDeviceToHost, ctx->stream);
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	UGRT_HIP(e);
	return UGRT_OK;
}

static int bits_of(u32 v)
{
	int b = 1;
	while (b < 32 && (1ull << b) < (unsigned long long)v)
		b++;
	return b;
}

// check_for_shadows, per_frame_funcs.h:139-159
extern "C" int ugrt_trace_shadow(ugrt_ctx *ctx, const unsigned *d_value_list, const float *d_vertlist,
				 const int *d_trilist, const unsigned *d_span, const unsigned *d_offset,
				 const float *d_t_value, const float *d_ray_dir, int *d_is_shadowed,
				 const unsigned *d_map, const unsigned *d_prefix_map, const float *d_cam_position,
				 unsigned num_chunks)
{
	if (!ctx || !d_value_list || !d_vertlist || !d_trilist || !d_span || !d_offset || !d_t_value || !d_ray_dir ||
	    !d_is_shadowed || !d_map || !d_prefix_map || !d_cam_position)
		return ugrt_fail(UGRT_EINVAL, "trace_shadow: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	const u32 C = (u32)ctx->cfg.light_nbx * (u32)ctx->cfg.light_nby;
	const u32 n = (u32)ctx->npix;
	// which chunks get traced: the reference launches nbx*nby blocks, block b
	// takes chunk b-1 and only blocks b < num_chunks work (light_kernel.cu:76-85)
	u32 traced;
	const bool on_device = num_chunks == UGRT_CHUNKS_ON_DEVICE; // ugrt_sort_rays(..., NULL) left the count there
	const u32 launch_cap = (ctx->cfg.flags & UGRT_FLAG_SHADOW_ALL_CHUNKS) ? 0xFFFFFFFFu : (u32)ctx->nbx * (u32)ctx->nby;
	const u32 *nchunks_dev = nullptr;
	u32 nchunks_arg = num_chunks;
	if (on_device) {
		if (!ctx->ray_sort_pending && !ctx->cbase.p)
			return ugrt_fail(UGRT_EINVAL, "trace_shadow: UGRT_CHUNKS_ON_DEVICE without a ugrt_sort_rays before");
		if (ctx->chunk_prefix != d_prefix_map || ctx->chunk_map != d_map)
			return ugrt_fail(UGRT_EINVAL, "trace_shadow: UGRT_CHUNKS_ON_DEVICE refers to the last ugrt_sort_rays, which "
						      "sorted other arrays");
		traced = num_chunks; // the keys kernel applies the launch rule itself
		if (ctx->ray_sort_pending)
			nchunks_arg = 0u; // every chunk is traced and none was formed: d_map is the unsorted map, all n rays count
		else
			nchunks_dev = (const u32 *)ctx->cbase.p + C; // inclusive scan of the chunks per light cell, last entry
	} else if (launch_cap == 0xFFFFFFFFu) {
		traced = num_chunks;
	} else {
		u32 lim = num_chunks < launch_cap ? num_chunks : launch_cap;
		traced = lim ? lim - 1 : 0;
	}
	ctx->stats[2] = traced;
	ctx->stats[1] = 0;
	ctx->stats[6] = ctx->stats[7] = 0;
	if (traced == 0 || n == 0)
		return UGRT_OK;
	const bool use_rec = ctx->rec_valid && ctx->rec_verts == d_vertlist && ctx->rec_tris == d_trilist;
	const float4 *rec = use_rec ? (const float4 *)ctx->trirec.p : (const float4 *)nullptr;
	int rc;
	// NUM_SLABS > 1: span/offset hold C * slabs entries; the block of a chunk walks all slabs of its cell
	// (light_kernel.cu:105-113) and a ray is shadowed by an occluder in any of them: the cell's list is the
	// union of its slabs' runs, which lie next to each other
	if (ctx->cfg.slabs > 1 && (rc = ugrt_slab_union(ctx, d_span, d_offset, C, (u32)ctx->cfg.slabs, &d_span, &d_offset)))
		return rc;
	for (int i = 0; i < 2; i++) {
		if ((rc = ugrt_buf_reserve(ctx, ctx->skey[i], (size_t)n * 8)))
			return rc;
		if ((rc = ugrt_buf_reserve(ctx, ctx->sval[i], (size_t)n * 4)))
			return rc;
	}
	const u32 ncellk = C + 2; // + sentinel + "not traced"
	const size_t maxg = (size_t)n / 64 + C + 1; // beams
	if ((rc = ugrt_buf_reserve(ctx, ctx->sstart, (size_t)ncellk * 8))) // run starts, then run ends
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->sbase, (size_t)C * 8)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->sdesc, maxg * sizeof(GBox))))
		return rc;
	// candidate run starts, then ends, per beam; then one "all rays flagged" word per 64-ray sub-group
	if ((rc = ugrt_buf_reserve(ctx, ctx->tbcnt, maxg * 8 + maxg * 128 * 4)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->witems, maxg * 8)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->sray, (size_t)n * 16))) // rebuilt shadow rays in beam order
		return rc;
	void *k0 = ctx->skey[0].p, *k1 = ctx->skey[1].p;
	u32 *v0 = (u32 *)ctx->sval[0].p, *v1 = (u32 *)ctx->sval[1].p;
	u32 *rstart = (u32 *)ctx->sstart.p, *rend = rstart + ncellk;
	u32 *gincl = (u32 *)ctx->sbase.p, *iincl = gincl + C;
	u32 *pstart = (u32 *)ctx->tbcnt.p, *pend = pstart + maxg;
	GBox *boxes = (GBox *)ctx->sdesc.p;
	unsigned long long *wcnt = (unsigned long long *)(ctx->d_small + UGRT_DSMALL_SHADOW_WORK); // [0] cull tests, [1] staged candidates
	// (the two work counters + the candidate-pair cursor behind them, and the cull pass's output cursors, are cleared by
	// the keys kernel: two fills less per pass)
	if ((rc = ugrt_buf_reserve(ctx, ctx->pseg, (size_t)PAIR_SEGS * PAIR_SEG_STRIDE * 4)))
		return rc;
	ugrt_prof_begin(ctx, UGRT_ST_SHADOW_PREP);
	// 1. rays: (cell, direction code) order, runs per cell, beams
	// key = (light cell, direction code): 32 bits when the cell index leaves >= 12 bits for the code
	const u32 cellbits = (u32)bits_of(ncellk);
	bool key64 = cellbits > 20u;
	key64 = key64 || ctx->opt[UGRT_OPT_SHADOW_KEY64] == 1;
	u32 mbits = 32u - cellbits;
	if (ctx->opt[UGRT_OPT_SHADOW_MBITS] > 0 && (u32)ctx->opt[UGRT_OPT_SHADOW_MBITS] < mbits)
		mbits = (u32)ctx->opt[UGRT_OPT_SHADOW_MBITS];
	mbits = mbits > 24u ? 24u : mbits;
	const bool own_sort = ctx->opt[UGRT_OPT_SORT_LIBRARY] != 1;
	const u32 kblocks = (u32)((n + WL_THREADS - 1) / WL_THREADS) < 768u ? (u32)((n + WL_THREADS - 1) / WL_THREADS) : 768u;
	if (key64) {
		hipLaunchKernelGGL(k_shadow_keys<true>, dim3(kblocks), dim3(WL_THREADS), 0, st,
				   ctx->cam, d_t_value, d_ray_dir, d_map, d_prefix_map, nchunks_arg, nchunks_arg ? traced : 0u, n, C, d_span,
				   d_cam_position, 30u, k0, v0, rstart, 2u * ncellk, nchunks_dev, launch_cap, ctx->chunk_capacity,
				   (u32 *)wcnt, 5u, (u32 *)ctx->pseg.p, PAIR_SEGS * PAIR_SEG_STRIDE, RsFirst{ nullptr });
		UGRT_HIP(hipGetLastError());
		if ((rc = ugrt_prim_sort_pairs64(ctx, (const u64 *)k0, (u64 *)k1, v0, v1, n, 30 + (int)cellbits)))
			return rc;
	} else {
		// (the kernels that write this pass's sort keys count their first digit: no histogram kernel before the sorts)
		RsFirst hs = { nullptr };
		if (own_sort && (rc = ugrt_sort_first_digit(ctx, &hs)))
			return rc;
		hipLaunchKernelGGL(k_shadow_keys<false>, dim3(kblocks), dim3(WL_THREADS), 0, st,
				   ctx->cam, d_t_value, d_ray_dir, d_map, d_prefix_map, nchunks_arg, nchunks_arg ? traced : 0u, n, C, d_span,
				   d_cam_position, mbits, k0, v0, rstart, 2u * ncellk, nchunks_dev, launch_cap, ctx->chunk_capacity,
				   (u32 *)wcnt, 5u, (u32 *)ctx->pseg.p, PAIR_SEGS * PAIR_SEG_STRIDE, hs);
		UGRT_HIP(hipGetLastError());
		if ((rc = own_sort ? ugrt_sort_pairs_u32(ctx, (const u32 *)k0, (u32 *)k1, v0, v1, n, (int)(mbits + cellbits), nullptr, true)
				   : ugrt_prim_sort_pairs(ctx, (const u32 *)k0, (u32 *)k1, v0, v1, n, (int)(mbits + cellbits))))
			return rc;
	}
	if (key64)
		hipLaunchKernelGGL(k_shadow_runs<u64>, dim3((n + WL_THREADS - 1) / WL_THREADS), dim3(WL_THREADS), 0, st,
				   (const u64 *)k1, n, 30u, rstart, rend);
	else
		hipLaunchKernelGGL(k_shadow_runs<u32>, dim3((n + WL_THREADS - 1) / WL_THREADS), dim3(WL_THREADS), 0, st,
				   (const u32 *)k1, n, mbits, rstart, rend);
	UGRT_HIP(hipGetLastError());
	// rays per beam: the cull pass costs (triangles of the cell) x (beams of the cell); the exact pass
	// re-culls the beam's candidates against each 64-ray sub-group, so its cost barely depends on the
	// beam size.  ~1000 rays per beam is the measured optimum on the 1 M-triangle scene (tools/beam_sweep.py)
	u32 beam = ctx->opt[UGRT_OPT_SHADOW_BEAM] > 0 ? (u32)ctx->opt[UGRT_OPT_SHADOW_BEAM] : 2048u;
	beam = beam < 64u ? 64u : (beam > 8192u ? 8192u : (beam + 63u) / 64u * 64u);
	// candidates per exact-pass work item: a 64-ray sub-group stops at the first batch after which all its
	// rays are flagged, so long items cost little where everything is in shadow; short items bound the
	// work of a sub-group that stays lit (128 since the later segments' items go through sieve waves: twice the items
	// were twice the waves to start before - 256 then; profiles/r04_shadow_sieve_sweep.txt)
	u32 XSEG = ctx->opt[UGRT_OPT_SHADOW_XSEG] > 0 ? (u32)ctx->opt[UGRT_OPT_SHADOW_XSEG] : 128u;
	XSEG = XSEG < 64u ? 64u : (XSEG + 63u) / 64u * 64u;
	{
		// (the beams and the cull items per cell: two scans over the light cells in one launch)
		const ShadowCountLoad beams = { d_span, rstart, rend, beam, wcnt, false };
		const ShadowCountLoad items = { d_span, rstart, rend, beam, wcnt, true };
		if ((rc = ugrt_scan_launch_pair<true>(ctx, beams, gincl, items, iincl, C)))
			return rc;
	}
	if ((rc = ugrt_buf_reserve(ctx, ctx->citem, (size_t)CULL_TABLE * sizeof(CullItem))))
		return rc;
	hipLaunchKernelGGL(k_shadow_boxes, dim3(launch_blocks_for((u32)maxg)), dim3(64 * BOX_WAVES), 0, st, ctx->cam,
			   (const u32 *)gincl, C, (const u32 *)rstart, (const u32 *)rend, (const u32 *)v1, d_t_value,
			   d_ray_dir, d_cam_position, boxes, beam, pstart, (u32)(2 * maxg + maxg * (beam / 64u)),
			   (float4 *)ctx->sray.p, (const u32 *)iincl, d_span, d_offset, (CullItem *)ctx->citem.p);
	UGRT_HIP(hipGetLastError());
	ugrt_prof_end(ctx, UGRT_ST_SHADOW_PREP);
	u32 sbits = ctx->opt[UGRT_OPT_SHADOW_SIZEBITS] >= 0 ? (u32)ctx->opt[UGRT_OPT_SHADOW_SIZEBITS] : 4u;
	sbits = sbits > 8u ? 8u : sbits;
	// 2. cull pass -> (beam, triangle) candidate pairs.  Synchronous form: the pair count is read back (it sizes the
	// launches that follow), and the pass is repeated with a larger buffer if it was too small.  Asynchronous form
	// (option "async_build", once a synchronous pass has left estimates): no read-back; the launches are sized by
	// the previous pass's counts plus a quarter, the kernels take the real counts from the device, and counts
	// beyond the capacities raise a status bit instead (UGRT_EOVERFLOW at the next synchronisation).
	u32 *pcount = ctx->d_small + UGRT_DSMALL_PAIRS; // right behind the work counters: cleared with them
	// (the report is written by the kernels straight into the pinned host words: no copy behind the pass)
	u32 *status = ctx->d_small + UGRT_DSMALL_STATUS, *pg = ctx->d_small + UGRT_DSMALL_SHADOW, *report = ctx->h_pinned + UGRT_PIN_SHADOW;
	if (ctx->shadow_async_pending) { // what the last asynchronous pass needed (possibly a frame old)
		ctx->est_pairs = ctx->h_pinned[UGRT_PIN_SHADOW];
		ctx->est_beams = ctx->h_pinned[UGRT_PIN_SHADOW + 1];
	}
	const bool async = ctx->opt[UGRT_OPT_ASYNC_BUILD] == 1 && ctx->have_shadow_est && ugrt_reported_status(ctx) == 0u &&
			   !ctx->overflow_seen;
	if (!async && ugrt_reported_status(ctx) != 0u)
		ctx->overflow_seen = true;
	u32 P = 0, G = 0, Gcap = 0, xcap = 0;
	const u32 *pgp = nullptr; // device counts (asynchronous form)
	for (int attempt = 0; attempt < 3; attempt++) {
		size_t cap = ctx->tkey[0].cap / 4;
		size_t want0 = (size_t)4 << 22;
		if (async && ((size_t)ctx->est_pairs + ctx->est_pairs / 2 + 65536) * 4 > want0)
			want0 = ((size_t)ctx->est_pairs + ctx->est_pairs / 2 + 65536) * 4;
		if (cap * 4 < want0) {
			if ((rc = ugrt_buf_reserve(ctx, ctx->tkey[0], want0)) || (rc = ugrt_buf_reserve(ctx, ctx->tkey[1], want0)) ||
			    (rc = ugrt_buf_reserve(ctx, ctx->tval[0], want0)) || (rc = ugrt_buf_reserve(ctx, ctx->tval[1], want0)))
				return rc;
		}
		cap = ctx->tkey[0].cap / 4;
		if (ctx->tkey[1].cap / 4 < cap)
			cap = ctx->tkey[1].cap / 4;
		if (ctx->tval[0].cap / 4 < cap)
			cap = ctx->tval[0].cap / 4;
		if (ctx->tval[1].cap / 4 < cap)
			cap = ctx->tval[1].cap / 4;
		if (cap > 0xFFFFFFF0u)
			cap = 0xFFFFFFF0u;
		u32 *segcnt = (u32 *)ctx->pseg.p;
		const u32 segcap = (u32)(cap / PAIR_SEGS);
		if (attempt > 0) // (the first attempt's cursors were cleared by the keys kernel)
			UGRT_HIP(hipMemsetAsync(segcnt, 0, (size_t)PAIR_SEGS * PAIR_SEG_STRIDE * 4, st));
		PairCheck chk = { nullptr, 0u, 0u, nullptr, nullptr, nullptr };
		if (async) {
			// beams: a power of two above the estimate keeps the sort at the key width the real count needs
			u32 gb = 1;
			while (gb < ctx->est_beams + ctx->est_beams / 4u + 1u)
				gb <<= 1;
			Gcap = (u32)maxg;
			G = gb < Gcap ? gb : Gcap; // only its bit width is used below
			// launch size of the per-pair kernels; the check is made against it, not against the (larger) buffers:
			// the sort and the run kernel work on P pairs, so a count between the two would lose candidates
			const size_t lp = (size_t)ctx->est_pairs + ctx->est_pairs / 4 + 65536;
			P = (u32)(lp < cap ? lp : cap);
			chk = PairCheck{ (const u32 *)(gincl + (C - 1)), P, G, pg, status, report }; // (made by the compaction's first workgroup)
		}
		RsFirst hsp = { nullptr };
		if (own_sort && (rc = ugrt_sort_first_digit(ctx, &hsp)))
			return rc;
		ugrt_prof_begin(ctx, UGRT_ST_SHADOW_CULL);
		if (use_rec)
			hipLaunchKernelGGL(k_shadow_cull<true>, dim3(launch_blocks_for(0xFFFFFFFFu, ctx->opt[UGRT_OPT_SHADOW_WAVES])), dim3(64), 0, st, ctx->cam, (const u32 *)iincl,
					   (const u32 *)gincl, C, d_span, d_offset, d_value_list, rec, d_vertlist, d_trilist,
					   (const GBox *)boxes, segcnt, segcap, (u32 *)ctx->tkey[1].p, (u32 *)ctx->tval[1].p, sbits,
					   (const CullItem *)ctx->citem.p);
		else
			hipLaunchKernelGGL(k_shadow_cull<false>, dim3(launch_blocks_for(0xFFFFFFFFu, ctx->opt[UGRT_OPT_SHADOW_WAVES])), dim3(64), 0, st, ctx->cam, (const u32 *)iincl,
					   (const u32 *)gincl, C, d_span, d_offset, d_value_list, rec, d_vertlist, d_trilist,
					   (const GBox *)boxes, segcnt, segcap, (u32 *)ctx->tkey[1].p, (u32 *)ctx->tval[1].p, sbits,
					   (const CullItem *)ctx->citem.p);
		hipLaunchKernelGGL(k_pair_compact, dim3(PAIR_SEGS * 16u), dim3(256), 0, st, (const u32 *)segcnt, segcap,
				   (const u32 *)ctx->tkey[1].p, (const u32 *)ctx->tval[1].p, (u32 *)ctx->tkey[0].p,
				   (u32 *)ctx->tval[0].p, pcount, chk, hsp);
		ugrt_prof_end(ctx, UGRT_ST_SHADOW_CULL);
		UGRT_HIP(hipGetLastError());
		if (async) {
			pgp = pg; // (the report travels to the host with the copy behind the exact pass)
			xcap = (ctx->est_beams + ctx->est_beams / 4u + 64u + P / XSEG) * (beam / 64u);
			ctx->shadow_async_pending = true;
			break;
		}
		UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_PAIRS, pcount, 4, hipMemcpyDeviceToHost, st));
		UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_BEAMS, gincl + (C - 1), 4, hipMemcpyDeviceToHost, st));
		UGRT_HIP(hipStreamSynchronize(st));
		P = ctx->h_pinned[UGRT_PIN_PAIRS];
		if ((size_t)P <= cap)
			break;
		if (attempt == 2)
			return ugrt_fail(UGRT_ENOMEM, "trace_shadow: %u candidate pairs do not fit", P);
		size_t want = (size_t)P * 4 + ((size_t)P * 4) / 4;
		if ((rc = ugrt_buf_reserve(ctx, ctx->tkey[0], want)) || (rc = ugrt_buf_reserve(ctx, ctx->tkey[1], want)) ||
		    (rc = ugrt_buf_reserve(ctx, ctx->tval[0], want)) || (rc = ugrt_buf_reserve(ctx, ctx->tval[1], want)))
			return rc;
	}
	if (!async) {
		G = ctx->h_pinned[UGRT_PIN_BEAMS];
		Gcap = G;
		ctx->stats[1] = G;
		ctx->est_pairs = P;
		ctx->est_beams = G;
		ctx->have_shadow_est = true;
		// (the slots the asynchronous form reports into: never older than this pass)
		ctx->h_pinned[UGRT_PIN_SHADOW] = P;
		ctx->h_pinned[UGRT_PIN_SHADOW + 1] = G;
		ctx->shadow_async_pending = false;
		if (P == 0 || G == 0)
			return UGRT_OK;
		xcap = (G + P / XSEG) * (beam / 64u); // >= number of exact-pass items
	}
	// 3. candidates by beam
	ugrt_prof_begin(ctx, UGRT_ST_SHADOW_PREP);
	if ((rc = own_sort ? ugrt_sort_pairs_u32(ctx, (const u32 *)ctx->tkey[0].p, (u32 *)ctx->tkey[1].p, (const u32 *)ctx->tval[0].p,
						 (u32 *)ctx->tval[1].p, P, bits_of(G) + (int)sbits, pgp, true)
			   : ugrt_prim_sort_pairs(ctx, (const u32 *)ctx->tkey[0].p, (u32 *)ctx->tkey[1].p, (const u32 *)ctx->tval[0].p,
						  (u32 *)ctx->tval[1].p, P, bits_of(G) + (int)sbits, pgp)))
		return rc;
	{
		u32 pb = (P + WL_THREADS - 1) / WL_THREADS;
		hipLaunchKernelGGL(k_pair_runs, dim3(pb ? pb : 1u), dim3(WL_THREADS), 0, st, (const u32 *)ctx->tkey[1].p, P, sbits,
				   pstart, pend, pgp);
	}
	UGRT_HIP(hipGetLastError());
	u32 *xincl = (u32 *)ctx->witems.p;
	{
		const PairItemLoad load = { pstart, pend, boxes, G, wcnt + 1, XSEG, pgp };
		if ((rc = ugrt_scan_launch<true>(ctx, load, xincl, Gcap, ScanTailNone())))
			return rc;
	}
	if ((rc = ugrt_buf_reserve(ctx, ctx->sitem, (size_t)xcap * 16)))
		return rc;
	u32 *iseg0 = (u32 *)ctx->sitem.p, *isub0 = iseg0 + xcap, *iseg1 = isub0 + xcap, *isub1 = iseg1 + xcap;
	const bool item_sort = ctx->opt[UGRT_OPT_SHADOW_ITEMSORT] != 0;
	RsFirst hsi = { nullptr };
	if (item_sort && own_sort && (rc = ugrt_sort_first_digit(ctx, &hsi)))
		return rc;
	{
		const u32 ib = (xcap + WL_THREADS - 1) / WL_THREADS;
		hipLaunchKernelGGL(k_pair_items, dim3(ib < 512u ? (ib ? ib : 1u) : 512u), dim3(WL_THREADS), 0, st,
				   (const u32 *)xincl, Gcap, xcap, (const u32 *)pstart, (const u32 *)pend, (const GBox *)boxes, XSEG,
				   iseg0, isub0, async ? status : (u32 *)nullptr, hsi);
	}
	UGRT_HIP(hipGetLastError());
	if (item_sort) {
		if ((rc = own_sort ? ugrt_sort_pairs_u32(ctx, iseg0, iseg1, isub0, isub1, xcap, 8, nullptr, true)
				   : ugrt_prim_sort_pairs(ctx, iseg0, iseg1, isub0, isub1, xcap, 8)))
			return rc;
	} else {
		iseg1 = iseg0;
		isub1 = isub0;
	}
	ugrt_prof_end(ctx, UGRT_ST_SHADOW_PREP);
	// 4. exact pass
	ugrt_prof_begin(ctx, UGRT_ST_TRACE_SHADOW);
	// One wave per item, runs of `x_run` neighbouring items per XCD in turn (as the primary tracer; alone 0.254 -> 0.216
	// ms, profiles/r03_shadow_waves.txt); "shadow_xcd_run" 0 restores round 2's persistent waves ("shadow_waves" of
	// them, which the cull pass always runs on) with a contiguous slice of the list per XCD
	const int x_opt = ctx->opt[UGRT_OPT_SHADOW_XCD_RUN];
	const bool x_persistent = x_opt == 0;
	u32 x_run = x_opt > 0 ? (u32)x_opt : 128u, x_run_log2 = 0;
	while (x_run >> (x_run_log2 + 1u))
		x_run_log2++;
	x_run = 1u << x_run_log2; // (a power of two: rounded down)
	// (the first W0 items - at least the first segments of all sub-groups, of which there are at most rays / 64 + beams -
	// get a wave each, the rest go through sieve waves of `x_sieve` items: see the kernel)
	const u32 x_sieve = ctx->opt[UGRT_OPT_SHADOW_SIEVE] >= 0 ? (u32)ctx->opt[UGRT_OPT_SHADOW_SIEVE] : 8u;
	u32 xw0 = xcap, xnsieve = 0u;
	if (!x_persistent && x_sieve > 1u) {
		const unsigned long long firsts = (unsigned long long)n / 64u + G + 1u; // (asynchronous form: G is the bound the beams were checked against)
		xw0 = firsts < xcap ? (u32)firsts : xcap;
		xnsieve = (xcap - xw0 + x_sieve - 1u) / x_sieve;
	}
	const u32 xwaves = x_persistent ? (u32)launch_blocks_for(xcap, ctx->opt[UGRT_OPT_SHADOW_WAVES])
					: (u32)(((size_t)xw0 + xnsieve + 8u * x_run - 1) / (8u * x_run) * (8u * x_run));
	const u32 xslices = x_persistent ? 1u : (x_run_log2 + 1u) << 1;
#ifdef UGRT_SHADOW_TIMELINE
	unsigned long long *tlbuf = nullptr;
	if (getenv("UGRT_SHADOW_TIMELINE_FILE")) {
		UGRT_HIP(hipMalloc((void **)&tlbuf, (size_t)xwaves * 16));
		UGRT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_shadow_tl), &tlbuf, sizeof tlbuf, 0, hipMemcpyHostToDevice, st));
		fprintf(stderr, "[shadow timeline] waves %u single-item waves %u sieve waves %u of %u items, %u item slots\n", xwaves, xw0, xnsieve, x_sieve, xcap);
	}
#endif
	if (use_rec)
		hipLaunchKernelGGL(k_trace_shadow<true>, dim3(xwaves), dim3(64), 0, st, ctx->cam,
				   (const u32 *)xincl, Gcap, (const u32 *)iseg1, (const u32 *)isub1, (const GBox *)boxes, (const u32 *)pstart, (const u32 *)pend,
				   (const u32 *)ctx->tval[1].p, d_vertlist, d_trilist, rec, d_t_value, d_ray_dir, d_is_shadowed,
				   (const u32 *)v1, d_cam_position, XSEG, pend + maxg, beam / 64u,
				   (const float4 *)ctx->sray.p, xcap, report, (const u32 *)status, (const unsigned long long *)wcnt, xslices, xw0,
				   x_sieve > 64u ? 64u : x_sieve, xnsieve);
	else
		hipLaunchKernelGGL(k_trace_shadow<false>, dim3(xwaves), dim3(64), 0, st, ctx->cam,
				   (const u32 *)xincl, Gcap, (const u32 *)iseg1, (const u32 *)isub1, (const GBox *)boxes, (const u32 *)pstart, (const u32 *)pend,
				   (const u32 *)ctx->tval[1].p, d_vertlist, d_trilist, rec, d_t_value, d_ray_dir, d_is_shadowed,
				   (const u32 *)v1, d_cam_position, XSEG, pend + maxg, beam / 64u,
				   (const float4 *)ctx->sray.p, xcap, report, (const u32 *)status, (const unsigned long long *)wcnt, xslices, xw0,
				   x_sieve > 64u ? 64u : x_sieve, xnsieve);
	ugrt_prof_end(ctx, UGRT_ST_TRACE_SHADOW);
	UGRT_HIP(hipGetLastError());
#ifdef UGRT_SHADOW_TIMELINE
	if (tlbuf) {
		UGRT_HIP(hipStreamSynchronize(st));
		unsigned long long *h = (unsigned long long *)malloc((size_t)xwaves * 16), *none = nullptr;
		UGRT_HIP(hipMemcpy(h, tlbuf, (size_t)xwaves * 16, hipMemcpyDeviceToHost));
		UGRT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_shadow_tl), &none, sizeof none));
		FILE *f = fopen(getenv("UGRT_SHADOW_TIMELINE_FILE"), "wb");
		if (f) {
			const unsigned long long hdr[4] = { xwaves, xw0, xnsieve, x_run_log2 };
			fwrite(hdr, 8, 4, f);
			fwrite(h, 16, xwaves, f);
			fclose(f);
		}
		free(h);
		(void)hipFree(tlbuf);
	}
#endif
	// (the pass's report -- {pairs, beams} as found in the asynchronous form: what the next pass is sized by; the status
	// word; the work counters of ugrt_stats_get -- is in the pinned host words when the stream has got this far: the
	// compaction and the exact pass write it there themselves)
	return UGRT_OK;
}

// work counters of the primary tracer's last counting launch (UGRT_FLAG_COUNT_WORK), in the order of the PS_* enum
extern "C" int ugrt_stats_primary(ugrt_ctx *ctx, unsigned long long *stats, int n)
{
	if (!ctx || !stats || n < 0)
		return ugrt_fail(UGRT_EINVAL, "stats_primary: bad argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	for (int i = 0; i < n; i++)
		stats[i] = i < UGRT_PRIMARY_STATS ? ctx->primary_stats[i] : 0ull;
	return UGRT_OK;
}
