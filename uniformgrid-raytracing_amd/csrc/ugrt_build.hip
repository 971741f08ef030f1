// ugrt_build.hip -- triangle -> cell binning (grid construction).
//
// Reference pipeline (frustum_grid.h:210-366, same for :368-532):
//   DSKernel -> [D2H + host z loop] -> cudppScan -> [D2H R] -> 4x cudaMalloc ->
//   SlabKernel -> DSFillkernel -> cudppPlan + cudppSort(32 bit) -> do_scan_dump ->
//   cudppPlan + cudppCompact -> [D2H] -> set_as_zero -> create_histogram -> cudppScan
// Here: count (stores the cell range) -> rocPRIM inclusive scan -> [D2H R, 4 B]
//   -> fill (one thread per REF, coalesced, no recomputation of the bbox) ->
//   rocPRIM radix sort on ceil(log2 C) bits -> one boundary kernel -> one
//   per-cell kernel -> rocPRIM exclusive scan.  Outputs are identical arrays:
//   value[R], key[R], span[C], offset[C].
// Triangles whose cell range is the WHOLE grid (the reference clamps the screen box of a triangle that
//   straddles the eye plane to the full screen, SURVEY.md Q9: 325 of 1 M triangles make 10.3 M of the
//   12.2 M references of the bench scene) are not pushed through fill + sort: they are kept as a short
//   sorted id list and merged into every cell's run after the sort of the remaining references.  The
//   output arrays are the same, element for element.
// With NUM_SLABS = 1 the z-slab stage (SlabKernel, grid_kernel.cu:334, and the
// host zMin/zMax loop, frustum_grid.h:221-241) always yields slab 0 and is
// dropped.
#include "ugrt_dev.h"
#include "ugrt_rs_hist.h"
#include "ugrt_scan.h"

// cell range of one triangle: {x0 | x1 << 16, y0 | y1 << 16, z0 | z1 << 16}
struct Rng {
	u32 x, y, z;
};

#define BUILD_THREADS 256

// Triangle records for the tracers: {v0, e1 = v1 - v0, e2 = v2 - v0, 0,0,0}, the same
// subtractions the reference does while staging (trace_kernel.cu:159-169, light_kernel.cu:131-142),
// done once per build instead of once per (cell, triangle) reference.
__global__ __launch_bounds__(BUILD_THREADS) void k_tri_records(const int *__restrict__ faces,
								const float *__restrict__ verts, int F,
								float4 *__restrict__ rec, u32 *__restrict__ wcount)
{
	int f = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (f == 0)
		*wcount = 0; // the count kernel that follows appends the wide triangles
	if (f >= F)
		return;
	int i1 = 3 * faces[f * 3 + 0], i2 = 3 * faces[f * 3 + 1], i3 = 3 * faces[f * 3 + 2];
	float v0x = verts[i1], v0y = verts[i1 + 1], v0z = verts[i1 + 2];
	float e1x = verts[i2] - v0x, e1y = verts[i2 + 1] - v0y, e1z = verts[i2 + 2] - v0z;
	float e2x = verts[i3] - v0x, e2y = verts[i3 + 1] - v0y, e2z = verts[i3 + 2] - v0z;
	rec[f * 3 + 0] = make_float4(v0x, v0y, v0z, e1x);
	rec[f * 3 + 1] = make_float4(e1y, e1z, e2x, e2y);
	rec[f * 3 + 2] = make_float4(e2z, 0.0f, 0.0f, 0.0f);
}

// A triangle that covers all `full` active cells goes to the wide list and contributes no references
// to fill + sort (one atomic per wave that holds any).
__device__ __forceinline__ u32 d_split_wide(int f, u32 size, u32 full, u32 *__restrict__ wide,
					     u32 *__restrict__ wcount)
{
	const bool w = size == full && full > 1u;
	const unsigned long long mask = __ballot(w);
	if (mask == 0ull)
		return size;
	const int leader = (int)__builtin_ctzll(mask);
	const int lane = (int)(threadIdx.x & 63u);
	u32 base = 0;
	if (lane == leader)
		base = atomicAdd(wcount, (u32)__popcll(mask));
	base = __shfl(base, leader);
	if (w) {
		wide[base + (u32)__popcll(mask & ((1ull << lane) - 1ull))] = (u32)f;
		return 0u;
	}
	return size;
}

// DSKernel, grid_kernel.cu:164-243 (+ the band clamp of the multi-GPU split)
__global__ __launch_bounds__(BUILD_THREADS) void k_count_persp(CamBlock cam, const int *__restrict__ faces,
								const float *__restrict__ verts, int F,
								int gy_lo, int gy_hi, Rng *__restrict__ rng,
								u32 *__restrict__ sizes, u32 *__restrict__ wide,
								u32 *__restrict__ wcount, float *__restrict__ projz)
{
	int f = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (f >= F)
		f = F - 1; // keeps the wave whole for d_split_wide; the duplicate is dropped below
	const bool dup = (int)(blockIdx.x * BUILD_THREADS + threadIdx.x) >= F;
	int i1 = 3 * faces[f * 3 + 0], i2 = 3 * faces[f * 3 + 1], i3 = 3 * faces[f * 3 + 2];
	float v1[3], v2[3], v3[3];
	d_transformed_vertex(cam, verts[i1], verts[i1 + 1], verts[i1 + 2], v1);
	d_transformed_vertex(cam, verts[i2], verts[i2 + 1], verts[i2 + 2], v2);
	d_transformed_vertex(cam, verts[i3], verts[i3 + 1], verts[i3 + 2], v3);
	float xmin = d_min3(v1[0], v2[0], v3[0]);
	float ymin = d_min3(v1[1], v2[1], v3[1]);
	float xmax = d_max3(v1[0], v2[0], v3[0]);
	float ymax = d_max3(v1[1], v2[1], v3[1]);
	int nbx = cam.nbx, nby = cam.nby;
	int gxmin = ugrt_floor2i(((xmin + 1.0f) / 2.0f) * (float)nbx);
	int gymin = ugrt_floor2i(((ymin + 1.0f) / 2.0f) * (float)nby);
	int gxmax = ugrt_floor2i(((xmax + 1.0f) / 2.0f) * (float)nbx);
	int gymax = ugrt_floor2i(((ymax + 1.0f) / 2.0f) * (float)nby);
	gxmin = d_clampi(gxmin, 0, nbx - 1);
	gymin = d_clampi(gymin, 0, nby - 1);
	gxmax = d_clampi(gxmax, 0, nbx - 1);
	gymax = d_clampi(gymax, 0, nby - 1);
	u32 size = 0;
	if (!(gymax < gy_lo || gymin >= gy_hi)) {
		gymin = gymin < gy_lo ? gy_lo : gymin;
		gymax = gymax > gy_hi - 1 ? gy_hi - 1 : gymax;
		size = (u32)((gxmax - gxmin + 1) * (gymax - gymin + 1));
	}
	Rng r;
	r.x = (u32)gxmin | ((u32)gxmax << 16);
	r.y = (u32)gymin | ((u32)gymax << 16);
	r.z = 0;
	// (with z-slabs a triangle lies in ONE slab of every cell it covers: no wide list then)
	size = d_split_wide(f, dup ? 0u : size, projz ? 0u : (u32)(nbx * (gy_hi - gy_lo)), wide, wcount);
	if (dup)
		return;
	rng[f] = r;
	sizes[f] = size;
	if (projz)
		projz[f] = d_min3(v1[2], v2[2], v3[2]); // projCoordZ, grid_kernel.cu:203,212
}

// DS_spherical_Kernel, grid_kernel.cu:481-659
__global__ __launch_bounds__(BUILD_THREADS) void k_count_sph(CamBlock cam, const int *__restrict__ faces,
							      const float *__restrict__ verts, int F, int lnbx,
							      int lnby, float xM, float yM, Rng *__restrict__ rng,
							      u32 *__restrict__ sizes, u32 *__restrict__ wide,
							      u32 *__restrict__ wcount, float *__restrict__ projz, int f_lo, int f_hi)
{
	int f = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (f >= F)
		f = F - 1;
	// triangles outside [f_lo, f_hi) belong to another rank's shard of the build: no references here
	const bool dup = (int)(blockIdx.x * BUILD_THREADS + threadIdx.x) >= F;
	const bool outside = f < f_lo || f >= f_hi;
	int blx[3], bly[3];
	float rad[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		int idx = 3 * faces[f * 3 + k];
		float point[3];
		point[0] = verts[idx + 0] - cam.cc[0];
		point[1] = verts[idx + 1] - cam.cc[1];
		point[2] = verts[idx + 2] - cam.cc[2];
		float radius = d_magnitude(point);
		rad[k] = radius;
		point[0] /= radius;
		point[1] /= radius;
		point[2] /= radius;
		blx[k] = (int)d_effective_x(cam, point, xM, lnbx / 2);
		bly[k] = (int)d_effective_y(cam, point, yM, lnby / 2);
	}
	int gxmin = d_clampi(d_imin3(blx[0], blx[1], blx[2]), 0, lnbx - 1);
	int gymin = d_clampi(d_imin3(bly[0], bly[1], bly[2]), 0, lnby - 1);
	int gxmax = d_clampi(d_imax3(blx[0], blx[1], blx[2]), 0, lnbx - 1);
	int gymax = d_clampi(d_imax3(bly[0], bly[1], bly[2]), 0, lnby - 1);
	Rng r;
	r.x = (u32)gxmin | ((u32)gxmax << 16);
	r.y = (u32)gymin | ((u32)gymax << 16);
	r.z = 0;
	u32 size = outside ? 0u : (u32)((gxmax - gxmin + 1) * (gymax - gymin + 1));
	size = d_split_wide(f, dup ? 0u : size, projz ? 0u : (u32)(lnbx * lnby), wide, wcount);
	if (dup)
		return;
	rng[f] = r;
	sizes[f] = size;
	if (projz)
		projz[f] = d_min3(rad[0], rad[1], rad[2]); // grid_kernel.cu:617-618,651
}

// ---------------------------------------------------------------------------
// z-slabs (NUM_SLABS > 1): the host loop over h_projCoordZ (frustum_grid.h:221-241, :384-404) as a reduction,
// then SlabKernel (grid_kernel.cu:334-352).  zr[0], zr[1] hold zMin, zMax as order-preserving integers.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int d_ordered_f(float f)
{
	const int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float d_unordered_f(int b) { return __int_as_float(b ^ ((b >> 31) & 0x7FFFFFFF)); }

__global__ void k_zrange_init(int *__restrict__ zr, float zmin_init, float zmax_init)
{
	zr[0] = d_ordered_f(zmin_init);
	zr[1] = d_ordered_f(zmax_init);
}

// zMin = smallest value >= 0 (or the initial value), zMax = largest value (NaNs never win a comparison of the
// host loop, so they are left out)
__global__ __launch_bounds__(BUILD_THREADS) void k_zrange(const float *__restrict__ projz, int F, int *__restrict__ zr)
{
	int lo = 0x7FFFFFFF, hi = (int)0x80000000;
	for (int f = blockIdx.x * BUILD_THREADS + threadIdx.x; f < F; f += gridDim.x * BUILD_THREADS) {
		const float z = projz[f];
		if (z >= 0.0f) {
			const int o = d_ordered_f(z);
			lo = o < lo ? o : lo;
		}
		if (z == z) {
			const int o = d_ordered_f(z);
			hi = o > hi ? o : hi;
		}
	}
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) {
		const int ol = __shfl_xor(lo, m), oh = __shfl_xor(hi, m);
		lo = ol < lo ? ol : lo;
		hi = oh > hi ? oh : hi;
	}
	if ((threadIdx.x & 63) == 0) {
		if (lo != 0x7FFFFFFF)
			atomicMin(&zr[0], lo);
		if (hi != (int)0x80000000)
			atomicMax(&zr[1], hi);
	}
}

// SlabKernel; a triangle with a negative depth keeps slab 0 (the reference leaves its entry uninitialised,
// SURVEY.md Q20).  The slab goes into the z range of the triangle's cell box: the fill then writes
// key = (gx*ny + gy)*slabs + slab (grid_kernel.cu:310,322).
__global__ __launch_bounds__(BUILD_THREADS) void k_slab(const float *__restrict__ projz, int F, int slabs,
							 const int *__restrict__ zr, Rng *__restrict__ rng)
{
	const int f = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (f >= F)
		return;
	const float zMin = d_unordered_f(zr[0]), zMax = d_unordered_f(zr[1]);
	const float pCoord = projz[f];
	u32 binID = 0;
	if (pCoord >= 0.0f) {
		binID = ugrt_f2u((float)slabs * (float)((pCoord - zMin) / (zMax - zMin)));
		if (binID >= (u32)slabs)
			binID = (u32)slabs - 1u;
	}
	rng[f].z = binID | (binID << 16);
}

// span/offset of a light cell's slabs taken together: a cell's runs are adjacent in the sorted lists
__global__ __launch_bounds__(BUILD_THREADS) void k_slab_union(const u32 *__restrict__ span, const u32 *__restrict__ offset,
							       u32 C, u32 slabs, u32 *__restrict__ uspan,
							       u32 *__restrict__ uoffset)
{
	const u32 c = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (c >= C)
		return;
	const u32 first = c * slabs, last = first + slabs - 1u;
	uoffset[c] = offset[first];
	uspan[c] = offset[last] + span[last] - offset[first];
}

int ugrt_slab_union(ugrt_ctx *ctx, const u32 *d_span, const u32 *d_offset, u32 C, u32 slabs, const u32 **uspan,
		    const u32 **uoffset)
{
	Grid &G = ctx->grid[UGRT_GRID_SPHERICAL];
	int rc = ugrt_buf_reserve(ctx, G.uspan, (size_t)C * 8);
	if (rc)
		return rc;
	u32 *us = (u32 *)G.uspan.p, *uo = us + C;
	hipLaunchKernelGGL(k_slab_union, dim3((C + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0, ctx->stream,
			   d_span, d_offset, C, slabs, us, uo);
	UGRT_HIP(hipGetLastError());
	*uspan = us;
	*uoffset = uo;
	return UGRT_OK;
}

// the slab stage between the count kernel and the scan/fill
static int build_slabs(ugrt_ctx *ctx, Grid &G, int F, float zmin_init, float zmax_init)
{
	int *zr = (int *)((float *)G.projz.p + F);
	hipLaunchKernelGGL(k_zrange_init, dim3(1), dim3(1), 0, ctx->stream, zr, zmin_init, zmax_init);
	int blocks = (F + BUILD_THREADS - 1) / BUILD_THREADS;
	hipLaunchKernelGGL(k_zrange, dim3(blocks < 1024 ? blocks : 1024), dim3(BUILD_THREADS), 0, ctx->stream,
			   (const float *)G.projz.p, F, zr);
	hipLaunchKernelGGL(k_slab, dim3(blocks), dim3(BUILD_THREADS), 0, ctx->stream, (const float *)G.projz.p, F,
			   ctx->cfg.slabs, (const int *)zr, (Rng *)G.rng.p);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

struct UGrid {
	float lo[3], cs[3], inv[3];
	int dims[3];
};

__device__ __forceinline__ int d_ucell(const UGrid &g, int k, float p)
{
	int c = ugrt_floor2i((p - g.lo[k]) * g.inv[k]);
	return d_clampi(c, 0, g.dims[k] - 1);
}

// uniform grid: world-space bbox -> 3-D cell range
__global__ __launch_bounds__(BUILD_THREADS) void k_count_uniform(UGrid g, const int *__restrict__ faces,
								  const float *__restrict__ verts, int F,
								  Rng *__restrict__ rng, u32 *__restrict__ sizes,
								  u32 *__restrict__ wide, u32 *__restrict__ wcount, int f_lo, int f_hi)
{
	int f = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (f >= F)
		f = F - 1;
	const bool dup = (int)(blockIdx.x * BUILD_THREADS + threadIdx.x) >= F;
	const bool outside = f < f_lo || f >= f_hi;
	int i1 = 3 * faces[f * 3 + 0], i2 = 3 * faces[f * 3 + 1], i3 = 3 * faces[f * 3 + 2];
	u32 packed[3], size = 1;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		float a = verts[i1 + k], b = verts[i2 + k], c = verts[i3 + k];
		int lo = d_ucell(g, k, d_min3(a, b, c));
		int hi = d_ucell(g, k, d_max3(a, b, c));
		packed[k] = (u32)lo | ((u32)hi << 16);
		size *= (u32)(hi - lo + 1);
	}
	Rng r;
	r.x = packed[0];
	r.y = packed[1];
	r.z = packed[2];
	size = d_split_wide(f, (dup || outside) ? 0u : size, (u32)g.dims[0] * (u32)g.dims[1] * (u32)g.dims[2], wide, wcount);
	if (dup)
		return;
	rng[f] = r;
	sizes[f] = size;
}

// DSFillkernel (grid_kernel.cu:245-332) inverted: one thread per reference.
// ref r belongs to the triangle f with scan[f-1] <= r < scan[f]; its position
// inside the triangle's box is x-major, then y, then z (:318-325), and
// key = ((gx*ny + gy)*nz + gz).  Writes are fully coalesced and the work per
// thread no longer depends on how many cells a triangle covers (border cells
// collect thousands: SURVEY.md Q9).
// The search for f is split: k_fill_parts finds, for every workgroup of k_fill, the triangle of its first
// reference (one 20-step search per 256 references); the workgroup then holds its short run of the scan
// in LDS and its threads search that (a 20-step search of the global scan per REFERENCE diverges to 64
// cache lines per wave in its last steps and was bound by the address path, not by bytes).
#define FILL_LDS 1024
// (counts that the host does not know - an asynchronous build - are read from `rw` = {narrow references, wide
// triangles}, already checked against the capacities by k_fill_parts)
// (asynchronous build: `chk` carries the check -- the counts behind the scan against the
// capacities the launches were sized for; every thread derives the checked count itself, thread 0 publishes it)
struct BuildCheck {
	const u32 *scan_last; // {narrow references, wide triangles}; nullptr = the host knows the counts
	u32 capRn, capW;
	unsigned long long capR, active;
	u32 *rw, *status, *report;
};
__global__ __launch_bounds__(BUILD_THREADS) void k_fill_parts(const u32 *__restrict__ scan, int F, u32 R, u32 nparts,
							       u32 *__restrict__ parts, BuildCheck chk)
{
	const u32 b = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (chk.scan_last) {
		u32 Rn = chk.scan_last[0], W = chk.scan_last[1];
		const bool over = Rn > chk.capRn || W > chk.capW || (unsigned long long)Rn + chk.active * W > chk.capR;
		if (b == 0u) {
			chk.report[0] = Rn; // what the build needed: the next build's estimate
			chk.report[1] = W;
			if (over)
				atomicOr(chk.status, UGRT_STATUS_BUILD_OVERFLOW);
			chk.rw[0] = over ? 0u : Rn; // an overflowing build is emptied: nothing is written out of bounds
			chk.rw[1] = over ? 0u : W;
		}
		R = over ? 0u : Rn;
		nparts = (R + BUILD_THREADS - 1) / BUILD_THREADS;
	}
	if (b > nparts || R == 0u)
		return;
	u32 target = b * BUILD_THREADS; // first reference of workgroup b; parts[nparts] closes the last one
	if (target >= R)
		target = R - 1;
	int lo = 0, hi = F - 1; // smallest f with scan[f] > target
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (scan[mid] > target)
			hi = mid;
		else
			lo = mid + 1;
	}
	parts[b] = (u32)lo;
}

// (grid-stride over the workgroup-sized parts: a bounded number of workgroups -- fewer when they also count the keys'
// first digit for the sort that follows, ugrt_rs_hist.h: each then ends with up to 256 global adds)
#define FILL_MAX_BLOCKS 8192u
#define FILL_MAX_BLOCKS_COUNTING 4096u
__global__ __launch_bounds__(BUILD_THREADS) void k_fill(const u32 *__restrict__ scan, const Rng *__restrict__ rng,
							 const u32 *__restrict__ parts, u32 R, int ny, int nz,
							 u32 *__restrict__ keys, u32 *__restrict__ vals,
							 u32 *__restrict__ zero, u32 nzero, const u32 *__restrict__ rw, RsFirst hs)
{
	__shared__ u32 s_scan[FILL_LDS];
	__shared__ u32 s_rsh[RS_BINS * RS_PRIV];
	d_rs_zero(s_rsh, hs); // (the barrier at the head of the loop stands between this and the first count)
	// the per-cell words the boundary kernels start from (run ends, run starts, cells_used) are cleared here
	// instead of by a fill of their own
	for (u32 i = blockIdx.x * BUILD_THREADS + threadIdx.x; i < nzero; i += gridDim.x * BUILD_THREADS)
		zero[i] = 0;
	if (rw)
		R = rw[0];
	const u32 nparts = (R + BUILD_THREADS - 1) / BUILD_THREADS;
	for (u32 part = blockIdx.x; part < nparts; part += gridDim.x) {
		const u32 r = part * BUILD_THREADS + threadIdx.x;
		const int f_first = (int)parts[part], f_last = (int)parts[part + 1];
		const int nrun = f_last - f_first + 1;
		const bool in_lds = nrun <= FILL_LDS;
		__syncthreads(); // the previous part's searches are done with s_scan
		if (in_lds)
			for (int i = threadIdx.x; i < nrun; i += BUILD_THREADS)
				s_scan[i] = scan[f_first + i];
		__syncthreads();
		if (r < R) {
			int lo = f_first, hi = f_last; // smallest f with scan[f] > r
			if (in_lds) {
				while (lo < hi) {
					int mid = (lo + hi) >> 1;
					if (s_scan[mid - f_first] > r)
						hi = mid;
					else
						lo = mid + 1;
				}
			} else {
				while (lo < hi) {
					int mid = (lo + hi) >> 1;
					if (scan[mid] > r)
						hi = mid;
					else
						lo = mid + 1;
				}
			}
			const int f = lo;
			const u32 base = (in_lds && f > f_first) ? s_scan[f - 1 - f_first] : (f ? scan[f - 1] : 0);
			const u32 local = r - base;
			const Rng g = rng[f];
			u32 x0 = g.x & 0xFFFFu, y0 = g.y & 0xFFFFu, y1 = g.y >> 16, z0 = g.z & 0xFFFFu, z1 = g.z >> 16;
			u32 sy = y1 - y0 + 1, sz = z1 - z0 + 1;
			u32 k = local % sz;
			u32 ij = local / sz;
			u32 j = ij % sy;
			u32 i = ij / sy;
			const u32 key = ((x0 + i) * (u32)ny + (y0 + j)) * (u32)nz + (z0 + k);
			keys[r] = key;
			vals[r] = (u32)f;
			if (hs.hist)
				d_rs_count(s_rsh, key & 0xFFu, true);
		}
	}
	__syncthreads();
	d_rs_flush(s_rsh, hs);
}

// do_scan_dump + cudppCompact + create_histogram (misc_kernel.cu:4-60,
// frustum_grid.h:334) in one pass over the sorted keys: run heads record the
// run start, run tails the run end; used[0] counts the runs.
__global__ __launch_bounds__(BUILD_THREADS) void k_bounds(const u32 *__restrict__ keys, u32 R,
							   u32 *__restrict__ cstart, u32 *__restrict__ cend,
							   const u32 *__restrict__ rw)
{
	if (rw)
		R = rw[0];
	u32 i = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (i >= R)
		return;
	u32 k = keys[i];
	bool head = (i == 0) || (keys[i - 1] != k);
	bool tail = (i == R - 1) || (keys[i + 1] != k);
	if (head)
		cstart[k] = i;
	if (tail)
		cend[k] = i + 1;
}

// span = run length (0 for cells without a run: set_as_zero, misc_kernel.cu:26); also counts the
// occupied cells ("Number of actual cells", frustum_grid.h:337) with one atomic per block
// A cell is "active" when its y index lies in [ylo, yhi] (the whole grid, or this rank's band of the
// screen grid); every active cell also holds the W wide triangles.
struct WideBox {
	u32 W, ny, nz, ylo, yhi;
	const u32 *rw; // asynchronous build: {narrow references, wide triangles} on the device (W above is unused then)
};
__device__ __forceinline__ u32 d_wide_count(const WideBox &wb) { return wb.rw ? wb.rw[1] : wb.W; }

__device__ __forceinline__ bool d_cell_active(const WideBox &wb, u32 c)
{
	u32 y = (c / wb.nz) % wb.ny;
	return y >= wb.ylo && y <= wb.yhi;
}

// The cells' spans are formed INSIDE the scan that turns them into offsets (ugrt_scan.h: a load functor instead of
// a kernel of its own): span = run end - run start (+ the wide triangles of an active cell), written back for the
// tracers, the occupied cells counted with one atomic per tile of 4096 cells.
struct SpanLoad {
	const u32 *cstart;
	u32 *span_io; // holds the run ends on entry
	u32 *used;
	WideBox wb;
	__device__ __forceinline__ void operator()(u32 base, u32 n, u32 (&v)[SC_ITEMS]) const
	{
		__shared__ u32 s_used[SC_WAVES];
		const u32 W = d_wide_count(wb);
		u32 mine = 0;
#pragma unroll
		for (int i = 0; i < SC_ITEMS; i++) {
			const u32 c = base + (u32)i;
			u32 sp = 0;
			if (c < n) {
				sp = span_io[c] - cstart[c];
				if (W && d_cell_active(wb, c))
					sp += W;
				span_io[c] = sp;
				mine += sp != 0u ? 1u : 0u;
			}
			v[i] = sp;
		}
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1)
			mine += (u32)__shfl_xor((int)mine, m);
		if ((threadIdx.x & 63u) == 0u)
			s_used[threadIdx.x >> 6] = mine;
		__syncthreads();
		if (threadIdx.x == 0) {
			u32 tot = 0;
			for (int w = 0; w < SC_WAVES; w++)
				tot += s_used[w];
			if (tot)
				atomicAdd(used, tot);
		}
	}
};
// what the last tile of that scan does: the tail of an asynchronous build's report (cells used, the status word at
// its end) and the wide-triangle counter back to zero for the next build of this grid
struct SpanTail {
	static constexpr bool active = true;
	u32 *report_tail;       // nullptr: no report
	const u32 *used, *status;
	u32 *zero;              // nullptr: nothing to clear
	__device__ __forceinline__ void operator()() const
	{
		if (report_tail) {
			report_tail[0] = __hip_atomic_load(used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			report_tail[1] = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (zero)
			*zero = 0u;
	}
};

// ascending order of the (few) wide triangle ids: rank = number of smaller ids
__global__ __launch_bounds__(BUILD_THREADS) void k_wide_rank(const u32 *__restrict__ wl, u32 W,
							      u32 *__restrict__ sorted, const u32 *__restrict__ rw)
{
	if (rw)
		W = rw[1];
	u32 i = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (i >= W)
		return;
	u32 v = wl[i], r = 0;
	for (u32 j = 0; j < W; j++)
		r += wl[j] < v ? 1u : 0u;
	sorted[r] = v;
}

// number of elements of the ascending list a[0..n) that are smaller than v; the trip count depends on n
// only (uniform across the wave), pow2 = smallest power of two >= n
template <typename P>
__device__ __forceinline__ u32 d_lower_bound(P a, u32 n, u32 pow2, u32 v)
{
	u32 pos = 0;
	for (u32 step = pow2; step; step >>= 1) {
		u32 t = pos + step;
		if (t <= n && a[t - 1] < v)
			pos = t;
	}
	return pos;
}

__device__ __forceinline__ u32 d_pow2_ge(u32 n)
{
	return n <= 1u ? n : 1u << (32 - __builtin_clz(n - 1u));
}

// Merge of the sorted narrow references with the sorted wide list.  Within a cell both are ascending id
// lists without common elements: element i of one lands at i + (smaller elements of the other).
// Narrow side: one thread per reference (a cell may hold thousands).
#define MERGE_LDS 1024
__global__ __launch_bounds__(BUILD_THREADS) void k_merge_narrow(const u32 *__restrict__ nkeys,
								 const u32 *__restrict__ nvals, u32 Rn,
								 const u32 *__restrict__ cstart,
								 const u32 *__restrict__ offset,
								 const u32 *__restrict__ wl, WideBox wb,
								 u32 *__restrict__ okeys, u32 *__restrict__ ovals)
{
	__shared__ u32 s_w[MERGE_LDS];
	const u32 W = d_wide_count(wb);
	if (wb.rw)
		Rn = wb.rw[0];
	if (blockIdx.x * BUILD_THREADS >= Rn)
		return;
	const bool w_lds = W <= MERGE_LDS;
	if (w_lds)
		for (u32 k = threadIdx.x; k < W; k += BUILD_THREADS)
			s_w[k] = wl[k];
	__syncthreads();
	const u32 j = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (j >= Rn)
		return;
	const u32 c = nkeys[j], id = nvals[j];
	const u32 wp2 = d_pow2_ge(W);
	u32 lb = 0;
	if (W && d_cell_active(wb, c))
		lb = w_lds ? d_lower_bound(s_w, W, wp2, id) : d_lower_bound(wl, W, wp2, id);
	const u32 pos = offset[c] + (j - cstart[c]) + lb;
	ovals[pos] = id;
	okeys[pos] = c;
}

// Wide side: one wave per active cell, the cell's narrow run is searched in LDS when it fits.
__global__ __launch_bounds__(64) void k_merge_wide(const u32 *__restrict__ nvals, const u32 *__restrict__ cstart,
						    const u32 *__restrict__ span, const u32 *__restrict__ offset,
						    const u32 *__restrict__ wl, u32 C, WideBox wb,
						    u32 *__restrict__ okeys, u32 *__restrict__ ovals,
						    u32 *__restrict__ report_tail, const u32 *__restrict__ used,
						    const u32 *__restrict__ status)
{
	__shared__ u32 s_n[MERGE_LDS];
	const u32 lane = threadIdx.x;
	// the last kernel of an asynchronous build completes its report (one copy then takes it to the host)
	if (report_tail && blockIdx.x == 0 && lane == 0) {
		report_tail[0] = *used;
		report_tail[1] = *status;
	}
	const u32 W = d_wide_count(wb);
	if (W == 0u)
		return;
	for (u32 c = blockIdx.x; c < C; c += gridDim.x) {
		if (!d_cell_active(wb, c))
			continue;
		const u32 ns = span[c] - W;
		const u32 *nv = nvals + cstart[c];
		const u32 out = offset[c];
		const bool n_lds = ns <= MERGE_LDS;
		const u32 np2 = d_pow2_ge(ns);
		__syncthreads(); // the previous cell's s_n is no longer read
		if (n_lds)
			for (u32 i = lane; i < ns; i += 64u)
				s_n[i] = nv[i];
		__syncthreads();
		for (u32 k = lane; k < W; k += 64u) {
			const u32 id = wl[k];
			const u32 lb = n_lds ? d_lower_bound(s_n, ns, np2, id) : d_lower_bound(nv, ns, np2, id);
			const u32 pos = out + k + lb;
			ovals[pos] = id;
			okeys[pos] = c;
		}
	}
}

// number of wide triangles of the running build: kept right behind the scan of the sizes, so that
// total_refs (scan[F-1]) and this count come back in one 8-byte copy
static inline u32 *wide_counter(Grid &G, int F) { return (u32 *)G.scan.p + F; }

static int bits_for(u32 C)
{
	int b = 1;
	while (b < 32 && (1ull << b) < (unsigned long long)C)
		b++;
	return b;
}

// ---------------------------------------------------------------------------
// Asynchronous build (option "async_build"): no read-back, the host never waits.  The counts the reference reads
// back (total_triangles, frustum_grid.h:254) stay on the device; launches and buffers are sized by what the same
// grid needed in the previous build plus a margin, and every kernel takes the real counts from device memory.
// k_fill_parts compares them with the capacities first: when they do not fit, the build is emptied (nothing is
// written out of bounds) and a status bit is raised, which the host sees at its next synchronisation
// (UGRT_EOVERFLOW; the following build of the grid runs synchronously and sizes the buffers exactly).
// ---------------------------------------------------------------------------
// (three parts: everything before the sort, the sort, everything behind it -- so that the builds of a batch,
// ugrt_grid_build_batch_begin / _end, can share the launches of their sorts)
static int build_async_begin(ugrt_ctx *ctx, AsyncBuild &b)
{
	Grid &G = *b.G;
	const int F = b.F, ny = b.ny, nz = b.nz, ylo = b.ylo, yhi = b.yhi;
	const u32 C = b.C;
	hipStream_t st = ctx->stream;
	const int gi = (int)(&G - ctx->grid);
	int rc;
	G.valid = false;
	G.C = C;
	G.F = F;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_SCAN);
	rc = ugrt_prim_inclusive_scan(ctx, (const u32 *)G.sizes.p, (u32 *)G.scan.p, (size_t)F);
	ugrt_prof_end(ctx, UGRT_ST_BUILD_SCAN);
	if (rc)
		return rc;
	// capacities: the last known need of this grid + a quarter; the wide list goes through the rank kernel
	const unsigned long long active = (unsigned long long)(C / ((u32)ny * (u32)nz)) * (u32)(yhi - ylo + 1) * (u32)nz;
	const u32 estRn = G.est_rn, estW = G.est_w;
	// A grid that had no wide triangle in its last build (the light grid and the uniform grid of the bench scene) is
	// built for none: the sorted references are the final lists, so the rank kernel and the two merge kernels (for
	// such a grid: a copy of every reference) do not run.  A wide triangle that does turn up exceeds the capacity
	// like any other count: the build is emptied and flagged, and the next one expects it.
	const bool no_wide = estW == 0u;
	u32 capRn = estRn + estRn / 4u + 65536u, capW = no_wide ? 0u : estW + estW / 4u + 16u;
	if (capW > 4096u)
		capW = 4096u;
	unsigned long long capR = (unsigned long long)capRn + active * capW;
	if (capR > 0xFFFFFFF0ull)
		capR = 0xFFFFFFF0ull;
	// never shrink below what the buffers already hold: a grow-only arena
	if ((size_t)capRn * 4 < G.key[1].cap && G.key[1].cap / 4 < 0xFFFFFFF0ull)
		capRn = (u32)(G.key[1].cap / 4);
	if (capR * 4 < G.key[0].cap)
		capR = G.key[0].cap / 4 < 0xFFFFFFF0ull ? G.key[0].cap / 4 : 0xFFFFFFF0ull;
	if ((rc = ugrt_buf_reserve(ctx, G.key[0], (size_t)capR * 4)) || (rc = ugrt_buf_reserve(ctx, G.val[0], (size_t)capR * 4)) ||
	    (rc = ugrt_buf_reserve(ctx, G.key[1], (size_t)capRn * 4)) || (rc = ugrt_buf_reserve(ctx, G.val[1], (size_t)capRn * 4)) ||
	    (rc = ugrt_buf_reserve(ctx, G.parts, ((size_t)capRn / BUILD_THREADS + 2) * 4)) ||
	    (rc = ugrt_buf_reserve(ctx, G.span, (size_t)C * 8 + 16)) || (rc = ugrt_buf_reserve(ctx, G.offset, (size_t)C * 4)))
		return rc;
	// the capacities actually available (buffers may be larger than asked for)
	capRn = (u32)((G.key[1].cap < G.val[1].cap ? G.key[1].cap : G.val[1].cap) / 4 < 0xFFFFFFF0ull
			      ? (G.key[1].cap < G.val[1].cap ? G.key[1].cap : G.val[1].cap) / 4 : 0xFFFFFFF0ull);
	if ((size_t)capRn / BUILD_THREADS + 2 > G.parts.cap / 4)
		capRn = (u32)((G.parts.cap / 4 - 2) * BUILD_THREADS);
	capR = (G.key[0].cap < G.val[0].cap ? G.key[0].cap : G.val[0].cap) / 4;
	if (capR > 0xFFFFFFF0ull)
		capR = 0xFFFFFFF0ull;
	u32 *k0 = (u32 *)G.key[0].p, *v0 = (u32 *)G.val[0].p;
	u32 *rw = ctx->d_small + UGRT_DSMALL_RW + 2 * gi, *status = ctx->d_small + UGRT_DSMALL_STATUS;
	// (written by the build's kernels straight into the pinned host words: no copy behind the build)
	u32 *report = ctx->h_pinned + UGRT_PIN_REPORT + 4 * gi;
	// launch sizes: the estimate plus the margin (every kernel stops at the real count).  The check below is made
	// against THIS size, not against the (larger, grow-only) buffers: fill, sort, bounds and merge run one thread per
	// reference of the launch, so a count between the two would leave references unfilled and unsorted.
	const u32 launchRn = estRn + estRn / 4u + 65536u < capRn ? estRn + estRn / 4u + 65536u : capRn;
	BuildCheck chk = { (const u32 *)G.scan.p + (F - 1), launchRn, capW, capR, active, rw, status, report };
	const u32 nparts = (launchRn + BUILD_THREADS - 1) / BUILD_THREADS;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_FILL);
	hipLaunchKernelGGL(k_fill_parts, dim3((nparts + BUILD_THREADS) / BUILD_THREADS), dim3(BUILD_THREADS), 0, st,
			   (const u32 *)G.scan.p, F, 0u, 0u, (u32 *)G.parts.p, chk);
	// the fill counts the first digit of its keys for the build's own sort, unless that sort shares its launches with
	// another build's (ugrt_grid_build_batch_begin: the lists of a batch are counted by one histogram kernel)
	RsFirst hs = { nullptr };
	if (ctx->opt[UGRT_OPT_SORT_LIBRARY] != 1 && !ctx->batch_open && (rc = ugrt_sort_first_digit(ctx, &hs)))
		return rc;
	b.prehist = hs.hist != nullptr;
	const u32 fill_blocks = hs.hist ? FILL_MAX_BLOCKS_COUNTING : FILL_MAX_BLOCKS;
	hipLaunchKernelGGL(k_fill, dim3(nparts < fill_blocks ? nparts : fill_blocks), dim3(BUILD_THREADS), 0, st,
			   (const u32 *)G.scan.p, (const Rng *)G.rng.p, (const u32 *)G.parts.p, 0u, ny, nz, k0, v0, (u32 *)G.span.p,
			   2u * C + 1u, (const u32 *)rw, hs);
	ugrt_prof_end(ctx, UGRT_ST_BUILD_FILL);
	UGRT_HIP(hipGetLastError());
	b.launchRn = launchRn, b.capW = capW, b.capR = capR, b.active = active, b.no_wide = no_wide, b.nparts = nparts;
	return UGRT_OK;
}

// the sort of one or two begun builds, in shared launches
static int build_async_sort(ugrt_ctx *ctx, AsyncBuild *b, int n)
{
	int rc = UGRT_OK;
	const bool own_sort = ctx->opt[UGRT_OPT_SORT_LIBRARY] != 1;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_SORT);
	RsJob jobs[2];
	for (int i = 0; i < n; i++) {
		Grid &G = *b[i].G;
		const int gi = (int)(&G - ctx->grid);
		jobs[i] = RsJob{ (const u32 *)G.key[0].p, (const u32 *)G.val[0].p, (u32 *)G.key[1].p, (u32 *)G.val[1].p, b[i].launchRn,
				 bits_for(b[i].C), (const u32 *)(ctx->d_small + UGRT_DSMALL_RW + 2 * gi) };
	}
	if (own_sort) {
		rc = ugrt_sort_pairs_batch(ctx, jobs, n, n == 1 && b[0].prehist);
	} else {
		for (int i = 0; i < n && !rc; i++)
			rc = ugrt_prim_sort_pairs(ctx, jobs[i].kin, jobs[i].kout, jobs[i].vin, jobs[i].vout, jobs[i].n, jobs[i].end_bit, jobs[i].n_dev);
	}
	if (rc)
		return rc;
	for (int i = 0; i < n; i++) {
		Grid &G = *b[i].G;
		const int gi = (int)(&G - ctx->grid);
		if (!b[i].no_wide)
			hipLaunchKernelGGL(k_wide_rank, dim3((b[i].capW + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0, ctx->stream,
					   (const u32 *)G.wide.p, 0u, (u32 *)G.wide.p + b[i].F, (const u32 *)(ctx->d_small + UGRT_DSMALL_RW + 2 * gi));
	}
	ugrt_prof_end(ctx, UGRT_ST_BUILD_SORT);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

static int build_async_end(ugrt_ctx *ctx, AsyncBuild &b)
{
	Grid &G = *b.G;
	const int F = b.F;
	const u32 C = b.C;
	hipStream_t st = ctx->stream;
	const int gi = (int)(&G - ctx->grid);
	int rc;
	u32 *k0 = (u32 *)G.key[0].p, *k1 = (u32 *)G.key[1].p, *v0 = (u32 *)G.val[0].p, *v1 = (u32 *)G.val[1].p;
	u32 *wsorted = (u32 *)G.wide.p + F;
	u32 *rw = ctx->d_small + UGRT_DSMALL_RW + 2 * gi, *status = ctx->d_small + UGRT_DSMALL_STATUS;
	u32 *report = ctx->h_pinned + UGRT_PIN_REPORT + 4 * gi;
	const bool no_wide = b.no_wide;
	WideBox wb;
	wb.W = 0;
	wb.ny = (u32)b.ny;
	wb.nz = (u32)b.nz;
	wb.ylo = (u32)b.ylo;
	wb.yhi = (u32)b.yhi;
	wb.rw = rw;
	const u32 nparts = b.nparts;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_BOUNDS);
	u32 *cstart = (u32 *)G.span.p + C, *used = cstart + C;
	hipLaunchKernelGGL(k_bounds, dim3(nparts), dim3(BUILD_THREADS), 0, st, (const u32 *)k1, 0u, cstart, (u32 *)G.span.p,
			   (const u32 *)rw);
	// spans + offsets in one kernel; a grid without wide triangles ends its report there (with them: k_merge_wide)
	{
		SpanLoad sl = { (const u32 *)cstart, (u32 *)G.span.p, used, wb };
		SpanTail tl = { no_wide ? report + 2 : (u32 *)nullptr, (const u32 *)used, (const u32 *)status, wide_counter(G, F) };
		if ((rc = ugrt_scan_launch<false>(ctx, sl, (u32 *)G.offset.p, (size_t)C, tl)))
			return rc;
		G.wide_zeroed = wide_counter(G, F);
	}
	if (no_wide) {
		G.keys = k1;
		G.vals = v1;
	} else {
		// the merged lists go to key[0]/val[0]
		hipLaunchKernelGGL(k_merge_narrow, dim3(nparts), dim3(BUILD_THREADS), 0, st, (const u32 *)k1, (const u32 *)v1, 0u,
				   (const u32 *)cstart, (const u32 *)G.offset.p, (const u32 *)wsorted, wb, k0, v0);
		u32 blocks = C < 256u * 32u ? C : 256u * 32u;
		hipLaunchKernelGGL(k_merge_wide, dim3(blocks), dim3(64), 0, st, (const u32 *)v1, (const u32 *)cstart,
				   (const u32 *)G.span.p, (const u32 *)G.offset.p, (const u32 *)wsorted, C, wb, k0, v0, report + 2,
				   (const u32 *)used, (const u32 *)status);
		G.keys = k0;
		G.vals = v0;
	}
	UGRT_HIP(hipGetLastError());
	ugrt_prof_end(ctx, UGRT_ST_BUILD_BOUNDS);
	G.R = no_wide ? b.launchRn : (u32)b.capR; // an upper bound; the exact count is in the pinned report words once the stream has got here
	G.r_exact = false;
	G.active_cells = b.active;
	G.async_pending = true;
	G.valid = true;
	return UGRT_OK;
}

static int build_common_async(ugrt_ctx *ctx, Grid &G, int F, u32 C, int ny, int nz, int ylo, int yhi)
{
	AsyncBuild b = {};
	b.G = &G, b.F = F, b.C = C, b.ny = ny, b.nz = nz, b.ylo = ylo, b.yhi = yhi;
	int rc = build_async_begin(ctx, b);
	if (rc)
		return rc;
	// inside ugrt_grid_build_batch_begin / _end the build stops here: its sort shares its launches with the next build's
	if (ctx->batch_open && ctx->nbatch < 2) {
		ctx->batch[ctx->nbatch++] = b;
		return UGRT_OK;
	}
	if ((rc = build_async_sort(ctx, &b, 1)))
		return rc;
	return build_async_end(ctx, b);
}

// Two grid builds that depend on the geometry only (the light grid and the uniform grid of a frame) between
// ugrt_grid_build_batch_begin and _end: each ugrt_grid_build_* call in between enqueues its count, scan and fill and
// returns; _end sorts both reference lists in shared launches (one histogram kernel, one kernel per pass level) and
// completes the builds.  A build that cannot run in the asynchronous form (the first of a grid, after an overflow,
// slabs > 1) is simply built at once.  Until _end has returned the grids of the batch must not be used.
extern "C" int ugrt_grid_build_batch_begin(ugrt_ctx *ctx)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "grid_build_batch_begin: null context");
	if (ctx->batch_open)
		return ugrt_fail(UGRT_EINVAL, "grid_build_batch_begin: a batch is open already");
	ctx->batch_open = true;
	ctx->nbatch = 0;
	return UGRT_OK;
}

extern "C" int ugrt_grid_build_batch_end(ugrt_ctx *ctx)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "grid_build_batch_end: null context");
	if (!ctx->batch_open)
		return ugrt_fail(UGRT_EINVAL, "grid_build_batch_end: no batch is open");
	ctx->batch_open = false;
	const int n = ctx->nbatch;
	ctx->nbatch = 0;
	if (n == 0)
		return UGRT_OK;
	UGRT_HIP(hipSetDevice(ctx->device));
	int rc = build_async_sort(ctx, ctx->batch, n);
	for (int i = 0; i < n && !rc; i++)
		rc = build_async_end(ctx, ctx->batch[i]);
	return rc;
}

// shared tail of the three builders: sizes/rng/wide list are filled, ny/nz give the key layout,
// [ylo, yhi] the y range of the cells a wide triangle covers
static int build_common(ugrt_ctx *ctx, Grid &G, int F, u32 C, int ny, int nz, int ylo, int yhi)
{
	hipStream_t st = ctx->stream;
	int rc;
	const int gidx = (int)(&G - ctx->grid);
	if (G.async_pending) { // what the previous asynchronous build of this grid reported (possibly a frame old)
		G.est_rn = ctx->h_pinned[UGRT_PIN_REPORT + 4 * gidx];
		G.est_w = ctx->h_pinned[UGRT_PIN_REPORT + 4 * gidx + 1];
	}
	// asynchronous when asked for, when this grid has been built before (an estimate exists), when no overflow is
	// pending, and when the wide list fits the rank kernel
	if (ctx->opt[UGRT_OPT_ASYNC_BUILD] == 1 && G.have_est && G.est_w <= 3000u && ctx->cfg.slabs == 1 &&
	    ugrt_reported_status(ctx) == 0u && !ctx->overflow_seen)
		return build_common_async(ctx, G, F, C, ny, nz, ylo, yhi);
	if (ugrt_reported_status(ctx) != 0u)
		ctx->overflow_seen = true; // reported by ugrt_ctx_synchronize; until then every call waits and sizes exactly
	G.async_pending = false;
	G.valid = false;
	G.C = C;
	G.F = F;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_SCAN);
	rc = ugrt_prim_inclusive_scan(ctx, (const u32 *)G.sizes.p, (u32 *)G.scan.p, (size_t)F);
	ugrt_prof_end(ctx, UGRT_ST_BUILD_SCAN);
	if (rc)
		return rc;
	// total_triangles, frustum_grid.h:254 (the one unavoidable read-back: it sizes the lists), here
	// as narrow references + number of wide triangles
	UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_RW, (u32 *)G.scan.p + (F - 1), 8, hipMemcpyDeviceToHost, st));
	UGRT_HIP(hipStreamSynchronize(st));
	const u32 Rn = ctx->h_pinned[UGRT_PIN_RW], W = ctx->h_pinned[UGRT_PIN_RW + 1];
	WideBox wb;
	wb.W = W;
	wb.ny = (u32)ny;
	wb.nz = (u32)nz;
	wb.ylo = (u32)ylo;
	wb.yhi = (u32)yhi;
	wb.rw = nullptr;
	const unsigned long long active = (unsigned long long)(C / ((u32)ny * (u32)nz)) * (u32)(yhi - ylo + 1) * (u32)nz;
	const unsigned long long Rtot = (unsigned long long)Rn + active * W;
	if (Rtot > 0xFFFFFFF0ull)
		return ugrt_fail(UGRT_ENOMEM, "grid build: %llu references exceed the 32-bit lists", Rtot);
	const u32 R = (u32)Rtot;
	G.R = R;
	G.r_exact = true;
	G.est_rn = Rn;
	G.est_w = W;
	G.have_est = true;
	// (the slots an asynchronous build reports into: never older than this build)
	ctx->h_pinned[UGRT_PIN_REPORT + 4 * gidx] = Rn;
	ctx->h_pinned[UGRT_PIN_REPORT + 4 * gidx + 1] = W;
	// the sort goes key[0] -> key[1]; with wide triangles the merged lists are written back into key[0]
	size_t rb1 = (size_t)(Rn ? Rn : 1) * 4, rb0 = W ? (size_t)(R ? R : 1) * 4 : rb1;
	if ((rc = ugrt_buf_reserve(ctx, G.key[0], rb0)) || (rc = ugrt_buf_reserve(ctx, G.val[0], rb0)) ||
	    (rc = ugrt_buf_reserve(ctx, G.key[1], rb1)) || (rc = ugrt_buf_reserve(ctx, G.val[1], rb1)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.parts, ((size_t)Rn / BUILD_THREADS + 2) * 4))) // first triangle per fill workgroup
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.span, (size_t)C * 8 + 16))) // span[C], run starts[C], cells_used
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.offset, (size_t)C * 4)))
		return rc;
	u32 *k0 = (u32 *)G.key[0].p, *k1 = (u32 *)G.key[1].p, *v0 = (u32 *)G.val[0].p, *v1 = (u32 *)G.val[1].p;
	u32 *wl = (u32 *)G.wide.p, *wsorted = wl + F;
	if (Rn) {
		ugrt_prof_begin(ctx, UGRT_ST_BUILD_FILL);
		const u32 nparts = (Rn + BUILD_THREADS - 1) / BUILD_THREADS;
		hipLaunchKernelGGL(k_fill_parts, dim3((nparts + BUILD_THREADS) / BUILD_THREADS), dim3(BUILD_THREADS), 0, st,
				   (const u32 *)G.scan.p, F, Rn, nparts, (u32 *)G.parts.p, BuildCheck{ nullptr, 0u, 0u, 0ull, 0ull, nullptr, nullptr, nullptr });
		const bool own_sort = ctx->opt[UGRT_OPT_SORT_LIBRARY] != 1 && Rn <= (1u << 30);
		RsFirst hs = { nullptr }; // the fill counts the first digit of its keys for the sort
		if (own_sort && (rc = ugrt_sort_first_digit(ctx, &hs)))
			return rc;
		const u32 fill_blocks = hs.hist ? FILL_MAX_BLOCKS_COUNTING : FILL_MAX_BLOCKS;
		hipLaunchKernelGGL(k_fill, dim3(nparts < fill_blocks ? nparts : fill_blocks), dim3(BUILD_THREADS), 0, st,
				   (const u32 *)G.scan.p, (const Rng *)G.rng.p, (const u32 *)G.parts.p, Rn, ny, nz, k0, v0,
				   (u32 *)G.span.p, 2u * C + 1u, (const u32 *)nullptr, hs);
		ugrt_prof_end(ctx, UGRT_ST_BUILD_FILL);
		UGRT_HIP(hipGetLastError());
		ugrt_prof_begin(ctx, UGRT_ST_BUILD_SORT);
		rc = own_sort ? ugrt_sort_pairs_u32(ctx, k0, k1, v0, v1, Rn, bits_for(C), nullptr, true)
			      : ugrt_prim_sort_pairs(ctx, k0, k1, v0, v1, Rn, bits_for(C));
		ugrt_prof_end(ctx, UGRT_ST_BUILD_SORT);
		if (rc)
			return rc;
	}
	if (W) {
		ugrt_prof_begin(ctx, UGRT_ST_BUILD_SORT);
		if (W <= 4096u) {
			hipLaunchKernelGGL(k_wide_rank, dim3((W + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
					   st, (const u32 *)wl, W, wsorted, (const u32 *)nullptr);
			UGRT_HIP(hipGetLastError());
		} else {
			// many wide triangles (a tiny grid): radix sort of the ids, the values are not used
			if ((rc = ugrt_prim_sort_pairs(ctx, wl, wsorted, wl, (u32 *)G.sizes.p, W, bits_for((u32)F))))
				return rc;
		}
		ugrt_prof_end(ctx, UGRT_ST_BUILD_SORT);
	}
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_BOUNDS);
	u32 *cstart = (u32 *)G.span.p + C, *used = cstart + C;
	if (!Rn)
		UGRT_HIP(hipMemsetAsync(G.span.p, 0, (size_t)C * 8 + 4, st)); // (otherwise cleared by k_fill)
	if (R) {
		if (Rn) {
			hipLaunchKernelGGL(k_bounds, dim3((Rn + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
					   st, (const u32 *)k1, Rn, cstart, (u32 *)G.span.p, (const u32 *)nullptr);
			UGRT_HIP(hipGetLastError());
		}
	}
	{
		// spans + offsets in one kernel (an empty build: the cleared words are the spans)
		SpanLoad sl = { (const u32 *)cstart, (u32 *)G.span.p, used, wb };
		SpanTail tl = { (u32 *)nullptr, (const u32 *)used, (const u32 *)nullptr, wide_counter(G, F) };
		if ((rc = ugrt_scan_launch<false>(ctx, sl, (u32 *)G.offset.p, (size_t)C, tl)))
			return rc;
		G.wide_zeroed = wide_counter(G, F);
	}
	if (W) {
		if (Rn) {
			hipLaunchKernelGGL(k_merge_narrow, dim3((Rn + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS),
					   0, st, (const u32 *)k1, (const u32 *)v1, Rn, (const u32 *)cstart,
					   (const u32 *)G.offset.p, (const u32 *)wsorted, wb, k0, v0);
			UGRT_HIP(hipGetLastError());
		}
		u32 blocks = C < 256u * 32u ? C : 256u * 32u;
		hipLaunchKernelGGL(k_merge_wide, dim3(blocks), dim3(64), 0, st, (const u32 *)v1, (const u32 *)cstart,
				   (const u32 *)G.span.p, (const u32 *)G.offset.p, (const u32 *)wsorted, C, wb, k0, v0,
				   (u32 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr);
		UGRT_HIP(hipGetLastError());
		G.keys = k0;
		G.vals = v0;
	} else {
		G.keys = k1;
		G.vals = v1;
	}
	ugrt_prof_end(ctx, UGRT_ST_BUILD_BOUNDS);
	// "Number of actual cells" (frustum_grid.h:337): fetched lazily by ugrt_grid_get_info
	UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_CELLS_USED + (&G - ctx->grid), used, 4, hipMemcpyDeviceToHost, st));
	G.valid = true;
	return UGRT_OK;
}

static int build_prologue(ugrt_ctx *ctx, Grid &G, const int *d_facelist, const float *d_vertlist, int F,
			  const char *who)
{
	if (!ctx || !d_facelist || !d_vertlist)
		return ugrt_fail(UGRT_EINVAL, "%s: null argument", who);
	if (F <= 0)
		return ugrt_fail(UGRT_EINVAL, "%s: num_faces must be positive", who);
	UGRT_HIP(hipSetDevice(ctx->device));
	int rc;
	if ((rc = ugrt_buf_reserve(ctx, G.rng, (size_t)F * sizeof(Rng))))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.sizes, (size_t)F * 4)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.scan, (size_t)(F + 1) * 4))) // + the wide-triangle counter
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, G.wide, (size_t)F * 8))) // wide triangle ids as found, then ascending
		return rc;
	// the geometry may have changed since the last build (animation, a new frame file): refresh the records,
	// unless the caller vouches for it (UGRT_FLAG_STATIC_GEOMETRY) and these are the arrays last seen
	if ((ctx->cfg.flags & UGRT_FLAG_STATIC_GEOMETRY) && ctx->rec_valid && ctx->rec_verts == d_vertlist &&
	    ctx->rec_tris == d_facelist && ctx->rec_faces == F) {
		if (G.wide_zeroed != wide_counter(G, F)) // (the last build of this grid left it at zero: SpanTail)
			UGRT_HIP(hipMemsetAsync(wide_counter(G, F), 0, 4, ctx->stream));
		G.wide_zeroed = nullptr; // the count kernel dirties it; the build's last scan clears it again
		return UGRT_OK;
	}
	ctx->rec_valid = false;
	G.wide_zeroed = nullptr;
	if ((rc = ugrt_buf_reserve(ctx, ctx->trirec, (size_t)F * 48)))
		return rc;
	hipLaunchKernelGGL(k_tri_records, dim3((F + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
			   ctx->stream, d_facelist, d_vertlist, F, (float4 *)ctx->trirec.p, wide_counter(G, F));
	UGRT_HIP(hipGetLastError());
	ctx->rec_verts = d_vertlist;
	ctx->rec_tris = d_facelist;
	ctx->rec_faces = F;
	ctx->rec_valid = true;
	return UGRT_OK;
}

// FrustumGrid::buildGrid, frustum_grid.h:210
extern "C" int ugrt_grid_build_perspective(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int F)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "grid_build_perspective: null context");
	Grid &G = ctx->grid[UGRT_GRID_PERSPECTIVE];
	int rc = build_prologue(ctx, G, d_facelist, d_vertlist, F, "grid_build_perspective");
	if (rc)
		return rc;
	const int K = ctx->cfg.slabs;
	G.slabs = K;
	if (K > 1 && (rc = ugrt_buf_reserve(ctx, G.projz, (size_t)F * 4 + 8)))
		return rc;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_COUNT);
	hipLaunchKernelGGL(k_count_persp, dim3((F + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
			   ctx->stream, ctx->cam, d_facelist, d_vertlist, F, ctx->cfg.row_begin, ctx->cfg.row_end,
			   (Rng *)G.rng.p, (u32 *)G.sizes.p, (u32 *)G.wide.p, wide_counter(G, F),
			   K > 1 ? (float *)G.projz.p : (float *)nullptr);
	UGRT_HIP(hipGetLastError());
	if (K > 1 && (rc = build_slabs(ctx, G, F, 2.0f, -2.0f))) // frustum_grid.h:223
		return rc;
	ugrt_prof_end(ctx, UGRT_ST_BUILD_COUNT);
	G.dims[0] = ctx->nbx;
	G.dims[1] = ctx->nby;
	G.dims[2] = K;
	return build_common(ctx, G, F, (u32)ctx->nbx * (u32)ctx->nby * (u32)K, ctx->nby, K, ctx->cfg.row_begin,
			    ctx->cfg.row_end - 1);
}

// FrustumGrid::buildSphericalGrid, frustum_grid.h:368
extern "C" int ugrt_grid_build_spherical(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int F,
					 float xM, float yM)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "grid_build_spherical: null context");
	Grid &G = ctx->grid[UGRT_GRID_SPHERICAL];
	int rc = build_prologue(ctx, G, d_facelist, d_vertlist, F, "grid_build_spherical");
	if (rc)
		return rc;
	int lx = ctx->cfg.light_nbx, ly = ctx->cfg.light_nby;
	const int K = ctx->cfg.slabs;
	G.slabs = K;
	if (K > 1 && (rc = ugrt_buf_reserve(ctx, G.projz, (size_t)F * 4 + 8)))
		return rc;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_COUNT);
	hipLaunchKernelGGL(k_count_sph, dim3((F + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
			   ctx->stream, ctx->cam, d_facelist, d_vertlist, F, lx, ly, xM, yM, (Rng *)G.rng.p,
			   (u32 *)G.sizes.p, (u32 *)G.wide.p, wide_counter(G, F),
			   K > 1 ? (float *)G.projz.p : (float *)nullptr, ctx->face_lo, ctx->face_hi >= 0 ? ctx->face_hi : F);
	UGRT_HIP(hipGetLastError());
	if (K > 1 && (rc = build_slabs(ctx, G, F, 9999.9f, -9999.9f))) // frustum_grid.h:386
		return rc;
	ugrt_prof_end(ctx, UGRT_ST_BUILD_COUNT);
	G.dims[0] = lx;
	G.dims[1] = ly;
	G.dims[2] = K;
	return build_common(ctx, G, F, (u32)lx * (u32)ly * (u32)K, ly, K, 0, ly - 1);
}

// uniform grid over the scene box (Model::{x,y,z}{Min,Max}, scene.h:273-292),
// padded by 1e-4 of the extent + 1e-4 so that no vertex lies on the boundary
extern "C" int ugrt_grid_build_uniform(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int F,
				       const float bbmin[3], const float bbmax[3])
{
	if (!ctx || !bbmin || !bbmax)
		return ugrt_fail(UGRT_EINVAL, "grid_build_uniform: null argument");
	Grid &G = ctx->grid[UGRT_GRID_UNIFORM];
	int rc = build_prologue(ctx, G, d_facelist, d_vertlist, F, "grid_build_uniform");
	if (rc)
		return rc;
	UGrid g;
	for (int k = 0; k < 3; k++) {
		float ext = bbmax[k] - bbmin[k];
		float pad = ext * 1e-4f + 1e-4f;
		float lo = bbmin[k] - pad, hi = bbmax[k] + pad;
		float cs = (hi - lo) / (float)ctx->cfg.uniform_dims[k];
		g.lo[k] = lo;
		g.cs[k] = cs;
		g.inv[k] = 1.0f / cs;
		g.dims[k] = ctx->cfg.uniform_dims[k];
		G.dims[k] = g.dims[k];
		G.ug[k] = lo;
		G.ug[3 + k] = cs;
		G.ug[6 + k] = g.inv[k];
	}
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_COUNT);
	hipLaunchKernelGGL(k_count_uniform, dim3((F + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0,
			   ctx->stream, g, d_facelist, d_vertlist, F, (Rng *)G.rng.p, (u32 *)G.sizes.p, (u32 *)G.wide.p,
			   wide_counter(G, F), ctx->face_lo, ctx->face_hi >= 0 ? ctx->face_hi : F);
	ugrt_prof_end(ctx, UGRT_ST_BUILD_COUNT);
	UGRT_HIP(hipGetLastError());
	return build_common(ctx, G, F, (u32)g.dims[0] * (u32)g.dims[1] * (u32)g.dims[2], g.dims[1], g.dims[2], 0,
			    g.dims[1] - 1);
}

// ---------------------------------------------------------------------------
// Sharded build: every rank builds the light / uniform grid for its window of the triangle
// list (ugrt_ctx_set_face_window) and the ranks exchange their results.  A shard's lists are complete grid
// arrays of its triangles: sorted by cell, ascending id inside a cell.  The windows are disjoint and ascending
// with the rank, so the full build's run of a cell is the concatenation of the shards' runs in rank order:
//   span[c] = sum_r span_r[c],  offset = exclusive scan,
//   element i of shard r (cell c = key_r[i]) lands at offset[c] + sum_{r' < r} span_r'[c] + (i - offset_r[c]).
// ---------------------------------------------------------------------------
#define MERGE_MAX_PARTS 16
struct MergeParts {
	const u32 *keys[MERGE_MAX_PARTS], *vals[MERGE_MAX_PARTS], *span[MERGE_MAX_PARTS];
	u32 count[MERGE_MAX_PARTS];
	int n;
};

// total span per cell + the occupied cells; before[r][c] = sum of the spans of the parts before r
__global__ __launch_bounds__(BUILD_THREADS) void k_shard_spans(MergeParts mp, u32 C, u32 *__restrict__ span,
								u32 *__restrict__ before, u32 *__restrict__ used)
{
	u32 mine = 0;
	for (u32 c = blockIdx.x * BUILD_THREADS + threadIdx.x; c < C; c += gridDim.x * BUILD_THREADS) {
		u32 acc = 0;
		for (int r = 0; r < mp.n; r++) {
			before[(size_t)r * C + c] = acc;
			acc += mp.span[r][c];
		}
		span[c] = acc;
		mine += acc != 0u ? 1u : 0u;
	}
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		mine += (u32)__shfl_xor((int)mine, m);
	if ((threadIdx.x & 63) == 0 && mine)
		atomicAdd(used, mine);
}

// start of every cell's run inside part r (exclusive scan of span_r), written by the run heads of the part's keys
__global__ __launch_bounds__(BUILD_THREADS) void k_shard_starts(const u32 *__restrict__ keys, u32 n,
								 u32 *__restrict__ start)
{
	const u32 i = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (i >= n)
		return;
	const u32 k = keys[i];
	if (i == 0 || keys[i - 1] != k)
		start[k] = i;
}

__global__ __launch_bounds__(BUILD_THREADS) void k_shard_scatter(const u32 *__restrict__ keys, const u32 *__restrict__ vals,
								  u32 n, const u32 *__restrict__ start,
								  const u32 *__restrict__ before, const u32 *__restrict__ offset,
								  u32 *__restrict__ okeys, u32 *__restrict__ ovals)
{
	const u32 i = blockIdx.x * BUILD_THREADS + threadIdx.x;
	if (i >= n)
		return;
	const u32 c = keys[i];
	const u32 pos = offset[c] + before[c] + (i - start[c]);
	okeys[pos] = c;
	ovals[pos] = vals[i];
}

extern "C" int ugrt_grid_merge_shards(ugrt_ctx *ctx, int which, int nparts, const unsigned *const *d_keys,
				      const unsigned *const *d_vals, const unsigned *const *d_span,
				      const unsigned *counts)
{
	if (!ctx || !d_keys || !d_vals || !d_span || !counts || nparts < 1 || nparts > MERGE_MAX_PARTS)
		return ugrt_fail(UGRT_EINVAL, "grid_merge_shards: bad argument (1..%d parts)", MERGE_MAX_PARTS);
	if (which != UGRT_GRID_SPHERICAL && which != UGRT_GRID_UNIFORM)
		return ugrt_fail(UGRT_EINVAL, "grid_merge_shards: the light grid and the uniform grid are built in shards");
	Grid &G = ctx->grid[which];
	if (!G.valid)
		return ugrt_fail(UGRT_EINVAL, "grid_merge_shards: build this rank's shard first (it defines the cells)");
	UGRT_HIP(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	const u32 C = G.C;
	MergeParts mp;
	mp.n = nparts;
	unsigned long long Rtot = 0;
	for (int r = 0; r < nparts; r++) {
		if (!d_span[r] || (counts[r] && (!d_keys[r] || !d_vals[r])))
			return ugrt_fail(UGRT_EINVAL, "grid_merge_shards: null part %d", r);
		mp.keys[r] = d_keys[r];
		mp.vals[r] = d_vals[r];
		mp.span[r] = d_span[r];
		mp.count[r] = counts[r];
		Rtot += counts[r];
	}
	if (Rtot > 0xFFFFFFF0ull)
		return ugrt_fail(UGRT_ENOMEM, "grid_merge_shards: %llu references exceed the 32-bit lists", Rtot);
	int rc;
	// outputs: the context's grid arrays (the shard this context built is replaced; the parts are the caller's
	// buffers and must not be these arrays: checked against the arrays as they are NOW, before a reserve below can
	// free and re-allocate them)
	for (int r = 0; r < nparts; r++)
		for (int i = 0; i < 2; i++)
			if (d_keys[r] == (const unsigned *)G.key[i].p || d_vals[r] == (const unsigned *)G.val[i].p ||
			    d_span[r] == (const unsigned *)G.span.p)
				return ugrt_fail(UGRT_EINVAL, "grid_merge_shards: part %d aliases the context's own grid arrays", r);
	const size_t rb = (size_t)(Rtot ? Rtot : 1) * 4;
	if ((rc = ugrt_buf_reserve(ctx, G.key[0], rb)) || (rc = ugrt_buf_reserve(ctx, G.val[0], rb)) ||
	    (rc = ugrt_buf_reserve(ctx, G.span, (size_t)C * 8 + 16)) || (rc = ugrt_buf_reserve(ctx, G.offset, (size_t)C * 4)) ||
	    (rc = ugrt_buf_reserve(ctx, G.parts, (size_t)(nparts + 1) * C * 4)))
		return rc;
	u32 *span = (u32 *)G.span.p, *used = span + 2 * (size_t)C, *before = (u32 *)G.parts.p, *start = before + (size_t)nparts * C;
	ugrt_prof_begin(ctx, UGRT_ST_BUILD_BOUNDS);
	UGRT_HIP(hipMemsetAsync(used, 0, 4, st));
	const u32 cblocks = (C + BUILD_THREADS - 1) / BUILD_THREADS;
	hipLaunchKernelGGL(k_shard_spans, dim3(cblocks < 512u ? cblocks : 512u), dim3(BUILD_THREADS), 0, st, mp, C, span, before,
			   used);
	UGRT_HIP(hipGetLastError());
	if ((rc = ugrt_prim_exclusive_scan(ctx, (const u32 *)span, (u32 *)G.offset.p, (size_t)C)))
		return rc;
	for (int r = 0; r < nparts; r++) {
		if (!counts[r])
			continue;
		const u32 n = counts[r];
		hipLaunchKernelGGL(k_shard_starts, dim3((n + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0, st,
				   mp.keys[r], n, start);
		hipLaunchKernelGGL(k_shard_scatter, dim3((n + BUILD_THREADS - 1) / BUILD_THREADS), dim3(BUILD_THREADS), 0, st,
				   mp.keys[r], mp.vals[r], n, (const u32 *)start, (const u32 *)(before + (size_t)r * C),
				   (const u32 *)G.offset.p, (u32 *)G.key[0].p, (u32 *)G.val[0].p);
		UGRT_HIP(hipGetLastError());
	}
	ugrt_prof_end(ctx, UGRT_ST_BUILD_BOUNDS);
	UGRT_HIP(hipMemcpyAsync(ctx->h_pinned + UGRT_PIN_CELLS_USED + which, used, 4, hipMemcpyDeviceToHost, st));
	G.keys = (u32 *)G.key[0].p;
	G.vals = (u32 *)G.val[0].p;
	G.R = (u32)Rtot;
	G.r_exact = true; // (the merged count, not the estimate of an asynchronous shard build)
	G.async_pending = false;
	return UGRT_OK;
}

extern "C" int ugrt_ctx_set_face_window(ugrt_ctx *ctx, int begin, int end)
{
	// end < 0: up to the last triangle ((0, -1) = every triangle); begin == end: an empty shard
	if (!ctx || begin < 0 || (end >= 0 && end < begin))
		return ugrt_fail(UGRT_EINVAL, "set_face_window: bad argument");
	ctx->face_lo = begin;
	ctx->face_hi = end;
	return UGRT_OK;
}

extern "C" int ugrt_geometry_changed(ugrt_ctx *ctx)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "geometry_changed: null context");
	ctx->rec_valid = false;
	return UGRT_OK;
}
