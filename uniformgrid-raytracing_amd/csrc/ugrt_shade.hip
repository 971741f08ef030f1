// ugrt_shade.hip -- per-pixel stages: light-space ray mapping, ray re-ordering,
// shading, secondary-ray generation, vertex animation.
#include "ugrt_dev.h"
#include "ugrt_scan.h"

#define PX_THREADS 256

// mapSort_Effective_kernel, misc_kernel.cu:255-296 (cam = LIGHT camera).
// The reference launches one 8x8 block per tile; per-pixel work has no tile
// structure, so this is a flat, fully coalesced pass over the band's pixels.
__global__ __launch_bounds__(PX_THREADS) void k_map_rays(CamBlock cam, const float *__restrict__ t_value_list,
							  const float *__restrict__ ray_direction,
							  u32 *__restrict__ d_map, const float *__restrict__ cmPt,
							  float xM, float yM, int lnbx, int lnby, int p0, int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int pixelId = p0 + i;
	float tVal = t_value_list[pixelId];
	float pI[3], lrd[3];
	pI[0] = cmPt[0] + tVal * ray_direction[pixelId * 3 + 0];
	pI[1] = cmPt[1] + tVal * ray_direction[pixelId * 3 + 1];
	pI[2] = cmPt[2] + tVal * ray_direction[pixelId * 3 + 2];
	lrd[0] = pI[0] - cam.cc[0];
	lrd[1] = pI[1] - cam.cc[1];
	lrd[2] = pI[2] - cam.cc[2];
	D_NORMALIZE(lrd);
	int blx = (int)d_effective_x(cam, lrd, xM, lnbx / 2);
	int bly = (int)d_effective_y(cam, lrd, yM, lnby / 2);
	int blockIndex;
	if (blx >= 0 && blx < lnbx && bly >= 0 && bly < lnby)
		blockIndex = blx * lnby + bly;
	else
		blockIndex = lnbx * lnby;
	d_map[i] = (u32)pixelId;
	d_map[n + i] = (u32)blockIndex;
}

// getEffectiveRayGridMapping, per_frame_funcs.h:97-114
extern "C" int ugrt_map_rays_to_light(ugrt_ctx *ctx, const float *d_t_value, const float *d_ray_dir, unsigned *d_map,
				      const float *d_cam_position, float xM, float yM)
{
	if (!ctx || !d_t_value || !d_ray_dir || !d_map || !d_cam_position)
		return ugrt_fail(UGRT_EINVAL, "map_rays_to_light: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_MAP_RAYS);
	hipLaunchKernelGGL(k_map_rays, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, ctx->cam, d_t_value, d_ray_dir, d_map, d_cam_position, xM, yM,
			   ctx->cfg.light_nbx, ctx->cfg.light_nby, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_MAP_RAYS);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// processData, per_frame_funcs.h:116-137.  Reference: 15-bit radix sort of N
// rays, then five N-sized passes (blockScan, cudppSegmentedScan,
// preStreamCompaction, tag_thread, cudppCompact: decision_data.h:171-271) to
// find the first ray of every 64-ray chunk.  Here: the same stable sort, ONE
// N-sized pass that records where each light cell's run starts and ends, and
// the chunk starts are then generated per CELL (run start + 64*j), which is
// the same ascending list.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PX_THREADS) void k_ray_runs(const u32 *__restrict__ keys, u32 n, u32 *__restrict__ rstart,
							  u32 *__restrict__ rend)
{
	u32 i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	u32 k = keys[i];
	if (i == 0 || keys[i - 1] != k)
		rstart[k] = i;
	if (i == n - 1 || keys[i + 1] != k)
		rend[k] = i + 1;
}

// chunks per light cell, formed where the scan of the counts loads them (ugrt_scan.h)
struct ChunkLoad {
	const u32 *rstart, *rend;
	__device__ __forceinline__ void operator()(u32 base, u32 n, u32 (&v)[SC_ITEMS]) const
	{
#pragma unroll
		for (int k = 0; k < SC_ITEMS; k++) {
			const u32 c = base + (u32)k;
			v[k] = c < n ? (rend[c] - rstart[c] + 63u) / 64u : 0u;
		}
	}
};

// one thread per CHUNK (a cell with thousands of rays would otherwise serialise one thread)
__global__ __launch_bounds__(PX_THREADS) void k_chunk_emit(const u32 *__restrict__ rstart,
							    const u32 *__restrict__ rend,
							    const u32 *__restrict__ incl, u32 ncell, u32 cap,
							    u32 *__restrict__ prefix, u32 *__restrict__ host_total)
{
	const u32 total = incl[ncell - 1];
	u32 k = blockIdx.x * PX_THREADS + threadIdx.x;
	if (k == 0u)
		*host_total = total; // h_numCudaBlocks, decision_data.h:264: straight into the pinned host word
	if (k >= total || k >= cap)
		return;
	u32 lo = 0, hi = ncell - 1; // smallest c with incl[c] > k
	while (lo < hi) {
		u32 mid = (lo + hi) >> 1;
		if (incl[mid] > k)
			hi = mid;
		else
			lo = mid + 1;
	}
	const u32 c = lo;
	const u32 j = k - (incl[c] - (rend[c] - rstart[c] + 63u) / 64u);
	prefix[k] = rstart[c] + 64u * j;
}

static int key_bits(u32 nkeys)
{
	int b = 1;
	while (b < 32 && (1ull << b) < (unsigned long long)nkeys)
		b++;
	return b;
}

// the sort, the runs per light cell and the chunk starts (enqueued; the chunk count goes to the pinned host word)
static int sort_rays_now(ugrt_ctx *ctx, unsigned *d_map, unsigned *d_prefix_map, unsigned prefix_capacity)
{
	hipStream_t st = ctx->stream;
	const u32 n = (u32)ctx->npix;
	const u32 ncell = (u32)ctx->cfg.light_nbx * (u32)ctx->cfg.light_nby + 1u; // + sentinel
	int rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->rmap[0], (size_t)n * 8)))
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->rstart, (size_t)ncell * 8))) // run starts, then run ends
		return rc;
	if ((rc = ugrt_buf_reserve(ctx, ctx->cbase, (size_t)ncell * 4)))
		return rc;
	u32 *tmp = (u32 *)ctx->rmap[0].p;
	ugrt_prof_begin(ctx, UGRT_ST_SORT_RAYS);
	// cudppSort(sortPlan, &d_map[IMAGE_SIZE], &d_map[0], 15, IMAGE_SIZE), decision_data.h:177: in place.  With an even
	// number of passes the built-in sort reads its input in the first pass only (and in the histogram kernel before
	// it), writes its own buffers there, and may write the caller's arrays in its last pass; otherwise sort into a
	// copy and copy back.
	const int kb = key_bits(ncell);
	if (((kb + 7) / 8) % 2 == 0 && ctx->opt[UGRT_OPT_SORT_LIBRARY] != 1 && n <= (1u << 30)) {
		rc = ugrt_sort_pairs_u32(ctx, d_map + n, d_map + n, d_map, d_map, n, kb);
		if (rc)
			return rc;
	} else {
		rc = ugrt_prim_sort_pairs(ctx, d_map + n, tmp + n, d_map, tmp, n, kb);
		if (rc)
			return rc;
		UGRT_HIP(hipMemcpyAsync(d_map, tmp, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
	}
	u32 *rstart = (u32 *)ctx->rstart.p, *rend = rstart + ncell;
	UGRT_HIP(hipMemsetAsync(rstart, 0, (size_t)ncell * 8, st));
	hipLaunchKernelGGL(k_ray_runs, dim3((n + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0, st,
			   (const u32 *)(d_map + n), n, rstart, rend);
	UGRT_HIP(hipGetLastError());
	{
		const ChunkLoad load = { rstart, rend };
		if ((rc = ugrt_scan_launch<true>(ctx, load, (u32 *)ctx->cbase.p, ncell, ScanTailNone())))
			return rc;
	}
	{
		// at most n/64 + ncell chunks exist; the kernel reads the exact count on the device
		u32 maxchunks = n / 64u + ncell;
		if (maxchunks > prefix_capacity)
			maxchunks = prefix_capacity;
		hipLaunchKernelGGL(k_chunk_emit, dim3((maxchunks + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0, st,
				   (const u32 *)rstart, (const u32 *)rend, (const u32 *)ctx->cbase.p,
				   ncell, prefix_capacity, d_prefix_map, ctx->h_pinned + UGRT_PIN_CHUNKS);
	}
	UGRT_HIP(hipGetLastError());
	ugrt_prof_end(ctx, UGRT_ST_SORT_RAYS);
	ctx->ray_sort_pending = false;
	return UGRT_OK;
}

extern "C" int ugrt_sort_rays(ugrt_ctx *ctx, unsigned *d_map, unsigned *d_prefix_map, unsigned prefix_capacity,
			      unsigned *num_chunks)
{
	if (!ctx || !d_map || !d_prefix_map)
		return ugrt_fail(UGRT_EINVAL, "sort_rays: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ctx->chunk_capacity = prefix_capacity;
	ctx->chunk_prefix = d_prefix_map;
	ctx->chunk_map = d_map;
	// Deferred form under UGRT_FLAG_SHADOW_ALL_CHUNKS: the chunk list only says WHICH rays the reference's launch traces,
	// and with that flag it is all of them -- the shadow tracer reads (pixel, light cell) pairs in any order and puts them
	// in (cell, direction) order itself.  The sort is then carried out when its results are asked for
	// (ugrt_sort_rays_chunks), not before: two radix passes and five more launches that a frame does not need.
	if (!num_chunks && (ctx->cfg.flags & UGRT_FLAG_SHADOW_ALL_CHUNKS) && ctx->opt[UGRT_OPT_RAY_SORT] != 1) {
		ctx->ray_sort_pending = true;
		return UGRT_OK;
	}
	int rc = sort_rays_now(ctx, d_map, d_prefix_map, prefix_capacity);
	if (rc || !num_chunks)
		return rc; // deferred: the count stays on the device (UGRT_CHUNKS_ON_DEVICE) until ugrt_sort_rays_chunks
	return ugrt_sort_rays_chunks(ctx, num_chunks);
}

extern "C" int ugrt_sort_rays_chunks(ugrt_ctx *ctx, unsigned *num_chunks)
{
	if (!ctx || !num_chunks)
		return ugrt_fail(UGRT_EINVAL, "sort_rays_chunks: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	if (ctx->ray_sort_pending) { // the sort that ugrt_sort_rays(..., NULL) put off: its arrays still hold the unsorted map
		int rc = sort_rays_now(ctx, const_cast<unsigned *>(ctx->chunk_map), const_cast<unsigned *>(ctx->chunk_prefix), ctx->chunk_capacity);
		if (rc)
			return rc;
	}
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	*num_chunks = ctx->h_pinned[UGRT_PIN_CHUNKS];
	if (*num_chunks > ctx->chunk_capacity)
		return ugrt_fail(UGRT_EINVAL, "sort_rays: %u chunks do not fit prefix_capacity %u", *num_chunks,
				 ctx->chunk_capacity);
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// shading: lambertian_shade :165-221, spot_shade :275-345, shadow_kernel :347
// ---------------------------------------------------------------------------
// shader_kernel.cu:223-273 get_along_x / get_along_y
__device__ __forceinline__ float d_along_x(const CamBlock &cam, const float *vec)
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
	return (rightDotValue > 0) ? angle : -1.0f * angle;
}
__device__ __forceinline__ float d_along_y(const CamBlock &cam, const float *vec)
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
	return (upDotValue > 0) ? angle : -1.0f * angle;
}

template <bool SPOT>
__global__ __launch_bounds__(PX_THREADS) void k_shade(CamBlock cam, unsigned char *__restrict__ d_img,
						       const float *__restrict__ dd_normal,
						       const float *__restrict__ dd_t_value, const float *__restrict__ dd_dir,
						       int *__restrict__ dd_intersect_id, const float *__restrict__ d_cam_pos,
						       const int *__restrict__ mat_idx, const float *__restrict__ mat_list,
						       int mat_count, float *__restrict__ dump, int p0, int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int pixelID = p0 + i;
	float color[3] = { 0.0f, 0.0f, 0.0f }, drop_off = 1.0f;
	int tri_intersected = dd_intersect_id[pixelID];
	// the reference reads mat_idx[-2] for a miss (shader_kernel.cu:170); a miss keeps its id and shades black
	int idx = tri_intersected >= 0 ? mat_idx[tri_intersected] : tri_intersected;
	float t_value = dd_t_value[pixelID];
	float dir[3] = { dd_dir[pixelID * 3 + 0], dd_dir[pixelID * 3 + 1], dd_dir[pixelID * 3 + 2] };
	float point[3];
	point[0] = d_cam_pos[0] + t_value * dir[0];
	point[1] = d_cam_pos[1] + t_value * dir[1];
	point[2] = d_cam_pos[2] + t_value * dir[2];
	if (SPOT) {
		float lrd[3];
		lrd[0] = point[0] - cam.cc[0];
		lrd[1] = point[1] - cam.cc[1];
		lrd[2] = point[2] - cam.cc[2];
		D_NORMALIZE(lrd);
		float x = d_along_x(cam, lrd), y = d_along_y(cam, lrd);
		if (dump) {
			dump[pixelID * 2 + 0] = x;
			dump[pixelID * 2 + 1] = y;
		}
		const float qpi = (float)(3.14159265358979323846 / 4);
		drop_off = (x < qpi && x > -qpi && y < qpi && y > -qpi) ? 1.0f : 0.25f;
	}
	dd_intersect_id[pixelID] = idx;
	if (idx >= 0 && idx < mat_count && (SPOT || t_value > 0)) {
		float material[6];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			material[k] = mat_list[idx * 6 + 3 + k]; // ambient uses Kd too (:180-186)
			material[3 + k] = mat_list[idx * 6 + 3 + k];
		}
		float nrm[3] = { dd_normal[pixelID * 3 + 0], dd_normal[pixelID * 3 + 1], dd_normal[pixelID * 3 + 2] };
		d_lambert<SPOT>(cam, point, nrm, color, material, drop_off);
#pragma unroll
		for (int k = 0; k < 3; k++)
			color[k] = color[k] > 1.0f ? 1.0f : color[k];
	}
	d_img[pixelID * 3 + 0] = d_to_u8(color[0]);
	d_img[pixelID * 3 + 1] = d_to_u8(color[1]);
	d_img[pixelID * 3 + 2] = d_to_u8(color[2]);
}

static int shade_args_ok(ugrt_ctx *ctx, const void *a, const void *b, const void *c, const void *d, const void *e,
			 const void *f, const void *g, const void *h, const char *who)
{
	if (!ctx || !a || !b || !c || !d || !e || !f || !g || !h)
		return ugrt_fail(UGRT_EINVAL, "%s: null argument", who);
	return UGRT_OK;
}

// Shader::simpleShade, shader.h:68-86
extern "C" int ugrt_shade_simple(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal, const float *d_t_value,
				 const float *d_ray_dir, int *d_intersect_id, const float *d_cam_position,
				 const int *d_mat_idx, const float *d_mat_list, int num_materials)
{
	int rc = shade_args_ok(ctx, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position, d_mat_idx,
			       d_mat_list, "shade_simple");
	if (rc)
		return rc;
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_SHADE);
	hipLaunchKernelGGL(k_shade<false>, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, ctx->cam, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position,
			   d_mat_idx, d_mat_list, num_materials, (float *)nullptr, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_SHADE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// Shader::spotlight_shade, shader.h:88-113 (d_dump may be null; the reference
// copies it back and prints one line per pixel, shader.h:108-112)
extern "C" int ugrt_shade_spotlight(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal,
				    const float *d_t_value, const float *d_ray_dir, int *d_intersect_id,
				    const float *d_cam_position, const int *d_mat_idx, const float *d_mat_list,
				    int num_materials, float *d_dump)
{
	int rc = shade_args_ok(ctx, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position, d_mat_idx,
			       d_mat_list, "shade_spotlight");
	if (rc)
		return rc;
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_SHADE);
	hipLaunchKernelGGL(k_shade<true>, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, ctx->cam, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position,
			   d_mat_idx, d_mat_list, num_materials, d_dump, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_SHADE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// shadow_kernel, shader_kernel.cu:347-359
__global__ __launch_bounds__(PX_THREADS) void k_add_shadows(unsigned char *__restrict__ d_img,
							     const int *__restrict__ is_shadowed, int p0, int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int pixelID = p0 + i;
	if (is_shadowed[pixelID] == 1) {
		d_img[pixelID * 3 + 0] /= 3;
		d_img[pixelID * 3 + 1] /= 3;
		d_img[pixelID * 3 + 2] /= 3;
	}
}

// Shader::add_shadows, shader.h:58-66
extern "C" int ugrt_shade_add_shadows(ugrt_ctx *ctx, unsigned char *d_img, const int *d_is_shadowed)
{
	if (!ctx || !d_img || !d_is_shadowed)
		return ugrt_fail(UGRT_EINVAL, "shade_add_shadows: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_SHADE);
	hipLaunchKernelGGL(k_add_shadows, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, d_img, d_is_shadowed, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_SHADE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// shader_kernel.cu:4-44 Noise / InterPolation / PerlinNoise with octaves = 1
__device__ __forceinline__ float d_noise(int x)
{
	u32 ux = (u32)x;
	ux = (ux << 13) ^ ux;
	ux = (ux * (ux * ux * 15731u + 789221u) + 1376312589u) & 0x7fffffffu;
	return (float)(int)ux / 2147483648.0f;
}
__device__ __forceinline__ float d_interp(float a, float b, float c) { return a + (b - a) * c * c * (3 - 2 * c); }
__device__ __forceinline__ float d_perlin(float x, float y, int width, int seed, float periode)
{
	float freq = 1.0f / periode;
	int num = ugrt_f2i((float)width * freq);
	int step_x = ugrt_f2i(x * freq), step_y = ugrt_f2i(y * freq);
	float zone_x = x * freq - (float)step_x;
	float zone_y = y * freq - (float)step_y;
	int noisedata = step_x + step_y * num + seed;
	float a = d_interp(d_noise(noisedata), d_noise(noisedata + 1), zone_x);
	float b = d_interp(d_noise(noisedata + num), d_noise(noisedata + 1 + num), zone_x);
	return d_interp(a, b, zone_y) * 324.0f;
}

// perlin_noise_shade, shader_kernel.cu:505-547
__global__ __launch_bounds__(PX_THREADS) void k_shade_perlin(unsigned char *__restrict__ d_img,
							      const int *__restrict__ dd_intersect_id, int W, int p0,
							      int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int pixelID = p0 + i;
	float x = (float)(pixelID % W), y = (float)(pixelID / W);
	float v1 = d_perlin(x, y, 12413, 63, 100.0f), v2 = d_perlin(x, y, 12413, 63, 25.0f);
	float v3 = d_perlin(x, y, 12413, 63, 12.5f), v4 = d_perlin(x, y, 12413, 63, 6.25f);
	float v5 = d_perlin(x, y, 12413, 63, 3.125f), v6 = d_perlin(x, y, 12413, 63, 1.56f);
	float tmp = (float)(ugrt_f2i(v1) + ugrt_f2i(v2 * 0.25f) + ugrt_f2i(v3 * 0.125f) + ugrt_f2i(v4 * 0.0625f) +
			    ugrt_f2i(v5 * 0.03125f) + ugrt_f2i(v6 * 0.0156f));
	int r = ugrt_f2i(tmp * (1 - 0.0f) + 0.0f * 0.0f);
	int g = ugrt_f2i(0.0f * (1 - 0.0f) + tmp * 0.0f);
	int b = ugrt_f2i(0.0f * (1 - tmp) + 0.0f * tmp);
	r = r > 255 ? 255 : r;
	g = g > 255 ? 255 : g;
	b = b > 255 ? 255 : b;
	bool hit = dd_intersect_id[pixelID] >= 0;
	d_img[pixelID * 3 + 0] = hit ? (unsigned char)r : 0;
	d_img[pixelID * 3 + 1] = hit ? (unsigned char)g : 0;
	d_img[pixelID * 3 + 2] = hit ? (unsigned char)b : 0;
}

// Shader::perlinShade, shader.h:115-131 (t, dir and camera are unused by the kernel, as in the reference)
extern "C" int ugrt_shade_perlin(ugrt_ctx *ctx, unsigned char *d_img, const float *d_t_value, const float *d_ray_dir,
				 const float *d_cam_position, const int *d_intersect_id)
{
	(void)d_t_value;
	(void)d_ray_dir;
	(void)d_cam_position;
	if (!ctx || !d_img || !d_intersect_id)
		return ugrt_fail(UGRT_EINVAL, "shade_perlin: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_SHADE);
	hipLaunchKernelGGL(k_shade_perlin, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, d_img, d_intersect_id, ctx->cfg.width, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_SHADE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// reflection bounce helpers (not in the reference)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PX_THREADS) void k_reflect_rays(const float *__restrict__ cam_pos,
							      const float *__restrict__ t_list,
							      const float *__restrict__ dir_list,
							      const int *__restrict__ id_list, const int *__restrict__ mat_idx,
							      const float *__restrict__ reflect, int mat_count,
							      const float *__restrict__ verts, const int *__restrict__ tris,
							      float eps, float *__restrict__ rays, int *__restrict__ active,
							      int p0, int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int p = p0 + i;
	int id = id_list[p];
	float t = t_list[p];
	float out[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
	int act = 0;
	if (t > 0 && id >= 0) {
		int m = mat_idx[id];
		if (m >= 0 && m < mat_count && reflect[m] > 0) {
			float tri[9], nn[3], d[3], P[3];
			d_stage_triangle(verts, tris, (u32)id, 0.0f, 0.0f, 0.0f, tri);
			const float *e1 = &tri[3], *e2 = &tri[6];
#pragma unroll
			for (int k = 0; k < 3; k++) {
				d[k] = dir_list[p * 3 + k];
				P[k] = cam_pos[k] + t * d[k];
			}
			D_CROSS(nn, e1, e2);
			D_NORMALIZE(nn);
			float dn = D_DOT(d, nn);
			if (dn > 0) {
				nn[0] = -nn[0];
				nn[1] = -nn[1];
				nn[2] = -nn[2];
				dn = -dn;
			}
#pragma unroll
			for (int k = 0; k < 3; k++) {
				out[k] = P[k] + eps * nn[k];
				out[3 + k] = d[k] - (2.0f * dn) * nn[k];
			}
			act = 1;
		}
	}
#pragma unroll
	for (int k = 0; k < 6; k++)
		rays[p * 6 + k] = out[k];
	active[p] = act;
}

extern "C" int ugrt_reflect_rays(ugrt_ctx *ctx, const float *d_cam_position, const float *d_t_value,
				 const float *d_ray_dir, const int *d_intersect_id, const int *d_mat_idx,
				 const float *d_reflect, int num_materials, const float *d_vertlist, const int *d_trilist,
				 float eps, float *d_rays, int *d_active)
{
	if (!ctx || !d_cam_position || !d_t_value || !d_ray_dir || !d_intersect_id || !d_mat_idx || !d_reflect ||
	    !d_vertlist || !d_trilist || !d_rays || !d_active)
		return ugrt_fail(UGRT_EINVAL, "reflect_rays: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_REFLECT_GEN);
	hipLaunchKernelGGL(k_reflect_rays, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, d_cam_position, d_t_value, d_ray_dir, d_intersect_id, d_mat_idx, d_reflect,
			   num_materials, d_vertlist, d_trilist, eps, d_rays, d_active, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_REFLECT_GEN);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

struct ReflIn {
	const float *reflect;
	const float *verts;
	const int *tris;
	const float *rays;
	const int *active;
	const float *hit_t;
	const int *hit_id;
};

__global__ __launch_bounds__(PX_THREADS) void k_shade_reflect(CamBlock cam, unsigned char *__restrict__ d_img,
							       const float *__restrict__ dd_normal,
							       const float *__restrict__ dd_t_value,
							       const float *__restrict__ dd_dir,
							       int *__restrict__ dd_intersect_id,
							       const float *__restrict__ d_cam_pos,
							       const int *__restrict__ mat_idx,
							       const float *__restrict__ mat_list, int mat_count, ReflIn in,
							       int p0, int n)
{
	int i = blockIdx.x * PX_THREADS + threadIdx.x;
	if (i >= n)
		return;
	int pixelID = p0 + i;
	float color[3] = { 0.0f, 0.0f, 0.0f };
	int tri = dd_intersect_id[pixelID];
	int idx = tri >= 0 ? mat_idx[tri] : tri;
	dd_intersect_id[pixelID] = idx;
	if (idx >= 0 && idx < mat_count) {
		float t_value = dd_t_value[pixelID];
		float material[6];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			material[k] = mat_list[idx * 6 + 3 + k];
			material[3 + k] = mat_list[idx * 6 + 3 + k];
		}
		if (t_value > 0) {
			float point[3], nrm[3];
#pragma unroll
			for (int k = 0; k < 3; k++) {
				point[k] = d_cam_pos[k] + t_value * dd_dir[pixelID * 3 + k];
				nrm[k] = dd_normal[pixelID * 3 + k];
			}
			d_lambert<false>(cam, point, nrm, color, material, 1.0f);
#pragma unroll
			for (int k = 0; k < 3; k++)
				color[k] = color[k] > 1.0f ? 1.0f : color[k];
		}
		if (in.active[pixelID]) {
			float kr = in.reflect[idx], rc[3] = { 0.0f, 0.0f, 0.0f };
			int hid = in.hit_id[pixelID];
			if (hid >= 0) {
				int hm = mat_idx[hid];
				if (hm >= 0 && hm < mat_count) {
					float t9[9], nn[3], hp[3], hmat[6];
					float ht = in.hit_t[pixelID];
					d_stage_triangle(in.verts, in.tris, (u32)hid, 0.0f, 0.0f, 0.0f, t9);
					float *e1 = &t9[3], *e2 = &t9[6];
#pragma unroll
					for (int k = 0; k < 3; k++) {
						hp[k] = in.rays[pixelID * 6 + k] + ht * in.rays[pixelID * 6 + 3 + k];
						hmat[k] = mat_list[hm * 6 + 3 + k];
						hmat[3 + k] = mat_list[hm * 6 + 3 + k];
					}
					D_NORMALIZE(e1);
					D_NORMALIZE(e2);
					D_CROSS(nn, e1, e2);
					D_NORMALIZE(nn);
					d_lambert<false>(cam, hp, nn, rc, hmat, 1.0f);
#pragma unroll
					for (int k = 0; k < 3; k++)
						rc[k] = rc[k] > 1.0f ? 1.0f : rc[k];
				}
			}
#pragma unroll
			for (int k = 0; k < 3; k++)
				color[k] = (1.0f - kr) * color[k] + kr * rc[k];
		}
	}
	d_img[pixelID * 3 + 0] = d_to_u8(color[0]);
	d_img[pixelID * 3 + 1] = d_to_u8(color[1]);
	d_img[pixelID * 3 + 2] = d_to_u8(color[2]);
}

extern "C" int ugrt_shade_reflect(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal, const float *d_t_value,
				  const float *d_ray_dir, int *d_intersect_id, const float *d_cam_position,
				  const int *d_mat_idx, const float *d_mat_list, const float *d_reflect, int num_materials,
				  const float *d_vertlist, const int *d_trilist, const float *d_rays, const int *d_active,
				  const float *d_hit_t, const int *d_hit_id)
{
	int rc = shade_args_ok(ctx, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position, d_mat_idx,
			       d_mat_list, "shade_reflect");
	if (rc)
		return rc;
	if (!d_reflect || !d_vertlist || !d_trilist || !d_rays || !d_active || !d_hit_t || !d_hit_id)
		return ugrt_fail(UGRT_EINVAL, "shade_reflect: null argument");
	UGRT_HIP(hipSetDevice(ctx->device));
	ReflIn in = { d_reflect, d_vertlist, d_trilist, d_rays, d_active, d_hit_t, d_hit_id };
	ugrt_prof_begin(ctx, UGRT_ST_SHADE);
	hipLaunchKernelGGL(k_shade_reflect, dim3((ctx->npix + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0,
			   ctx->stream, ctx->cam, d_img, d_normal, d_t_value, d_ray_dir, d_intersect_id, d_cam_position,
			   d_mat_idx, d_mat_list, num_materials, in, ctx->p0, ctx->npix);
	ugrt_prof_end(ctx, UGRT_ST_SHADE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// copy_data_transform, transformation_kernel.cu:4-18.  cosf/sinf of the frame's
// angle are evaluated once on the host (the reference evaluates them per vertex
// per component) and enter the kernel as scalars.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PX_THREADS) void k_animate(float *__restrict__ vertexlist,
							 const float *__restrict__ orig_list, int size, int offset,
							 float cr, float sr)
{
	int vert = blockIdx.x * PX_THREADS + threadIdx.x;
	if (vert >= size)
		return;
	float x = ((orig_list[vert * 3 + 0] - 12.0f) / 12.0f);
	float y = ((orig_list[vert * 3 + 1] - 11.0f) / 12.0f);
	float z = ((orig_list[vert * 3 + 2] - 4.5f) / 12.0f);
	vertexlist[(offset + vert) * 3 + 0] = (x * cr - y * sr) * 9.0f + 14.5f;
	vertexlist[(offset + vert) * 3 + 1] = (x * sr + y * cr) * 9.0f + 13.0f;
	vertexlist[(offset + vert) * 3 + 2] = z * 9.0f + 4.0f;
}

// Model::rotate_bunny, scene.h:122-139
extern "C" int ugrt_animate(ugrt_ctx *ctx, float *d_vertlist, const float *d_orig_list, int size, int offset,
			    float rot_factor)
{
	if (!ctx || !d_vertlist || !d_orig_list || size < 0 || offset < 0)
		return ugrt_fail(UGRT_EINVAL, "animate: bad argument");
	if (size == 0)
		return UGRT_OK;
	float c, s;
	ugrt_rot_cos_sin(rot_factor, &c, &s);
	ctx->rec_valid = false; // vertices change: the triangle records are stale until the next grid build
	UGRT_HIP(hipSetDevice(ctx->device));
	ugrt_prof_begin(ctx, UGRT_ST_ANIMATE);
	hipLaunchKernelGGL(k_animate, dim3((size + PX_THREADS - 1) / PX_THREADS), dim3(PX_THREADS), 0, ctx->stream,
			   d_vertlist, d_orig_list, size, offset, c, s);
	ugrt_prof_end(ctx, UGRT_ST_ANIMATE);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}
