// ugrt_context.hip -- device context: buffers, stream, camera block, profiler.
//
// The reference keeps this state in globals and per-class members
// (main.cu.h:58-63, frustum_grid.h:31-66, decision_data.h:13-40) and
// re-allocates the ref-sized lists on every build (frustum_grid.h:260-272).
// Here one context owns grow-only device buffers; nothing is allocated in the
// steady state of a frame loop.
#include "ugrt_ctx.h"

static int check_device(int device)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return ugrt_fail(UGRT_ENODEV, "no HIP device available (%s); libugrt has no CPU fallback",
				 e != hipSuccess ? hipGetErrorString(e) : "device count 0");
	if (device < 0 || device >= n)
		return ugrt_fail(UGRT_EINVAL, "device %d out of range (%d devices)", device, n);
	return UGRT_OK;
}

int ugrt_buf_reserve(ugrt_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap && b.p)
		return UGRT_OK;
	size_t want = bytes + bytes / 2;
	if (want < 256)
		want = 256;
	void *np = nullptr;
	hipError_t e = hipMalloc(&np, want);
	if (e != hipSuccess)
		return ugrt_fail(UGRT_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
	if (b.p) {
		// the old buffer may still be in use by enqueued work
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipFree(b.p);
	}
	b.p = np;
	b.cap = want;
	return UGRT_OK;
}

static void buf_free(DevBuf &b)
{
	if (b.p)
		(void)hipFree(b.p);
	b.p = nullptr;
	b.cap = 0;
}

extern "C" int ugrt_ctx_create(ugrt_ctx **out, int device, const ugrt_config *cfg)
{
	if (!out || !cfg)
		return ugrt_fail(UGRT_EINVAL, "ctx_create: null argument");
	int rc = check_device(device);
	if (rc)
		return rc;
	if (cfg->tile != 8)
		return ugrt_fail(UGRT_EINVAL, "tile must be 8 (one 8x8 tile = one wavefront; main.cu.h:25-26)");
	if (cfg->slabs < 1 || cfg->slabs > 64)
		return ugrt_fail(UGRT_EINVAL, "slabs must be within [1,64] (NUM_SLABS, main.cu.h:18)");
	if (cfg->width <= 0 || cfg->height <= 0 || cfg->width % 8 || cfg->height % 8)
		return ugrt_fail(UGRT_EINVAL, "width and height must be positive multiples of 8");
	int nbx = cfg->width / 8, nby = cfg->height / 8;
	if (nbx > 65535 || nby > 65535)
		return ugrt_fail(UGRT_EINVAL, "image too large");
	if (cfg->row_begin < 0 || cfg->row_end > nby || cfg->row_begin >= cfg->row_end)
		return ugrt_fail(UGRT_EINVAL, "row band [%d,%d) outside [0,%d)", cfg->row_begin, cfg->row_end, nby);
	if (cfg->light_nbx < 2 || cfg->light_nby < 2 || (cfg->light_nbx & 1) || (cfg->light_nby & 1) ||
	    cfg->light_nbx > 4096 || cfg->light_nby > 4096)
		return ugrt_fail(UGRT_EINVAL, "light grid must be even and within [2,4096]^2");
	for (int k = 0; k < 3; k++)
		if (cfg->uniform_dims[k] < 1 || cfg->uniform_dims[k] > 1024)
			return ugrt_fail(UGRT_EINVAL, "uniform_dims[%d] must be within [1,1024]", k);
	UGRT_HIP(hipSetDevice(device));
	ugrt_ctx *ctx = new ugrt_ctx();
	ctx->cfg = *cfg;
	ctx->device = device;
	for (int i = 0; i < UGRT_OPT_COUNT; i++)
		ctx->opt[i] = -1;
	ctx->nbx = nbx;
	ctx->nby = nby;
	ctx->p0 = cfg->row_begin * 8 * cfg->width;
	ctx->npix = (cfg->row_end - cfg->row_begin) * 8 * cfg->width;
	memset(&ctx->cam, 0, sizeof(ctx->cam));
	ctx->cam.W = cfg->width;
	ctx->cam.H = cfg->height;
	ctx->cam.nbx = nbx;
	ctx->cam.nby = nby;
	ctx->cam.strict_tex = (cfg->flags & UGRT_FLAG_STRICT_TEXTURE) ? 1 : 0;
	hipError_t e = hipHostMalloc((void **)&ctx->h_pinned, UGRT_PIN_WORDS * sizeof(u32), hipHostMallocDefault);
	if (e == hipSuccess)
		memset(ctx->h_pinned, 0, UGRT_PIN_WORDS * sizeof(u32));
	if (e == hipSuccess)
		e = hipMalloc((void **)&ctx->d_small, UGRT_DSMALL_WORDS * sizeof(u32));
	if (e == hipSuccess)
		e = hipMemset(ctx->d_small, 0, UGRT_DSMALL_WORDS * sizeof(u32));
	if (e != hipSuccess) {
		ugrt_ctx_destroy(ctx);
		return ugrt_fail(UGRT_EHIP, "ctx_create: %s", hipGetErrorString(e));
	}
	// per-pixel (t,ref) merge slots of split cells: all ones = "no hit"
	rc = ugrt_buf_reserve(ctx, ctx->best, (size_t)ctx->npix * sizeof(u64));
	if (!rc) {
		e = hipMemset(ctx->best.p, 0xFF, (size_t)ctx->npix * sizeof(u64));
		if (e != hipSuccess)
			rc = ugrt_fail(UGRT_EHIP, "ctx_create: %s", hipGetErrorString(e));
	}
	if (rc) {
		ugrt_ctx_destroy(ctx);
		return rc;
	}
	*out = ctx;
	return UGRT_OK;
}

extern "C" void ugrt_ctx_destroy(ugrt_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	for (int g = 0; g < 3; g++) {
		Grid &G = ctx->grid[g];
		buf_free(G.rng);
		buf_free(G.sizes);
		buf_free(G.scan);
		buf_free(G.wide);
		buf_free(G.parts);
		buf_free(G.key[0]);
		buf_free(G.key[1]);
		buf_free(G.val[0]);
		buf_free(G.val[1]);
		buf_free(G.span);
		buf_free(G.offset);
		buf_free(G.projz);
		buf_free(G.uspan);
		}
	buf_free(ctx->temp);
	buf_free(ctx->rs_state);
	for (int j = 0; j < 2; j++) {
		buf_free(ctx->rs_tmp[j][0]);
		buf_free(ctx->rs_tmp[j][1]);
	}
	buf_free(ctx->trirec);
	buf_free(ctx->witems);
	buf_free(ctx->wscan);
	buf_free(ctx->ubitmap);
	buf_free(ctx->scan_state);
	buf_free(ctx->dsort);
	buf_free(ctx->dsplit);
	buf_free(ctx->best);
	buf_free(ctx->rmap[0]);
	buf_free(ctx->rmap[1]);
	buf_free(ctx->rstart);
	buf_free(ctx->cbase);
	buf_free(ctx->skey[0]);
	buf_free(ctx->skey[1]);
	buf_free(ctx->sval[0]);
	buf_free(ctx->sval[1]);
	buf_free(ctx->sdesc);
	buf_free(ctx->sstart);
	buf_free(ctx->sbase);
	buf_free(ctx->tkey[0]);
	buf_free(ctx->tkey[1]);
	buf_free(ctx->tval[0]);
	buf_free(ctx->tval[1]);
	buf_free(ctx->tbcnt);
	buf_free(ctx->sitem);
	buf_free(ctx->citem);
	buf_free(ctx->pseg);
	buf_free(ctx->sray);
	if (ctx->h_pinned)
		(void)hipHostFree(ctx->h_pinned);
	if (ctx->d_small)
		(void)hipFree(ctx->d_small);
	for (int s = 0; s < UGRT_ST_COUNT; s++)
		for (auto &p : ctx->prof[s]) {
			(void)hipEventDestroy(p.a);
			(void)hipEventDestroy(p.b);
		}
	for (auto &p : ctx->prof_pool) {
		(void)hipEventDestroy(p.a);
		(void)hipEventDestroy(p.b);
	}
	delete ctx;
}

// option keys in the order of the UGRT_OPT_* enum, with their accepted ranges
static const struct {
	const char *key;
	int lo, hi;
} k_opt[UGRT_OPT_COUNT] = {
	{ "dda_rays_per_wave", 0, 64 }, { "dda_coop", 1, 1 << 30 },     { "dda_kernel", 0, 1 },
	{ "dda_cull_min", 1, 1 << 30 }, { "dda_blocks", 1, 1 << 20 },
	{ "primary_seg", 64, 1 << 20 }, { "shadow_beam", 64, 8192 },
	{ "shadow_xseg", 64, 1 << 20 }, { "shadow_sizebits", 0, 8 },    { "shadow_itemsort", 0, 1 },
	{ "shadow_mbits", 1, 24 },      { "shadow_key64", 0, 1 },       { "sort_library", 0, 1 },
	{ "async_build", 0, 1 },        { "primary_waves", 64, 1 << 20 },
	{ "shadow_waves", 64, 1 << 20 }, { "dda_sort", 0, 1 }, { "primary_order", 0, 1 }, { "primary_chunk", 4, 64 }, { "sort_items", 8, 16 }, { "dda_cull_work", 1, 1 << 30 },
	{ "dda_split", 0, 4 }, { "dda_split_load", 50, 100000 }, { "dda_split_segments", 1, 4 }, { "primary_xcd_run", 0, 4096 }, { "shadow_xcd_run", 0, 4096 },
	{ "primary_centre", 0, 1 }, { "sort_rank", 0, 1 }, { "ray_sort", 0, 1 }, { "shadow_sieve", 0, 64 },
};

extern "C" int ugrt_ctx_get_state(ugrt_ctx *ctx, const char *key, long long *value)
{
	if (!ctx || !key || !value)
		return ugrt_fail(UGRT_EINVAL, "ctx_get_state: null argument");
	if (strcmp(key, "radix_launches") == 0)
		*value = (long long)ctx->rs_launches;
	else if (strcmp(key, "sort_rank_atomic") == 0)
		*value = ctx->rs_atomic_rank < 0 ? -1 : (ctx->rs_atomic_rank == 1 && ctx->opt[UGRT_OPT_SORT_RANK] != 0 ? 1 : 0);
	else if (strcmp(key, "recip_mismatches") == 0) {
		// exhaustive: every float through the tracers' short reciprocal against the division (ugrt_dev.h d_recip_det)
		unsigned long long bad = 0;
		const int rc = ugrt_recip_selftest(ctx, &bad);
		if (rc != UGRT_OK)
			return rc;
		*value = (long long)bad;
	} else if (strcmp(key, "f2i_mismatches") == 0) {
		// the one-instruction device forms of ugrt_f2i / ugrt_f2u / ugrt_floor2i against the portable ones, every float
		unsigned long long bad = 0;
		const int rc = ugrt_f2i_selftest(ctx, &bad);
		if (rc != UGRT_OK)
			return rc;
		*value = (long long)bad;
	} else if (strcmp(key, "lane_reduce_mismatches") == 0) {
		// the DPP / permlane-swap reductions of the tracers against the same by __shfl_xor (ugrt_packet.h)
		unsigned long long bad = 0;
		const int rc = ugrt_lane_reduce_selftest(ctx, &bad);
		if (rc != UGRT_OK)
			return rc;
		*value = (long long)bad;
	} else
		return ugrt_fail(UGRT_EINVAL, "ctx_get_state: unknown key '%s'", key);
	return UGRT_OK;
}

extern "C" int ugrt_ctx_set_option(ugrt_ctx *ctx, const char *key, int value)
{
	if (!ctx || !key)
		return ugrt_fail(UGRT_EINVAL, "ctx_set_option: null argument");
	for (int i = 0; i < UGRT_OPT_COUNT; i++)
		if (strcmp(key, k_opt[i].key) == 0) {
			if (value >= 0 && (value < k_opt[i].lo || value > k_opt[i].hi))
				return ugrt_fail(UGRT_EINVAL, "ctx_set_option: %s %d outside [%d,%d]", key, value, k_opt[i].lo,
						 k_opt[i].hi);
			// 0 keeps meaning "default" for the rays-per-wave option of version 100
			ctx->opt[i] = (value < 0 || (i == UGRT_OPT_DDA_RPW && value == 0)) ? -1 : value;
			return UGRT_OK;
		}
	return ugrt_fail(UGRT_EINVAL, "ctx_set_option: unknown key '%s'", key);
}

extern "C" int ugrt_ctx_set_stream(ugrt_ctx *ctx, void *hip_stream)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "set_stream: null ctx");
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	ctx->stream = (hipStream_t)hip_stream;
	return UGRT_OK;
}

extern "C" int ugrt_ctx_synchronize(ugrt_ctx *ctx)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "synchronize: null ctx");
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	// asynchronous builds / shadow passes: a count that did not fit the capacity it was given (sized by the call
	// before) left that call's results incomplete.  Reported once; the next build and shadow pass run in the
	// synchronous form and size the buffers exactly.
	if (ugrt_reported_status(ctx) != 0u || ctx->overflow_seen) {
		const unsigned bits = ugrt_reported_status(ctx);
		ctx->overflow_seen = false;
		for (int g = 0; g < 3; g++)
			ctx->h_pinned[UGRT_PIN_REPORT + 4 * g + 3] = 0u;
		ctx->h_pinned[UGRT_PIN_SHADOW + 2] = 0u;
		UGRT_HIP(hipMemsetAsync(ctx->d_small + UGRT_DSMALL_STATUS, 0, 4, ctx->stream));
		UGRT_HIP(hipStreamSynchronize(ctx->stream));
		for (int g = 0; g < 3; g++)
			ctx->grid[g].have_est = false;
		ctx->have_shadow_est = false;
		return ugrt_fail(UGRT_EOVERFLOW, "an asynchronous call needed more room than its estimate gave it (status %u: "
						  "1 grid build, 2 shadow candidate pairs, 4 shadow work items): the frames since the last "
						  "synchronisation are incomplete, repeat them", bits);
	}
	return UGRT_OK;
}

// the direction table travels as a by-value kernel argument and is written to
// device memory in stream order, so back-to-back camera changes cannot race
struct TexArg {
	float v[100];
};
__global__ void k_store_table(TexArg t, float *dst)
{
	int i = threadIdx.x;
	if (i < 100)
		dst[i] = t.v[i];
}

// device copy of the direction table of the current camera; written (in stream order) when a tracer first asks for
// it: the light's camera block is uploaded every frame too, but only the camera pass samples the table
float *ugrt_ctx_tex(ugrt_ctx *ctx)
{
	float *dst = (float *)(ctx->d_small + UGRT_DSMALL_TEX);
	if (ctx->tex_dirty) {
		TexArg t;
		memcpy(t.v, ctx->tex_host, sizeof(t.v));
		hipLaunchKernelGGL(k_store_table, dim3(1), dim3(128), 0, ctx->stream, t, dst);
		ctx->tex_dirty = false;
	}
	return dst;
}

// per_frame_funcs.h:18-43 fillCoordinatesData (+ setDirectionTexture :161)
extern "C" int ugrt_upload_camera(ugrt_ctx *ctx, const float camcoords[64])
{
	if (!ctx || !camcoords)
		return ugrt_fail(UGRT_EINVAL, "upload_camera: null argument");
	memcpy(ctx->cam.cc, camcoords, sizeof(float) * 64);
	int rc = ugrt_camera_direction_table(camcoords, ctx->tex_host);
	if (rc)
		return rc;
	ctx->tex_dirty = true; // stored on the device by the next primary trace (ugrt_ctx_tex)
	return UGRT_OK;
}

// per_frame_funcs.h:6-16 updateLightPosition
extern "C" int ugrt_set_light_position(ugrt_ctx *ctx, const float pos[3])
{
	if (!ctx || !pos)
		return ugrt_fail(UGRT_EINVAL, "set_light_position: null argument");
	ctx->cam.light[0] = pos[0];
	ctx->cam.light[1] = pos[1];
	ctx->cam.light[2] = pos[2];
	ctx->cam.light[3] = 0.0f;
	return UGRT_OK;
}

extern "C" int ugrt_grid_get_info(ugrt_ctx *ctx, int which, ugrt_grid_info *out)
{
	if (!ctx || !out || which < 0 || which > 2)
		return ugrt_fail(UGRT_EINVAL, "grid_get_info: bad argument");
	Grid &G = ctx->grid[which];
	if (!G.valid)
		return ugrt_fail(UGRT_EINVAL, "grid_get_info: grid %d has not been built", which);
	// pointers and total_refs are known on the host since the build; cells_used is fetched lazily and
	// is final after the next ugrt_ctx_synchronize (the reference only prints it, frustum_grid.h:338)
	out->d_triangle_value_list = G.vals;
	out->d_triangle_key_list = G.keys;
	out->d_span = (unsigned *)G.span.p;
	out->d_offset = (unsigned *)G.offset.p;
	out->total_refs = G.R;
	if (!G.r_exact) // asynchronous build: what the build reported (final once the stream has been synchronised)
		out->total_refs = (unsigned)(ctx->h_pinned[UGRT_PIN_REPORT + 4 * which] +
					     G.active_cells * ctx->h_pinned[UGRT_PIN_REPORT + 4 * which + 1]);
	out->num_cells = G.C;
	out->cells_used = G.r_exact ? ctx->h_pinned[UGRT_PIN_CELLS_USED + which] : ctx->h_pinned[UGRT_PIN_REPORT + 4 * which + 2];
	return UGRT_OK;
}

// the z-slab stage of a build: DSKernel's projCoordZ, the host loop's zMin/zMax (frustum_grid.h:221-241),
// SlabKernel's zList
extern "C" int ugrt_grid_get_slabs(ugrt_ctx *ctx, int which, ugrt_slab_info *out)
{
	if (!ctx || !out || which < 0 || which > 1)
		return ugrt_fail(UGRT_EINVAL, "grid_get_slabs: bad argument (the perspective and the spherical grid have slabs)");
	Grid &G = ctx->grid[which];
	if (!G.valid)
		return ugrt_fail(UGRT_EINVAL, "grid_get_slabs: grid %d has not been built", which);
	out->slabs = G.slabs;
	out->d_proj_coord_z = nullptr;
	out->z_min = out->z_max = 0.0f;
	if (G.slabs > 1) {
		int z[2];
		UGRT_HIP(hipSetDevice(ctx->device));
		UGRT_HIP(hipMemcpyAsync(z, (const char *)G.projz.p + (size_t)G.F * 4, 8, hipMemcpyDeviceToHost,
					ctx->stream));
		UGRT_HIP(hipStreamSynchronize(ctx->stream));
		for (int k = 0; k < 2; k++) {
			const int b = z[k] ^ ((z[k] >> 31) & 0x7FFFFFFF);
			memcpy(k ? &out->z_max : &out->z_min, &b, 4);
		}
		out->d_proj_coord_z = (float *)G.projz.p;
	}
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// profiler: hipEvent pairs on the context's stream
// ---------------------------------------------------------------------------
void ugrt_prof_begin(ugrt_ctx *ctx, int stage)
{
	if (!((ctx->prof_mask >> stage) & 1u))
		return;
	ProfPair p;
	if (!ctx->prof_pool.empty()) {
		p = ctx->prof_pool.back();
		ctx->prof_pool.pop_back();
	} else {
		if (hipEventCreate(&p.a) != hipSuccess)
			return;
		if (hipEventCreate(&p.b) != hipSuccess) {
			(void)hipEventDestroy(p.a);
			return;
		}
	}
	(void)hipEventRecord(p.a, ctx->stream);
	ctx->prof[stage].push_back(p);
}

void ugrt_prof_end(ugrt_ctx *ctx, int stage)
{
	if (!((ctx->prof_mask >> stage) & 1u) || ctx->prof[stage].empty())
		return;
	(void)hipEventRecord(ctx->prof[stage].back().b, ctx->stream);
}

extern "C" int ugrt_prof_enable(ugrt_ctx *ctx, int on)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "prof_enable: null ctx");
	// 0: off, 1: every stage, otherwise bit (s + 1) selects stage s (event pairs cost a few
	// microseconds each, so a timed loop may want only the kernels it reports)
	ctx->prof_mask = on == 0 ? 0u : (on == 1 ? 0xFFFFFFFFu : ((unsigned)on >> 1));
	return UGRT_OK;
}

extern "C" int ugrt_prof_reset(ugrt_ctx *ctx)
{
	if (!ctx)
		return ugrt_fail(UGRT_EINVAL, "prof_reset: null ctx");
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	for (int s = 0; s < UGRT_ST_COUNT; s++) {
		for (auto &p : ctx->prof[s])
			ctx->prof_pool.push_back(p);
		ctx->prof[s].clear();
	}
	return UGRT_OK;
}

extern "C" int ugrt_prof_get(ugrt_ctx *ctx, int stage, double *ms_total, int *launches)
{
	if (!ctx || stage < 0 || stage >= UGRT_ST_COUNT)
		return ugrt_fail(UGRT_EINVAL, "prof_get: bad argument");
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	double tot = 0.0;
	int n = 0;
	for (auto &p : ctx->prof[stage]) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
			tot += ms;
			n++;
		}
	}
	if (ms_total)
		*ms_total = tot;
	if (launches)
		*launches = n;
	return UGRT_OK;
}

extern "C" int ugrt_stats_get(ugrt_ctx *ctx, unsigned long long stats[8])
{
	if (!ctx || !stats)
		return ugrt_fail(UGRT_EINVAL, "stats_get: null argument");
	UGRT_HIP(hipStreamSynchronize(ctx->stream));
	memcpy(stats, ctx->stats, sizeof(ctx->stats));
	if (ctx->shadow_async_pending)
		stats[1] = ctx->h_pinned[UGRT_PIN_SHADOW + 1];
	if (ctx->stats[2]) { // the shadow tracer ran: its work counters were copied to pinned memory
		memcpy(&stats[6], ctx->h_pinned + UGRT_PIN_SHADOW_WORK, 8);
		memcpy(&stats[7], ctx->h_pinned + UGRT_PIN_SHADOW_WORK + 2, 8);
	}
	return UGRT_OK;
}
