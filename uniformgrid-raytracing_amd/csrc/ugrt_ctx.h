// ugrt_ctx.h -- device context of libugrt.so (HIP translation units only)
#ifndef UGRT_CTX_H
#define UGRT_CTX_H

#include <hip/hip_runtime.h>

#include <vector>

#include "ugrt_internal.h"

// The reference keeps the camera in __constant__ dd_camcoords[64], the light in
// dd_light_position[6] and the ray targets in a 5x5 texture (main.cu.h:58-63).
// Here the whole block (688 B) is a by-value kernel argument: it lives in the
// kernarg segment and is read with scalar loads into SGPRs.
struct CamBlock {
	float cc[64];   // dd_camcoords
	float light[4]; // dd_light_position[0..2]
	int W, H, nbx, nby;
	int strict_tex; // UGRT_FLAG_STRICT_TEXTURE: the direction fetch with the texture unit's 8-bit weights
};

// launch-shape options (ugrt_ctx_set_option): none changes a result
enum {
	UGRT_OPT_DDA_RPW = 0,      // "dda_rays_per_wave"
	UGRT_OPT_DDA_COOP,         // "dda_coop": list length from which a lone ray's cell is tested by the whole wave
	UGRT_OPT_DDA_KERNEL,       // "dda_kernel": 0 = window kernel, 1 = per-ray kernel of round 1 (the cross-check)
	UGRT_OPT_DDA_CULL_MIN,     // "dda_cull_min": list length from which a shared cell is culled before the exact tests
	UGRT_OPT_DDA_BLOCKS,       // "dda_blocks": upper bound of the persistent waves of ugrt_trace_dda
	UGRT_OPT_PRIMARY_SEG,      // "primary_seg"
	UGRT_OPT_SHADOW_BEAM,      // "shadow_beam"
	UGRT_OPT_SHADOW_XSEG,      // "shadow_xseg"
	UGRT_OPT_SHADOW_SIZEBITS,  // "shadow_sizebits"
	UGRT_OPT_SHADOW_ITEMSORT,  // "shadow_itemsort"
	UGRT_OPT_SHADOW_MBITS,     // "shadow_mbits"
	UGRT_OPT_SHADOW_KEY64,     // "shadow_key64"
	UGRT_OPT_SORT_LIBRARY,     // "sort_library": 1 = rocPRIM radix sort instead of the built-in one
	UGRT_OPT_ASYNC_BUILD,      // "async_build": 1 = grid builds and the shadow tracer never wait for the device
	UGRT_OPT_PRIMARY_WAVES,    // "primary_waves": single-wave workgroups of the primary tracer
	UGRT_OPT_SHADOW_WAVES,     // "shadow_waves": the same for the two shadow kernels
	UGRT_OPT_DDA_SORT,         // "dda_sort": 1 = the bounce's ray list is sorted by (entry cell, octant) instead of tile order
	UGRT_OPT_PRIMARY_ORDER,    // "primary_order": 0 = a flush's jobs run in list order (default 1: nearest triangles first)
	UGRT_OPT_PRIMARY_CHUNK,    // "primary_chunk": jobs between two looks at the rays' closest hits (4..64)
	UGRT_OPT_SORT_ITEMS,       // "sort_items": pairs per thread of a radix pass, 16 (tiles of 8192) or 8 (4096)
	UGRT_OPT_DDA_CULL_WORK,    // "dda_cull_work": window kernel: (triangles x rays) of a job from which its list is culled first
	UGRT_OPT_DDA_SPLIT,        // "dda_split": window kernel: 0 = no split walks, 1 = long groups of the last launch cut into segments (default), 2..4 = every group (tests)
	UGRT_OPT_DDA_SPLIT_LOAD,   // "dda_split_load": jobs of a group, in percent of the average group's, per segment it is cut into (default 400)
	UGRT_OPT_DDA_SPLIT_SEGMENTS, // "dda_split_segments": segments a group is cut into at most (1..4; 1 = none is cut, the long groups are only started first)
	UGRT_OPT_PRIMARY_XCD_RUN,  // "primary_xcd_run": primary tracer, one wave per item: neighbouring items per XCD in turn (default 128; 0 = one)
	UGRT_OPT_SHADOW_XCD_RUN,   // "shadow_xcd_run": exact shadow pass: one wave per item, this many neighbouring items per XCD in turn (default 128); 0 = the persistent waves of round 2
	UGRT_OPT_PRIMARY_CENTRE,   // "primary_centre": primary tracer, one wave per item: 0 = the runs of items in list order (default 1: from the middle of the list outwards)
	UGRT_OPT_SORT_RANK,        // "sort_rank": radix pass: 0 = ranks by ballots, 1 / default = by LDS atomics where the device's self-test allows it
	UGRT_OPT_RAY_SORT,         // "ray_sort": 1 = the deferred ugrt_sort_rays sorts at once; 0 / default = on demand (see ugrt_sort_rays)
	UGRT_OPT_SHADOW_SIEVE,     // "shadow_sieve": items a sieve wave of the exact shadow pass looks at (default 16; 0 / 1 = a wave per item)
	UGRT_OPT_COUNT
};

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
};

struct Grid {
	DevBuf rng, sizes, scan;          // per triangle
	DevBuf parts;                     // fill: triangle of the first reference of every workgroup
	DevBuf wide;                      // ids of the triangles that cover every cell: [F] as found, [F] ascending
	DevBuf key[2], val[2];            // per ref, ping-pong for the radix sort
	DevBuf span, offset;              // per cell (span buffer also holds run starts + cells_used)
	DevBuf projz;                     // NUM_SLABS > 1: projCoordZ per triangle, then {zMin, zMax} as ordered integers
	DevBuf uspan;                     // NUM_SLABS > 1: span/offset of a cell's slabs taken together (shadow tracer)
	int slabs = 1, F = 0;
	// asynchronous builds: what the grid needed last time (narrow references, wide triangles)
	u32 est_rn = 0, est_w = 0;
	bool have_est = false, async_pending = false, r_exact = true;
	unsigned long long active_cells = 0;
	u32 *keys = nullptr, *vals = nullptr; // sorted result (one of key[i]/val[i])
	u32 R = 0, C = 0, cells_used = 0;
	int dims[3] = { 0, 0, 0 };
	float ug[12] = { 0 }; // uniform grid: lo[3], cell[3], inv cell[3]
	bool valid = false;
	const u32 *wide_zeroed = nullptr; // the wide-triangle counter the last build's scan left at zero (build_prologue)
};

// layout of ugrt_ctx::d_small (u32 words of device scratch)
#define UGRT_DSMALL_DDA_RAYS 2     // secondary rays in the DDA's ray list
#define UGRT_DSMALL_WIDE 3         // wide triangles of the running build
#define UGRT_DSMALL_DDA_RAYS_B 4   // the DDA's other ray counter (two in turn: a launch clears the next one's)
#define UGRT_DSMALL_TICKET 8       // group ticket of the beam DDA
#define UGRT_DSMALL_STATUS 14      // status bits of asynchronous calls
#define UGRT_DSMALL_SHADOW_WORK 16 // u64 [2]: cull tests, staged candidates of the shadow pass (FLAG_COUNT_WORK)
#define UGRT_DSMALL_PAIRS 20       // candidate pairs of the shadow pass (cleared with the two counters before it)
#define UGRT_DSMALL_TEX 32         // [100] the 5x5x4 direction table
#define UGRT_DSMALL_DDA 132        // u64 [UGRT_DDA_STATS] work counters of the DDA
#define UGRT_DSMALL_RW 236         // [3][2] checked {narrow references, wide triangles} of an asynchronous build
#define UGRT_DSMALL_REPORT 242     // [3][4] report of a grid's asynchronous build: {narrow references, wide triangles} as
                                   // found (what the next build is sized by), cells used, the status word at its end
#define UGRT_DSMALL_SHADOW 254     // [10] shadow tracer: {pairs, beams} as checked, then its report: {pairs, beams} as
                                   // found, status word, pad, the two u64 work counters at the start of the exact pass
#define UGRT_DSMALL_PRIMARY 264    // u64 [UGRT_PRIMARY_STATS] work counters of the primary tracer (FLAG_COUNT_WORK)
#define UGRT_DSMALL_WORDS (132 + 104 + 28 + 32)
// layout of ugrt_ctx::h_pinned (u32 words of pinned host memory for small read-backs)
#define UGRT_PIN_RW 0           // {narrow references, wide triangles} of the running (waiting) build
#define UGRT_PIN_CELLS_USED 4   // [3] per grid
#define UGRT_PIN_REFS 8         // ugrt_refs_of: last span, last offset
#define UGRT_PIN_PAIRS 10       // candidate pairs of the shadow pass
#define UGRT_PIN_CHUNKS 11      // chunks of ugrt_sort_rays
#define UGRT_PIN_BEAMS 12       // beams of the shadow pass
#define UGRT_PIN_SHADOW_WORK (UGRT_PIN_SHADOW + 4) // u64 [2]: cull tests, staged candidates
#define UGRT_PIN_REPORT 32      // [3][4] copies of UGRT_DSMALL_REPORT (one copy per build)
#define UGRT_PIN_SHADOW 44      // [8] copy of the shadow tracer's report (UGRT_DSMALL_SHADOW + 2 ...)
#define UGRT_PIN_WORDS 64
#define UGRT_STATUS_BUILD_OVERFLOW 1u
#define UGRT_STATUS_PAIR_OVERFLOW 2u
#define UGRT_STATUS_ITEM_OVERFLOW 4u
#define UGRT_DDA_STATS 46
#define UGRT_PRIMARY_STATS 16
static_assert(UGRT_DSMALL_DDA + 2 * UGRT_DDA_STATS <= UGRT_DSMALL_RW, "the DDA's counters run into the build counts");
static_assert(UGRT_DSMALL_REPORT + 12 <= UGRT_DSMALL_SHADOW && UGRT_DSMALL_SHADOW + 10 <= UGRT_DSMALL_PRIMARY && UGRT_DSMALL_PRIMARY + 2 * UGRT_PRIMARY_STATS <= UGRT_DSMALL_WORDS && UGRT_PIN_SHADOW + 8 <= UGRT_PIN_WORDS && UGRT_PIN_REPORT + 12 <= UGRT_PIN_SHADOW, "scratch layout");

// an asynchronous grid build between its fill and its sort (ugrt_build.hip: build_async_begin / _sort / _end)
struct AsyncBuild {
	Grid *G;
	int F, ny, nz, ylo, yhi;
	u32 C, launchRn, capW, nparts;
	unsigned long long capR, active;
	bool no_wide;
	bool prehist; // the fill has counted the first digit of the keys for the sort (ugrt_rs_hist.h)
};

struct ProfPair {
	hipEvent_t a, b;
};

struct ugrt_ctx {
	ugrt_config cfg;
	int device = 0;
	hipStream_t stream = nullptr;
	CamBlock cam;
	float tex_host[100]; // 5x5x4 direction table of the current camera (setDirectionTexture)
	bool tex_dirty = false;
	int nbx = 0, nby = 0; // screen grid
	int face_lo = 0, face_hi = -1; // ugrt_ctx_set_face_window: triangles the light / uniform builds bin (hi < 0 = to the last)
	int p0 = 0, npix = 0; // first pixel and pixel count of this context's band
	Grid grid[3];
	DevBuf temp;                  // rocPRIM temporary storage
	DevBuf scan_state;            // own scan: ticket + done counter (64 B), then one epoch-tagged state word per tile
	u32 scan_epoch = 0;           // tag of the last scan's state words
	DevBuf rs_state, rs_tmp[2][2]; // own radix sort: histogram rows + tickets, look-back words; ping-pong buffers per list of a batch
	u32 rs_tiles = 0;             // tiles per list the look-back words are laid out for
	u32 rs_epoch = 0;             // tag of the last pass's look-back words
	bool batch_open = false;      // ugrt_grid_build_batch_begin: the builds that follow stop in front of their sorts
	int nbatch = 0;
	AsyncBuild batch[2];
	unsigned long long rs_launches = 0; // histogram + pass kernels enqueued so far (ugrt_ctx_get_state "radix_launches")
	bool rs_prehist = false;      // the first pass's histogram rows hold the counts of a producer whose sort has not run yet
	int rs_atomic_rank = -1;      // k_rs_selftest: 1 = LDS add-with-return serves equal addresses in lane order on this device
	// per-triangle records {v0, v1-v0, v2-v0} (48 B), rewritten by every grid build; the tracers
	// gather ONE record per reference instead of 3 indices + 3 vertices
	DevBuf trirec;
	const float *rec_verts = nullptr;
	const int *rec_tris = nullptr;
	int rec_faces = 0;
	bool rec_valid = false;
	DevBuf witems, wscan; // tracer work lists
	DevBuf ubitmap;               // bounce: occupancy bitmap of the uniform grid's cells (1 bit per cell)
	DevBuf dsort;                 // bounce, option dda_sort: keys + sorted keys + sorted list
	DevBuf dsplit;                // bounce, split walks: work items, the groups' job history, merge state (ugrt_dda_walk.hip)
	u32 dda_turn = 0;             // which of the two ray counters the last bounce used
	u32 dsplit_rpw = 0, dsplit_turn = 0; // rays per wave the history was laid out for; launches since
	DevBuf best;                  // u64 per pixel: (t bits << 32 | ref) for split cells
	DevBuf rmap[2];               // ray sort ping-pong (2n u32 each)
	DevBuf rstart, cbase; // ray runs per light cell (sort_rays)
	DevBuf skey[2], sval[2], sdesc, sstart, sbase; // shadow tracer: re-grouped rays, beams, counts
	DevBuf tkey[2], tval[2], tbcnt;                      // shadow tracer: candidate pairs, runs per beam
	DevBuf sray;                                         // shadow tracer: rebuilt rays {direction, distance}, beam order
	DevBuf citem;                                        // shadow tracer: the cull items (CullItem table)
	DevBuf pseg;                                         // shadow tracer: cursors of the cull pass's output segments
	DevBuf sitem;                                        // shadow tracer: exact-pass item list (segment, beam|sub) x2
	u32 *h_pinned = nullptr; // pinned host words for small read-backs (UGRT_PIN_*)
	u32 *d_small = nullptr;  // device scratch words (UGRT_DSMALL_*)
	unsigned prof_mask = 0; // bit s = stage s is timed
	unsigned chunk_capacity = 0; // prefix_capacity of the last ugrt_sort_rays
	const unsigned *chunk_prefix = nullptr, *chunk_map = nullptr; // and the arrays it sorted / wrote
	bool ray_sort_pending = false; // the deferred ugrt_sort_rays of these arrays has not been carried out (yet)
	int opt[UGRT_OPT_COUNT];     // ugrt_ctx_set_option; -1 = the built-in default
	std::vector<ProfPair> prof[UGRT_ST_COUNT];
	std::vector<ProfPair> prof_pool;
	bool overflow_seen = false; // an asynchronous call exceeded a capacity: reported by ugrt_ctx_synchronize
	// asynchronous shadow pass: candidate pairs and beams of the last pass
	u32 est_pairs = 0, est_beams = 0;
	bool have_shadow_est = false, shadow_async_pending = false;
	unsigned long long stats[8] = { 0 };
	unsigned long long dda_stats[UGRT_DDA_STATS] = { 0 }; // ugrt_stats_dda
	unsigned long long primary_stats[UGRT_PRIMARY_STATS] = { 0 }; // ugrt_stats_primary
};

#define UGRT_HIP(call)                                                                            \
	do {                                                                                      \
		hipError_t e_ = (call);                                                           \
		if (e_ != hipSuccess)                                                             \
			return ugrt_fail(UGRT_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
					 hipGetErrorString(e_));                                  \
	} while (0)

int ugrt_buf_reserve(ugrt_ctx *ctx, DevBuf &b, size_t bytes);
// NUM_SLABS > 1: span/offset over all slabs of each of the C light cells (ugrt_build.hip)
int ugrt_slab_union(ugrt_ctx *ctx, const u32 *d_span, const u32 *d_offset, u32 C, u32 slabs, const u32 **uspan,
		    const u32 **uoffset);
float *ugrt_ctx_tex(ugrt_ctx *ctx); // device copy of the 5x5x4 direction table
// status words the asynchronous calls have reported so far (each report carries the device's word as it stood then)
static inline u32 ugrt_reported_status(const ugrt_ctx *ctx)
{
	const u32 *p = ctx->h_pinned;
	return p[UGRT_PIN_REPORT + 3] | p[UGRT_PIN_REPORT + 7] | p[UGRT_PIN_REPORT + 11] | p[UGRT_PIN_SHADOW + 2];
}
static inline u32 *ugrt_wide_counter(ugrt_ctx *ctx) { return ctx->d_small + UGRT_DSMALL_WIDE; }
void ugrt_prof_begin(ugrt_ctx *ctx, int stage);
void ugrt_prof_end(ugrt_ctx *ctx, int stage);

// prefix sums (ugrt_scan.hip) and the library wrappers (ugrt_prims.hip); all enqueue on ctx->stream
int ugrt_prim_inclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n);
int ugrt_prim_exclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n);
// stable LSD radix sort of (key,value) pairs on key bits [0,end_bit)
// (n_dev != nullptr: the pair count lives on the device and n is the capacity the launches are sized for)
int ugrt_prim_sort_pairs(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n,
			 int end_bit, const u32 *n_dev = nullptr);
// the same on the library's onesweep (UGRT_SORT=rocprim, and the reference point of the sort tests)
int ugrt_prim_sort_pairs_rocprim(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n,
				 int end_bit);
// ugrt_sort.hip
int ugrt_sort_pairs_u32(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n, int end_bit,
			const u32 *n_dev = nullptr, bool prehist = false);
// up to two independent lists in shared launches (ugrt_sort.hip)
struct RsJob {
	const u32 *kin, *vin;
	u32 *kout, *vout;
	size_t n;
	int end_bit;
	const u32 *n_dev;
};
int ugrt_sort_pairs_batch(ugrt_ctx *ctx, const RsJob *jobs, int njobs, bool prehist = false);
int ugrt_lane_reduce_selftest(ugrt_ctx *ctx, unsigned long long *mismatches); // ugrt_trace.hip: DPP / permlane-swap reductions against __shfl_xor
int ugrt_f2i_selftest(ugrt_ctx *ctx, unsigned long long *mismatches); // ugrt_trace.hip: device float -> int forms against the portable ones
int ugrt_recip_selftest(ugrt_ctx *ctx, unsigned long long *mismatches); // ugrt_trace.hip: d_recip_det against 1.0f / x, all floats
int ugrt_prim_sort_pairs64(ugrt_ctx *ctx, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n,
			   int end_bit);

#endif
