// ugrt_scan.hip -- prefix sums of u32 arrays in ONE kernel.
//
// Replaces cudppScan (cudpp/cudpp.h:426-471; call sites frustum_grid.h:249 inclusive over the triangles' cell
// counts, frustum_grid.h:361 exclusive over the cells' spans, decision_data.h:209 the rays' in-run ranks) for the
// eleven scans of a frame.  The library's scan is two launches (it initialises its look-back state in a kernel of
// its own); the arrays here are 16 K - 1 M words, so a scan is bound by its launches, not by its bytes.
// Tiles of 4096 words (16 consecutive words per thread, 16-byte accesses) with a decoupled look-back: tiles are
// taken from a ticket in launch order (a tile only waits for tiles that run or have finished); the state words
// carry the EPOCH of the launch beside their value, so nothing has to be cleared between scans, and the last tile
// to finish resets the ticket.  (One workgroup walking a 16-32 K array by itself - no state at all - was measured
// too: 15-20 us per scan against 5-6, the frame 0.04 ms slower.)
#include "ugrt_ctx.h"

#define SC_THREADS 256
#define SC_WAVES (SC_THREADS / 64)
#define SC_ITEMS 16
#define SC_TILE (SC_THREADS * SC_ITEMS)

// state word of tile i: epoch << 34 | flag << 32 | value; flag 1 = the tile's own sum, 2 = the sum of all tiles up to it
#define SC_FLAG_SUM 1ull
#define SC_FLAG_PREFIX 2ull

__device__ __forceinline__ u32 d_wave_incl_scan(u32 v, u32 lane)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) {
		const u32 o = (u32)__shfl_up((int)v, m);
		if (lane >= (u32)m)
			v += o;
	}
	return v;
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(SC_THREADS) void k_scan_tiles(const u32 *__restrict__ in, u32 *__restrict__ out, u32 n,
							    unsigned long long *state, u32 *ctrl, u32 epoch, u32 vec)
{
	__shared__ u32 s_tile, s_wave[SC_WAVES], s_prefix;
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	if (t == 0)
		s_tile = atomicAdd(&ctrl[0], 1u);
	__syncthreads();
	const u32 tile = s_tile, base = tile * SC_TILE + t * SC_ITEMS;
	u32 v[SC_ITEMS];
	if (vec && base + SC_ITEMS <= n) {
#pragma unroll
		for (int q = 0; q < SC_ITEMS / 4; q++) {
			const uint4 x = reinterpret_cast<const uint4 *>(in + base)[q];
			v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
		}
	} else {
#pragma unroll
		for (int i = 0; i < SC_ITEMS; i++)
			v[i] = base + (u32)i < n ? in[base + i] : 0u;
	}
	u32 sum = 0;
#pragma unroll
	for (int i = 0; i < SC_ITEMS; i++)
		sum += v[i];
	const u32 incl = d_wave_incl_scan(sum, lane);
	if (lane == 63u)
		s_wave[wave] = incl;
	__syncthreads();
	u32 run = incl - sum, total = 0;
#pragma unroll
	for (u32 w = 0; w < SC_WAVES; w++) {
		run += w < wave ? s_wave[w] : 0u;
		total += s_wave[w];
	}
	// the sum of all tiles before this one: wave 0 looks back 64 tiles at a time
	if (wave == 0u) {
		const unsigned long long tag = (unsigned long long)epoch << 34;
		u32 prefix = 0;
		if (tile == 0u) {
			if (lane == 0u)
				__hip_atomic_store(state, tag | (SC_FLAG_PREFIX << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			if (lane == 0u)
				__hip_atomic_store(state + tile, tag | (SC_FLAG_SUM << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			int look = (int)tile - 1 - (int)lane; // lane 0 looks at the nearest predecessor
			for (;;) {
				unsigned long long w = tag | (SC_FLAG_PREFIX << 32); // tiles before the first: nothing to add
				if (look >= 0) {
					do
						w = __hip_atomic_load(state + look, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					while ((w >> 34) != (unsigned long long)epoch);
				}
				const bool full = ((w >> 32) & 3ull) == SC_FLAG_PREFIX;
				const unsigned long long fm = __ballot(full);
				// lanes up to the nearest tile that knows its whole prefix contribute
				const u32 stop = fm ? (u32)__builtin_ctzll(fm) : 63u;
				u32 part = lane <= stop ? (u32)w : 0u;
#pragma unroll
				for (int m = 32; m >= 1; m >>= 1)
					part += (u32)__shfl_xor((int)part, m);
				prefix += part;
				if (fm)
					break;
				look -= 64;
			}
			if (lane == 0u)
				__hip_atomic_store(state + tile, tag | (SC_FLAG_PREFIX << 32) | (unsigned long long)(prefix + total),
						   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (lane == 0u)
			s_prefix = prefix;
	}
	__syncthreads();
	run += s_prefix;
	u32 o[SC_ITEMS];
#pragma unroll
	for (int i = 0; i < SC_ITEMS; i++) {
		o[i] = INCLUSIVE ? run + v[i] : run;
		run += v[i];
	}
	if (vec && base + SC_ITEMS <= n) {
#pragma unroll
		for (int q = 0; q < SC_ITEMS / 4; q++)
			reinterpret_cast<uint4 *>(out + base)[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
	} else {
#pragma unroll
		for (int i = 0; i < SC_ITEMS; i++)
			if (base + (u32)i < n)
				out[base + i] = o[i];
	}
	// the last tile to get here leaves the ticket at zero for the next scan
	if (t == 0) {
		if (atomicAdd(&ctrl[1], 1u) == gridDim.x - 1u) {
			ctrl[0] = 0u;
			ctrl[1] = 0u;
		}
	}
}

static int scan_u32(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n, bool inclusive)
{
	if (n == 0)
		return UGRT_OK;
	if (n > 0xFFFFFFF0ull)
		return ugrt_fail(UGRT_EINVAL, "scan: %zu words exceed the 32-bit index", n);
	hipStream_t st = ctx->stream;
	// 16-byte accesses when both arrays allow them (hipMalloc is 256-B aligned; callers also pass offsets into buffers)
	const u32 vec = ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0 ? 1u : 0u;
	const u32 tiles = (u32)((n + SC_TILE - 1) / SC_TILE);
	const void *before = ctx->scan_state.p;
	int rc = ugrt_buf_reserve(ctx, ctx->scan_state, (size_t)tiles * 8 + 64);
	if (rc)
		return rc;
	if (ctx->scan_state.p != before) { // a new allocation: ticket, done counter and every epoch tag start at zero
		UGRT_HIP(hipMemsetAsync(ctx->scan_state.p, 0, ctx->scan_state.cap, st));
		ctx->scan_epoch = 0;
	}
	if (++ctx->scan_epoch >= (1u << 30)) { // (the tag has 30 bits)
		UGRT_HIP(hipMemsetAsync(ctx->scan_state.p, 0, ctx->scan_state.cap, st));
		ctx->scan_epoch = 1;
	}
	u32 *ctrl = (u32 *)ctx->scan_state.p;
	unsigned long long *state = (unsigned long long *)((char *)ctx->scan_state.p + 64);
	if (inclusive)
		hipLaunchKernelGGL(k_scan_tiles<true>, dim3(tiles), dim3(SC_THREADS), 0, st, in, out, (u32)n, state, ctrl, ctx->scan_epoch, vec);
	else
		hipLaunchKernelGGL(k_scan_tiles<false>, dim3(tiles), dim3(SC_THREADS), 0, st, in, out, (u32)n, state, ctrl, ctx->scan_epoch, vec);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

int ugrt_prim_inclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n) { return scan_u32(ctx, in, out, n, true); }
int ugrt_prim_exclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n) { return scan_u32(ctx, in, out, n, false); }
