// ugrt_scan.h -- the prefix-sum kernel of ugrt_scan.hip as a template: LOAD produces a thread's 16 consecutive input
// words (a plain array, or values computed on the fly: the grid builds form the cells' spans from the run bounds
// inside the scan that turns them into offsets; the work lists' counts per cell or beam are formed the same way),
// STORE may consume the sums where they are produced, TAIL runs once in the last tile to finish (the tail of an
// asynchronous build's report).
#ifndef UGRT_SCAN_H
#define UGRT_SCAN_H

#include "ugrt_ctx.h"

#define SC_THREADS 256
#define SC_WAVES (SC_THREADS / 64)
#define SC_ITEMS 16
#define SC_TILE (SC_THREADS * SC_ITEMS)

// state word of tile i: epoch << 34 | flag << 32 | value; flag 1 = the tile's own sum, 2 = the sum of all tiles up to it
#define SC_FLAG_SUM 1ull
#define SC_FLAG_PREFIX 2ull

struct ScanLoadArray {
	const u32 *in;
	u32 vec; // 16-byte loads allowed
	__device__ __forceinline__ void operator()(u32 base, u32 n, u32 (&v)[SC_ITEMS]) const
	{
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
	}
};
struct ScanTailNone {
	static constexpr bool active = false;
	__device__ __forceinline__ void operator()() const {}
};
// STORE (optional) sees a thread's 16 inputs beside their prefix sums, after `out` has been written: the kernel that
// would turn counts and offsets into a list runs inside the scan
struct ScanStoreNone {
	static constexpr bool active = false;
	__device__ __forceinline__ void operator()(u32, u32, const u32 (&)[16], const u32 (&)[16]) const {}
};

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

// one tile of a scan of `tiles` tiles (the workgroups of a launch may serve two scans: k_scan_pair)
template <bool INCLUSIVE, typename LOAD, typename TAIL, typename STORE>
__device__ __forceinline__ void d_scan_tile(const LOAD &load, u32 *__restrict__ out, u32 n, unsigned long long *state, u32 *ctrl,
					    u32 epoch, u32 vec, const TAIL &tail, const STORE &store, u32 tiles)
{
	__shared__ u32 s_tile, s_wave[SC_WAVES], s_prefix;
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	if (t == 0)
		s_tile = atomicAdd(&ctrl[0], 1u);
	__syncthreads();
	const u32 tile = s_tile, base = tile * SC_TILE + t * SC_ITEMS;
	u32 v[SC_ITEMS];
	load(base, n, v); // this thread's 16 consecutive words (0 beyond n)
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
	if (STORE::active)
		store(base, n, v, o);
	// the last tile to get here leaves the ticket at zero for the next scan and runs the caller's epilogue.  What the
	// epilogue reads from other tiles are device-scope atomics (performed in L2), so no cache has to be written back:
	// it is enough that this thread's own earlier atomics have been acknowledged before it counts itself in
	if (t == 0) {
		if (TAIL::active)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const u32 done = atomicAdd(&ctrl[1], 1u);
		if (done == tiles - 1u) {
			ctrl[0] = 0u;
			ctrl[1] = 0u;
			tail();
		}
	}
}

template <bool INCLUSIVE, typename LOAD, typename TAIL, typename STORE>
__global__ __launch_bounds__(SC_THREADS) void k_scan_tiles(LOAD load, u32 *__restrict__ out, u32 n, unsigned long long *state,
							    u32 *ctrl, u32 epoch, u32 vec, TAIL tail, STORE store)
{
	d_scan_tile<INCLUSIVE>(load, out, n, state, ctrl, epoch, vec, tail, store, gridDim.x);
}

// two scans of n words each in ONE launch (their inputs must not depend on each other): the first `tiles` workgroups
// serve scan A, the others scan B, each with its own ticket and state words
template <bool INCLUSIVE, typename LOAD>
__global__ __launch_bounds__(SC_THREADS) void k_scan_pair(LOAD load_a, u32 *__restrict__ out_a, LOAD load_b, u32 *__restrict__ out_b, u32 n,
							   unsigned long long *state, u32 *ctrl, u32 epoch, u32 vec, u32 tiles)
{
	if (blockIdx.x < tiles)
		d_scan_tile<INCLUSIVE>(load_a, out_a, n, state, ctrl, epoch, vec, ScanTailNone(), ScanStoreNone(), tiles);
	else
		d_scan_tile<INCLUSIVE>(load_b, out_b, n, state + tiles, ctrl + 2, epoch, vec, ScanTailNone(), ScanStoreNone(), tiles);
}

// enqueues the scan on ctx->stream; out must be 16-byte aligned for `vec_out`
template <bool INCLUSIVE, typename LOAD, typename TAIL, typename STORE = ScanStoreNone>
static int ugrt_scan_launch(ugrt_ctx *ctx, const LOAD &load, u32 *out, size_t n, const TAIL &tail, const STORE &store = STORE())
{
	if (n == 0)
		return UGRT_OK;
	if (n > 0xFFFFFFF0ull)
		return ugrt_fail(UGRT_EINVAL, "scan: %zu words exceed the 32-bit index", n);
	hipStream_t st = ctx->stream;
	const u32 vec = (((uintptr_t)out) & 15u) == 0 ? 1u : 0u;
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
	hipLaunchKernelGGL((k_scan_tiles<INCLUSIVE, LOAD, TAIL, STORE>), dim3(tiles), dim3(SC_THREADS), 0, st, load, out, (u32)n, state,
			   ctrl, ctx->scan_epoch, vec, tail, store);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

// two scans of n words in one launch (k_scan_pair)
template <bool INCLUSIVE, typename LOAD>
static int ugrt_scan_launch_pair(ugrt_ctx *ctx, const LOAD &load_a, u32 *out_a, const LOAD &load_b, u32 *out_b, size_t n)
{
	if (n == 0)
		return UGRT_OK;
	if (n > 0x7FFFFFF0ull)
		return ugrt_fail(UGRT_EINVAL, "scan: %zu words exceed the 32-bit index", n);
	hipStream_t st = ctx->stream;
	const u32 vec = ((((uintptr_t)out_a) | ((uintptr_t)out_b)) & 15u) == 0 ? 1u : 0u;
	const u32 tiles = (u32)((n + SC_TILE - 1) / SC_TILE);
	const void *before = ctx->scan_state.p;
	int rc = ugrt_buf_reserve(ctx, ctx->scan_state, (size_t)tiles * 16 + 64);
	if (rc)
		return rc;
	if (ctx->scan_state.p != before) {
		UGRT_HIP(hipMemsetAsync(ctx->scan_state.p, 0, ctx->scan_state.cap, st));
		ctx->scan_epoch = 0;
	}
	if (++ctx->scan_epoch >= (1u << 30)) {
		UGRT_HIP(hipMemsetAsync(ctx->scan_state.p, 0, ctx->scan_state.cap, st));
		ctx->scan_epoch = 1;
	}
	u32 *ctrl = (u32 *)ctx->scan_state.p;
	unsigned long long *state = (unsigned long long *)((char *)ctx->scan_state.p + 64);
	hipLaunchKernelGGL((k_scan_pair<INCLUSIVE, LOAD>), dim3(2u * tiles), dim3(SC_THREADS), 0, st, load_a, out_a, load_b, out_b, (u32)n,
			   state, ctrl, ctx->scan_epoch, vec, tiles);
	UGRT_HIP(hipGetLastError());
	return UGRT_OK;
}

#endif
