// ugrt_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass.
//
// Replaces cudppSort (cudpp/cudpp.h:426-471; call sites frustum_grid.h:298, decision_data.h:177) for the
// seven sorts of a frame (three grid builds, the ray sort, the shadow tracer's three private sorts).
// They are small (0.2-3 M pairs on 8-32 key bits), so the fixed cost per pass matters as much as the
// bandwidth: a library onesweep spends one histogram kernel, one digit-scan kernel and 1 + 2*passes
// buffer fills per sort besides the passes.  Here: ONE fill (all counters of all passes), ONE histogram
// kernel (all digits in one read of the keys) and one kernel per pass, which scans the 256 digit totals
// itself.
//
// Pass kernel, one workgroup of 512 threads per tile of 8192 pairs:
//   - tiles are taken in launch order from an atomic ticket, so a tile only ever waits for tiles that
//     already run.  The tiles of these small sorts all start together, so a chained look-back (wait for
//     the predecessor's inclusive prefix) would propagate through the tiles one round trip at a time.
//     Instead the digit counts are combined in two levels without a chain: every tile publishes its
//     counts, the last tile of each chunk of 16 publishes the chunk's sum, and a tile's offset is the
//     sum of the chunk sums before its chunk + the counts of the tiles before it inside the chunk
//     (<= 15 + tiles/16 independent loads per digit, about three round trips in all);
//   - ranking is wave-synchronous: the 64 lanes of a wave find their equal-digit group with 8 ballots,
//     the group's first lane bumps the wave's digit counter in LDS; items are visited in memory order,
//     which makes the sort stable;
//   - the tile is put in digit order in LDS and written out in runs, so the scatter is coalesced.
#include "ugrt_ctx.h"

#define RS_THREADS 512
#define RS_WAVES (RS_THREADS / 64)
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_BINS 256
#define RS_MAXPASS 4
#define RS_CHUNK 16 // tiles per chunk of the two-level offset computation

#define RS_FLAG 0x80000000u // set in a published count (counts are < 2^31)
#define RS_VALUE 0x7FFFFFFFu

// digit histograms of all passes in one read of the keys
// (n_dev: the number of pairs when only the device knows it; n is then the capacity the launch was sized for)
// It also clears what the kernels after it expect to be zero, so that a sort needs no fill: the look-back words of
// this sort's passes (`look`) and the histogram + ticket block of the NEXT sort (`next_head`; the two blocks of the
// state alternate from sort to sort, both on the context's stream).
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const u32 *__restrict__ keys, u32 n, int passes, u32 end_bit,
							 u32 *__restrict__ hist, const u32 *__restrict__ n_dev,
							 u32 *__restrict__ next_head, u32 head_words, u32 *__restrict__ look,
							 u32 look_words)
{
	for (u32 i = blockIdx.x * RS_THREADS + threadIdx.x; i < head_words + look_words; i += gridDim.x * RS_THREADS) {
		if (i < head_words)
			next_head[i] = 0u;
		else
			look[i - head_words] = 0u;
	}
	if (n_dev)
		n = *n_dev < n ? *n_dev : n;
	__shared__ u32 s_h[RS_MAXPASS][RS_BINS];
	if (threadIdx.x < RS_BINS)
		for (int p = 0; p < passes; p++)
			s_h[p][threadIdx.x] = 0;
	__syncthreads();
	// few workgroups: every one ends with up to 256 * passes global adds on the same 256 * passes words
	const u32 stride = gridDim.x * RS_THREADS;
	for (u32 i0 = blockIdx.x * RS_THREADS + threadIdx.x; i0 < n; i0 += 4u * stride) {
		u32 k4[4];
#pragma unroll
		for (u32 u = 0; u < 4; u++)
			k4[u] = i0 + u * stride < n ? keys[i0 + u * stride] : 0u;
#pragma unroll
		for (u32 u = 0; u < 4; u++) {
			const bool ok = i0 + u * stride < n;
			const unsigned long long act = __ballot(ok);
			if (!ok)
				continue;
			for (int p = 0; p < passes; p++) {
				const u32 bits = end_bit - 8u * (u32)p < 8u ? end_bit - 8u * (u32)p : 8u;
				const u32 d = (k4[u] >> (8 * p)) & ((1u << bits) - 1u);
				// neighbouring keys mostly share their upper digits (cell ids in fill order): one add per wave then
				const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
				if (__ballot(d == d0) == act) {
					if ((threadIdx.x & 63u) == (u32)__builtin_ctzll(act))
						atomicAdd(&s_h[p][d0], (u32)__popcll(act));
				} else {
					atomicAdd(&s_h[p][d], 1u);
				}
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < RS_BINS)
		for (int p = 0; p < passes; p++) {
			const u32 c = s_h[p][threadIdx.x];
			if (c)
				atomicAdd(&hist[p * RS_BINS + threadIdx.x], c);
		}
}

// exclusive scan of one value per thread over the threads of the block (s_part: RS_WAVES words of LDS)
__device__ __forceinline__ u32 d_block_excl_scan(u32 v, u32 *s_part)
{
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	u32 incl = v;
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) {
		u32 o = (u32)__shfl_up((int)incl, m);
		if (lane >= (u32)m)
			incl += o;
	}
	__syncthreads(); // s_part may still be read from the previous scan
	if (lane == 63u)
		s_part[wave] = incl;
	__syncthreads();
	u32 base = 0;
#pragma unroll
	for (u32 w = 0; w < RS_WAVES; w++)
		base += w < wave ? s_part[w] : 0u;
	return base + incl - v;
}

__global__ __launch_bounds__(RS_THREADS, 4) void k_rs_pass(const u32 *__restrict__ kin, const u32 *__restrict__ vin,
							 u32 *__restrict__ kout, u32 *__restrict__ vout, u32 n, u32 shift,
							 u32 dmask, const u32 *__restrict__ hist, u32 *look, u32 *look2,
							 u32 *ticket, const u32 *__restrict__ n_dev)
{
	if (n_dev)
		n = *n_dev < n ? *n_dev : n;
	__shared__ u32 s_keys[RS_TILE], s_vals[RS_TILE];
	__shared__ u32 s_cnt[RS_WAVES][RS_BINS]; // per wave: digit counters while ranking, then the wave's offset inside the digit
	__shared__ u32 s_lstart[RS_BINS]; // first slot of the digit inside the sorted tile
	__shared__ u32 s_base[RS_BINS];   // global position of slot j of digit d = s_base[d] + j
	__shared__ u32 s_part[RS_WAVES];
	__shared__ u32 s_tile;
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	if (t == 0)
		s_tile = atomicAdd(ticket, 1u);
	for (u32 i = t; i < RS_WAVES * RS_BINS; i += RS_THREADS)
		(&s_cnt[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = s_tile;
	const u32 base = tile * RS_TILE;
	if (base >= n)
		return; // a launch sized by the capacity: this tile holds nothing, and no tile waits for a later one
	u32 k[RS_ITEMS], r[RS_ITEMS];
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		const u32 idx = base + wave * (64u * RS_ITEMS) + (u32)i * 64u + lane;
		k[i] = idx < n ? kin[idx] : 0xFFFFFFFFu;
	}
	// wave-synchronous ranking, items in memory order
	volatile u32 *cnt = s_cnt[wave];
	const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		const u32 idx = base + wave * (64u * RS_ITEMS) + (u32)i * 64u + lane;
		const bool ok = idx < n;
		const u32 d = (k[i] >> shift) & dmask;
		unsigned long long grp = __ballot(ok);
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __ballot(bit);
			grp &= bit ? m : ~m;
		}
		u32 old = 0;
		if (ok)
			old = cnt[d];
		r[i] = old + (u32)__popcll(grp & lt);
		__builtin_amdgcn_wave_barrier(); // every lane of the group has read the counter
		if (ok && (grp & lt) == 0ull)
			cnt[d] = old + (u32)__popcll(grp);
		__builtin_amdgcn_wave_barrier();
	}
	__syncthreads();
	// digit d = thread d (the upper half of the block only takes part in the barriers): totals, offsets of
	// the waves inside the digit, position of the digit in the tile
	const bool digit = t < RS_BINS;
	u32 total = 0;
	if (digit) {
#pragma unroll
		for (int w = 0; w < RS_WAVES; w++) {
			const u32 c = s_cnt[w][t];
			s_cnt[w][t] = total;
			total += c;
		}
	}
	const u32 lstart = d_block_excl_scan(total, s_part);
	const u32 gdigit = d_block_excl_scan(digit ? hist[t] : 0u, s_part);
	if (digit) {
	// digit t: offset of this tile = counts of all tiles before it, combined in two levels (no chain)
	const u32 chunk = tile / RS_CHUNK, first = chunk * RS_CHUNK;
	__hip_atomic_store(look + (size_t)tile * RS_BINS + t, total | RS_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	u32 excl = 0;
	{
		// the tiles before this one inside its chunk (they hold lower tickets: they run or have finished)
		u32 s[RS_CHUNK - 1];
		const u32 nb = tile - first;
		bool again;
		do {
			again = false;
#pragma unroll
			for (u32 w = 0; w < RS_CHUNK - 1; w++)
				s[w] = w < nb ? __hip_atomic_load(look + (size_t)(first + w) * RS_BINS + t, __ATOMIC_RELAXED,
								  __HIP_MEMORY_SCOPE_AGENT)
					      : RS_FLAG;
#pragma unroll
			for (u32 w = 0; w < RS_CHUNK - 1; w++)
				again = again || (s[w] & RS_FLAG) == 0u;
		} while (again);
#pragma unroll
		for (u32 w = 0; w < RS_CHUNK - 1; w++)
			excl += s[w] & RS_VALUE;
	}
	if (tile - first == RS_CHUNK - 1u) // the chunk is complete with this tile: publish its sum
		__hip_atomic_store(look2 + (size_t)chunk * RS_BINS + t, (excl + total) | RS_FLAG, __ATOMIC_RELAXED,
				   __HIP_MEMORY_SCOPE_AGENT);
	// the chunks before this one (their last tiles hold lower tickets)
	for (u32 c0 = 0; c0 < chunk; c0 += RS_CHUNK) {
		u32 s[RS_CHUNK];
		bool again;
		do {
			again = false;
#pragma unroll
			for (u32 w = 0; w < RS_CHUNK; w++)
				s[w] = c0 + w < chunk ? __hip_atomic_load(look2 + (size_t)(c0 + w) * RS_BINS + t, __ATOMIC_RELAXED,
									  __HIP_MEMORY_SCOPE_AGENT)
						      : RS_FLAG;
#pragma unroll
			for (u32 w = 0; w < RS_CHUNK; w++)
				again = again || (s[w] & RS_FLAG) == 0u;
		} while (again);
#pragma unroll
		for (u32 w = 0; w < RS_CHUNK; w++)
			excl += s[w] & RS_VALUE;
	}
	s_lstart[t] = lstart;
	s_base[t] = gdigit + excl - lstart;
	} // digit
	__syncthreads();
	// tile in digit order in LDS (the values are fetched only now: they would occupy registers all the way)
	u32 v[RS_ITEMS];
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		const u32 idx = base + wave * (64u * RS_ITEMS) + (u32)i * 64u + lane;
		v[i] = idx < n ? vin[idx] : 0u;
	}
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		const u32 idx = base + wave * (64u * RS_ITEMS) + (u32)i * 64u + lane;
		if (idx < n) {
			const u32 d = (k[i] >> shift) & dmask;
			const u32 slot = s_lstart[d] + s_cnt[wave][d] + r[i];
			s_keys[slot] = k[i];
			s_vals[slot] = v[i];
		}
	}
	__syncthreads();
	const u32 ntile = (n - base) < (u32)RS_TILE ? (n - base) : (u32)RS_TILE;
	for (u32 j = t; j < ntile; j += RS_THREADS) {
		const u32 key = s_keys[j];
		const u32 pos = s_base[(key >> shift) & dmask] + j;
		kout[pos] = key;
		vout[pos] = s_vals[j];
	}
}

// stable sort of n pairs on key bits [0, end_bit); kin/vin are left untouched, the result is in kout/vout
int ugrt_sort_pairs_u32(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n, int end_bit,
			const u32 *n_dev)
{
	if (n == 0)
		return UGRT_OK;
	if (n > ((size_t)1 << 30))
		return ugrt_fail(UGRT_EINVAL, "sort: %zu pairs exceed 2^30", n);
	if (end_bit < 1)
		end_bit = 1;
	if (end_bit > 32)
		end_bit = 32;
	const int passes = (end_bit + 7) / 8;
	const u32 tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
	hipStream_t st = ctx->stream;
	int rc;
	// state: two blocks of {[RS_MAXPASS][256] histograms, [256] tickets} that alternate from sort to sort, then per
	// pass [tiles][256] tile counts and [chunks][256] chunk sums
	const u32 chunks = (tiles + RS_CHUNK - 1) / RS_CHUNK;
	const size_t per_pass = (size_t)(tiles + chunks) * RS_BINS;
	const size_t head = (size_t)RS_MAXPASS * RS_BINS + RS_BINS;
	const size_t words = 2 * head + (size_t)passes * per_pass;
	const void *before = ctx->rs_state.p;
	if ((rc = ugrt_buf_reserve(ctx, ctx->rs_state, words * 4)))
		return rc;
	if (passes > 1) {
		if ((rc = ugrt_buf_reserve(ctx, ctx->rs_tmp[0], n * 4)) || (rc = ugrt_buf_reserve(ctx, ctx->rs_tmp[1], n * 4)))
			return rc;
	}
	if (ctx->rs_state.p != before) { // a new allocation: nothing has cleared the first block yet
		UGRT_HIP(hipMemsetAsync(ctx->rs_state.p, 0, 2 * head * 4, st));
		ctx->rs_flip = 0;
	}
	u32 *hist = (u32 *)ctx->rs_state.p + (size_t)ctx->rs_flip * head, *ticket = hist + (size_t)RS_MAXPASS * RS_BINS;
	u32 *next_head = (u32 *)ctx->rs_state.p + (size_t)(ctx->rs_flip ^ 1) * head, *look = (u32 *)ctx->rs_state.p + 2 * head;
	ctx->rs_flip ^= 1;
	u32 hblocks = (u32)((n + RS_THREADS * 8 - 1) / (RS_THREADS * 8));
	hblocks = hblocks > 256u ? 256u : hblocks;
	hipLaunchKernelGGL(k_rs_hist, dim3(hblocks), dim3(RS_THREADS), 0, st, kin, (u32)n, passes, (u32)end_bit, hist, n_dev,
			   next_head, (u32)head, look, (u32)((size_t)passes * per_pass));
	UGRT_HIP(hipGetLastError());
	const u32 *ki = kin, *vi = vin;
	for (int p = 0; p < passes; p++) {
		// the buffers alternate so that the last pass writes the caller's output
		const bool to_out = ((passes - 1 - p) & 1) == 0;
		u32 *ko = to_out ? kout : (u32 *)ctx->rs_tmp[0].p, *vo = to_out ? vout : (u32 *)ctx->rs_tmp[1].p;
		const u32 bits = (u32)(end_bit - 8 * p) < 8u ? (u32)(end_bit - 8 * p) : 8u;
		hipLaunchKernelGGL(k_rs_pass, dim3(tiles), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, (u32)n, (u32)(8 * p),
				   (1u << bits) - 1u, (const u32 *)(hist + (size_t)p * RS_BINS),
				   look + (size_t)p * per_pass, look + (size_t)p * per_pass + (size_t)tiles * RS_BINS, ticket + p, n_dev);
		UGRT_HIP(hipGetLastError());
		ki = ko;
		vi = vo;
	}
	return UGRT_OK;
}
