// ugrt_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass.
//
// Replaces cudppSort (cudpp/cudpp.h:426-471; call sites frustum_grid.h:298, decision_data.h:177) for the
// seven sorts of a frame (three grid builds, the ray sort, the shadow tracer's three private sorts).
// They are small (0.2-3 M pairs on 8-32 key bits), so the fixed cost per pass matters as much as the
// bandwidth: a library onesweep spends one histogram kernel, one digit-scan kernel and 1 + 2*passes
// buffer fills per sort besides the passes.  Here a sort is its passes and nothing else:
//   - a histogram kernel counts the FIRST pass's digit only; every pass counts the digit of the pass that follows
//     while it holds the keys (LDS counters, one global add per digit and tile, spread over RS_COPIES rows).  Round 3
//     counted all digits up front: 14-17 us per sort, as much as a pass (2-4 LDS atomics per key, mostly on one
//     address per wave); counting in the kernels that WRITE the keys was tried too and lost to the global adds of their
//     thousands of workgroups;
//   - the state needs no clearing: the look-back words carry the EPOCH of their pass beside the count,
//     and the last workgroup of a pass to finish zeroes that pass's histogram rows, ticket and counter.
//
// Pass kernel, one workgroup of 512 threads per tile of 8192 pairs:
//   - tiles are taken in launch order from an atomic ticket, so a tile only ever waits for tiles that
//     already run.  The tiles of these small sorts all start together, so a chained look-back (wait for
//     the predecessor's inclusive prefix) would propagate through the tiles one round trip at a time.
//     Instead the digit counts are combined in two levels without a chain: every tile publishes its
//     counts, the last tile of each chunk of 16 publishes the chunk's sum, and a tile's offset is the
//     sum of the chunk sums before its chunk + the counts of the tiles before it inside the chunk.
//     Rows are tile-major ([tile][digit]): the 256 digit threads of a workgroup read and write a row as one
//     coalesced 2-KB access; the rows of a level are awaited eight at a time (uniform row bases, one per-thread
//     offset: all fifteen at once cost an address register pair each and 29 spilled registers);
//   - ranking is wave-synchronous: the 64 lanes of a wave find their equal-digit group with 8 ballots,
//     the group's first lane bumps the wave's digit counter in LDS; items are visited in memory order,
//     which makes the sort stable;
//   - the tile is put in digit order in LDS and written out in runs, so the scatter is coalesced.
#include "ugrt_rs_hist.h"

#define RS_THREADS 512
#define RS_WAVES (RS_THREADS / 64)
#define RS_ITEMS_MAX 16 // pairs per thread: 16 (tiles of 8192 pairs) or 8 (4096: option "sort_items")
#define RS_CHUNK 16 // tiles per chunk of the two-level offset computation


typedef unsigned long long u64w; // look-back word: epoch << 32 | count

// One launch may serve up to RS_MAXSEG independent sorts (segments): the light grid's and the uniform grid's references
// depend on the geometry only, and their sorts share the launches of every pass level (ugrt_sort_pairs_batch) -- twice
// the tiles in flight per launch hide the per-tile chain, and a frame has three launches less.
#define RS_MAXSEG 2
struct RsSeg {
	const u32 *kin, *vin;
	u32 *kout, *vout;
	const u32 *n_dev;      // the pair count when only the device knows it (n is then the capacity the launch was sized for)
	u32 *hist, *hist_next; // this pass's rows [copy][digit] / the rows of the pass that follows
	u64w *look, *look2;    // [tile][digit] tile counts, [chunk][digit] chunk sums
	u32 n, dmask, nmask;   // nmask: digit mask of the pass that follows (0: this is the segment's last pass)
	u32 blocks;            // workgroups of this launch that belong to the segment
};
struct RsBatch {
	RsSeg s[RS_MAXSEG];
	u32 nseg;
};

// histogram of the first pass's digit
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(RsBatch b)
{
	const u32 si = b.nseg > 1u && blockIdx.x >= b.s[0].blocks ? 1u : 0u;
	const RsSeg &g = b.s[si];
	const u32 blk = blockIdx.x - (si ? b.s[0].blocks : 0u);
	const u32 *__restrict__ keys = g.kin;
	u32 n = g.n;
	if (g.n_dev)
		n = *g.n_dev < n ? *g.n_dev : n;
	__shared__ u32 s_h[RS_BINS * RS_PRIV];
	for (u32 i = threadIdx.x; i < RS_BINS * RS_PRIV; i += RS_THREADS)
		s_h[i] = 0u;
	__syncthreads();
	// few workgroups: every one ends with up to 256 global adds
	const u32 stride = g.blocks * RS_THREADS;
	for (u32 i0 = blk * RS_THREADS + threadIdx.x; i0 < n; i0 += 4u * stride) {
		u32 k4[4];
#pragma unroll
		for (u32 u = 0; u < 4; u++)
			k4[u] = i0 + u * stride < n ? keys[i0 + u * stride] : 0u;
#pragma unroll
		for (u32 u = 0; u < 4; u++)
			d_rs_count(s_h, k4[u] & g.dmask, i0 + u * stride < n);
	}
	__syncthreads();
	if (threadIdx.x < RS_BINS) {
		const u32 c = d_rs_count_sum(s_h, threadIdx.x);
		if (c)
			atomicAdd(&g.hist[(blk % RS_COPIES) * RS_BINS + threadIdx.x], c);
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

// sum over `count` rows of the look-back word of this thread's digit (rows `stride` words apart, the first at `row`),
// each awaited until it carries this pass's epoch; GROUP loads in flight at a time
template <int GROUP>
__device__ __forceinline__ u32 d_rs_wait_sum(const u64w *row, u32 count, u32 stride, u32 epoch)
{
	const u64w tag = (u64w)epoch << 32;
	u32 sum = 0;
	for (u32 c0 = 0; c0 < count; c0 += (u32)GROUP) {
		u64w s[GROUP];
		bool again;
		do {
			again = false;
#pragma unroll
			for (u32 w = 0; w < (u32)GROUP; w++)
				s[w] = c0 + w < count ? __hip_atomic_load(row + (size_t)(c0 + w) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
						      : tag;
#pragma unroll
			for (u32 w = 0; w < (u32)GROUP; w++)
				again = again || (u32)(s[w] >> 32) != epoch;
		} while (again);
#pragma unroll
		for (u32 w = 0; w < (u32)GROUP; w++)
			sum += (u32)s[w];
	}
	return sum;
}

// Rank of every pair among the pairs of its digit inside its wave's 64 * RS_ITEMS pairs, in memory order, and the
// wave's digit counts in cnt[].  Two forms:
//   ATOMIC: one ds_add_rtn_u32 per pair on the wave's counter of its digit.  A wave's LDS instructions execute in
//     program order, and inside one instruction the lanes that hit the same counter are served in lane order (the
//     hardware's conflict resolution: lowest lane first) -- that order is not in the ISA manual, so every context
//     checks it on its device before it uses this form (k_rs_selftest; "sort_rank" 0 forces the other form);
//   ballots: the lanes of a pair's digit group are found with one ballot per digit bit (on 32-bit halves: one
//     v_bitop3 per half and bit), the group's first lane bumps the counter.
// 3 vector instructions a pair against ~45 (round 3's form: ~79, more than half of the kernel's 2000 per wave and tile).
template <int RS_ITEMS, bool ATOMIC, bool FULL>
__device__ __forceinline__ void d_rs_rank(const u32 (&k)[RS_ITEMS], u32 (&r)[RS_ITEMS], u32 *cnt, u32 shift, u32 dmask, u32 first, u32 n,
					  u32 lane)
{
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		const bool ok = FULL || first + (u32)i * 64u + lane < n;
		const u32 d = (k[i] >> shift) & dmask;
		if (ATOMIC) {
			// (a wave whose 64 keys share the digit -- sorted runs, constant upper digits -- would have its lanes served one
			// after the other: there the rank is the lane and one lane adds 64)
			const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
			if (FULL && __ballot(d == d0) == ~0ull) {
				u32 old = 0;
				if (lane == 0u)
					old = atomicAdd(&cnt[d0], 64u);
				r[i] = (u32)__builtin_amdgcn_readfirstlane((int)old) + lane;
			} else {
				r[i] = 0;
				if (ok)
					r[i] = atomicAdd(&cnt[d], 1u);
			}
		} else {
			const unsigned long long act = FULL ? ~0ull : __ballot(ok);
			u32 glo = (u32)act, ghi = (u32)(act >> 32);
#pragma unroll
			for (int b = 0; b < 8; b++) {
				const u32 sb = (u32)((int)(d << (31 - b)) >> 31); // all ones where the bit is set
				const unsigned long long m = __ballot(sb != 0u);
				glo &= ~((u32)m ^ sb); // lanes whose bit equals this lane's
				ghi &= ~((u32)(m >> 32) ^ sb);
			}
			const u32 before = __builtin_amdgcn_mbcnt_hi(ghi, __builtin_amdgcn_mbcnt_lo(glo, 0u)); // group lanes below this one
			volatile u32 *vc = cnt;
			u32 old = 0;
			if (ok)
				old = vc[d];
			r[i] = old + before;
			__builtin_amdgcn_wave_barrier(); // every lane of the group has read the counter
			if (ok && before == 0u)
				vc[d] = old + (u32)__popc(glo) + (u32)__popc(ghi);
			__builtin_amdgcn_wave_barrier();
		}
	}
}

// the work of one tile that holds pairs: keys and values in, ranks, offsets (look-back), LDS reorder, scatter
template <int RS_ITEMS, bool ATOMIC, bool FULL>
__device__ __forceinline__ void d_rs_tile(const u32 *__restrict__ kin, const u32 *__restrict__ vin, u32 *__restrict__ kout,
					  u32 *__restrict__ vout, u32 n, u32 shift, u32 dmask, u32 gdigit_in, u32 *hist_next, u32 nmask,
					  u64w *look, u64w *look2, u32 epoch, u32 tile, u32 *s_keys, u32 *s_vals, u32 (*s_cnt)[RS_BINS],
					  u32 *s_base, u32 *s_part, u32 *s_next)
{
	constexpr u32 RS_TILE = RS_THREADS * RS_ITEMS;
	const u32 t = threadIdx.x, lane = t & 63u;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(t >> 6));
	const u32 base = tile * RS_TILE;
	const u32 first = base + wave * (64u * RS_ITEMS); // (uniform: the loads take a scalar base and one lane offset)
	u32 k[RS_ITEMS], v[RS_ITEMS], r[RS_ITEMS];
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++)
		k[i] = FULL || first + (u32)i * 64u + lane < n ? kin[first + (u32)i * 64u + lane] : 0xFFFFFFFFu;
	// (the values are requested with the keys: the ranks no longer need the registers they used to)
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++)
		v[i] = FULL || first + (u32)i * 64u + lane < n ? vin[first + (u32)i * 64u + lane] : 0u;
	d_rs_rank<RS_ITEMS, ATOMIC, FULL>(k, r, s_cnt[wave], shift, dmask, first, n, lane);
	// the digit of the pass that follows, counted while the keys are here (nmask == 0: this is the last pass)
	if (nmask) {
#pragma unroll
		for (int i = 0; i < RS_ITEMS; i++)
			d_rs_count(s_next, (k[i] >> (shift + 8u)) & nmask, FULL || first + (u32)i * 64u + lane < n);
	}
	__syncthreads();
	// digit d = thread d (the upper half of the block only takes part in the barriers): totals, offsets of
	// the waves inside the digit, position of the digit in the tile
	const bool digit = t < RS_BINS;
	u32 total = 0, woff[RS_WAVES];
	if (digit) {
#pragma unroll
		for (int w = 0; w < RS_WAVES; w++) {
			woff[w] = total;
			total += s_cnt[w][t];
		}
	}
	if (nmask && digit) { // (the adds are on their way while the tile goes on)
		const u32 c = d_rs_count_sum(s_next, t);
		if (c)
			atomicAdd(&hist_next[(tile % RS_COPIES) * RS_BINS + t], c);
	}
	const u32 lstart = d_block_excl_scan(total, s_part); // (its barriers also stand between the reads of s_next above and the reorder below)
	const u32 gdigit = d_block_excl_scan(gdigit_in, s_part);
	if (digit) {
		// slot of a pair inside the sorted tile = s_cnt[its wave][its digit] + its rank
#pragma unroll
		for (int w = 0; w < RS_WAVES; w++)
			s_cnt[w][t] = lstart + woff[w];
		// digit t: offset of this tile = counts of all tiles before it, combined in two levels (no chain)
		const u32 chunk = tile / RS_CHUNK, firstt = chunk * RS_CHUNK, nb = tile - firstt;
		const u64w tag = (u64w)epoch << 32;
		__hip_atomic_store(look + (size_t)tile * RS_BINS + t, tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		// the tiles before this one inside its chunk (they hold lower tickets: they run or have finished)
		u32 excl = d_rs_wait_sum<(RS_ITEMS > 8 ? 8 : 4)>(look + (size_t)firstt * RS_BINS + t, nb, RS_BINS, epoch);
		if (nb == RS_CHUNK - 1u) // the chunk is complete with this tile: publish its sum
			__hip_atomic_store(look2 + (size_t)chunk * RS_BINS + t, tag | (u64w)(excl + total), __ATOMIC_RELAXED,
					   __HIP_MEMORY_SCOPE_AGENT);
		// the chunks before this one (their last tiles hold lower tickets)
		excl += d_rs_wait_sum<(RS_ITEMS > 8 ? 8 : 4)>(look2 + t, chunk, RS_BINS, epoch);
		s_base[t] = gdigit + excl - lstart; // global position of slot j of digit d = s_base[d] + j
	}
	__syncthreads();
	// tile in digit order in LDS
#pragma unroll
	for (int i = 0; i < RS_ITEMS; i++) {
		if (FULL || first + (u32)i * 64u + lane < n) {
			const u32 d = (k[i] >> shift) & dmask;
			const u32 slot = s_cnt[wave][d] + r[i];
			s_keys[slot] = k[i];
			s_vals[slot] = v[i];
		}
	}
	__syncthreads();
	const u32 ntile = FULL ? (u32)RS_TILE : n - base;
#pragma unroll 4
	for (u32 j = t; j < ntile; j += RS_THREADS) {
		const u32 key = s_keys[j];
		const u32 pos = s_base[(key >> shift) & dmask] + j;
		kout[pos] = key;
		vout[pos] = s_vals[j];
	}
}

// ctl: {ticket, finished workgroups} of the launch.  Tickets are dealt over the segments in turn (segment 0's tiles
// first): inside a segment the tiles still come in ticket order, so a tile only ever waits for tiles that already run.
template <int RS_ITEMS, bool ATOMIC>
__global__ __launch_bounds__(RS_THREADS, RS_ITEMS == 16 ? 4 : 6) void k_rs_pass(RsBatch b, u32 shift, u32 *ctl, u32 epoch)
{
	constexpr u32 RS_TILE = RS_THREADS * RS_ITEMS;
	__shared__ u32 s_keys[RS_TILE], s_vals[RS_TILE];
	__shared__ u32 s_cnt[RS_WAVES][RS_BINS]; // per wave: digit counters while ranking, then the first slot of the wave's pairs of the digit
	__shared__ u32 s_base[RS_BINS];
	u32 *s_next = s_keys; // the following pass's digit is counted in the tile's key space, which is free until the reorder (RS_PRIV copies)
	static_assert(RS_BINS * RS_PRIV <= RS_THREADS * 8, "the privatised histogram must fit the smallest tile's keys");
	__shared__ u32 s_part[RS_WAVES];
	__shared__ u32 s_tile, s_last;
	const u32 t = threadIdx.x;
	if (t == 0)
		s_tile = atomicAdd(&ctl[0], 1u);
	// this pass's digit totals (complete: the kernel before wrote the rows) are asked for before anything else, for both
	// segments: which one this workgroup serves is known with the ticket
	u32 gd0 = 0, gd1 = 0;
	if (t < RS_BINS) {
#pragma unroll
		for (int c = 0; c < RS_COPIES; c++)
			gd0 += b.s[0].hist[c * RS_BINS + t];
		if (b.nseg > 1u) {
#pragma unroll
			for (int c = 0; c < RS_COPIES; c++)
				gd1 += b.s[1].hist[c * RS_BINS + t];
		}
	}
	for (u32 i = t; i < RS_BINS * RS_PRIV; i += RS_THREADS)
		s_next[i] = 0u;
	for (u32 i = t; i < RS_WAVES * RS_BINS; i += RS_THREADS)
		(&s_cnt[0][0])[i] = 0;
	__syncthreads();
	const u32 ticket = (u32)__builtin_amdgcn_readfirstlane((int)s_tile); // (uniform: row bases and loop bounds stay in scalar registers)
	const u32 si = b.nseg > 1u && ticket >= b.s[0].blocks ? 1u : 0u;
	const RsSeg &g = b.s[si];
	const u32 tile = ticket - (si ? b.s[0].blocks : 0u);
	const u32 gdigit_in = si ? gd1 : gd0;
	u32 n = g.n;
	if (g.n_dev)
		n = *g.n_dev < n ? *g.n_dev : n;
	const u32 base = tile * RS_TILE;
	// (a launch sized by the capacity: a tile beyond the pairs holds nothing, and no tile waits for a later one)
	if (base < n) {
		if (n - base >= RS_TILE) // (no bounds checks in a full tile: all but the last)
			d_rs_tile<RS_ITEMS, ATOMIC, true>(g.kin, g.vin, g.kout, g.vout, n, shift, g.dmask, gdigit_in, g.hist_next, g.nmask, g.look,
							  g.look2, epoch, tile, s_keys, s_vals, s_cnt, s_base, s_part, s_next);
		else
			d_rs_tile<RS_ITEMS, ATOMIC, false>(g.kin, g.vin, g.kout, g.vout, n, shift, g.dmask, gdigit_in, g.hist_next, g.nmask, g.look,
							   g.look2, epoch, tile, s_keys, s_vals, s_cnt, s_base, s_part, s_next);
	}
	// the last workgroup to finish leaves this pass's histogram rows, ticket and counter at zero for the next sort
	// (every workgroup has read the rows by now: a workgroup counts itself in after its own reads)
	__syncthreads();
	if (t == 0)
		s_last = atomicAdd(&ctl[1], 1u) == gridDim.x - 1u ? 1u : 0u;
	__syncthreads();
	if (s_last) {
		for (u32 i = t; i < RS_COPIES * RS_BINS; i += RS_THREADS) {
			b.s[0].hist[i] = 0u;
			if (b.nseg > 1u)
				b.s[1].hist[i] = 0u;
		}
		if (t == 0) {
			ctl[0] = 0u;
			ctl[1] = 0u;
		}
	}
}

// Does an LDS add-with-return serve the lanes of one instruction that hit the same word in lane order?  512 threads rank
// 32 rounds of pseudo-random digits (1 to 256 distinct ones per round, so every conflict degree occurs) both ways;
// bad[0] counts the ranks that differ.
__global__ __launch_bounds__(RS_THREADS) void k_rs_selftest(u32 *bad)
{
	__shared__ u32 s_a[RS_WAVES][RS_BINS], s_b[RS_WAVES][RS_BINS];
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	u32 wrong = 0;
	for (u32 round = 0; round < 32u; round++) {
		for (u32 i = t; i < RS_WAVES * RS_BINS; i += RS_THREADS) {
			(&s_a[0][0])[i] = 0;
			(&s_b[0][0])[i] = 0;
		}
		__syncthreads();
		u32 k[16], ra[16], rb[16];
		const u32 sel = round & 3u;
		const u32 dmask = sel == 0u ? (2u << ((round >> 2) & 7u)) - 1u : 255u; // random digits: 2, 4, ... 256 distinct ones
#pragma unroll
		for (int i = 0; i < 16; i++) {
			u32 x = (t * 16u + (u32)i) * 2654435761u + round * 40503u;
			x ^= x >> 15;
			x *= 2246822519u;
			x ^= x >> 13;
			k[i] = sel == 0u   ? x >> 7
			       : sel == 1u ? lane / (1u + (round >> 2)) + (u32)i * 3u // runs of 1..16 neighbouring lanes
			       : sel == 2u ? round                                    // every lane the same digit
					   : (x >> 9) & 3u;                           // four digits
		}
		d_rs_rank<16, true, true>(k, ra, s_a[wave], 0u, dmask, 0u, 0u, lane);
		d_rs_rank<16, false, true>(k, rb, s_b[wave], 0u, dmask, 0u, 0u, lane);
#pragma unroll
		for (int i = 0; i < 16; i++)
			wrong += ra[i] != rb[i] ? 1u : 0u;
		__syncthreads();
		for (u32 i = t; i < RS_WAVES * RS_BINS; i += RS_THREADS)
			wrong += (&s_a[0][0])[i] != (&s_b[0][0])[i] ? 1u : 0u;
		__syncthreads();
	}
	if (wrong)
		atomicAdd(bad, wrong);
}

// state: RS_MAXSEG heads (histogram rows [pass][copy][digit]), the launches' {ticket, finished} words per pass level,
// then per segment the look-back words of the running pass (tile counts, chunk sums) for `tiles` tiles each
static size_t rs_look_words64(u32 tiles)
{
	const u32 chunks = (tiles + RS_CHUNK - 1) / RS_CHUNK;
	return (size_t)chunks * RS_CHUNK * RS_BINS + (size_t)chunks * RS_BINS;
}
#define RS_HIST_WORDS (RS_MAXPASS * RS_COPIES * RS_BINS)
#define RS_HEADS_WORDS (RS_MAXSEG * RS_HIST_WORDS + 2 * RS_MAXPASS + 56)

static int rs_state(ugrt_ctx *ctx, u32 tiles)
{
	const size_t bytes = (size_t)RS_HEADS_WORDS * 4 + (size_t)RS_MAXSEG * rs_look_words64(tiles) * 8;
	if (bytes <= ctx->rs_state.cap && tiles <= ctx->rs_tiles)
		return UGRT_OK;
	if (tiles < ctx->rs_tiles)
		tiles = ctx->rs_tiles;
	// growing: between two sorts the heads are all zeros (every pass cleans up behind itself); only the first pass's rows
	// may hold what the producer of the coming sort's keys has counted
	DevBuf old = ctx->rs_state;
	ctx->rs_state = DevBuf();
	int rc = ugrt_buf_reserve(ctx, ctx->rs_state, (size_t)RS_HEADS_WORDS * 4 + (size_t)RS_MAXSEG * rs_look_words64(tiles) * 8);
	if (rc) {
		ctx->rs_state = old;
		return rc;
	}
	ctx->rs_tiles = tiles;
	UGRT_HIP(hipMemsetAsync(ctx->rs_state.p, 0, ctx->rs_state.cap, ctx->stream));
	if (ctx->rs_atomic_rank < 0) {
		// once per context: may the passes rank by LDS atomics on this device?  (the look-back words serve as scratch)
		u32 *bad = (u32 *)ctx->rs_state.p + RS_HEADS_WORDS, h_bad = 1;
		hipLaunchKernelGGL(k_rs_selftest, dim3(4), dim3(RS_THREADS), 0, ctx->stream, bad);
		UGRT_HIP(hipGetLastError());
		UGRT_HIP(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
		UGRT_HIP(hipMemsetAsync(bad, 0, 4, ctx->stream));
		UGRT_HIP(hipStreamSynchronize(ctx->stream));
		ctx->rs_atomic_rank = h_bad == 0u ? 1 : 0;
	}
	if (old.p) {
		if (ctx->rs_prehist)
			UGRT_HIP(hipMemcpyAsync(ctx->rs_state.p, old.p, (size_t)RS_COPIES * RS_BINS * 4, hipMemcpyDeviceToDevice, ctx->stream));
		UGRT_HIP(hipStreamSynchronize(ctx->stream)); // (a pass of the previous sort may still read the old words)
		(void)hipFree(old.p);
	}
	ctx->rs_epoch = 0;
	return UGRT_OK;
}

// the first pass's histogram rows, for the kernel that writes the next sort's keys (ugrt_rs_hist.h)
int ugrt_sort_first_digit(ugrt_ctx *ctx, RsFirst *out)
{
	int rc = rs_state(ctx, 1);
	if (rc)
		return rc;
	if (ctx->rs_prehist) // a producer ran and its sort never did (an error between them, or the producer is repeated)
		UGRT_HIP(hipMemsetAsync(ctx->rs_state.p, 0, (size_t)RS_COPIES * RS_BINS * 4, ctx->stream));
	ctx->rs_prehist = true;
	out->hist = (u32 *)ctx->rs_state.p;
	return UGRT_OK;
}

// Stable sorts of up to RS_MAXSEG independent lists of pairs on key bits [0, end_bit) in shared launches: one histogram
// kernel, one kernel per pass level (a list with fewer passes drops out).  kin/vin are left untouched, the results are in
// kout/vout.  n_dev != nullptr: the pair count lives on the device and n is the capacity the launches are sized for.
// prehist (a single list only): the kernel that wrote the keys has counted their first digit (ugrt_sort_first_digit).
int ugrt_sort_pairs_batch(ugrt_ctx *ctx, const RsJob *jobs, int njobs, bool prehist)
{
	if (njobs < 1 || njobs > RS_MAXSEG)
		return ugrt_fail(UGRT_EINVAL, "sort: %d lists in one batch (1..%d)", njobs, RS_MAXSEG);
	size_t nmax = 0, nsum = 0;
	for (int j = 0; j < njobs; j++) {
		nmax = jobs[j].n > nmax ? jobs[j].n : nmax;
		nsum += jobs[j].n;
	}
	if (ctx->rs_prehist && (!prehist || njobs != 1 || nsum == 0)) { // counts of a producer whose sort does not run: forget them
		UGRT_HIP(hipMemsetAsync(ctx->rs_state.p, 0, (size_t)RS_COPIES * RS_BINS * 4, ctx->stream));
		ctx->rs_prehist = false;
	}
	if (nsum == 0)
		return UGRT_OK;
	if (nmax > ((size_t)1 << 30))
		return ugrt_fail(UGRT_EINVAL, "sort: %zu pairs exceed 2^30", nmax);
	// pairs per thread: tiles of 4096 pairs finish a pass of up to ~1 M pairs sooner, tiles of 8192 are faster from 1 M
	// on (half the tickets and look-back rows; profiles/r04_sort_bench_*.json)
	const int items = ctx->opt[UGRT_OPT_SORT_ITEMS] > 0 ? (ctx->opt[UGRT_OPT_SORT_ITEMS] == 8 ? 8 : 16) : (nmax <= (3u << 18) ? 8 : 16);
	const u32 RS_TILE = (u32)(RS_THREADS * items);
	hipStream_t st = ctx->stream;
	int rc;
	if ((rc = rs_state(ctx, (u32)((nmax + RS_TILE - 1) / RS_TILE))))
		return rc;
	int end_bit[RS_MAXSEG], passes[RS_MAXSEG], maxpasses = 0;
	u32 tiles[RS_MAXSEG];
	for (int j = 0; j < njobs; j++) {
		end_bit[j] = jobs[j].end_bit < 1 ? 1 : (jobs[j].end_bit > 32 ? 32 : jobs[j].end_bit);
		passes[j] = jobs[j].n ? (end_bit[j] + 7) / 8 : 0;
		maxpasses = passes[j] > maxpasses ? passes[j] : maxpasses;
		tiles[j] = (u32)((jobs[j].n + RS_TILE - 1) / RS_TILE);
		if (passes[j] > 1 && ((rc = ugrt_buf_reserve(ctx, ctx->rs_tmp[j][0], jobs[j].n * 4)) ||
				      (rc = ugrt_buf_reserve(ctx, ctx->rs_tmp[j][1], jobs[j].n * 4))))
			return rc;
	}
	u32 *heads = (u32 *)ctx->rs_state.p, *ctl = heads + RS_MAXSEG * RS_HIST_WORDS;
	u64w *look0 = (u64w *)(heads + RS_HEADS_WORDS);
	const size_t look_stride = rs_look_words64(ctx->rs_tiles);
	auto bits_of_pass = [&](int j, int p) { return (u32)(end_bit[j] - 8 * p) < 8u ? (u32)(end_bit[j] - 8 * p) : 8u; };
	// the segments of a launch: the lists that still have a pass at level p, in job order
	auto batch_of = [&](int p, const u32 *const *ki, const u32 *const *vi, u32 *const *ko, u32 *const *vo, const u32 *blocks, RsBatch *b) {
		b->nseg = 0;
		for (int j = 0; j < njobs; j++) {
			if (passes[j] <= p)
				continue;
			RsSeg &g = b->s[b->nseg++];
			g.kin = ki[j], g.vin = vi[j], g.kout = ko[j], g.vout = vo[j];
			g.n_dev = jobs[j].n_dev;
			g.hist = heads + (size_t)j * RS_HIST_WORDS + (size_t)p * RS_COPIES * RS_BINS;
			g.hist_next = heads + (size_t)j * RS_HIST_WORDS + (size_t)(p + 1 < passes[j] ? p + 1 : p) * RS_COPIES * RS_BINS;
			g.look = look0 + (size_t)j * look_stride;
			g.look2 = g.look + (size_t)((ctx->rs_tiles + RS_CHUNK - 1) / RS_CHUNK) * RS_CHUNK * RS_BINS;
			g.n = (u32)jobs[j].n;
			g.dmask = (1u << bits_of_pass(j, p)) - 1u;
			g.nmask = p + 1 < passes[j] ? (1u << bits_of_pass(j, p + 1)) - 1u : 0u;
			g.blocks = blocks[j];
		}
		for (u32 k = b->nseg; k < RS_MAXSEG; k++)
			b->s[k] = b->s[0];
	};
	const u32 *ki[RS_MAXSEG], *vi[RS_MAXSEG];
	u32 *ko[RS_MAXSEG], *vo[RS_MAXSEG];
	for (int j = 0; j < njobs; j++)
		ki[j] = jobs[j].kin, vi[j] = jobs[j].vin, ko[j] = jobs[j].kout, vo[j] = jobs[j].vout;
	if (prehist && njobs == 1 && ctx->rs_prehist) {
		ctx->rs_prehist = false; // (the first pass reads the rows and its last workgroup clears them)
	} else {
		u32 hblocks[RS_MAXSEG], total = 0;
		for (int j = 0; j < njobs; j++) {
			u32 h = (u32)((jobs[j].n + RS_THREADS * 8 - 1) / (RS_THREADS * 8));
			hblocks[j] = passes[j] ? (h > 256u ? 256u : (h ? h : 1u)) : 0u;
			total += hblocks[j];
		}
		RsBatch b;
		batch_of(0, ki, vi, ko, vo, hblocks, &b);
		hipLaunchKernelGGL(k_rs_hist, dim3(total), dim3(RS_THREADS), 0, st, b);
		UGRT_HIP(hipGetLastError());
		ctx->rs_launches++;
	}
	// ranks by LDS atomics where the device serves them in lane order (checked once per context), unless "sort_rank" is 0
	const bool atomic_rank = ctx->rs_atomic_rank == 1 && ctx->opt[UGRT_OPT_SORT_RANK] != 0;
	for (int p = 0; p < maxpasses; p++) {
		u32 total = 0, blocks[RS_MAXSEG];
		for (int j = 0; j < njobs; j++) {
			// the buffers alternate so that a list's last pass writes the caller's output
			const bool to_out = ((passes[j] - 1 - p) & 1) == 0;
			ko[j] = to_out ? jobs[j].kout : (u32 *)ctx->rs_tmp[j][0].p;
			vo[j] = to_out ? jobs[j].vout : (u32 *)ctx->rs_tmp[j][1].p;
			blocks[j] = passes[j] > p ? tiles[j] : 0u;
			total += blocks[j];
		}
		RsBatch b;
		batch_of(p, ki, vi, ko, vo, blocks, &b);
		if (++ctx->rs_epoch == 0u) { // (2^32 passes later: old tags could be taken for new ones)
			UGRT_HIP(hipMemsetAsync(look0, 0, ctx->rs_state.cap - (size_t)RS_HEADS_WORDS * 4, st));
			ctx->rs_epoch = 1;
		}
#define RS_LAUNCH(ITEMS, ATOMIC) \
	hipLaunchKernelGGL((k_rs_pass<ITEMS, ATOMIC>), dim3(total), dim3(RS_THREADS), 0, st, b, (u32)(8 * p), ctl + 2 * p, ctx->rs_epoch)
		if (items == 8) {
			if (atomic_rank)
				RS_LAUNCH(8, true);
			else
				RS_LAUNCH(8, false);
		} else {
			if (atomic_rank)
				RS_LAUNCH(16, true);
			else
				RS_LAUNCH(16, false);
		}
#undef RS_LAUNCH
		ctx->rs_launches++;
		UGRT_HIP(hipGetLastError());
		for (int j = 0; j < njobs; j++)
			if (passes[j] > p)
				ki[j] = ko[j], vi[j] = vo[j];
	}
	return UGRT_OK;
}

// stable sort of n pairs on key bits [0, end_bit); kin/vin are left untouched, the result is in kout/vout
int ugrt_sort_pairs_u32(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n, int end_bit,
			const u32 *n_dev, bool prehist)
{
	const RsJob job = { kin, vin, kout, vout, n, end_bit, n_dev };
	return ugrt_sort_pairs_batch(ctx, &job, 1, prehist);
}
