// ugrt_rs_hist.h -- digit histograms of the radix sort (ugrt_sort.hip), accumulated by the kernel that WRITES the
// keys instead of by a histogram kernel that reads them again.  A producer kernel holds a block of
// RS_MAXPASS x 256 counters in LDS, feeds every key it stores through d_rs_hist_add and adds its counters to the
// sort's global histogram when it ends.  Producers run a bounded number of workgroups (grid-stride), so the
// global adds stay a few hundred thousand at most.
#ifndef UGRT_RS_HIST_H
#define UGRT_RS_HIST_H

#include "ugrt_ctx.h"

#define RS_BINS 256
#define RS_MAXPASS 4

// where a producer accumulates: the histogram rows of the sort that will run on its keys (passes == 0: nowhere)
struct RsHist {
	u32 *hist;
	u32 end_bit;
	int passes;
};

// all threads of the block; a barrier must follow before the first add
__device__ __forceinline__ void d_rs_hist_zero(u32 *s_h, const RsHist &h)
{
	for (int i = (int)threadIdx.x; i < h.passes * RS_BINS; i += (int)blockDim.x)
		s_h[i] = 0u;
}

// one key per lane (`ok` = this lane has one); neighbouring keys mostly share their upper digits (cell ids in
// fill order, beams in candidate order): one LDS add per wave then
__device__ __forceinline__ void d_rs_hist_add(u32 *s_h, const RsHist &h, u32 key, bool ok)
{
	const unsigned long long act = __ballot(ok);
	if (act == 0ull)
		return;
	for (int p = 0; p < h.passes; p++) {
		const u32 bits = h.end_bit - 8u * (u32)p < 8u ? h.end_bit - 8u * (u32)p : 8u;
		const u32 d = (key >> (8 * p)) & ((1u << bits) - 1u);
		const u32 d0 = (u32)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(act));
		if (__ballot(ok && d == d0) == act) {
			if ((threadIdx.x & 63u) == (u32)__builtin_ctzll(act))
				atomicAdd(&s_h[p * RS_BINS + d0], (u32)__popcll(act));
		} else if (ok) {
			atomicAdd(&s_h[p * RS_BINS + d], 1u);
		}
	}
}

// all threads of the block, behind a barrier that follows the last add
__device__ __forceinline__ void d_rs_hist_flush(const u32 *s_h, const RsHist &h)
{
	for (int i = (int)threadIdx.x; i < h.passes * RS_BINS; i += (int)blockDim.x) {
		const u32 c = s_h[i];
		if (c)
			atomicAdd(&h.hist[i], c);
	}
}

// sort sites: every producer -> sort pair of a frame has histogram rows of its own, so the producers of one sort
// may run before the passes of another
enum { RS_SITE_MISC = 0, RS_SITE_GRID0, RS_SITE_GRID1, RS_SITE_GRID2, RS_SITE_SHADOW_RAYS, RS_SITE_SHADOW_PAIRS,
       RS_SITE_SHADOW_ITEMS, RS_SITES };

// host side (ugrt_sort.hip): the rows a producer adds to, for a sort of `end_bit` key bits at `site`; the sort that
// follows must be told so (prehist = true).  A producer that cannot know the key width yet passes 32: its keys have
// no bits above the width the sort will be given, so the rows the sort uses are the same and the sort clears the
// others.  Fails only when the state cannot be allocated.
int ugrt_sort_hist_arg(ugrt_ctx *ctx, int site, int end_bit, RsHist *out);
int ugrt_sort_hist_reset(ugrt_ctx *ctx, int site);
// stable sort of n pairs on key bits [0, end_bit); kin/vin are left untouched, the result is in kout/vout
int ugrt_sort_pairs_site(ugrt_ctx *ctx, int site, bool prehist, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout,
			 size_t n, int end_bit, const u32 *n_dev);

#endif
