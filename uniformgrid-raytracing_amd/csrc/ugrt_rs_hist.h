// ugrt_rs_hist.h -- layout of the radix sort's state (ugrt_sort.hip) and the digit counting shared with the kernels that
// write sort keys
#ifndef UGRT_RS_HIST_H
#define UGRT_RS_HIST_H

#include "ugrt_ctx.h"

#define RS_BINS 256
#define RS_MAXPASS 4
// A pass's digit histogram is kept in RS_COPIES rows that are summed when it is read: the workgroups that add to it (the
// histogram kernel or the producer of the keys for the first pass, the tiles of pass p for pass p + 1) pick a row by
// their index, so that an address sees an eighth of the adds (same-address atomics take ~12 ns each, whoever issues them).
#define RS_COPIES 8

// One more count of digit d in the workgroup's LDS histogram.  Neighbouring keys mostly share their digit (cell ids in
// fill order, beams in candidate order), and lanes that add to one word -- or to one bank -- are served one after the
// other.  The histogram is therefore kept RS_PRIV-fold, lane l adds to copy l % RS_PRIV, and the copies of a digit lie
// next to each other (in different banks): 64 equal digits are four adds deep instead of 64.  (Sorting out the wave's
// groups first -- one add per group by a leader lane -- was measured too: the scalar loop cost more than the adds.)
#define RS_PRIV 16
__device__ __forceinline__ void d_rs_count(u32 *s_h, u32 d, bool ok)
{
	if (ok)
		atomicAdd(&s_h[d * RS_PRIV + (threadIdx.x & (RS_PRIV - 1u))], 1u);
}
__device__ __forceinline__ u32 d_rs_count_sum(const u32 *s_h, u32 d)
{
	u32 c = 0;
#pragma unroll
	for (u32 k = 0; k < RS_PRIV; k++)
		c += s_h[d * RS_PRIV + ((k + d) & (RS_PRIV - 1u))]; // (rotated: the digit threads of a wave start in different banks)
	return c;
}

// A kernel that WRITES the keys of a sort may count their first digit itself (the sort then runs without its histogram
// kernel): it is handed the first pass's rows, keeps RS_BINS * RS_PRIV counters in LDS (zeroed, then a barrier), feeds
// every key through d_rs_count(s_h, key & 0xFF, ok) and ends with d_rs_flush behind a barrier.  Only for kernels of a
// bounded number of workgroups (<= ~1000: each ends with up to 256 global adds; round 3 measured the loss with 7 400).
struct RsFirst {
	u32 *hist; // nullptr: do not count
};
__device__ __forceinline__ void d_rs_zero(u32 *s_h, const RsFirst &h)
{
	if (h.hist)
		for (u32 i = threadIdx.x; i < RS_BINS * RS_PRIV; i += blockDim.x)
			s_h[i] = 0u;
}
__device__ __forceinline__ void d_rs_flush(const u32 *s_h, const RsFirst &h)
{
	if (h.hist)
		for (u32 d = threadIdx.x; d < RS_BINS; d += blockDim.x) {
			const u32 c = d_rs_count_sum(s_h, d);
			if (c)
				atomicAdd(&h.hist[(blockIdx.x % RS_COPIES) * RS_BINS + d], c);
		}
}
// host side (ugrt_sort.hip): the rows for the producer of the next sort's keys; that sort must then be told (prehist)
int ugrt_sort_first_digit(ugrt_ctx *ctx, RsFirst *out);

#endif
