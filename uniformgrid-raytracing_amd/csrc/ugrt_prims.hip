// ugrt_prims.hip -- the data-parallel primitives of the grid build, on rocPRIM.
//
// Replaces the reference's CUDPP 1.1 call sites (cudpp/cudpp.h:426-471):
//   cudppScan inclusive/exclusive  frustum_grid.h:249,361  -> ugrt_scan.hip (own single-kernel scan)
//   cudppSort key-value radix      frustum_grid.h:298, decision_data.h:177
//                                  -> rocprim::radix_sort_pairs (stable LSD radix sort)
// The reference sorts all 32 key bits although keys < number of cells
// (frustum_grid.h:298); here only ceil(log2(cells)) bits are sorted, which
// gives the same permutation in half the passes.
#include <cstring> // must precede rocprim on ROCm 7.2

#include <rocprim/rocprim.hpp>

#include "ugrt_ctx.h"

// (the prefix sums live in ugrt_scan.hip)

int ugrt_prim_sort_pairs(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n,
			 int end_bit, const u32 *n_dev)
{
	if (!n_dev && (ctx->opt[UGRT_OPT_SORT_LIBRARY] == 1 || n > ((size_t)1 << 30)))
		return ugrt_prim_sort_pairs_rocprim(ctx, kin, kout, vin, vout, n, end_bit);
	return ugrt_sort_pairs_u32(ctx, kin, kout, vin, vout, n, end_bit, n_dev);
}

int ugrt_prim_sort_pairs_rocprim(ugrt_ctx *ctx, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n,
				 int end_bit)
{
	if (n == 0)
		return UGRT_OK;
	if (end_bit < 1)
		end_bit = 1;
	if (end_bit > 32)
		end_bit = 32;
	size_t bytes = 0;
	UGRT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit,
					   ctx->stream));
	int rc = ugrt_buf_reserve(ctx, ctx->temp, bytes);
	if (rc)
		return rc;
	UGRT_HIP(rocprim::radix_sort_pairs(ctx->temp.p, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit,
					   ctx->stream));
	return UGRT_OK;
}

// 64-bit keys: the shadow tracer's private re-grouping of rays (light cell, direction Morton code)
int ugrt_prim_sort_pairs64(ugrt_ctx *ctx, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n,
			   int end_bit)
{
	if (n == 0)
		return UGRT_OK;
	if (end_bit < 1)
		end_bit = 1;
	if (end_bit > 64)
		end_bit = 64;
	size_t bytes = 0;
	UGRT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit,
					   ctx->stream));
	int rc = ugrt_buf_reserve(ctx, ctx->temp, bytes);
	if (rc)
		return rc;
	UGRT_HIP(rocprim::radix_sort_pairs(ctx->temp.p, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit,
					   ctx->stream));
	return UGRT_OK;
}

extern "C" int ugrt_sort_pairs(ugrt_ctx *ctx, const unsigned *d_keys_in, unsigned *d_keys_out,
			       const unsigned *d_values_in, unsigned *d_values_out, size_t n, int key_bits,
			       int use_library)
{
	if (!ctx || (n && (!d_keys_in || !d_keys_out || !d_values_in || !d_values_out)))
		return ugrt_fail(UGRT_EINVAL, "sort_pairs: null argument");
	if (key_bits < 1 || key_bits > 32)
		return ugrt_fail(UGRT_EINVAL, "sort_pairs: key_bits %d outside [1,32]", key_bits);
	if (n && (d_keys_in == d_keys_out || d_values_in == d_values_out))
		return ugrt_fail(UGRT_EINVAL, "sort_pairs: outputs alias inputs");
	UGRT_HIP(hipSetDevice(ctx->device));
	if (use_library)
		return ugrt_prim_sort_pairs_rocprim(ctx, d_keys_in, d_keys_out, d_values_in, d_values_out, n, key_bits);
	return ugrt_prim_sort_pairs(ctx, d_keys_in, d_keys_out, d_values_in, d_values_out, n, key_bits);
}
