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
#include "ugrt_scan.h"

static int scan_u32(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n, bool inclusive)
{
	// 16-byte accesses when both arrays allow them (hipMalloc is 256-B aligned; callers also pass offsets into buffers)
	ScanLoadArray load = { in, ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0 ? 1u : 0u };
	if (inclusive)
		return ugrt_scan_launch<true>(ctx, load, out, n, ScanTailNone());
	return ugrt_scan_launch<false>(ctx, load, out, n, ScanTailNone());
}

int ugrt_prim_inclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n) { return scan_u32(ctx, in, out, n, true); }
int ugrt_prim_exclusive_scan(ugrt_ctx *ctx, const u32 *in, u32 *out, size_t n) { return scan_u32(ctx, in, out, n, false); }
