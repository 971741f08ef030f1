"""Timeline of the bounce kernel's ray groups (start, duration, work counts), read from a debug array.

Needs a THROWAWAY instrumentation of k_trace_dda_walk that is not in the tree: a `__device__ unsigned long long
g_dbg[16384 * 4]` written by lane 0 at the end of every group with {s_memrealtime at its start, at its end,
jobs << 32 | windows, rounds << 32 | busy windows}, and `extern "C" int ugrt_debug_read(void *dst, size_t bytes)`
(hipMemcpyFromSymbol).  (s_memtime is per XCD and cannot be compared across groups; s_memrealtime counts 10 ns.)
Output of the round-3 run: profiles/r03_dda_timeline.txt.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ugrt, bench
from ugrt import host
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(2):
    r.display(setup, reflect=True)
ctx.synchronize()
import ugrt.device as dev
lib = dev.lib
for rpw, blk in ((32, 3072), (32, 1024), (64, 3072), (16, 3072)):
    ctx.set_option("dda_rays_per_wave", rpw); ctx.set_option("dda_blocks", blk)
    uv, us, uo, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
    for _ in range(3):
        ctx.trace_dda(uv, us, uo, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
    ctx.synchronize()
    buf = np.zeros(16384 * 4, dtype=np.uint64)
    rc = lib.ugrt_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    n = (155627 + rpw - 1) // rpw
    b4 = buf.reshape(-1, 4)[:n]
    a = b4[:, :2].astype(np.int64)
    jobs = (b4[:, 2] >> np.uint64(32)).astype(np.int64); win = (b4[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    rounds = (b4[:, 3] >> np.uint64(32)).astype(np.int64); busy = (b4[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    du = a[:, 1] - a[:, 0]
    print("rpw %d waves %d: groups %d; group duration mean %d median %d p90 %d p99 %d max %d; sum/slots %d"
          % (rpw, blk, n, du.mean(), np.median(du), np.percentile(du, 90), np.percentile(du, 99), du.max(), du.sum() / min(blk, 3072)))
    st = a[:, 0] - a[:, 0].min(); en = a[:, 1] - a[:, 0].min()
    T = en.max()
    print("   span %d ticks of 10 ns; slots busy %.2f" % (T, du.sum() / T / min(blk, 3072)))
    order = np.argsort(-du)[:12]
    print("   twelve longest (group, start, duration): ", [(int(g), int(st[g]), int(du[g])) for g in order])
    order = np.argsort(-en)[:12]
    print("   twelve last to end (group, start, duration): ", [(int(g), int(st[g]), int(du[g])) for g in order])
    for f in (0.1, 0.25, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
        t = f * T
        print("   at %.2f of the span: %d groups running, %d not started" % (f, int(((st <= t) & (en > t)).sum()), int((st > t).sum())))
    for name, x in (("jobs", jobs), ("windows", win), ("busy windows", busy), ("rounds", rounds), ("jobs + 2 rounds", jobs + 2 * rounds), ("3 busy + jobs + rounds", 3 * busy + jobs + rounds)):
        c = np.corrcoef(x, du)[0, 1]
        top = np.argsort(-du)[:50]
        print("   %s: mean %.1f max %d, correlation with duration %.3f; of the 50 longest groups, %d are among the 100 largest by this count"
              % (name, x.mean(), x.max(), c, len(set(top) & set(np.argsort(-x)[:100]))))
    # what the span would be with the groups started longest first on the same number of slots (durations as measured)
    import heapq
    for name, seq in (("as run (index order)", range(n)), ("longest first", np.argsort(-du))):
        slots = [0] * min(blk, 3072); heapq.heapify(slots)
        for g in seq:
            t = heapq.heappop(slots); heapq.heappush(slots, t + int(du[g]))
        print("   list scheduling, %s: span %d" % (name, max(slots)))
