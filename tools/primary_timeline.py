"""Timeline of the primary tracer's work items (start, duration, triangle count), and list-scheduling simulations.

Needs a THROWAWAY instrumentation of k_trace_primary that is not in the tree: `__device__ unsigned long long
g_dbgp[65536 * 2]`, per item {s_memrealtime at its start, duration << 32 | count}, and
`extern "C" int ugrt_debug_read_primary(void *dst, size_t bytes)`.  Output of the round-3 run:
profiles/r03_primary_timeline.txt.  (The simulations assume an item's duration does not depend on what runs beside it;
the one ordering that was then built - cells above the average first - made the launch SLOWER, 0.304 -> 0.337 ms.)
"""
import ctypes, os, sys, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(3):
    r.display(setup, reflect=True)
ctx.synchronize()
import ugrt.device as dev
buf = np.zeros(65536 * 2, dtype=np.uint64)
dev.lib.ugrt_debug_read_primary(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
b = buf.reshape(-1, 2)
n = int((b[:, 0] != 0).sum())
st = b[:n, 0].astype(np.int64); du = (b[:n, 1] >> np.uint64(32)).astype(np.int64); cnt = (b[:n, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
st -= st.min(); en = st + du; T = en.max()
print("items %d; span %d ticks of 10 ns; item duration mean %.0f median %.0f p90 %.0f p99 %.0f max %d; count mean %.0f max %d; corr(count, duration) %.3f"
      % (n, T, du.mean(), np.median(du), np.percentile(du, 90), np.percentile(du, 99), du.max(), cnt.mean(), cnt.max(), np.corrcoef(cnt, du)[0, 1]))
for f in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    t = f * T
    print("   at %.2f of the span: %d items running, %d not started" % (f, int(((st <= t) & (en > t)).sum()), int((st > t).sum())))
order = np.argsort(-en)[:10]
print("   last to end (item, start, duration, count):", [(int(g), int(st[g]), int(du[g]), int(cnt[g])) for g in order])
order = np.argsort(-du)[:10]
print("   longest (item, start, duration, count):", [(int(g), int(st[g]), int(du[g]), int(cnt[g])) for g in order])
print("   busy: sum of durations / span / 4096 slots = %.2f" % (du.sum() / T / 4096))
def simulate(order, waves=16384, slots=4044):
    # waves are dispatched in index order to the first free slot; wave w runs items order[w], order[w + waves], ... in turn
    free = [0] * slots; heapq.heapify(free); end = 0
    for w in range(min(waves, len(order))):
        t = heapq.heappop(free)
        for k in range(w, len(order), waves):
            t += int(du[order[k]])
        heapq.heappush(free, t); end = max(end, t)
    return end
ident = np.arange(n)
print("   simulated span, list order, 16384 waves: %d" % simulate(ident))
print("   simulated span, items by count descending, 16384 waves: %d" % simulate(np.argsort(-cnt, kind="stable")))
print("   simulated span, items by duration descending (bound), 16384 waves: %d" % simulate(np.argsort(-du, kind="stable")))
print("   simulated span, list order, one item per wave: %d" % simulate(ident, waves=n))
print("   simulated span, by count descending, one item per wave: %d" % simulate(np.argsort(-cnt, kind="stable"), waves=n))
# buckets of the count (what a counting pass could do): 8 buckets by count >> 7
print("   simulated span, 8 count buckets descending, 16384 waves: %d" % simulate(np.argsort(-(cnt >> 7), kind="stable")))
print("   simulated span, 2 buckets (count > mean first), 16384 waves: %d" % simulate(np.argsort(-(cnt > cnt.mean()).astype(int), kind="stable")))
