import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(3):
    r.display(setup, reflect=True)
ctx.synchronize()
a = (C.c_ulonglong * (8 * 8192))()
ugrt.lib.ugrt_debug_cull_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
print("rc", ugrt.lib.ugrt_debug_cull_cycles(a, 8 * 8192))
v = np.array(a, dtype=np.int64).reshape(-1, 8)
cyc, items, iters = v[:, 0], v[:, 1], v[:, 2]
print("waves", (cyc > 0).sum(), "items total", items.sum(), "iterations total", iters.sum())
print("cycles (100 MHz ticks?) mean %.0f max %d  p50 %d p90 %d p99 %d" % (cyc.mean(), cyc.max(), np.percentile(cyc, 50), np.percentile(cyc, 90), np.percentile(cyc, 99)))
print("items per wave: min %d max %d; iterations per wave: mean %.1f max %d" % (items.min(), items.max(), iters.mean(), iters.max()))
order = np.argsort(-cyc)[:10]
print("slowest waves: (block, cycles, items, iters)", [(int(b), int(cyc[b]), int(items[b]), int(iters[b])) for b in order])
print("corr cycles~iters", np.corrcoef(cyc, iters)[0, 1])

setup_c, flush_c, fp = v[:, 3], v[:, 4], v[:, 5]
nfl, npairs = fp >> 32, fp & 0xFFFFFFFF
print("setup cycles per wave mean %.0f (%.0f per item); flush cycles mean %.0f; flushes total %d, pairs total %d" % (setup_c.mean(), setup_c.sum() / items.sum(), flush_c.mean(), nfl.sum(), npairs.sum()))
print("share of wave time: setup %.2f flush %.2f rest (beam loop) %.2f" % (setup_c.sum() / cyc.sum(), flush_c.sum() / cyc.sum(), 1 - (setup_c.sum() + flush_c.sum()) / cyc.sum()))

first, tstart = v[:, 6], v[:, 7]
print("prologue cycles mean %.0f max %d; start spread (max-min of t_start) %d; end spread: last end - first start %d" % (first.mean(), first.max(), tstart.max() - tstart.min(), (tstart + cyc).max() - tstart.min()))
