"""Sweep the uniform grid resolution of the reflection bounce on the bench workload (build + DDA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
for dims in ((64, 64, 32), (96, 96, 48), (128, 128, 64), (160, 160, 80), (192, 192, 96), (256, 256, 128), (256, 128, 128)):
    ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=dims)
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    for _ in range(3):
        r.display(setup, reflect=True)
    ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        r.display(setup, reflect=True)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    p = ctx.prof_get(); ctx.prof_enable(False)
    gi = ctx.grid_ptrs(ugrt.GRID_UNIFORM)[3]
    print("dims %-16s refs %9d  dda %.3f ms  frame(with events) %.3f ms" % (dims, gi.total_refs, p["trace_dda"][0] / 10, dt), flush=True)
    del r, ctx
