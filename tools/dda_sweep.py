"""Sweep the DDA tunables on the bench workload (one process per setting is not needed: read per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(3):
    r.display(setup, reflect=True)
ctx.synchronize()
uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
ctx.prof_enable(True)
for rpw in (4, 8, 16, 32, 64):
    for coop in (4, 8, 16, 32, 1 << 30):
        os.environ["UGRT_DDA_RPW"], os.environ["UGRT_DDA_COOP"] = str(rpw), str(coop)
        ctx.prof_reset()
        for _ in range(5):
            ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
        p = ctx.prof_get()["trace_dda"]
        print("rpw %2d coop %10d : %.3f ms" % (rpw, coop, p[0] / p[1]), flush=True)
