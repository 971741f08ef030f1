"""Sweep the primary tracer's segment length on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ref = None
for seg in (256, 512, 1024, 2048, 4096, 16384):
    ctx.set_option("primary_seg", seg)
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(10):
        r.display(setup, reflect=True)
    p = ctx.prof_get(); ctx.prof_enable(False)
    t = r.t.clone()
    if ref is None: ref = t
    print("seg %5d: primary %.3f ms worklist %.3f same=%s" % (seg, p["trace_primary"][0] / 10, p["worklist"][0] / 10, bool((t == ref).all())), flush=True)
