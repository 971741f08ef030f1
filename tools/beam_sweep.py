"""Sweep the shadow tracer's rays-per-beam on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ref = None
for beam in (64, 512, 768, 1024, 1536, 2048, 4096):
    ctx.set_option("shadow_beam", beam)
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        r.display(setup, reflect=True)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
    p = ctx.prof_get(); ctx.prof_enable(False)
    sh = r.is_shadowed.clone()
    if ref is None: ref = sh
    print("beam %4d: frame %.3f ms cull %.3f exact %.3f worklist %.3f pairs %d same=%s" % (beam, dt * 1e3, p["shadow_cull"][0] / 10, p["trace_shadow"][0] / 10, p["worklist"][0] / 10, ctx.stats()[7], bool((sh == ref).all())), flush=True)
