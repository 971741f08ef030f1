"""Sweep the exact pass's candidates-per-item on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ref = None
for xseg in (128, 256, 512, 1024, 4096, 1 << 20):
    os.environ["UGRT_SHADOW_XSEG"] = str(xseg)
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    ctx.prof_enable(True, stages=("trace_shadow", "shadow_prep")); ctx.prof_reset()
    for _ in range(10):
        r.display(setup, reflect=True)
    p = ctx.prof_get(); ctx.prof_enable(False)
    sh = r.is_shadowed.clone()
    if ref is None: ref = sh
    print("xseg %7d: exact %.3f ms prep %.3f same=%s" % (xseg, p["trace_shadow"][0] / 10, p["shadow_prep"][0] / 10, bool((sh == ref).all())), flush=True)
