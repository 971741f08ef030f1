"""Sweep one of the shadow tracer's options (ugrt_ctx_set_option) on the bench workload.

    python tools/env_sweep.py shadow_mbits 12 16 17
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
var, vals = sys.argv[1], sys.argv[2:]
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ref = None
st = ("trace_shadow", "shadow_prep", "shadow_cull")
for v in vals:
    ctx.set_option(var, int(v))
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    ctx.prof_enable(True, stages=st); ctx.prof_reset()
    for _ in range(10):
        r.display(setup, reflect=True)
    p = ctx.prof_get(); ctx.prof_enable(False)
    sh = r.is_shadowed.clone()
    if ref is None: ref = sh
    t = [p[k][0] / 10 for k in st]
    print("%s=%-8s exact %.3f prep %.3f cull %.3f sum %.3f ms  pairs %d same=%s" % (
        var, v, t[0], t[1], t[2], sum(t), ctx.stats()[7], bool((sh == ref).all())), flush=True)
