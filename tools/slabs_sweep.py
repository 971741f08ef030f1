"""z-slabs (NUM_SLABS, main.cu.h:18): does the reference's front-to-back slab walk ever beat one slab here?
Camera pass (perspective build + primary tracer) and light grid + shadow pass per slab count.

    python tools/slabs_sweep.py [--out FILE.json]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
res = {"rows": []}
for wl, W, H, scale in (("hall", 1024, 1024, 1.0), ("crash", 1920, 1080, 1.0)):
    s = bench.load_scene(ugrt, wl, scale, 0)
    setup = ugrt.FrameSetup.from_scene(s)
    for slabs in (1, 2, 4, 8):
        ctx = ugrt.Context(W, H, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64), slabs=slabs)
        r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
        for _ in range(2):
            r.display(setup, shadows=True)
        ctx.synchronize()
        ctx.prof_enable(True); ctx.prof_reset()
        n = 5
        for _ in range(n):
            r.display(setup, shadows=True)
        ctx.synchronize()
        p = {k: v[0] / n for k, v in ctx.prof_get().items() if v[1]}
        ctx.prof_enable(False)
        build = sum(p.get(k, 0) for k in ("build_count", "build_scan", "build_fill", "build_sort", "build_bounds"))
        shadow = sum(p.get(k, 0) for k in ("shadow_prep", "shadow_cull", "trace_shadow"))
        row = {"workload": wl, "slabs": slabs, "builds_ms": round(build, 4), "trace_primary_ms": round(p.get("trace_primary", 0), 4),
               "shadow_ms": round(shadow, 4), "frame_gpu_ms": round(sum(p.values()), 4)}
        res["rows"].append(row)
        print(row, flush=True)
        del r, ctx
if out:
    json.dump(res, open(out, "w"), indent=1)
