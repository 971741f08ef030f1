"""The primary tracer's flush order on the bench workload: list order against nearest-first, and the number of jobs
between two looks at the rays' closest hits.  Results are compared bit for bit; the work counters of both orders
come from a counting context.

    python tools/primary_sweep.py [--out FILE.json] [--workload crash|hall]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "crash"
s = bench.load_scene(ugrt, wl, 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
W, H = (1920, 1080) if wl == "crash" else (1024, 1024)
res = {"rows": []}
ctx = ugrt.Context(W, H, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ref = None
for rep in range(2):  # twice: the first pass of a process also warms the clocks
    for order, chunk in ((0, 64), (1, 64), (1, 32), (1, 16), (1, 8), (0, 32)):
        ctx.set_option("primary_order", order)
        ctx.set_option("primary_chunk", chunk)
        for _ in range(2):
            r.display(setup, shadows=False)
        ctx.synchronize()
        ctx.prof_enable(True); ctx.prof_reset()
        for _ in range(10):
            r.display(setup, shadows=False)
        p = ctx.prof_get(); ctx.prof_enable(False)
        cur = (r.t.clone(), r.intersect_id.clone(), r.normal.clone())
        if ref is None:
            ref = cur
        same = all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(cur, ref))
        ms = p["trace_primary"][0] / 10
        print("order %d chunk %2d: primary %.4f ms  identical=%s" % (order, chunk, ms, same), flush=True)
        if rep:
            res["rows"].append({"order": order, "chunk": chunk, "ms": ms, "identical": same})
for order, chunk in ((0, 64), (1, 32)):
    c = ugrt.Context(W, H, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_COUNT_WORK, uniform_dims=(128, 128, 64))
    c.set_option("primary_order", order)
    c.set_option("primary_chunk", chunk)
    cr = ugrt.Renderer(c, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    cr.display(setup, shadows=False)
    c.synchronize()
    d = c.stats_primary()
    print("order %d chunk %d:" % (order, chunk), d, flush=True)
    res["work_order%d_chunk%d" % (order, chunk)] = d
if out:
    json.dump(res, open(out, "w"), indent=1)
