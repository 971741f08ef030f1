"""Window kernel of the bounce alone on the bench workload: time at 32 and 64 rays per wave (identity against the per-ray
kernel checked), then the counting variant's phase stamps.  For before/after comparisons of a kernel change.

    python tools/dda_exp.py [--blocks N] [--out FILE.json]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench

out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
blocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 0
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)


def make(flags):
    ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=flags, uniform_dims=(128, 128, 64))
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    return ctx, r


def run(ctx, r, n=8):
    uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
    ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
    ctx.synchronize()
    ctx.prof_enable(True, stages=("trace_dda",))
    ctx.prof_reset()
    for _ in range(n):
        ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
    p = ctx.prof_get()["trace_dda"]
    ctx.prof_enable(False)
    return p[0] / p[1]


ctx, r = make(ugrt.FLAG_SHADOW_ALL_CHUNKS)
ctx.set_option("dda_kernel", 1)
ctx.set_option("dda_rays_per_wave", 32)
run(ctx, r, 1)
ref_t, ref_id = r.hit_t.clone(), r.hit_id.clone()
ctx.set_option("dda_kernel", 0)
res = {"rows": []}
splits = [int(x) for x in sys.argv[sys.argv.index("--splits") + 1].split(",")] if "--splits" in sys.argv else [0, 1]
loads = [int(x) for x in sys.argv[sys.argv.index("--loads") + 1].split(",")] if "--loads" in sys.argv else [200]
for rep in range(3):
    for rpw in (32, 64):
        for blk in ((blocks,) if blocks else (1024, 3072)):
            for split in splits:
                for load in (loads if split == 1 else loads[:1]):
                    ctx.set_option("dda_rays_per_wave", rpw)
                    ctx.set_option("dda_blocks", blk)
                    ctx.set_option("dda_split", split)
                    ctx.set_option("dda_split_load", load)
                    r.hit_t.fill_(7.0)
                    r.hit_id.fill_(7)
                    ms = run(ctx, r)
                    same = bool((r.hit_id == ref_id).all()) and bool((r.hit_t.view(torch.int32) == ref_t.view(torch.int32)).all())
                    print("rpw %2d waves %4d split %d load %4d: %.4f ms identical=%s %s" % (rpw, blk, split, load, ms, same, ctx.stats_dda_split() if split else ""), flush=True)
                    res["rows"].append({"rpw": rpw, "waves": blk, "split": split, "load": load, "ms": ms, "identical": same})
cctx, cr = make(ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_COUNT_WORK)
cctx.set_option("dda_kernel", 0)
for rpw in (32, 64):
    cctx.set_option("dda_rays_per_wave", rpw)
    uv, us, uo, _ = cctx.grid_ptrs(ugrt.GRID_UNIFORM)
    for blk in (3072, 1024):
        cctx.set_option("dda_blocks", blk)
        import time
        cctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            cctx.trace_dda(uv, us, uo, cr.d_verts, cr.d_faces, cr.rays, cr.active, cr.hit_t, cr.hit_id)
        cctx.synchronize()
        print("counting variant (host clock, with its prepare kernel and read-back), rpw %d, %d waves: %.4f ms"
              % (rpw, blk, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
    cctx.trace_dda(uv, us, uo, cr.d_verts, cr.d_faces, cr.rays, cr.active, cr.hit_t, cr.hit_id)
    d = cctx.stats_dda(kernel=0)
    print("rpw", rpw, json.dumps(d), flush=True)
    res["sharing_rpw%d" % rpw] = d
if out:
    json.dump(res, open(out, "w"), indent=1)
