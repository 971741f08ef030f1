"""DDA kernels on the bench workload, alone on the GPU: the window kernel against the per-ray kernel of round 1
(identical results checked), launch-shape sweep, and the window kernel's work-sharing counters.

    python tools/dda_sweep.py [--quick] [--out FILE.json]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench

quick = "--quick" in sys.argv
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
dims = (128, 128, 64)


def make(flags):
    ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=flags, uniform_dims=dims)
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    for _ in range(2):
        r.display(setup, reflect=True)
    ctx.synchronize()
    return ctx, r


ctx, r = make(ugrt.FLAG_SHADOW_ALL_CHUNKS)
uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)


def run(ctx, r, n=5):
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


res = {"rows": []}
ctx.set_option("dda_kernel", 1)
ctx.set_option("dda_rays_per_wave", 32)
ms_ray = run(ctx, r)
ref_t, ref_id = r.hit_t.clone(), r.hit_id.clone()
print("per-ray kernel, 32 rays per wave: %.3f ms" % ms_ray, flush=True)
res["per_ray_rpw32_ms"] = ms_ray
for kernel in (0,):  # the window kernel (round 2's beam kernel went in round 4)
    for sort in (0, 1):
        for rpw in ((64,) if (quick or sort) else (16, 32, 64)):
            for cull_min in ((8,) if (quick or sort) else (8, 4, 16, 1 << 30)):
                ctx.set_option("dda_kernel", kernel)
                ctx.set_option("dda_sort", sort)
                ctx.set_option("dda_rays_per_wave", rpw)
                ctx.set_option("dda_cull_min", cull_min)
                r.hit_t.fill_(7.0)
                r.hit_id.fill_(7)
                ms = run(ctx, r)
                same = bool((r.hit_id == ref_id).all()) and bool((r.hit_t.view(torch.int32) == ref_t.view(torch.int32)).all())
                print("kernel %d sort %d rpw %2d cull_min %10d : %.3f ms  identical=%s" % (kernel, sort, rpw, cull_min, ms, same),
                      flush=True)
                res["rows"].append({"kernel": kernel, "sort": sort, "rpw": rpw, "cull_min": cull_min, "ms": ms,
                                    "identical": same})
ctx.set_option("dda_sort", 0)
ctx.set_option("dda_kernel", 0)
for rpw in (32, 64):
    for cull_min, cull_work in ((8, 1), (8, 128), (8, 320), (8, 640), (8, 1280), (16, 320), (4, 320)):
        ctx.set_option("dda_rays_per_wave", rpw)
        ctx.set_option("dda_cull_min", cull_min)
        ctx.set_option("dda_cull_work", cull_work)
        ms = run(ctx, r)
        same = bool((r.hit_id == ref_id).all()) and bool((r.hit_t.view(torch.int32) == ref_t.view(torch.int32)).all())
        print("kernel 0 rpw %2d cull_min %3d cull_work %5d : %.3f ms  identical=%s" % (rpw, cull_min, cull_work, ms, same), flush=True)
        res["rows"].append({"kernel": 0, "rpw": rpw, "cull_min": cull_min, "cull_work": cull_work, "ms": ms, "identical": same})
ctx.set_option("dda_cull_work", -1)
ctx.set_option("dda_cull_min", -1)
ctx.set_option("dda_rays_per_wave", -1)
# work counters
cctx, cr = make(ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_COUNT_WORK)
for k, sort in ((1, 0), (2, 0), (0, 0), (0, 1)):
    cctx.set_option("dda_kernel", k)
    cctx.set_option("dda_sort", sort)
    uv, us, uo, _ = cctx.grid_ptrs(ugrt.GRID_UNIFORM)
    cctx.trace_dda(uv, us, uo, cr.d_verts, cr.d_faces, cr.rays, cr.active, cr.hit_t, cr.hit_id)
    st = cctx.stats()
    print("kernel %d sort %d: tests %d cells %d rays %d" % (k, sort, st[3], st[4], st[5]), flush=True)
    res["work_kernel%d_sort%d" % (k, sort)] = {"tests": st[3], "cells": st[4], "rays": st[5]}
    if k != 1:
        d = cctx.stats_dda(kernel=k)
        print("sharing:", d, flush=True)
        res["sharing_kernel%d_sort%d" % (k, sort)] = d
        res["algorithmic_bytes"] = 48 * st[5] + 8 * st[4] + 52 * st[3]
if out:
    json.dump(res, open(out, "w"), indent=1)
