"""What a sharded build of the light grid and the uniform grid costs and saves, measured on ONE GPU:
the build of a 1/N window of the triangle list against the full build, and the merge of N parts.  The exchange
itself (an all-gather of the shards over xGMI) cannot be measured here; its volume is printed.

    python tools/shard_cost.py [--out FILE.json]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, ugrt, bench
from ugrt import parallel

out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
W, H = 1920, 1080
ctx = ugrt.Context(W, H, light_grid=(128, 128), flags=ugrt.FLAG_STATIC_GEOMETRY, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
lcam = ugrt.renderer.make_camera(setup.light_camera, setup.fovy, r.aspect)
PI = float(np.float32(np.pi))
F = r.F


def build(which):
    if which == ugrt.GRID_SPHERICAL:
        ctx.upload_camera(lcam.camcoords)
        ctx.grid_build_spherical(r.d_faces, r.d_verts, F, PI, PI)
    else:
        ctx.grid_build_uniform(r.d_faces, r.d_verts, F, r.bbmin, r.bbmax)


def timed(fn, n=12):
    """GPU time of fn per call: the sum of the hipEvent-bracketed stages it runs (host waits between them not counted)."""
    for _ in range(2):
        fn()
    ctx.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(n):
        fn()
    ctx.synchronize()
    p = ctx.prof_get()
    ctx.prof_enable(False)
    return sum(v[0] for v in p.values()) / n


res = {"builds_ms": {}, "merge_ms": {}, "exchange_bytes_per_rank": {}}
for name, which in (("light", ugrt.GRID_SPHERICAL), ("uniform", ugrt.GRID_UNIFORM)):
    for world in (1, 2, 4, 8):
        # the slowest window of the partition sets the frame: time every window, keep the maximum
        worst, refs = 0.0, []
        for k in range(world):
            ctx.set_face_window(*parallel.face_window(k, world, F))
            ms = timed(lambda: build(which), 12)
            worst = max(worst, ms)
            refs.append(ctx.grid_info(which).total_refs)
        res["builds_ms"]["%s/%d" % (name, world)] = {"slowest_window_ms": round(worst, 4), "refs_per_window": refs}
        print("%s grid, %d windows: slowest window %.3f ms, refs %s" % (name, world, worst, refs), flush=True)
        if world > 1:
            parts = []
            for k in range(world):
                ctx.set_face_window(*parallel.face_window(k, world, F))
                build(which)
                value, key, span, offset, gi = ctx.grid_arrays(which)
                parts.append((key[:gi.total_refs].clone(), value[:gi.total_refs].clone(), span.clone(), gi.total_refs))
            ms = timed(lambda: ctx.grid_merge_shards(which, [p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts],
                                                     [p[3] for p in parts]), 12)
            C = parts[0][2].numel()
            vol = sum(8 * p[3] for p in parts) + 4 * C * world  # what every rank receives: keys + values + spans of all shards
            res["merge_ms"]["%s/%d" % (name, world)] = round(ms, 4)
            res["exchange_bytes_per_rank"]["%s/%d" % (name, world)] = int(vol * (world - 1) / world)
            print("   merge of %d parts %.3f ms; every rank receives %.1f MB from its %d peers" % (world, ms, vol * (world - 1) / world / 1e6, world - 1),
                  flush=True)
    ctx.set_face_window(0, -1)
if out:
    json.dump(res, open(out, "w"), indent=1)
