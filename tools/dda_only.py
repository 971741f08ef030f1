"""A few launches of the DDA alone on the bench workload (for rocprofv3 --pmc / --kernel-trace).

    python tools/dda_only.py KERNEL RPW [launches [dda_split]]      KERNEL 0 = window, 1 = per-ray
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
kernel, rpw = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ctx.set_option("dda_kernel", kernel)
ctx.set_option("dda_rays_per_wave", rpw)
if len(sys.argv) > 4:
    ctx.set_option("dda_split", int(sys.argv[4]))
r.display(setup, reflect=True)
ctx.synchronize()
uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
for _ in range(n):
    ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
ctx.synchronize()
print("done", flush=True)
