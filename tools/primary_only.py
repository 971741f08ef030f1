"""A few launches of the camera pass alone on the bench workload (for rocprofv3 --pmc / --kernel-trace).

    python tools/primary_only.py ORDER CHUNK [launches]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
order, chunk = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
ctx.set_option("primary_order", order)
ctx.set_option("primary_chunk", chunk)
for _ in range(n):
    r.display(setup, shadows=False)
ctx.synchronize()
print("done", flush=True)
