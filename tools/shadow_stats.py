"""Work accounting of the shadow tracer on the bench workload (COUNT_WORK context)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_COUNT_WORK, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
r.display(setup, reflect=True)
ctx.synchronize()
st = ctx.stats()
print("stats", st)
print("shadow beams", st[1], "chunks traced", st[2], "candidate pairs", st[7], "shadowed px", int(r.is_shadowed.sum()),
      "| dda tests", st[3], "cells", st[4], "rays", st[5])
