"""Frame time with and without the second-stream grid builds (bench workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
for overlap in (False, True, False, True):
    ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=overlap)
    for _ in range(3):
        r.display(setup, reflect=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        r.display(setup, reflect=True)
    torch.cuda.synchronize()
    print("overlap=%s: %.3f ms/frame" % (overlap, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del r, ctx
