"""Host cost of enqueuing one frame (tiny scene: the GPU never backs up, so this is Python + HIP launch overhead)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt
s = ugrt.scenes.crash(scale=0.01)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(256, 144, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY, uniform_dims=(32, 32, 16))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True, helper_thread=False)
for c in (ctx, r.aux):
    if c is not None:
        c.set_option("async_build", 1)  # (nothing in a frame waits for the device: what is timed is the enqueueing)
for _ in range(5):
    r.display(setup, shadows=True, reflect=True)
r.synchronize(); torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    r.display(setup, shadows=True, reflect=True)
t1 = time.perf_counter()
r.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/frame (done %.3f)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    r.display(setup, shadows=True, reflect=True)
pr.disable()
r.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
