timeout -k 10 300 python tools/host_rate.py 2>&1 | tail -5
# enqueue cost of a frame on a tiny image (GPU work negligible): pure host time
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 0.01, 0)
setup = ugrt.FrameSetup.from_scene(s)
flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY
ctx = ugrt.Context(64, 64, light_grid=(16, 16), flags=flags, uniform_dims=(8, 8, 8))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True, helper_thread=False)
for _ in range(10): r.display(setup, shadows=True, reflect=True)
r.synchronize(); torch.cuda.synchronize()
import cProfile, pstats
n=200
t0=time.perf_counter()
for _ in range(n): r.display(setup, shadows=True, reflect=True)
t1=time.perf_counter()
r.synchronize(); torch.cuda.synchronize()
print("tiny frame: host enqueue %.3f ms/frame" % ((t1-t0)/n*1e3))
pr=cProfile.Profile(); pr.enable()
for _ in range(100): r.display(setup, shadows=True, reflect=True)
pr.disable(); r.synchronize()
pstats.Stats(pr).sort_stats('cumtime').print_stats(14)
PY
