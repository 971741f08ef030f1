"""How long does the host take to ENQUEUE a frame (two-stream, builds that never wait)?  If that is about the frame
time with two frames in flight, the host is what bounds the throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY
rs = []
for i in range(2):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=flags, uniform_dims=(128, 128, 64))
        r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True, helper_thread=False)
        r._stream = st
        rs.append(r)
def step(k):
    r = rs[k % 2]
    with torch.cuda.stream(r._stream):
        r.display(setup, shadows=True, reflect=True)
for k in range(8):
    step(k)
torch.cuda.synchronize()
for n in (20, 40):
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d frames: enqueue %.3f ms/frame, done %.3f ms/frame" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
# the host alone: how fast can it enqueue when the GPU queue never fills?  (small image: the kernels are short)
import threading, queue
qs = [queue.Queue() for _ in rs]
done = queue.Queue()
def worker(i):
    torch.cuda.set_device(0)
    r = rs[i]
    while True:
        job = qs[i].get()
        if job is None:
            return
        with torch.cuda.stream(r._stream):
            r.display(setup, shadows=True, reflect=True)
        done.put(i)
ths = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(len(rs))]
for t in ths: t.start()
for n in (20, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        qs[k % 2].put(1)
    for k in range(n):
        done.get()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("threads: %d frames: enqueue %.3f ms/frame, done %.3f ms/frame" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
for q in qs: q.put(None)
