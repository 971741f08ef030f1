#!/usr/bin/env python3
"""Per-wave timeline of the exact shadow pass (k_trace_shadow) of the bench frame.

    make -C uniformgrid-raytracing_amd/csrc EXTRA=-DUGRT_SHADOW_TIMELINE -B     # an instrumented build
    python tools/shadow_timeline.py [--opt shadow_sieve=0 ...] > gpurun_out/shadow_timeline.txt
    make -C uniformgrid-raytracing_amd/csrc -B                                  # back to the product

The instrumented kernel leaves, per wave, its start and end (s_memrealtime, 10 ns) and the number of items it worked
on; this script renders two frames, takes the stamps of the third and prints when the waves started, how many were
resident over the launch, how long the waves with and without work lived and where the launch's tail comes from
(profiles/r04_shadow_exact_timeline.txt is its output, before and after the sieve waves)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import ugrt  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--file", default="gpurun_out/shadow_timeline.bin")
args = ap.parse_args()
s = bench.load_scene(ugrt, "crash", 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for kv in args.opt:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
os.makedirs(os.path.dirname(args.file) or ".", exist_ok=True)
for i in range(3):
    if i == 2:
        os.environ["UGRT_SHADOW_TIMELINE_FILE"] = args.file
    r.display(setup, shadows=True, reflect=False)
    ctx.synchronize()
if not os.path.exists(args.file):
    sys.exit("no stamps: libugrt.so was not built with -DUGRT_SHADOW_TIMELINE")
raw = np.fromfile(args.file, dtype=np.uint64)
nw, w0, nsieve, rl = (int(x) for x in raw[:4])
a = raw[4:].reshape(-1, 2)
t0 = (a[:, 0] & np.uint64((1 << 56) - 1)).astype(np.int64)
t1 = (a[:, 1] >> np.uint64(8)).astype(np.int64)
worked = (a[:, 1] & np.uint64(255)).astype(int)
base = t0.min()
t0 -= base
t1 -= base
life = (t1 - t0) / 100.0
b = np.arange(nw)
j = b >> 3
first = ((((j >> rl) << 3) + (b & 7)) << rl) + (j & ((1 << rl) - 1))  # the wave's index in the list order (XCD runs)
single = first < w0
sieve = (first >= w0) & (first < w0 + nsieve)
print("waves %d, launch %.1f us; single-item waves %d, sieve waves %d" % (nw, t1.max() / 100.0, single.sum(), sieve.sum()))
for name, m in (("single-item", single), ("sieve", sieve)):
    if not m.any():
        continue
    print("%s waves: %d, with work %d; life mean %.2f us, p99 %.1f, max %.1f; started %.0f-%.0f us"
          % (name, m.sum(), (worked[m] > 0).sum(), life[m].mean(), np.percentile(life[m], 99), life[m].max(),
             t0[m].min() / 100.0, t0[m].max() / 100.0))
    for k in range(0, 9):
        mk = m & (worked == k)
        if mk.any():
            print("   worked on %d item(s): %6d waves, life mean %.1f max %.1f us, the last ends at %.0f us"
                  % (k, mk.sum(), life[mk].mean(), life[mk].max(), t1[mk].max() / 100.0))
T = int(t1.max()) + 1
ev = np.zeros(T + 2, dtype=np.int64)
np.add.at(ev, t0, 1)
np.add.at(ev, t1 + 1, -1)
conc = np.cumsum(ev)[:T]
step = max(1, T // 20)
print("waves resident, per 5 %% of the launch: %s" % " ".join(str(int(conc[i:i + step].mean())) for i in range(0, T, step)))
print("wave-us in all: %.0f (single %.0f, sieve %.0f)" % (life.sum(), life[single].sum(), life[sieve].sum()))
late = t1 > 0.8 * t1.max()
print("waves that end in the last fifth of the launch: %d (single %d, sieve %d), lives of %.1f us on average, started at %.0f-%.0f us"
      % (late.sum(), (late & single).sum(), (late & sieve).sum(), life[late].mean(), t0[late].min() / 100.0, t0[late].max() / 100.0))
