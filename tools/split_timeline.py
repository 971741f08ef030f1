"""(needs the throwaway instrumentation described in dda_timeline.py, in its per-iteration form)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128, 128, 64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(2):
    r.display(setup, reflect=True)
ctx.synchronize()
import ugrt.device as dev
lib = dev.lib
for split in (0, 1):
    ctx.set_option("dda_rays_per_wave", 32); ctx.set_option("dda_blocks", 3072); ctx.set_option("dda_split", split)
    uv, us, uo, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
    buf = np.zeros(32768 * 4, dtype=np.uint64)
    for _ in range(4):
        ctx.trace_dda(uv, us, uo, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
        ctx.synchronize()
        lib.ugrt_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    ctx.trace_dda(uv, us, uo, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
    ctx.synchronize()
    lib.ugrt_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    b = buf.reshape(-1, 4)
    b = b[b[:, 0] != 0]; b = b[b[:, 0].astype(np.int64) > b[:, 0].astype(np.int64).max() - 100000]; n = len(b)
    st = b[:, 0].astype(np.int64); en = b[:, 1].astype(np.int64); t0 = st.min(); st -= t0; en -= t0; du = en - st
    jobs = (b[:, 2] >> np.uint64(32)).astype(np.int64); meta = (b[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    grp = meta & 0xFFFFFF; seg = (meta >> 24) & 15; nseg = meta >> 28
    x = b[:, 3]; isredo = (x >> np.uint64(63)).astype(np.int64); nredo = ((x >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64); nbeh = ((x >> np.uint64(56)) & np.uint64(0x7F)).astype(np.int64)
    wbeg = ((x >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64); wend = ((x >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64); widx = (x & np.uint64(0xFFFF)).astype(np.int64)
    print("split %d: iterations %d, span %d ticks; cut-group segments %d (groups %d), redo iterations %d (rays to redo: %d not covered, %d behind); duration mean %d p99 %d max %d; jobs total %d"
          % (split, n, en.max(), int((nseg > 1).sum()), len(set(grp[nseg > 1])), int(isredo.sum()), int(nredo.sum()), int(nbeh.sum()), du.mean(), np.percentile(du, 99), du.max(), jobs.sum()))
    order = np.argsort(-en)[:14]
    print("   last to end (grp, seg/nseg, wbeg-wend, windows run, start, duration, jobs, redo):")
    for g in order:
        print("     ", int(grp[g]), "%d/%d" % (seg[g], nseg[g]), "%d-%d" % (wbeg[g], wend[g]), int(widx[g]), int(st[g]), int(du[g]), int(jobs[g]), int(isredo[g]))
    for f in (0.25, 0.5, 0.7, 0.8, 0.9):
        t = f * en.max()
        print("   at %.2f of the span: %d running, %d not started" % (f, int(((st <= t) & (en > t)).sum()), int((st > t).sum())))
