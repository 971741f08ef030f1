#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow + 1 reflection bounce).
One "step" = one frame of display() (main.cu:59-302) over the C-ABI:
perspective grid build -> primary rays -> light-space mapping -> spherical grid build ->
ray sort + chunking -> shadow rays -> secondary rays -> uniform grid build -> 3D-DDA ->
shading -> (N > 1) RCCL gather of the RGB bands to rank 0.  All three grids are REBUILT every
frame (as the reference does), scene and buffers are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
N = 1: BASELINE configs[2] (1M-triangle "crashing" stand-in, 1920x1080).  N > 1: one process
per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE), the image is cut into bands of tile rows
and grows with N at fixed 16:9 aspect (weak scaling; N = 4 is configs[3]'s 3840x2160); the bands
are timed and their boundaries moved before the measurement (--balance-rounds).
Consecutive frames go to --frames-in-flight renderers with their own contexts and streams (every
frame is complete: all grids rebuilt); the time of a frame that has the GPU to itself is reported
beside the rate as ms_per_step_one_frame_in_flight.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DDA_WAVES_THROUGHPUT, DDA_WAVES_ONE_FRAME = 3072, 1024
DDA_RPW_THROUGHPUT = 64
DDA_SPLIT_THROUGHPUT = 1  # split walks of the bounce (ugrt.h "dda_split"): on (the default) beside other frames too since round 4
SHADOW_WAVES_THROUGHPUT = 8192  # (the cull pass; round 2: 4096.  The exact pass runs one wave per work item since round 3)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_hash():
    """Hash of the device-code sources: ties a committed PMC profile (profiles/traffic.json) to the build it was taken on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "uniformgrid-raytracing_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, W, H, scale, kernel):
    """HBM bytes per launch of `kernel` and per frame from the committed rocprofv3 PMC passes, or (None, None, why).
    The profile must have been taken on THIS build of the kernels: a stale number is refused, loudly."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None, "profiles/traffic.json is missing"
    rec = json.load(open(tpath))
    key = "%s:%dx%d:scale%g" % (workload, W, H, scale)
    ent = rec.get(key)
    if ent is None:
        return None, None, "profiles/traffic.json has no entry %r (have %s)" % (key, sorted(rec))
    if ent.get("kernel_source_hash") != kernel_source_hash():
        return None, None, ("profiles/traffic.json entry %r was measured on kernel sources %s (%s), this build is %s: "
                            "re-run tools/profile_round.sh" % (key, ent.get("kernel_source_hash"), ent.get("profile"),
                                                               kernel_source_hash()))
    per = ent.get("per_launch_bytes", {}).get(kernel)
    if per is None:
        return None, ent, "profile %s has no kernel %s" % (ent.get("profile"), kernel)
    return per, ent, None


def cpp_driver_leg(ugrt, torch, s, W, H, image, frames=40, time_from=10):
    """The frame loop in C++ (integration/display_main: display() of main.cu over the shim classes, two streams, one
    frame at a time, no Python in the process), started as a child process on the scene files of this run: its own
    ms per frame, and its PPM against the timed renderers' image."""
    import subprocess

    import numpy as np

    exe = os.path.join(ROOT, "integration", "display_main")
    if not os.path.exists(exe) or "files" not in s:
        return None
    d = s["files"]["dir"]
    cam, lcam = s["cameras"]["ref"], s["light_camera"]
    flat = lambda c: " ".join("%.9g" % v for v in (list(c["eye"]) + list(c["look"]) + list(c["up"]) + [c["near"], c["far"]]))
    params, out = os.path.join(d, "params.txt"), os.path.join(d, "out.ppm")
    with open(params, "w") as f:
        f.write("obj %s\nmat %s\nsize %d %d\ncamera %s\nlight_camera %s\nshading_light %s\nstreams 2\nreflect 1\nframes %d\n"
                "time_from %d\nflags %d\n" % (s["files"]["obj"], s["files"]["mat"], W, H, flat(cam), flat(lcam),
                                             " ".join("%.9g" % v for v in s["shading_light"]), frames, time_from,
                                             ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY))
    t0 = time.time()
    p = subprocess.run([exe, params, out], cwd=d, capture_output=True, text=True, timeout=300)
    res = {"rc": p.returncode, "wall_s": round(time.time() - t0, 1), "streams": 2, "frames_in_flight": 1,
           "what": "integration/display_main (C++ over the C-ABI, no Python): %d frames, the last %d timed as a whole"
                   % (frames, frames - time_from)}
    if p.returncode != 0:
        res["error"] = (p.stdout + p.stderr)[-400:]
        return res
    for line in p.stdout.splitlines():
        w = line.split()
        if len(w) == 4 and w[0] == "timed_frames":
            res["ms_per_frame"] = float(w[3])
    tok = open(out).read().split()
    got = np.array(tok[4:], dtype=np.int64).astype(np.uint8)
    res["ppm_equals_the_timed_renderers_image"] = bool(np.array_equal(got, image.cpu().numpy()))
    return res


def load_scene(ugrt, workload, scale, rank, keep_files=False):
    """Generate the procedural scene, write it as .obj/.mtl/.mat and load it through the product's loader."""
    t0 = time.time()
    d = tempfile.mkdtemp(prefix="ugrt_bench_r%d_" % rank)
    gen = {"crash": ugrt.scenes.crash, "hall": ugrt.scenes.hall}[workload]
    info = gen(d, scale=scale)
    m = ugrt.Model()
    m.some_material(info["mat"])
    m.load_model(info["obj"])
    s = dict(info)
    s["verts"] = m.h_vertexlist.reshape(-1, 3)
    s["faces"] = m.h_facelist.reshape(-1, 3)
    s["matidx"] = m.h_materiallist_index
    s["mat_list"] = m.h_materiallist.reshape(-1, 6)
    s["reflect"] = m.h_reflectlist
    if keep_files:  # (the C++ driver's leg loads the same files; removed at exit)
        import atexit
        import shutil

        s["files"] = {"dir": d, "obj": info["obj"], "mtl": info["mtl"], "mat": info["mat"]}
        atexit.register(shutil.rmtree, d, True)
    else:
        for f in (info["obj"], info["mtl"], info["mat"]):
            os.unlink(f)
        os.rmdir(d)
    log("[bench] rank %d: scene %s: %d triangles, generated + written + parsed in %.1f s"
        % (rank, info["name"], s["num_faces"], time.time() - t0))
    return s


def algorithmic_bytes(torch, ugrt, ctx, r, dda_counts):
    """SURVEY.md section 8(d) formulas, per launch of each tracer, for this rank's band."""
    out = {}
    rows = ctx.rows[1] - ctx.rows[0]
    C_band = ctx.nbx * rows
    gi = ctx.grid_info(ugrt.GRID_PERSPECTIVE)
    out["trace_primary"] = 8 * C_band + 52 * gi.total_refs + 36 * ctx.npix
    # shadow: 24 N + sum over traced chunks (8 + 52 span(cell(chunk)))
    gctx = r.aux if r.aux is not None else ctx  # the light and uniform grids live in the second context when overlapped
    _, _, lspan, _, lgi = gctx.grid_arrays(ugrt.GRID_SPHERICAL)
    n, nch = ctx.npix, r.num_chunks
    heads = r.prefix[:nch].long()
    cells = r.d_map[n:2 * n][heads].long()
    C_l = lgi.num_cells
    sp = torch.where(cells < C_l, lspan[cells.clamp(max=C_l - 1)].long(), torch.zeros_like(cells))
    # SURVEY A11 prices the shadow STAGE at 24 N + sum_chunks(8 + 52 span): every 64-ray chunk re-stages its
    # cell's whole list.  Here the stage is two kernels with their own units: the cull pass tests
    # (triangle, beam) pairs (52 B reference each), the exact pass rebuilds the N rays and stages the
    # candidate references of each beam once per 64-ray sub-group.
    st = ctx.stats()
    out["stage_shadow_A11"] = 24 * n + 8 * nch + 52 * int(sp.sum().item())
    out["_units_shadow_cull_tests"] = int(st[6])      # (triangle, beam) interval tests of the cull pass
    out["_units_shadow_candidates_staged"] = int(st[7])  # candidate references staged by the exact pass
    tests, cells_visited, active = dda_counts
    out["trace_dda"] = 48 * active + 8 * cells_visited + 52 * tests
    out["_R_perspective"], out["_R_spherical"] = gi.total_refs, lgi.total_refs
    try:
        out["_R_uniform"] = gctx.grid_info(ugrt.GRID_UNIFORM).total_refs
    except Exception:
        out["_R_uniform"] = 0
    out["_shadow_span_sum"], out["_chunks"] = int(sp.sum().item()), nch
    return out


def cpu_baseline(ugrt, s, setup, W, H, lg, udims, seconds):
    """The oracle (CPU restatement, kind "port") on the host cores, twice as BASELINE.md asks: all cores and one
    thread.  Grid builds are timed in full, tracing on a band of tile rows grown to about `seconds` of wall time
    and scaled to the frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # the checker, used here only as the timed CPU baseline

    nby = H // 8
    fixed = ("build_spherical", "build_uniform")

    def run(rows):
        t0 = time.perf_counter()
        fr = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True, reflect=True, uniform_dims=udims)
        wall = time.perf_counter() - t0
        rays = 2 * fr["n"] + int(fr["active"].sum())
        tf = sum(fr["times"][k] for k in fixed)
        return wall, tf, rays

    def sample(threads, budget):
        cores = O.set_threads(threads)
        mid = nby // 2
        nrows, wall, tf, rays, rows = 2, 0.0, 0.0, 0, (mid, mid + 2)
        for _ in range(4):  # grow the band until the sample costs about `budget` seconds (or is the frame)
            lo = max(0, mid - nrows // 2)
            rows = (lo, min(nby, lo + nrows))
            wall, tf, rays = run(rows)
            if wall >= 0.6 * budget or rows[1] - rows[0] >= nby:
                break
            per_row = max((wall - tf) / float(rows[1] - rows[0]), 1e-4)
            nrows = int(max(nrows + 1, min(nby, (budget - tf) / per_row)))
        scale = nby / float(rows[1] - rows[0])
        t_full = tf + (wall - tf) * scale
        value = rays * scale / t_full / 1e6
        return dict(value=round(value, 4), unit="Mrays/s", cores=cores,
                    sample="oracle frame on tile rows [%d,%d) of %d (%.1f s wall; spherical+uniform grid builds "
                           "%.2f s counted once, the rest scaled x%.2f to the frame)"
                           % (rows[0], rows[1], nby, wall, tf, scale))

    allc = sample(os.cpu_count() or 1, seconds)
    one = sample(1, seconds)
    out = dict(allc, kind="port")
    out["one_thread"] = dict(value=one["value"], unit="Mrays/s", cores=1, sample=one["sample"])
    return out


def measure_config(ugrt, torch, s, W, H, local, reflect, animate, steps, warmup, frames_in_flight=4):
    """One of BASELINE's other configurations, measured the way the headline is (two streams per renderer,
    `frames_in_flight` renderers, builds that never wait, every frame complete) and verified the same way: the buffers
    every renderer holds from its last frame against ONE sequential waiting-build context."""
    device = torch.device("cuda", local)
    setup = ugrt.FrameSetup.from_scene(s)
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY
    lg, udims = (128, 128), (128, 128, 64)
    rs = []
    for i in range(frames_in_flight):
        stream = torch.cuda.Stream(device) if i else None
        with torch.cuda.stream(stream):
            cx = ugrt.Context(W, H, device=local, light_grid=lg, flags=flags, uniform_dims=udims)
            rr = ugrt.Renderer(cx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True,
                               helper_thread=False)
        rr._stream = stream
        for c in [rr.ctx] + ([rr.aux] if rr.aux is not None else []):
            c.set_option("async_build", 1)
        if rr.aux is not None and frames_in_flight > 1:
            rr.aux.set_option("dda_blocks", DDA_WAVES_THROUGHPUT)
            rr.aux.set_option("dda_rays_per_wave", DDA_RPW_THROUGHPUT)
            rr.aux.set_option("dda_split", DDA_SPLIT_THROUGHPUT)
            rr.ctx.set_option("shadow_waves", SHADOW_WAVES_THROUGHPUT)
        if animate:
            rr.init_orig_list(s["animated_size"], s["animated_offset"])
        rs.append(rr)
    turn, frame_no = [0], [0]

    def step():
        rr = rs[turn[0] % len(rs)]
        turn[0] += 1
        with torch.cuda.stream(rr._stream):
            if animate:
                rr.rotate_bunny(1.81 + 0.05 * frame_no[0])
                frame_no[0] += 1
            rr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)

    def drain():
        for rr in rs:
            rr.synchronize()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 2 * len(rs))):
        step()
    drain()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    ms = (time.perf_counter() - t0) / steps * 1e3
    npix = rs[0].ctx.npix
    active = [int(rr.active.sum().item()) if reflect else 0 for rr in rs]
    rays = 2 * npix + sum(active) / float(len(active))
    # verification: one sequential context (one stream, builds that wait), the geometry of every renderer's last frame
    fctx = ugrt.Context(W, H, device=local, light_grid=lg, flags=flags, uniform_dims=udims)
    fr = ugrt.Renderer(fctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    names = ["image", "is_shadowed", "intersect_id", "t", "normal", "dir"] + (["hit_t", "hit_id", "active"] if reflect else [])
    bad = []
    for rr in rs:
        fr.d_verts.copy_(rr.d_verts)
        fctx.geometry_changed()
        fr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
        fctx.synchronize()
        torch.cuda.synchronize()
        bad.append([k for k in names if not torch.equal(getattr(fr, k).view(torch.uint8), getattr(rr, k).view(torch.uint8))])
    verified = all(not b for b in bad)
    out = {"ms_per_step": round(ms, 4), "frames_per_s": round(1e3 / ms, 2), "Mrays/s": round(rays / ms / 1e3, 1),
           "rays_per_frame": int(rays), "triangles": int(s["num_faces"]), "size": [W, H], "steps": steps,
           "frames_in_flight": len(rs), "verified": verified}
    if not verified:
        out["verify_mismatches_per_renderer"] = bad
    for rr in rs:
        rr.close()
    del fr, fctx, rs
    gc.collect()
    torch.cuda.synchronize()
    return out


def launch_ranks(n):
    """`bench.py --gpus N` without an external launcher: start N ranks (one process per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  This parent never touches the GPU and does not replace
    itself: the ranks are child processes, and the exit code is theirs."""
    import socket
    import subprocess

    with socket.socket() as so:  # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without WORLD_SIZE: starting %d ranks: %s" % (n, n, " ".join(cmd)))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in p.stdout:  # the ranks' stderr is inherited; stdout carries rank 0's line (and nothing else of ours)
        if ln.startswith('{"metric"'):
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="crash", choices=["crash", "hall"])
    ap.add_argument("--scale", type=float, default=1.0, help="triangle-count scale of the scene (1.0 = BASELINE)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--stages-json", default="", help="also write the per-stage table to this file")
    ap.add_argument("--no-reflect", action="store_true", help="primary + shadow only (BASELINE configs[1])")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream: every stage after the other (default: light/uniform grid builds and the bounce "
                         "on a second stream beside the camera and shadow passes)")
    ap.add_argument("--repeats", type=int, default=4, help="extra untimed repetitions of the K steps (spread of the figure)")
    ap.add_argument("--stream-order", default="A",
                    help="order in which the renderers' streams are created (the runtime deals hardware queues in that order): "
                         "A main0 side0 main1 side1 ...; B the main streams, then the side streams starting at frame 1; "
                         "C mains, then sides from frame 0; D mains, then sides from frame F/2; or an explicit list "
                         "such as s0,m1,s1,m2,s2,m3,s3 (x = a stream nothing runs on)")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="independent frames in flight: F renderers (own contexts, buffers and streams) take the steps in "
                         "turn, so the GPU works on the tail of one frame and the head of the next (1 = a frame is "
                         "finished before the next one starts; that figure is reported too, as latency)")
    ap.add_argument("--waiting-builds", action="store_true",
                    help="grid builds and the shadow pass read their counts back as the reference does (default: option "
                         "async_build, no host wait inside a frame)")
    ap.add_argument("--config3", action="store_true",
                    help="BASELINE configs[3] as stated: ONE 3840x2160 frame cut into N bands (strong scaling)")
    ap.add_argument("--shard-builds", action="store_true",
                    help="the light grid and the uniform grid are built in N shards of the triangle list, exchanged "
                         "and merged (SURVEY 8f.1; one-stream frame)")
    ap.add_argument("--no-static-geometry", action="store_true",
                    help="rebuild the per-triangle records in every grid build, as a caller that rewrites the vertex array "
                         "behind the library's back must (default: UGRT_FLAG_STATIC_GEOMETRY, the renderer is the only writer)")
    ap.add_argument("--verify", dest="verify", action="store_true", default=True, help="(the default; kept for older command lines)")
    ap.add_argument("--no-verify", dest="verify", action="store_false",
                    help="skip the check behind the timed region (default: on).  N = 1: the buffers every renderer in flight "
                         "holds from its last frame (image, shadow flags, ids, t, normals, directions, bounce hits) are "
                         "compared bit for bit with ONE sequential context whose builds wait (one stream, one frame, no "
                         "estimates); N > 1: rank 0 renders the whole frame that way and compares the gathered image")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="launch-shape option for every context (ugrt_ctx_set_option), e.g. dda_kernel=1")
    ap.add_argument("--uniform-grid", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"),
                    help="cells of the bounce's uniform grid (A13 is this repo's own spec: any resolution gives the same hits)")
    ap.add_argument("--balance-rounds", type=int, default=4,
                    help="N > 1: rounds of timing the bands and moving their boundaries before the measurement (0 = equal bands)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false", default=True,
                    help="N = 1, default workload: skip the other BASELINE configurations behind the headline measurement "
                         "(hall 1024x1024 primary + shadow = configs[1], the 3840x2160 frame on this one GPU = configs[3]'s "
                         "per-GPU ceiling, the animated rebuild = configs[4]; ~10 verified steps each, reported under "
                         "other_configs; the profile passes skip them so that their kernel averages are the headline's)")
    ap.add_argument("--batch-builds", dest="batch_builds", action="store_true", default=False,
                    help="the light grid and the uniform grid are built as one batch whose sorts share their launches "
                         "(ugrt_grid_build_batch_begin / _end: three radix launches less per frame; measured 2 %% slower with "
                         "four frames in flight, profiles/r04_batched_builds.txt; default: one after the other)")
    ap.add_argument("--bands-in-frame", type=int, nargs="*", default=[],
                    help="N = 1: also time ONE frame at a time cut into this many bands of tile rows on streams of their own "
                         "(BandedRenderer; reported under one_frame_in_bands).  Default: none -- measured: 2 bands 1.45 against "
                         "1.46 ms, 3 and 4 bands 2.0 ms (the one host thread then enqueues 160-200 launches a frame at ~6 us "
                         "each: profiles/r04_banded_frame.txt)")
    ap.add_argument("--animate", action="store_true",
                    help="BASELINE configs[4]: transform the animated sub-range every frame (rot = 1.81 + 0.05*frame)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # before anything that could initialise the GPU in this process

    # The frames in flight use 2 streams each; the HIP runtime spreads streams over GPU_MAX_HW_QUEUES hardware queues
    # (default 4), so with four frames two streams share a queue: kernels of different streams in one queue still
    # overlap, but the chip works on fewer of them at once.  That is the fast arrangement here (1.28 ms; 3 queues
    # 1.46, 5-8 queues 1.55-1.60, i.e. every stream with a queue of its own is SLOWER), so it is pinned, not assumed.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    import numpy as np
    import torch

    import ugrt
    from ugrt import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a job of %d ranks (WORLD_SIZE)" % (args.gpus, world))
    dist = None
    # Rehearsal on a ONE-GPU box (UGRT_BENCH_REHEARSE=1): every rank uses device 0 and the collectives go
    # through gloo with host staging.  It exercises the multi-rank code path, not RCCL, and is never a result.
    rehearse = os.environ.get("UGRT_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local = 0
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU: libugrt has no CPU fallback"
    torch.cuda.set_device(local)

    if args.config3:
        args.width, args.height = 3840, 2160
    if args.width and args.height:
        W, H = args.width, args.height
    elif args.workload == "hall":
        W, H = parallel.weak_scaling_resolution(world, base=(1024, 1024))
    else:
        W, H = parallel.weak_scaling_resolution(world)
    nby = H // 8
    lg, udims = (128, 128), (128, 128, 64)
    if args.uniform_grid:
        udims = tuple(args.uniform_grid)

    s = load_scene(ugrt, args.workload, args.scale, rank, keep_files=(world == 1 and args.other_configs))
    setup = ugrt.FrameSetup.from_scene(s)
    # the renderer is the only writer of the vertex array (ugrt_animate): triangle records survive between builds
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | (0 if args.no_static_geometry else ugrt.FLAG_STATIC_GEOMETRY)
    opts = ([] if args.waiting_builds else ["async_build=1"]) + args.opt
    reflect = not args.no_reflect
    device = torch.device("cuda", local)
    shards = parallel.GridShards(dist, torch, device, rank, world, host_staging=rehearse) if args.shard_builds else None

    def make_renderers(rows):
        """One renderer per frame in flight for this rank's band `rows`, each with its own contexts and stream."""
        out = []
        F = max(1, args.frames_in_flight)
        # --stream-order other than A: the streams are made (and used once) in the given order before any renderer
        # exists; renderer 0's main stream is the default stream.  A: every renderer makes its streams as it is built
        # (the runtime then finds the earlier renderers' queues busy and gives a renderer's side stream a queue other
        # than its main stream's: one frame in flight 1.67 ms, against 2.28 ms when both sit on one queue)
        made = {}
        if args.stream_order != "A":
            if "," in args.stream_order:
                roles = [({"m": "main", "s": "side", "x": "idle"}[w[0]], int(w[1:]) if len(w) > 1 else n)
                         for n, w in enumerate(args.stream_order.split(","))]
            else:
                rot = {"B": 1, "C": 0, "D": F // 2}[args.stream_order]
                roles = [("main", i) for i in range(1, F)] + [("side", (i + rot) % F) for i in range(F)]
            for role in roles:
                made[role] = torch.cuda.Stream(device)
                with torch.cuda.stream(made[role]):
                    torch.zeros(1, device=device)
        for i in range(F):
            stream = (made[("main", i)] if made else torch.cuda.Stream(device)) if i else None
            with torch.cuda.stream(stream):
                cx = ugrt.Context(W, H, device=local, light_grid=lg, rows=rows, flags=flags, uniform_dims=udims)
                rr = ugrt.Renderer(cx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"],
                                   overlap=not args.no_overlap and shards is None, shards=shards,
                                   helper_thread=args.waiting_builds, aux_stream=made.get(("side", i)))
            rr.batch_builds = args.batch_builds
            if stream is not None:
                rr._stream = stream
            # the bounce's persistent waves: with several frames in flight every ray group gets a wave of its own (the
            # chip holds 3072 of them: 1.37 -> 1.33 ms); a renderer on its own caps them at 1024 so that a sort pass of
            # its main stream finds room (its default, which the one-frame latency below is measured with)
            # The shadow cull pass: 8192 persistent waves (its default) beside other frames too (round 2 ran both shadow
            # kernels on 4096 there; with the exact pass on one wave per item: 1.168 ms on 4096, 1.159 on 8192).
            if rr.aux is not None and args.frames_in_flight > 1:
                rr.aux.set_option("dda_blocks", DDA_WAVES_THROUGHPUT)
                rr.ctx.set_option("shadow_waves", SHADOW_WAVES_THROUGHPUT)
                # 64 rays per wave of the bounce: the kernel itself is ~3 % slower than at 32 (its default, which the
                # one-frame figure below is measured with), but it issues fewer instructions in all, and beside three
                # other frames that is what counts (1.214 -> 1.195 ms per frame)
                rr.aux.set_option("dda_rays_per_wave", DDA_RPW_THROUGHPUT)
                # split walks (the long ray groups of the last bounce cut into segments on different waves) shorten the
                # bounce by 5-15 % and one frame in flight by 2-3 %, but add a launch and the segments' extra windows.  Round 3
                # kept them off beside other frames (4 runs: 1.198 against 1.183 ms per frame); with round 4's frame the cost is
                # within the run-to-run spread (0.977-0.993 against 0.979-1.000 ms) and the bounce is 0.31 instead of 0.36 ms
                # there (0.44 instead of 0.38 of the 8 TB/s line): on, the library's default
                rr.aux.set_option("dda_split", DDA_SPLIT_THROUGHPUT)
            for kv in opts:
                k, v = kv.split("=")
                for c in [rr.ctx] + ([rr.aux] if rr.aux is not None else []):
                    c.set_option(k, int(v))
            if args.animate:
                rr.init_orig_list(s["animated_size"], s["animated_offset"])
            out.append(rr)
        return out

    def frames_ms(rs, frames):
        """ms per frame of `frames` pipelined frames on the renderers `rs` (no gather): this rank's band alone."""
        for rr in rs:
            rr.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            rr = rs[k % len(rs)]
            with torch.cuda.stream(getattr(rr, "_stream", None)):
                rr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
        for rr in rs:
            rr.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3

    # Bands: SURVEY.md 8(e) shards the image by tile rows and the slowest rank sets the frame time (8 equal bands of
    # the 5432x3056 frame take 1.09 ... 1.71 ms, tools/band_balance.py).  Before the measurement the ranks time their
    # bands, all-gather the times and move the boundaries (parallel.balanced_bounds: a pure function of the gathered
    # times, so every rank arrives at the same bands); at most --balance-rounds times.
    bounds = parallel.equal_bounds(world, nby)
    balance_log = []
    if world > 1 and args.balance_rounds > 0 and shards is None:
        best = None  # (slowest band's time, bounds) of the best partition MEASURED so far: timings are noisy
        for _ in range(args.balance_rounds):
            rs = make_renderers((bounds[rank], bounds[rank + 1]))
            frames_ms(rs, 6)
            mine = frames_ms(rs, 12)
            for rr in rs:
                rr.close()
            del rs
            gc.collect()  # the round's contexts and their device buffers go now, not at some later collection
            times = [None] * world
            dist.all_gather_object(times, float(mine))
            balance_log.append({"bounds": list(bounds), "band_ms": [round(t, 4) for t in times]})
            if best is None or max(times) < best[0]:
                best = (max(times), list(bounds))
            if max(times) < 1.03 * (sum(times) / world):
                break
            new_bounds = parallel.balanced_bounds(bounds, times)
            if new_bounds == bounds:
                break
            bounds = new_bounds
        bounds = best[1]  # (the boundaries derived from the last round's times were never timed themselves)
    rows = (bounds[rank], bounds[rank + 1])
    renderers = make_renderers(rows)
    r = renderers[0]
    ctx = r.ctx
    gather = parallel.BandGather(dist, torch, ctx.device, W, nby, rank, world, host_staging=rehearse, bounds=bounds)

    frame_no = [0]

    turn = [0]

    def step():
        rr = renderers[turn[0] % len(renderers)]
        turn[0] += 1
        if args.animate:  # Model::rotate_bunny(lightRotFactor), main.cu:68 + per_frame_funcs.h:15
            rr.rotate_bunny(1.81 + 0.05 * frame_no[0])
            frame_no[0] += 1
        if getattr(rr, "_stream", None) is not None:
            with torch.cuda.stream(rr._stream):  # the gather reads the band on the stream that rendered it
                rr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
                gather.gather(rr.image)
        else:
            rr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
            gather.gather(rr.image)

    # the tracers' event pairs are bracketed from the first warm-up step on; after the W warm-up steps every context
    # is run until its pool holds the pairs K timed steps take (hipEventCreate inside the timed region cost ~4 %)
    tracers = ("trace_primary", "shadow_cull", "trace_shadow", "trace_dda")
    profiled = [c for rr in renderers for c in [rr.ctx] + ([rr.aux] if rr.aux is not None else [])]
    for c in profiled:
        c.prof_enable(True, stages=tracers)
    warm = max(1, args.warmup)
    for _ in range(warm):
        step()
    if not os.environ.get("UGRT_BENCH_COLD_EVENTS"):
        for _ in range(max(0, args.steps - warm)):
            step()
            warm += 1
    gather.finish()
    for rr in renderers:
        rr.synchronize()
    torch.cuda.synchronize()

    # rays per frame of this rank: primary + shadow (one per pixel, misses included: misc_kernel.cu:255)
    # + secondary rays actually shot
    active = int(r.active[ctx.p0:ctx.p0 + ctx.npix].sum().item()) if reflect else 0
    rays_rank = 2 * ctx.npix + active

    # work counters for the DDA's algorithmic bytes: a counting context, outside the timed region
    cctx = ugrt.Context(W, H, device=local, light_grid=lg, rows=rows, flags=flags | ugrt.FLAG_COUNT_WORK,
                        uniform_dims=udims)
    cr = ugrt.Renderer(cctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    cr.d_verts.copy_(r.d_verts)  # same geometry as the timed renderer (matters with --animate)
    cctx.geometry_changed()
    cr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
    cctx.synchronize()
    st = cctx.stats()
    dda_counts = (st[3], st[4], st[5]) if reflect else (0, 0, 0)
    if reflect and not args.animate:
        assert st[5] == active, (st[5], active)
    del cr, cctx
    gc.collect()  # the counting context's buffers are freed HERE (hipFree synchronises the device), not by a collection inside the timed region
    torch.cuda.synchronize()
    abytes = algorithmic_bytes(torch, ugrt, ctx, r, dda_counts)

    # ---- timed region: exactly K steps between barrier + synchronize pairs -------------------
    # hipEvent pairs bracket the four tracer kernels inside the timed loop (the roofline figure needs the
    # dominant kernel's live launch time); every other stage is timed in a separate, untimed pass below,
    # because ~40 event pairs per frame would themselves cost about 7 % of the frame.
    # events are recorded on the stream a kernel runs on

    def merged_prof():
        out = {}
        for c in profiled:
            for k, v in c.prof_get().items():
                a = out.get(k, (0.0, 0))
                out[k] = (a[0] + v[0], a[1] + v[1])
        return out

    # (the counting pass above ran other work and left the chip idle while the host summed up: a few more untimed
    # frames bring the pipeline of frames in flight back to its steady state before the clock starts)
    for _ in range(2 * len(renderers)):
        step()
        warm += 1
    gather.finish()
    for c in profiled:
        c.prof_enable(True, stages=tracers)
        c.prof_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    radix0 = sum(c.get_state("radix_launches") for c in profiled)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gather.finish()  # the last frame's bands are in rank 0's image
    for rr in renderers:
        rr.synchronize()  # (raises if an asynchronous build or shadow pass overflowed its estimate: no silent loss)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    radix_per_frame = (sum(c.get_state("radix_launches") for c in profiled) - radix0) / float(args.steps)
    prof = merged_prof()
    # spread of the figure: the same K steps a few more times, outside the reported region
    repeats = []
    for _ in range(args.repeats):
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        gather.finish()
        for rr in renderers:
            rr.synchronize()
        torch.cuda.synchronize()
        repeats.append((time.perf_counter() - r0) / args.steps * 1e3)
    # The frame leaves processData's sort to whoever asks for its outputs (ugrt_sort_rays, deferred form with
    # SHADOW_ALL_CHUNKS).  Beside the figure: the same K steps with that sort carried out in every frame ("ray_sort" 1).
    ray_sort_kept = []
    for _ in range(min(args.repeats, 2)):
        for rr in renderers:
            rr.ctx.set_option("ray_sort", 1)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        gather.finish()
        for rr in renderers:
            rr.synchronize()
        torch.cuda.synchronize()
        ray_sort_kept.append((time.perf_counter() - r0) / args.steps * 1e3)
    for rr in renderers:
        rr.ctx.set_option("ray_sort", -1)
    split_stats = None
    # latency: the same K steps with ONE frame in flight (every frame is finished before the next one is started)
    latency_ms = None
    latency_repeats = []
    if len(renderers) > 1:
        user_waves = [kv for kv in opts if kv.startswith(("dda_blocks=", "shadow_waves=", "dda_rays_per_wave=", "dda_split="))]
        if renderers[0].aux is not None and not user_waves:
            renderers[0].aux.set_option("dda_blocks", DDA_WAVES_ONE_FRAME)
            renderers[0].aux.set_option("dda_rays_per_wave", -1)
            renderers[0].aux.set_option("dda_split", -1)
            renderers[0].ctx.set_option("shadow_waves", -1)  # the default
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            renderers[0].display(setup, frame_cnt=1, shadows=True, reflect=reflect)
            gather.gather(renderers[0].image)
        gather.finish()
        renderers[0].synchronize()
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - r0) / args.steps * 1e3
        for _ in range(min(args.repeats, 2)):  # (spread of the one-frame figure)
            r0 = time.perf_counter()
            for _ in range(args.steps):
                renderers[0].display(setup, frame_cnt=1, shadows=True, reflect=reflect)
                gather.gather(renderers[0].image)
            gather.finish()
            renderers[0].synchronize()
            torch.cuda.synchronize()
            latency_repeats.append((time.perf_counter() - r0) / args.steps * 1e3)
        # what the bounce's split walks did in the last of those frames
        if reflect:
            c0 = renderers[0].aux if renderers[0].aux is not None else renderers[0].ctx
            split_stats = c0.stats_dda_split()
        if renderers[0].aux is not None and not user_waves:
            renderers[0].aux.set_option("dda_blocks", DDA_WAVES_THROUGHPUT)
            renderers[0].aux.set_option("dda_rays_per_wave", DDA_RPW_THROUGHPUT)
            renderers[0].aux.set_option("dda_split", DDA_SPLIT_THROUGHPUT)
            renderers[0].ctx.set_option("shadow_waves", SHADOW_WAVES_THROUGHPUT)
    # untimed pass: the full stage table (with two streams the stages overlap: their sum exceeds the frame)
    for c in profiled:
        c.prof_enable(True)
        c.prof_reset()
    nfull = min(args.steps, 10)
    for _ in range(nfull):
        step()
    gather.finish()
    for rr in renderers:
        rr.synchronize()
    prof_full = merged_prof()
    for c in profiled:
        c.prof_enable(False)
    # the same stage table with every kernel alone on the chip: one context, one stream, one frame at a time
    # (what a stage costs, as opposed to how long it lasts beside the kernels of three other streams)
    alone = None
    if rank == 0 and not args.animate and shards is None:
        actx = ugrt.Context(W, H, device=local, light_grid=lg, rows=rows, flags=flags, uniform_dims=udims)
        ar = ugrt.Renderer(actx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
        for kv in opts:
            k, v = kv.split("=")
            if k != "dda_blocks":
                actx.set_option(k, int(v))
        for i in range(3 + nfull):
            if i == 3:
                actx.synchronize()
                actx.prof_enable(True)
                actx.prof_reset()
            ar.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
        actx.synchronize()
        alone = {k: round(v[0] / nfull, 4) for k, v in sorted(actx.prof_get().items()) if v[1]}
        del ar, actx

    verified, verify_detail = None, None
    if args.verify:
        # one more step, gathered synchronously; then the same frame(s) by ONE sequential context: one stream, one frame
        # at a time, builds that wait and size everything exactly (no estimates, no second stream)
        last = renderers[turn[0] % len(renderers)]
        step()
        gather.finish()
        for rr in renderers:
            rr.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if rank == 0:
            fctx = ugrt.Context(W, H, device=local, light_grid=lg, flags=flags, uniform_dims=udims)
            fr = ugrt.Renderer(fctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])

            def same(a, b):
                return bool(torch.equal(a.view(torch.uint8), b.view(torch.uint8)))

            verify_detail = []
            # N = 1: every renderer in flight holds the buffers of its last frame; N > 1: the gathered image
            if os.environ.get("UGRT_BENCH_FORCE_MISMATCH") == "1":  # (tests: a frame that differs must fail the run)
                last.image[0:3] = last.image[0:3] ^ 0xFF
            for rr in (renderers if world == 1 else [last]):
                fr.d_verts.copy_(rr.d_verts)
                fctx.geometry_changed()
                fr.display(setup, frame_cnt=1, shadows=True, reflect=reflect)
                fctx.synchronize()
                torch.cuda.synchronize()
                names = ["image"]
                if world == 1:
                    names += ["is_shadowed", "intersect_id", "t", "normal", "dir"] + (["hit_t", "hit_id", "active"] if reflect else [])
                bad = [k for k in names if not same(getattr(fr, k), getattr(rr, k))]
                verify_detail.append(bad)
            verified = all(not b for b in verify_detail)
            log("[bench] verify: %d renderer(s) against one sequential waiting-build context: %s"
                % (len(verify_detail), "all buffers equal" if verified else "DIFFERENT: %r" % (verify_detail,)))
            del fr, fctx
    tot = torch.tensor([elapsed, float(rays_rank)], dtype=torch.float64, device="cpu" if rehearse else ctx.device)
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
        rays_total = float(tot[1].item())
    else:
        rays_total = float(rays_rank)

    # a frame that differs from the sequential context's is not a result: every rank leaves with exit code 1 (rank 0
    # after printing its line, with "value" null and an "error" field)
    failed = False
    if args.verify:
        flag = torch.tensor([1.0 if (rank == 0 and verified is not True) else 0.0], dtype=torch.float64,
                            device="cpu" if rehearse else ctx.device)
        if dist is not None:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = bool(flag.item() > 0)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        if failed:
            sys.exit(1)
        return

    stages = {k: dict(ms_per_launch=v[0] / v[1], launches_per_step=v[1] / float(nfull),
                      ms_per_step=v[0] / nfull) for k, v in prof_full.items() if v[1]}
    for k in tracers:  # the tracers' numbers come from the timed loop itself
        if prof.get(k, (0, 0))[1]:
            v = prof[k]
            stages[k] = dict(ms_per_launch=v[0] / v[1], launches_per_step=v[1] / float(args.steps),
                             ms_per_step=v[0] / args.steps)
    gpu_ms = sum(v["ms_per_step"] for v in stages.values())
    # the roofline figure is quoted on the longest tracer whose algorithmic bytes are bytes the kernel's algorithm
    # really addresses (every staged reference of the primary pass, every cell and triangle a bounce ray meets); the
    # two shadow kernels only have a work count in units of the reference's algorithm (work_reduction below)
    dom = max(("trace_primary", "trace_dda") if reflect else ("trace_primary",),
              key=lambda k: stages.get(k, {}).get("ms_per_step", 0))
    dom_ms = stages[dom]["ms_per_launch"]
    achieved = abytes[dom] / (dom_ms * 1e-3) / 1e9
    kname = {"trace_primary": "k_trace_primary", "trace_shadow": "k_trace_shadow", "shadow_cull": "k_shadow_cull",
             "trace_dda": "k_trace_dda"}[dom]
    traffic, tent, twhy = pmc_traffic(args.workload, W, H, args.scale, kname) if world == 1 else (None, None, "N > 1")
    if twhy:
        log("[bench] roofline.traffic = null: " + twhy)
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    algorithmic_bytes_per_launch=int(abytes[dom]), ms_per_launch=round(dom_ms, 4),
                    traffic_source=(dict(profile=tent["profile"], kernel_source_hash=tent["kernel_source_hash"])
                                    if tent else None),
                    traffic_error=twhy,
                    note="achieved = algorithmic bytes of THIS kernel (SURVEY 8(d): 48 B per ray + 8 B per cell visited + "
                         "52 B per triangle tested, as the per-ray algorithm counts them) / its launch time from hipEvents "
                         "on its own stream inside the timed loop; traffic = HBM bytes per launch from rocprofv3 PMC "
                         "passes of the same build (2*FETCH_SIZE + WRITE_SIZE, KiB units).  The factor 2 on FETCH_SIZE is the "
                         "guide's correction for wide coalesced streaming reads; this kernel gathers (4-48 B per lane), for "
                         "which the factor is uncalibrated: read traffic as an UPPER bound (between FETCH_SIZE + WRITE_SIZE "
                         "and the figure given)",
                    traffic_is_upper_bound=True)
    # (the other tracer with byte-exact algorithmic bytes, for the record: which of the two is the longer one changes
    # with the workload and from round to round)
    others = {}
    for k in (("trace_primary", "trace_dda") if reflect else ("trace_primary",)):
        if k != dom and k in stages and k in abytes:
            ms = stages[k]["ms_per_launch"]
            others[{"trace_primary": "k_trace_primary", "trace_dda": "k_trace_dda"}[k]] = dict(
                achieved=round(abytes[k] / (ms * 1e-3) / 1e9, 2), frac=round(abytes[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                algorithmic_bytes_per_launch=int(abytes[k]), ms_per_launch=round(ms, 4))
    roofline["other_kernels"] = others
    # whole-frame HBM figure: all kernels' PMC bytes of one frame over the frame time
    frame_hbm = None
    if tent and tent.get("frame_bytes"):
        fb = float(tent["frame_bytes"])
        frame_hbm = dict(bytes=int(fb), GBps=round(fb / (elapsed / args.steps) / 1e9, 1),
                         frac=round(fb / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4), profile=tent["profile"])
    # NOT a roofline: how much less data the restructured shadow stage moves than the reference's algorithm would
    # (every 64-ray chunk re-stages its cell's whole list).  The "GB/s" of that formula exceed the HBM
    # peak because most of those bytes are never moved; reported as a work-reduction factor.
    a11_ms = sum(stages.get(k, {}).get("ms_per_step", 0.0) for k in ("shadow_prep", "shadow_cull", "trace_shadow"))
    work_reduction = None
    if a11_ms > 0:
        ref_bytes = abytes["stage_shadow_A11"]
        moved = None
        if tent:
            pl = tent.get("per_frame_bytes", {})
            mv = [pl.get(k) for k in ("k_shadow_keys", "k_shadow_boxes", "k_shadow_cull", "k_trace_shadow")]
            moved = int(sum(x for x in mv if x)) if any(mv) else None
        work_reduction = dict(stage="shadow (SURVEY A11: regroup + cull + exact kernels)",
                              reference_algorithm_bytes=int(ref_bytes), ms=round(a11_ms, 4),
                              pmc_bytes_of_the_stage_kernels=moved,
                              bytes_not_moved_factor=(round(ref_bytes / float(moved), 1) if moved else None),
                              note="reference_algorithm_bytes / ms = %.0f GB/s would be %.2f x the HBM peak: it is a count of "
                                   "work the restructured stage avoids, not a bandwidth"
                                   % (ref_bytes / (a11_ms * 1e-3) / 1e9, ref_bytes / (a11_ms * 1e-3) / 1e9 / HBM_PEAK_GBS))

    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(ugrt, s, setup, W, H, lg, udims, args.cpu_seconds)
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            cpu = dict(value=None, unit="Mrays/s", cores=0, kind="port", sample="failed: %r" % (e,))

    # one frame at a time, cut into bands of tile rows on streams of their own (renderer.BandedRenderer): the light grid,
    # the uniform grid and the bounce whole on one side context.  Verified against the renderers' frame.
    banded = None
    if world == 1 and args.bands_in_frame and reflect and not args.animate and shards is None and not args.no_overlap:
        banded = {}
        try:
            for nb in args.bands_in_frame:
                br = ugrt.BandedRenderer(ugrt.Context, W, H, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"],
                                         bands=nb, device=local, light_grid=lg, uniform_dims=udims, flags=flags)
                for _ in range(6):
                    br.display(setup, shadows=True, reflect=True)
                br.synchronize()
                torch.cuda.synchronize()
                times = []
                for _ in range(3):
                    r0 = time.perf_counter()
                    for _ in range(args.steps):
                        br.display(setup, shadows=True, reflect=True)
                    br.synchronize()
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - r0) / args.steps * 1e3)
                names = ["image", "is_shadowed", "intersect_id", "t", "normal", "dir", "hit_t", "hit_id", "active"]
                same = all(bool(torch.equal(getattr(br, k).view(torch.uint8), getattr(renderers[0], k).view(torch.uint8)))
                           for k in names) if not args.animate else None
                banded["bands_%d" % nb] = {"ms_per_step": round(times[0], 4), "repeat_ms_per_step": [round(x, 4) for x in times[1:]],
                                           "equal_to_the_timed_renderers_frame": same}
                if same is False:
                    failed = True
                del br
                gc.collect()
        except Exception as e:
            banded["error"] = repr(e)
            failed = True

    # the same frame from the C++ host (child process), one frame at a time
    cpp = None
    if world == 1 and args.other_configs and reflect and not args.animate and args.scale == 1.0 and not (args.width or args.height):
        try:
            cpp = cpp_driver_leg(ugrt, torch, s, W, H, renderers[0].image)
        except Exception as e:
            cpp = {"error": repr(e)}
        if cpp and cpp.get("ppm_equals_the_timed_renderers_image") is False:
            failed = True

    # the other BASELINE configurations, in this same process, behind the headline measurement (each verified)
    other = None
    default_run = (world == 1 and args.workload == "crash" and args.scale == 1.0 and not (args.width or args.height) and reflect
                   and not args.animate and not args.no_overlap and shards is None and not args.waiting_builds and not args.opt)
    if args.other_configs and default_run:
        for rr in renderers:
            rr.close()
        gc.collect()
        other = {}
        t_other = time.time()
        try:
            hall = load_scene(ugrt, "hall", 1.0, rank)
            other["hall_1024"] = dict(measure_config(ugrt, torch, hall, 1024, 1024, local, False, False, 10, 3),
                                      config="BASELINE configs[1] stand-in: procedural hall, primary + shadow (1 light, all chunks)")
            del hall
            other["crash_4k"] = dict(measure_config(ugrt, torch, s, 3840, 2160, local, True, False, 10, 3),
                                     config="BASELINE configs[3]'s frame (3840x2160, primary + shadow + bounce) on ONE GPU: "
                                            "the per-GPU ceiling of the 8-GPU configuration")
            other["animated"] = dict(measure_config(ugrt, torch, s, W, H, local, True, True, 10, 3),
                                     config="BASELINE configs[4]: animated sub-range transformed and all three grids rebuilt "
                                            "every frame, 1920x1080, primary + shadow + bounce")
        except Exception as e:  # the headline stands; what failed is said
            other["error"] = repr(e)
        other["wall_s"] = round(time.time() - t_other, 1)
        bad_other = [k for k, v in other.items() if isinstance(v, dict) and v.get("verified") is False]
        if bad_other or "error" in other:
            failed = True
            verify_detail = (verify_detail or []) + [{"other_configs": bad_other or other.get("error")}]

    value = rays_total / elapsed * args.steps / 1e6
    line = {
        "metric": "Mrays/s (primary+shadow+1-bounce)",
        "value": round(value, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (world > 1 and args.width and args.height) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a result)",
        "config": {
            "workload": ("BASELINE configs[%d] stand-in: procedural '%s' scene, %d triangles, %dx%d, primary + "
                         "shadow (1 light, all chunks traced)%s; all three grids rebuilt every frame%s%s"
                         % (4 if args.animate else (2 if reflect else 1), s["name"], s["num_faces"], W, H,
                            " + 1 reflection bounce" if reflect else "",
                            "; animated sub-range transformed every frame" if args.animate else "",
                            "" if args.no_static_geometry else " (per-triangle records kept between builds: static geometry flag)")),
            "frames_per_s": round(args.steps / elapsed, 2),
            "rays_per_frame": int(rays_total),
            "tile": 8, "light_grid": list(lg), "uniform_grid": list(udims),
            "streams": (1 if (args.no_overlap or args.shard_builds) else 2) * len(renderers),
            "frames_in_flight": len(renderers),
            "bounce_waves": {"frames_in_flight": DDA_WAVES_THROUGHPUT if len(renderers) > 1 else DDA_WAVES_ONE_FRAME,
                             "one_frame_in_flight": DDA_WAVES_ONE_FRAME},
            "bounce_rays_per_wave": {"frames_in_flight": DDA_RPW_THROUGHPUT if len(renderers) > 1 else 32,
                                     "one_frame_in_flight": 32},
            "shadow_waves": {"frames_in_flight": SHADOW_WAVES_THROUGHPUT if len(renderers) > 1 else 8192,
                             "one_frame_in_flight": 8192},
            "bounce_split_walks": {"frames_in_flight": bool(DDA_SPLIT_THROUGHPUT) if len(renderers) > 1 else True,
                                   "one_frame_in_flight": True, "last_one_frame_in_flight": split_stats},
            "host_waits_inside_a_frame": bool(args.waiting_builds),
            # processData's sort of the ray map by light cell: made when its outputs are asked for (every chunk is traced and
            # the shadow tracer orders the rays itself); ms_per_step_with_the_ray_sort_in_every_frame is the other way
            "ray_sort": "on demand" if not any(kv.startswith("ray_sort=") for kv in opts) else "as set by --opt",
            "static_geometry": not args.no_static_geometry,  # UGRT_FLAG_STATIC_GEOMETRY: triangle records kept between builds
            "ms_per_step_one_frame_in_flight": round(latency_ms, 4) if latency_ms else None,
            "n_gt_1_default": "weak scaling: the image grows with N at 16:9 (N = 4 is configs[3]'s 3840x2160); "
                              "--config3 = ONE 3840x2160 frame cut into N bands",
            "grid_builds": "light + uniform grid in %d shards of the triangle list, all-gathered and merged" % world
                           if args.shard_builds else "replicated per rank",
            "parallelism": "image bands of tile rows, 1 process per GPU, RCCL gather of RGB" if world > 1 else "1 GPU",
        },
        "roofline": roofline,
        "frame_hbm": frame_hbm,
        "work_reduction": work_reduction,
        "cpu_baseline": cpu,
        "verified_against_single_context_frame": verified,
        "verify_mismatches_per_renderer": verify_detail,
        "repeat_ms_per_step": [round(x, 4) for x in repeats],
        "ms_per_step_with_the_ray_sort_in_every_frame": [round(x, 4) for x in ray_sort_kept],
        "repeat_ms_per_step_one_frame_in_flight": [round(x, 4) for x in latency_repeats],
        "ms_per_step_one_frame_in_flight": round(latency_ms, 4) if latency_ms else None,
        "gpu_ms_per_step_in_kernels": round(gpu_ms, 4),
        "radix_launches_per_step": round(radix_per_frame, 2),  # histogram + pass kernels of the built-in sort (ugrt_ctx_get_state)
        "sort_rank_atomic": ctx.get_state("sort_rank_atomic"),
        "stages_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(stages.items())},
        "stages_ms_per_step_alone_on_one_stream": alone,
        "warmup_steps_run": warm,
        "band_bounds_tile_rows": list(bounds) if world > 1 else None,
        "band_balance_rounds": balance_log if world > 1 else None,
        "algorithmic_bytes": {k: int(v) for k, v in abytes.items()},
        "other_configs": other,
        "one_frame_in_bands": banded,
        "cpp_driver": cpp,
    }
    if failed:
        line["error"] = ("verification failed: the timed frames differ from the sequential waiting-build context in %r; "
                         "the measured %.3f Mrays/s are withheld" % (verify_detail, value))
        line["value"] = None
    if args.stages_json:
        with open(args.stages_json, "w") as fp:
            json.dump(dict(stages=stages, abytes=abytes, line=line), fp, indent=1)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
