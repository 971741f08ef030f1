"""GPU parity: every stage of the hot path through the C-ABI (libugrt.so, HIP on gfx950)
against the CPU oracle on the same seeded inputs.  Integer / index outputs must be equal,
float outputs bit-equal (both sides are IEEE fp32 without FMA contraction), colours equal
as uint8 -- tighter than the 1e-4 the north star allows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what):
    a, b = bits(a), bits(b)
    bad = np.flatnonzero(a != b)
    assert bad.size == 0, "%s: %d of %d floats differ, first at %d" % (what, bad.size, a.size, bad[0])


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


SCENES = {}


def scene(ugrt, name):
    if name not in SCENES:
        SCENES[name] = {"cornell": lambda: ugrt.scenes.cornell(), "hall": lambda: ugrt.scenes.hall(scale=0.1),
                        "crash": lambda: ugrt.scenes.crash(scale=0.02)}[name]()
    return SCENES[name]


CASES = [
    # scene, camera, W, H, light grid
    ("cornell", "A", 256, 256, (128, 128)),
    ("cornell", "B", 256, 256, (128, 128)),
    ("hall", "ref", 256, 256, (64, 64)),
    ("hall", "ref", 320, 200, (128, 128)),
    ("crash", "ref", 256, 144, (128, 128)),
]


def setup_for(ugrt, s, cam):
    return ugrt.FrameSetup(s["cameras"][cam], s["light_camera"], s["shading_light"])


def make(ugrt, s, W, H, lg, rows=None, flags=0, udims=(32, 32, 16)):
    ctx = ugrt.Context(W, H, light_grid=lg, rows=rows, flags=flags, uniform_dims=udims)
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    return ctx, r


@pytest.mark.parametrize("name,cam,W,H,lg", CASES)
def test_perspective_grid_build(ugrt, O, torch, name, cam, W, H, lg):
    s = scene(ugrt, name)
    ctx, r = make(ugrt, s, W, H, lg)
    c = ugrt.renderer.make_camera(s["cameras"][cam], 45.0, r.aspect)
    ctx.upload_camera(c.camcoords)
    ctx.grid_build_perspective(r.d_faces, r.d_verts, r.F)
    ctx.synchronize()
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
    g = O.grid_perspective(O.cam_from(s["cameras"][cam], 45.0, r.aspect).cc, s["faces"], s["verts"], W // 8, H // 8)
    assert gi.total_refs == g["R"] and gi.num_cells == (W // 8) * (H // 8) and gi.cells_used == g["used"]
    np.testing.assert_array_equal(u32(key), g["keys"])
    np.testing.assert_array_equal(u32(value), g["vals"])
    np.testing.assert_array_equal(u32(span), g["span"])
    np.testing.assert_array_equal(u32(offset), g["offset"])


@pytest.mark.parametrize("name,cam,W,H,lg", CASES)
def test_primary_trace_on_oracle_grid(ugrt, O, torch, name, cam, W, H, lg):
    """HIP primary tracer consuming a CPU-built grid."""
    s = scene(ugrt, name)
    ctx, r = make(ugrt, s, W, H, lg)
    ocam = O.cam_from(s["cameras"][cam], 45.0, r.aspect)
    g = O.grid_perspective(ocam.cc, s["faces"], s["verts"], W // 8, H // 8)
    want = O.trace_primary(ocam, W, H, g, s["verts"], s["faces"])
    ctx.upload_camera(ocam.cc)
    dv = ctx.upload(g["vals"].view(np.int32) if g["R"] else np.zeros(1, np.int32))
    ds, do = ctx.upload(g["span"].view(np.int32)), ctx.upload(g["offset"].view(np.int32))
    ctx.trace_primary(dv, ds, do, r.normal, r.t, r.dir, r.is_shadowed, r.intersect_id, r.d_verts, r.d_faces)
    ctx.synchronize()
    np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["id"])
    assert_bits_equal(r.t.cpu().numpy(), want["t"], "t")
    assert_bits_equal(r.dir.cpu().numpy(), want["dir"], "dir")
    assert_bits_equal(r.normal.cpu().numpy(), want["normal"], "normal")
    assert int(r.is_shadowed.abs().sum()) == 0
    assert (want["id"] >= 0).sum() > 0


@pytest.mark.parametrize("name,cam,W,H,lg", CASES)
@pytest.mark.parametrize("all_chunks", [False, True])
def test_full_frame(ugrt, O, torch, name, cam, W, H, lg, all_chunks):
    """display() order end to end: grids, primary, mapping, ray sort, chunks, shadows, shading."""
    s = scene(ugrt, name)
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS if all_chunks else 0
    ctx, r = make(ugrt, s, W, H, lg, flags=flags)
    setup = setup_for(ugrt, s, cam)
    r.display(setup, frame_cnt=1, shadows=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, all_chunks=all_chunks)
    pr = want["primary"]
    np.testing.assert_array_equal(r.t.cpu().numpy().view(np.uint32), bits(pr["t"]))
    assert_bits_equal(r.dir.cpu().numpy(), pr["dir"], "dir")
    assert_bits_equal(r.normal.cpu().numpy(), pr["normal"], "normal")
    # light grid
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_SPHERICAL)
    lg_want = want["lgrid"]
    assert gi.total_refs == lg_want["R"]
    np.testing.assert_array_equal(u32(key), lg_want["keys"])
    np.testing.assert_array_equal(u32(value), lg_want["vals"])
    np.testing.assert_array_equal(u32(span), lg_want["span"])
    np.testing.assert_array_equal(u32(offset), lg_want["offset"])
    # sorted rays + chunk starts
    np.testing.assert_array_equal(u32(r.d_map), want["map"])
    assert r.num_chunks == want["nchunks"]
    np.testing.assert_array_equal(u32(r.prefix)[:r.num_chunks], want["prefix"][:want["nchunks"]])
    # shadows, material ids, image
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
    np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
    assert want["is_shadowed"].sum() > 0 and want["image"].max() > 0


def test_spotlight_shading_frame2(ugrt, O, torch):
    """frames >= 2 switch to spot_shade (main.cu:205-219)."""
    s = scene(ugrt, "hall")
    W, H, lg = 256, 256, (64, 64)
    ctx, r = make(ugrt, s, W, H, lg)
    setup = setup_for(ugrt, s, "ref")
    r.display(setup, frame_cnt=2, shadows=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, frame_cnt=2)
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
    # the angle dump of spot_shade (shader.h:108-112)
    ctx2, r2 = make(ugrt, s, W, H, lg)
    r2.display(setup, frame_cnt=1, shadows=True, shade=False)
    dump = ctx2.empty(2 * W * H, torch.float32)
    ctx2.shade_spotlight(r2.image, r2.normal, r2.t, r2.dir, r2.intersect_id, r2.cam_pos, r2.d_matidx, r2.d_matlist,
                         r2.num_materials, dump)
    ctx2.synchronize()
    img = np.zeros(3 * W * H, np.uint8)
    odump = np.zeros(2 * W * H, np.float32)
    ids = want["primary"]["id"].copy()
    O.shade(want["lcam"].cc, setup.shading_light, img, want["primary"]["normal"], want["primary"]["t"],
            want["primary"]["dir"], ids, want["cam"].worldori[:3], s["matidx"], s["mat_list"], 0, W * H, spot=True,
            dump=odump)
    assert_bits_equal(dump.cpu().numpy(), odump, "spot angle dump")
    np.testing.assert_array_equal(r2.image.cpu().numpy(), img)


def test_perlin_shade(ugrt, O, torch):
    s = scene(ugrt, "cornell")
    ctx, r = make(ugrt, s, 256, 256, (128, 128))
    setup = setup_for(ugrt, s, "B")
    r.display(setup, shadows=False, shade=False)
    ctx.shade_perlin(r.image, r.t, r.dir, r.cam_pos, r.intersect_id)
    ctx.synchronize()
    img = np.zeros(3 * 256 * 256, np.uint8)
    O.shade_perlin(img, r.intersect_id.cpu().numpy(), 256, 0, 256 * 256)
    np.testing.assert_array_equal(r.image.cpu().numpy(), img)
    assert img.max() > 0


@pytest.mark.parametrize("name,cam,W,H", [("hall", "ref", 256, 256), ("crash", "ref", 256, 144)])
def test_reflection_bounce(ugrt, O, torch, name, cam, W, H):
    """uniform grid build + secondary rays + 3D-DDA + blended shading (no reference code: own spec)."""
    s = scene(ugrt, name)
    lg, ud = (64, 64), (32, 32, 16)
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    setup = setup_for(ugrt, s, cam)
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud)
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_UNIFORM)
    ug = want["ugrid"]
    assert gi.total_refs == ug["R"]
    np.testing.assert_array_equal(u32(key), ug["keys"])
    np.testing.assert_array_equal(u32(value), ug["vals"])
    np.testing.assert_array_equal(u32(span), ug["span"])
    np.testing.assert_array_equal(u32(offset), ug["offset"])
    np.testing.assert_array_equal(r.active.cpu().numpy(), want["active"])
    assert_bits_equal(r.rays.cpu().numpy(), want["rays"], "secondary rays")
    np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want["hit_id"])
    assert_bits_equal(r.hit_t.cpu().numpy(), want["hit_t"], "dda t")
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
    assert want["active"].sum() > 100 and (want["hit_id"] >= 0).sum() > 100


@pytest.mark.parametrize("name,W,H,ud", [("crash", 256, 144, (32, 32, 16)), ("crash", 320, 200, (64, 64, 32)),
                                         ("hall", 256, 256, (16, 16, 8)), ("cornell", 128, 128, (8, 8, 8)),
                                         ("cornell", 128, 128, (33, 17, 5)), ("cornell", 96, 64, (3, 1000, 2)),
                                         ("cornell", 72, 40, (8, 8, 8))])  # (the last: a band that ends inside a 512-pixel span of the ray list)
def test_bounce_kernels_agree(ugrt, O, torch, name, W, H, ud):
    """The two bounce kernels (0 window kernel, 1 per-ray kernel of round 1) at every launch shape, with
    and without the (entry cell, octant) ray sort: hit ids equal, t bit-equal to the oracle; the counting variants
    report the oracle's work counts (tests, cells, rays of the algorithmic-byte formula)."""
    s = dict(scene(ugrt, name))
    if name == "cornell":  # every material reflects: rays in all directions through a coarse grid
        s["reflect"] = np.full(len(np.asarray(s["mat_list"]).reshape(-1, 6)), 0.5, np.float32)
    lg = (64, 64)
    cam = "ref" if name != "cornell" else "B"
    setup = setup_for(ugrt, s, cam)
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud, shadows=False)
    assert want["active"].sum() > 100
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    r.display(setup, shadows=False, reflect=True)
    ctx.synchronize()
    uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
    shapes = [dict(dda_kernel=0), dict(dda_kernel=0, dda_rays_per_wave=16), dict(dda_kernel=0, dda_rays_per_wave=32),
              dict(dda_kernel=0, dda_cull_min=1, dda_cull_work=1), dict(dda_kernel=0, dda_cull_min=1 << 30),
              dict(dda_kernel=0, dda_cull_work=64), dict(dda_kernel=0, dda_sort=1),
              dict(dda_kernel=0, dda_sort=1, dda_rays_per_wave=24), dict(dda_kernel=0, dda_blocks=7),
              dict(dda_kernel=1), dict(dda_kernel=1, dda_sort=1)]
    # split walks (the long groups of the launch before cut into segments; dda_split >= 2: every group): launches in a
    # row, so that each cuts by the history the one before it left
    shapes += [dict(dda_kernel=0, dda_split=1, dda_split_load=50)] * 3 + [dict(dda_kernel=0, dda_split=k) for k in (2, 3, 4, 4)]
    shapes += [dict(dda_kernel=0, dda_split=1, dda_split_load=50, dda_rays_per_wave=64)] * 3 + [dict(dda_kernel=0, dda_split=0)]
    shapes += [dict(dda_kernel=0, dda_split=4, dda_blocks=3)] * 2  # (a cut group's segments wait for each other on no wave)
    for opts in shapes:
        for k in ("dda_kernel", "dda_rays_per_wave", "dda_cull_min", "dda_cull_work", "dda_sort", "dda_blocks", "dda_split", "dda_split_load"):
            ctx.set_option(k, opts.get(k, -1))
        r.hit_t.fill_(7.0)
        r.hit_id.fill_(7)
        ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
        ctx.synchronize()
        np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want["hit_id"], err_msg=str(opts))
        assert_bits_equal(r.hit_t.cpu().numpy(), want["hit_t"], "dda t %r" % (opts,))
    # counting variants
    cctx, cr = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_COUNT_WORK, udims=ud)
    cr.display(setup, shadows=False, reflect=True)
    cctx.synchronize()
    cv, cs, co, _ = cctx.grid_ptrs(ugrt.GRID_UNIFORM)
    for k in (0, 1):
        cctx.set_option("dda_kernel", k)
        cr.hit_t.fill_(7.0)
        cctx.trace_dda(cv, cs, co, cr.d_verts, cr.d_faces, cr.rays, cr.active, cr.hit_t, cr.hit_id)
        st = cctx.stats()
        assert [int(st[3]), int(st[4]), int(st[5])] == want["dda_counters"], (k, st[3:6], want["dda_counters"])
        np.testing.assert_array_equal(cr.hit_id.cpu().numpy(), want["hit_id"])
        assert_bits_equal(cr.hit_t.cpu().numpy(), want["hit_t"], "counting kernel %d" % k)


@pytest.mark.parametrize("name,ud,seed", [("crash", (32, 32, 16), 1), ("crash", (7, 64, 3), 2), ("hall", (16, 16, 8), 3),
                                          ("cornell", (5, 5, 5), 4)])
def test_bounce_kernels_on_synthetic_rays(ugrt, O, torch, name, ud, seed):
    """The bounce kernels fed rays no camera pass produces: origins inside, on the boundary of and far outside the
    grid, directions with zero components (axis-parallel rays), negative zeros, rays along cell boundaries and
    diagonals, rays that miss the grid, all mixed within a wave.  Hit ids equal and t bit-equal to the CPU
    restatement for every kernel, and the counting variant's work counts too."""
    s = scene(ugrt, name)
    W, H, lg = 256, 128, (32, 32)
    N = W * H
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    setup = setup_for(ugrt, s, "ref" if name != "cornell" else "B")
    r.display(setup, shadows=False, reflect=True)  # builds the uniform grid (and allocates the ray buffers)
    ctx.synchronize()
    rng = np.random.default_rng(seed)
    v3 = np.asarray(s["verts"], np.float32).reshape(-1, 3)
    lo, hi = v3.min(0), v3.max(0)
    ext = hi - lo
    rays = np.zeros((N, 6), np.float32)
    kind = rng.integers(0, 8, N)
    o = lo + rng.random((N, 3)) * ext                                   # inside
    far = lo - ext * 2 + rng.random((N, 3)) * ext * 5                  # anywhere around
    o = np.where((kind == 1)[:, None], far, o)
    cs = ext / np.asarray(ud, np.float32)
    snapped = lo + np.round((o - lo) / cs) * cs                         # on cell boundaries
    o = np.where((kind == 2)[:, None], snapped, o)
    d = rng.normal(size=(N, 3))
    axis = rng.integers(0, 3, N)
    ax_d = np.zeros((N, 3)); ax_d[np.arange(N), axis] = rng.choice([-1.0, 1.0], N)
    d = np.where((kind == 3)[:, None], ax_d, d)                         # axis-parallel
    plane = d.copy(); plane[np.arange(N), axis] = 0.0
    d = np.where((kind == 4)[:, None], plane, d)                        # one zero component
    d = np.where((kind == 5)[:, None], np.sign(d) * np.array([1.0, 1.0, 1.0]), d)  # exact diagonals
    d = np.where((kind == 6)[:, None], d * 1e-3, d)                     # short direction vectors (not normalised)
    nz = np.linalg.norm(d, axis=1) == 0
    d[nz] = (1.0, 0.0, 0.0)
    rays[:, :3], rays[:, 3:] = o, d
    negz = (kind == 7)
    rays[negz, 3 + axis[negz]] = -0.0                                   # negative zero on one axis
    active = (rng.random(N) < 0.7).astype(np.int32)
    ug = O.grid_uniform(s["faces"], s["verts"], lo, hi, ud)
    want_t, want_id, want_cnt = O.trace_dda(ug, s["verts"], s["faces"], rays.reshape(-1), active, 0, N, N)
    r.rays.copy_(torch.from_numpy(rays.reshape(-1)).to(r.rays.device))
    r.active.copy_(torch.from_numpy(active).to(r.active.device))
    uvalue, uspan, uoffset, _ = ctx.grid_ptrs(ugrt.GRID_UNIFORM)
    for opts in (dict(dda_kernel=0), dict(dda_kernel=0, dda_rays_per_wave=64), dict(dda_kernel=0, dda_cull_work=1, dda_cull_min=1),
                 dict(dda_kernel=0, dda_sort=1), dict(dda_kernel=1)):
        for k in ("dda_kernel", "dda_rays_per_wave", "dda_cull_min", "dda_cull_work", "dda_sort"):
            ctx.set_option(k, opts.get(k, -1))
        r.hit_t.fill_(7.0)
        r.hit_id.fill_(7)
        ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
        ctx.synchronize()
        np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want_id, err_msg=str(opts))
        assert_bits_equal(r.hit_t.cpu().numpy(), want_t, "dda t %r" % (opts,))
    assert (want_id >= 0).sum() > 200 and (want_id[active == 1] < 0).sum() > 200
    # Split walks of the window kernel: the long ray groups of the LAST launch are cut into segments of windows that
    # run on different waves and are merged per ray.  The same bits from every launch: the first one (no history), the
    # ones that cut by the history (dda_split_load 50: every group above half the average), every group cut by force
    # into two, three, four segments, a launch whose history is of OTHER rays, at 16 / 32 / 64 rays per wave.
    for k in ("dda_kernel", "dda_rays_per_wave", "dda_cull_min", "dda_cull_work", "dda_sort"):
        ctx.set_option(k, -1)
    flipped = torch.from_numpy((1 - active).astype(np.int32)).to(r.active.device)
    mine = r.active.clone()
    for rpw in (32, 64, 16):
        ctx.set_option("dda_rays_per_wave", rpw)
        for split, load, other_rays_first in ((1, 50, False), (1, 50, False), (1, 50, False), (2, -1, False), (3, -1, False),
                                              (4, -1, False), (4, -1, False), (1, 400, False), (1, 50, True), (4, -1, True), (0, -1, False)):
            ctx.set_option("dda_split", split)
            ctx.set_option("dda_split_load", load)
            if other_rays_first:
                r.active.copy_(flipped)
                ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
                r.active.copy_(mine)
            r.hit_t.fill_(7.0)
            r.hit_id.fill_(7)
            ctx.trace_dda(uvalue, uspan, uoffset, r.d_verts, r.d_faces, r.rays, r.active, r.hit_t, r.hit_id)
            ctx.synchronize()
            what = "rays per wave %d, dda_split %d, load %d, history of other rays %s" % (rpw, split, load, other_rays_first)
            np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want_id, err_msg=what)
            assert_bits_equal(r.hit_t.cpu().numpy(), want_t, "dda t, " + what)
    ctx.set_option("dda_split", -1)
    ctx.set_option("dda_split_load", -1)
    ctx.set_option("dda_rays_per_wave", -1)
    cctx, cr = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_COUNT_WORK, udims=ud)
    cr.display(setup, shadows=False, reflect=True)
    cctx.synchronize()
    cr.rays.copy_(r.rays)
    cr.active.copy_(r.active)
    cv, csn, co, _ = cctx.grid_ptrs(ugrt.GRID_UNIFORM)
    for k in (0, 1):
        cctx.set_option("dda_kernel", k)
        cctx.trace_dda(cv, csn, co, cr.d_verts, cr.d_faces, cr.rays, cr.active, cr.hit_t, cr.hit_id)
        st = cctx.stats()
        assert [int(st[3]), int(st[4]), int(st[5])] == want_cnt, (k, st[3:6], want_cnt)


def test_animation(ugrt, O, torch):
    """copy_data_transform (transformation_kernel.cu:4) then a rebuilt frame."""
    s = scene(ugrt, "crash")
    W, H, lg = 256, 144, (64, 64)
    ctx, r = make(ugrt, s, W, H, lg)
    r.init_orig_list(s["animated_size"], s["animated_offset"])
    rot = 1.81 + 0.05 * 3
    r.rotate_bunny(rot)
    ctx.synchronize()
    import ctypes

    cr, sr = ctypes.c_float(), ctypes.c_float()
    ugrt.lib.ugrt_rot_cos_sin(rot, ctypes.byref(cr), ctypes.byref(sr))
    verts = np.ascontiguousarray(s["verts"], np.float32).reshape(-1).copy()
    off, size = s["animated_offset"], s["animated_size"]
    orig = verts[3 * off:3 * (off + size)].copy()
    O.animate(verts, orig, size, off, cr.value, sr.value)
    assert_bits_equal(r.d_verts.cpu().numpy(), verts, "animated vertices")
    setup = setup_for(ugrt, s, "ref")
    r.display(setup, shadows=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, verts=verts)
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])


@pytest.mark.parametrize("rpw", [32, 64])
def test_bounce_split_walks_on_a_moving_scene(ugrt, torch, rpw):
    """Split walks with a history that does NOT hold: the animated part of the scene turns every frame, rays end
    elsewhere than the frame before, and the rays a cut group's segments could not vouch for go on from where the
    segments stopped looking.  A context with split walks (cutting every group above half the average) against one
    without, frame by frame: the bounce's results and the image bit for bit; and the cut groups did have rays that
    went on (the path is exercised), in both tile-row bands of a two-band split too."""
    s = scene(ugrt, "crash")
    W, H, lg = 512, 288, (64, 64)
    setup = setup_for(ugrt, s, "ref")
    for rows in (None, (H // 8 // 2, H // 8)):
        pair = []
        for split in (0, 1):
            ctx, r = make(ugrt, s, W, H, lg, rows=rows, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
            r.init_orig_list(s["animated_size"], s["animated_offset"])
            ctx.set_option("dda_split", split)
            ctx.set_option("dda_split_load", 50)
            ctx.set_option("dda_rays_per_wave", rpw)
            pair.append((ctx, r))
        went_on = cut_frames = 0
        for frame in range(8):
            rot = 1.81 + 0.05 * frame
            for ctx, r in pair:
                r.rotate_bunny(rot)
                r.display(setup, shadows=False, reflect=True)
                ctx.synchronize()
            (c0, r0), (c1, r1) = pair
            what = "frame %d, rows %r" % (frame, rows)
            lo, hi = (0, W * H) if rows is None else (rows[0] * 8 * W, rows[1] * 8 * W)  # (a band context writes its band)
            np.testing.assert_array_equal(r1.hit_id.cpu().numpy()[lo:hi], r0.hit_id.cpu().numpy()[lo:hi], err_msg=what)
            assert_bits_equal(r1.hit_t.cpu().numpy()[lo:hi], r0.hit_t.cpu().numpy()[lo:hi], "bounce t, " + what)
            np.testing.assert_array_equal(r1.image.cpu().numpy()[3 * lo:3 * hi], r0.image.cpu().numpy()[3 * lo:3 * hi], err_msg=what)
            st = c1.stats_dda_split()
            went_on += st["rays_walked_again"]
            cut_frames += 1 if st["segments"] else 0
        # (after a launch most of whose cut groups had such rays the context cuts nothing for 1, 2, 4 ... launches)
        assert went_on > 0 and cut_frames >= 2, (went_on, cut_frames)


def test_bounce_split_history_survives_toggles(ugrt, O, torch):
    """The split walks' per-pixel history is neither cleared nor refreshed while the option is off, and it is laid out
    for one rays-per-wave setting: dda_split 1 -> 0 -> 1 with a moved camera in between, and dda_rays_per_wave changed
    between launches (32 -> 64 -> 16 -> 32) -- every frame against the oracle's bounce, bit for bit."""
    s = scene(ugrt, "crash")
    W, H, lg, ud = 384, 216, (64, 64), (32, 32, 16)
    cams = [setup_for(ugrt, s, "ref")]
    moved = dict(s["cameras"]["ref"])
    moved["eye"] = tuple(np.asarray(moved["eye"], np.float32) + np.float32([0.6, 0.25, -0.4]))
    cams.append(ugrt.FrameSetup(moved, s["light_camera"], s["shading_light"]))
    want = [O.frame(s, c, W, H, light_grid=lg, reflect=True, uniform_dims=ud, all_chunks=True) for c in cams]
    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    ctx.set_option("dda_split_load", 50)  # cut every group above half the average: the path is exercised
    # (split, rays per wave, camera)
    plan = [(1, 32, 0), (1, 32, 0), (0, 32, 1), (1, 32, 1), (1, 32, 0), (1, 64, 0), (1, 64, 1), (0, 16, 1), (1, 16, 0),
            (1, 32, 0), (2, 32, 1), (1, 64, 1)]
    cut = 0
    for k, (split, rpw, cam) in enumerate(plan):
        ctx.set_option("dda_split", split)
        ctx.set_option("dda_rays_per_wave", rpw)
        r.display(cams[cam], shadows=True, reflect=True)
        ctx.synchronize()
        what = "launch %d: split %d, %d rays per wave, camera %d" % (k, split, rpw, cam)
        np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want[cam]["hit_id"], err_msg=what)
        assert_bits_equal(r.hit_t.cpu().numpy(), want[cam]["hit_t"], "bounce t, " + what)
        np.testing.assert_array_equal(r.image.cpu().numpy(), want[cam]["image"], err_msg=what)
        cut += 1 if (split and ctx.stats_dda_split()["segments"]) else 0
    assert cut >= 1, cut  # (a changed camera or rays-per-wave setting leaves launches without a usable history)


LAUNCH_SHAPES = [("primary_centre", 0), ("primary_waves", 64), ("primary_waves", 4096), ("primary_xcd_run", 0), ("primary_xcd_run", 1),
                 ("primary_xcd_run", 7), ("primary_xcd_run", 4096), ("shadow_xcd_run", 0), ("shadow_xcd_run", 1),
                 ("shadow_xcd_run", 4096), ("shadow_waves", 64), ("primary_order", 0), ("primary_chunk", 4),
                 ("primary_chunk", 64), ("primary_seg", 64), ("shadow_beam", 64), ("shadow_beam", 8192), ("shadow_xseg", 64),
                 ("shadow_sizebits", 0), ("shadow_itemsort", 0), ("shadow_mbits", 9), ("shadow_sieve", 0), ("shadow_sieve", 2),
                 ("shadow_sieve", 5), ("shadow_sieve", 64)]


@pytest.mark.parametrize("name,W,H", [("crash", 384, 216), ("hall", 256, 256)])
def test_launch_shape_options_do_not_change_a_result(ugrt, O, torch, name, W, H):
    """include/ugrt.h says no launch-shape option changes a result.  Every one that selects another code path of the
    primary tracer or the shadow pass (persistent waves with XCD slices, XCD runs of any length, the persistent exact
    pass, sieve waves of any width in the exact pass, list-order flushes, the padding-entry skip behind the item capacity) runs here, alone and all at once, waiting
    and asynchronous builds: ids, t, shadow flags and the image equal the oracle's."""
    s = scene(ugrt, name)
    lg, ud = (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=False, all_chunks=True)

    def check(r, what):
        np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"], err_msg=what)
        assert_bits_equal(r.t.cpu().numpy(), want["primary"]["t"], "t, " + what)
        assert_bits_equal(r.normal.cpu().numpy(), want["primary"]["normal"].reshape(-1), "normal, " + what)
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"], err_msg=what)
        np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"], err_msg=what)

    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    for async_build in (0, 1):
        ctx.set_option("async_build", async_build)
        for key, value in LAUNCH_SHAPES:
            ctx.set_option(key, value)
            for _ in range(2 if async_build else 1):  # (the second asynchronous frame runs on the first one's estimates)
                r.display(setup, shadows=True)
                try:
                    ctx.synchronize()
                except ugrt.UgrtError as e:
                    # an option that makes more beams / pairs / items than the frame before outgrows the asynchronous
                    # pass's estimate: that is reported (UGRT_EOVERFLOW), and the frame is to be repeated
                    assert async_build and e.code == 6, e
                    r.display(setup, shadows=True)
                    ctx.synchronize()
                check(r, "%s=%d async_build=%d" % (key, value, async_build))
            ctx.set_option(key, -1)
    # several at once
    for key, value in (("primary_waves", 192), ("shadow_xcd_run", 0), ("shadow_waves", 128), ("primary_order", 0),
                       ("primary_chunk", 8), ("shadow_beam", 128)):
        ctx.set_option(key, value)
    for _ in range(2):
        r.display(setup, shadows=True)
        try:
            ctx.synchronize()
        except ugrt.UgrtError as e:
            assert e.code == 6, e
            r.display(setup, shadows=True)
            ctx.synchronize()
        check(r, "combined")


def test_band_split_equals_full_frame(ugrt, O, torch):
    """Image-tile sharding (multi-GPU path): two contexts with complementary tile-row bands
    produce exactly the full frame's primary outputs and the oracle's band results."""
    s = scene(ugrt, "hall")
    W, H, lg = 256, 256, (64, 64)
    setup = setup_for(ugrt, s, "ref")
    full_ctx, full = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
    full.display(setup)
    full_ctx.synchronize()
    nby = H // 8
    img = np.zeros(3 * W * H, np.uint8)
    for rows in ((0, 13), (13, nby)):
        ctx, r = make(ugrt, s, W, H, lg, rows=rows, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
        r.display(setup)
        ctx.synchronize()
        want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True)
        a, b = 3 * ctx.p0, 3 * (ctx.p0 + ctx.npix)
        np.testing.assert_array_equal(r.image.cpu().numpy()[a:b], want["image"][a:b])
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy()[ctx.p0:ctx.p0 + ctx.npix],
                                      want["is_shadowed"][ctx.p0:ctx.p0 + ctx.npix])
        _, _, _, _, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
        assert gi.total_refs == want["grid"]["R"]
        img[a:b] = r.image.cpu().numpy()[a:b]
    np.testing.assert_array_equal(img, full.image.cpu().numpy())


def test_split_cells_merge(ugrt, O, torch):
    """Cells with more than 256 triangles are split into segments merged by atomicMin; the
    result must equal the sequential scan, including ties (coplanar duplicates)."""
    rng = np.random.default_rng(7)
    n = 3000
    ctr = rng.uniform(-0.3, 0.3, size=(n, 1, 3)) + np.array([0, 0, -5.0])
    tri = ctr + rng.normal(scale=0.25, size=(n, 3, 3))
    verts = np.concatenate([tri.reshape(-1, 3), tri.reshape(-1, 3)[:300]]).astype(np.float32)  # 100 duplicates
    faces = np.arange(len(verts)).reshape(-1, 3).astype(np.int32)
    s = dict(verts=verts, faces=faces, matidx=np.zeros(len(faces), np.int32),
             mat_list=np.array([[0.2, 0.2, 0.2, 0.8, 0.8, 0.8]], np.float32), reflect=np.zeros(1, np.float32))
    cam = dict(eye=(0, 0, 0), look=(0, 0, -1), up=(0, 1, 0), near=0.1, far=100.0)
    W = H = 64
    ctx, r = make(ugrt, s, W, H, (16, 16))
    ocam = O.cam_from(cam, 45.0, 1.0)
    g = O.grid_perspective(ocam.cc, faces, verts, 8, 8)
    assert g["span"].max() > 600
    want = O.trace_primary(ocam, W, H, g, verts, faces)
    for _ in range(2):  # twice: the merge slots must be re-armed
        ctx.upload_camera(ocam.cc)
        ctx.grid_build_perspective(r.d_faces, r.d_verts, r.F)
        value, _, span, offset, _ = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
        ctx.trace_primary(value, span, offset, r.normal, r.t, r.dir, r.is_shadowed, r.intersect_id, r.d_verts,
                          r.d_faces)
        ctx.synchronize()
        np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["id"])
        assert_bits_equal(r.t.cpu().numpy(), want["t"], "t")
        assert_bits_equal(r.normal.cpu().numpy(), want["normal"], "normal")


def _with_wide_triangles(s, cam, n_extra, seed=3):
    """The scene plus (a) backdrop triangles larger than the view of `cam` (every screen cell) and (b) triangles between opposite corners of the scene box (every uniform cell),
    interleaved with the scene's own triangles so that wide ids fall between narrow ids."""
    rng = np.random.default_rng(seed)
    verts, faces = s["verts"].astype(np.float32), s["faces"].astype(np.int32)
    e, look, up = (np.array(cam[k], np.float64) for k in ("eye", "look", "up"))
    f = (look - e) / np.linalg.norm(look - e)
    rt = np.cross(f, up)
    rt /= np.linalg.norm(rt)
    u = np.cross(rt, f)
    ext = []
    for k in range(n_extra):
        j = rng.uniform(0.8, 1.2, 6)
        d = 30.0 + 3.0 * k / n_extra  # backdrops behind the scene, larger than the view
        ext.append(np.stack([e + d * f - 100 * j[0] * rt - 100 * j[1] * u,
                             e + d * f + 100 * j[2] * rt - 100 * j[3] * u,
                             e + d * f + 10 * j[4] * rt + 100 * j[5] * u]))
    both = np.concatenate([verts] + ext)
    lo, hi = both.min(0), both.max(0)
    for k in range(n_extra):
        c = lo + (hi - lo) * rng.uniform(0.3, 0.7, 3)
        ext.append(np.stack([lo, hi, c]))
    ext = np.concatenate(ext).astype(np.float32)
    v2 = np.concatenate([verts, ext])
    extra_faces = (len(verts) + np.arange(len(ext), dtype=np.int32)).reshape(-1, 3)
    pos = np.sort(rng.integers(0, len(faces) + 1, len(extra_faces)))
    f2 = np.insert(faces, pos, extra_faces, axis=0).astype(np.int32)
    m2 = np.insert(s["matidx"].astype(np.int32), pos, 0)
    out = dict(s)
    out.update(verts=v2, faces=f2, matidx=m2)
    return out


@pytest.mark.parametrize("n_extra", [1, 7, 200])
def test_wide_triangles_are_merged_not_sorted(ugrt, O, torch, n_extra):
    """Triangles whose range is the whole grid bypass fill + sort and are merged into every cell's run:
    the lists must equal the oracle's sort of ALL references, for the screen grid, a band of it and the
    uniform grid; the frame on top of them must not change either."""
    s = scene(ugrt, "cornell")
    cam = s["cameras"]["B"]
    s = _with_wide_triangles(s, cam, n_extra)
    W, H, lg, ud = 128, 96, (32, 32), (8, 8, 4)
    for rows in (None, (3, 9)):
        ctx, r = make(ugrt, s, W, H, lg, rows=rows, udims=ud)
        ocam = O.cam_from(cam, 45.0, r.aspect)
        g = O.grid_perspective(ocam.cc, s["faces"], s["verts"], W // 8, H // 8, rows=rows)
        ncell_active = (W // 8) * ((rows[1] - rows[0]) if rows else H // 8)
        assert (np.bincount(g["vals"]) == ncell_active).sum() >= n_extra
        for _ in range(2):
            ctx.upload_camera(ocam.cc)
            ctx.grid_build_perspective(r.d_faces, r.d_verts, r.F)
            ctx.synchronize()
            value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
            assert gi.total_refs == g["R"] and gi.cells_used == g["used"]
            np.testing.assert_array_equal(u32(key), g["keys"])
            np.testing.assert_array_equal(u32(value), g["vals"])
            np.testing.assert_array_equal(u32(span), g["span"])
            np.testing.assert_array_equal(u32(offset), g["offset"])
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    setup = ugrt.FrameSetup(cam, s["light_camera"], s["shading_light"])
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud)
    for which, og in ((ugrt.GRID_UNIFORM, want["ugrid"]), (ugrt.GRID_SPHERICAL, want["lgrid"])):
        value, key, span, offset, gi = ctx.grid_arrays(which)
        assert gi.total_refs == og["R"]
        np.testing.assert_array_equal(u32(key), og["keys"])
        np.testing.assert_array_equal(u32(value), og["vals"])
        np.testing.assert_array_equal(u32(span), og["span"])
    ncell = ud[0] * ud[1] * ud[2]
    assert (np.bincount(want["ugrid"]["vals"]) == ncell).sum() >= n_extra
    np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])


def test_many_wide_triangles_tiny_grid(ugrt, O, torch):
    """More than 4096 wide triangles (a 2-cell screen grid): the id list goes through the radix sort."""
    rng = np.random.default_rng(11)
    n = 6000
    tri = np.zeros((n, 3, 3))
    tri[:, 0] = np.c_[rng.uniform(-1.5, -0.2, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-6, -4, n)]
    tri[:, 1] = np.c_[rng.uniform(0.2, 1.5, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-6, -4, n)]
    tri[:, 2] = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-6, -4, n)]
    tri[::5, 1, 0] = -0.1  # every fifth stays in the left cell
    verts = tri.reshape(-1, 3).astype(np.float32)
    faces = np.arange(len(verts)).reshape(-1, 3).astype(np.int32)
    s = dict(verts=verts, faces=faces, matidx=np.zeros(len(faces), np.int32),
             mat_list=np.array([[0.2, 0.2, 0.2, 0.8, 0.8, 0.8]], np.float32), reflect=np.zeros(1, np.float32))
    cam = dict(eye=(0, 0, 0), look=(0, 0, -1), up=(0, 1, 0), near=0.1, far=100.0)
    W, H = 16, 8
    ctx, r = make(ugrt, s, W, H, (16, 16))
    ocam = O.cam_from(cam, 45.0, r.aspect)
    g = O.grid_perspective(ocam.cc, faces, verts, 2, 1)
    assert (np.bincount(g["vals"]) == 2).sum() > 4096 and (np.bincount(g["vals"]) == 1).sum() > 500
    ctx.upload_camera(ocam.cc)
    ctx.grid_build_perspective(r.d_faces, r.d_verts, r.F)
    ctx.synchronize()
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
    assert gi.total_refs == g["R"] and gi.cells_used == g["used"]
    np.testing.assert_array_equal(u32(key), g["keys"])
    np.testing.assert_array_equal(u32(value), g["vals"])
    np.testing.assert_array_equal(u32(span), g["span"])
    np.testing.assert_array_equal(u32(offset), g["offset"])


@pytest.mark.parametrize("overlap", [False, True])
def test_static_geometry_flag(ugrt, O, torch, overlap):
    """(overlap: the two-stream frame, whose second context must hear of every geometry change too.)
    FLAG_STATIC_GEOMETRY keeps the triangle records between builds; ugrt_animate and geometry_changed()
    invalidate them.  Frames must equal the oracle's before and after both kinds of change."""
    s = scene(ugrt, "cornell")
    W, H, lg = 128, 128, (32, 32)
    ctx = ugrt.Context(W, H, light_grid=lg, flags=ugrt.FLAG_STATIC_GEOMETRY)
    r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=overlap)
    setup = setup_for(ugrt, s, "B")

    def check(verts):
        for _ in range(2):  # the second frame runs on kept records
            r.display(setup, shadows=True)
            r.synchronize()
        want = O.frame(s, setup, W, H, light_grid=lg, verts=verts)
        np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
        assert_bits_equal(r.t.cpu().numpy(), want["primary"]["t"], "t")
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
        np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])

    check(s["verts"])
    # the caller rewrites the vertices itself
    v2 = (np.asarray(s["verts"], np.float32) * np.float32(0.75)).astype(np.float32)
    r.d_verts.copy_(torch.from_numpy(v2.reshape(-1)).to(r.d_verts.device))
    ctx.geometry_changed()
    if r.aux is not None:
        r.aux.geometry_changed()
    check(v2)
    # ugrt_animate rewrites a sub-range
    r.init_orig_list(8, 16)
    r.rotate_bunny(0.3)
    ctx.synchronize()
    v3 = r.d_verts.cpu().numpy().reshape(-1, 3).copy()
    assert np.abs(v3 - v2).max() > 1e-3
    check(v3)
    r.close()


@pytest.mark.parametrize("n,bits,kind", [
    (1, 8, "rand"), (63, 1, "rand"), (255, 7, "rand"), (4096, 8, "rand"), (4097, 9, "rand"), (12345, 15, "rand"),
    (100000, 16, "rand"), (1000003, 20, "rand"), (2073600, 32, "rand"), (3000000, 14, "runs"),
    (2073600, 15, "few"), (500000, 24, "const"), (777777, 17, "desc"),
])
def test_radix_sort_is_the_stable_sort(ugrt, torch, n, bits, kind):
    """The built-in radix sort against numpy's stable argsort on the sorted bits, and against rocPRIM's
    through the same entry point: ragged tiles, every pass count, skewed digit histograms (all rays in a
    few light cells), already sorted runs (the fill kernel's output), bits above key_bits ignored."""
    rng = np.random.default_rng(n + bits)
    if kind == "rand":
        keys = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    elif kind == "runs":
        keys = np.repeat(rng.integers(0, 2**bits, n // 50 + 1, dtype=np.uint64), 50)[:n].astype(np.uint32)
    elif kind == "few":
        keys = rng.choice(rng.integers(0, 2**bits, 97), n).astype(np.uint32)
    elif kind == "const":
        keys = np.full(n, 0xABCDEF, np.uint32)
    else:
        keys = np.arange(n, 0, -1, dtype=np.uint32)
    vals = rng.permutation(n).astype(np.uint32)
    ctx = ugrt.Context(64, 64)
    dk, dv = ctx.upload(keys.view(np.int32)), ctx.upload(vals.view(np.int32))
    mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    # items: pairs per thread of a pass; rank: 0 = ranks by ballots, -1 = by LDS atomics where the context's self-test allows
    for library, items, rank in ((False, -1, -1), (False, 8, -1), (False, 16, -1), (False, 8, 0), (False, 16, 0), (True, -1, -1)):
        ctx.set_option("sort_items", items)
        ctx.set_option("sort_rank", rank)
        for rep in range(2):  # twice: the sort's state (tickets, histogram rows) must be clean again after a sort
            ok, ov = torch.empty_like(dk), torch.empty_like(dv)
            ctx.sort_pairs(dk, ok, dv, ov, bits, library=library)
            ctx.synchronize()
            np.testing.assert_array_equal(u32(ok), keys[order])
            np.testing.assert_array_equal(u32(ov), vals[order])
            np.testing.assert_array_equal(u32(dk), keys)  # inputs untouched
        if not library:
            assert ctx.get_state("sort_rank_atomic") == (0 if rank == 0 else ctx.get_state("sort_rank_atomic"))
    # the self-test has run with the first sort; on gfx950 the LDS serves equal addresses in lane order
    ctx.set_option("sort_rank", -1)
    assert ctx.get_state("sort_rank_atomic") == 1, "k_rs_selftest: LDS atomics not in lane order on this device (ballot ranks are used)"
    assert ctx.get_state("radix_launches") > 0


def test_known_answers_on_gpu(ugrt, torch):
    """centre tile, one triangle at z=-5 -> 64/64 hits, t=5, n=(0,0,1), dir=(0,0,-1), id 0."""
    verts = np.array([[-50, -50, -5], [50, -50, -5], [0, 50, -5]], np.float32)
    faces = np.array([[0, 1, 2]], np.int32)
    s = dict(verts=verts, faces=faces, matidx=np.zeros(1, np.int32),
             mat_list=np.array([[0.2, 0.2, 0.2, 0.8, 0.8, 0.8]], np.float32), reflect=np.zeros(1, np.float32))
    cam = dict(eye=(0, 0, 0), look=(0, 0, -1), up=(0, 1, 0), near=0.1, far=100.0)
    W = H = 64
    ctx, r = make(ugrt, s, W, H, (16, 16))
    r.display(ugrt.FrameSetup(cam, cam, (0, 0, 0)), shadows=False)
    ctx.synchronize()
    ids = r.intersect_id.cpu().numpy().reshape(H, W)
    t = r.t.cpu().numpy().reshape(H, W)
    n = r.normal.cpu().numpy().reshape(H, W, 3)
    d = r.dir.cpu().numpy().reshape(H, W, 3)
    assert (ids == 0).all()  # material index after shading
    assert abs(t[32, 32] - 5.0) < 1e-6 and np.allclose(d[32, 32], (0, 0, -1)) and np.allclose(n[32, 32], (0, 0, 1))
    centre = t[28:36, 28:36]
    assert (centre > 0).all() and np.allclose(centre * -d[28:36, 28:36, 2], 5.0, atol=1e-5)


def _frame_equals_oracle(ugrt, O, s, setup, W, H, lg, ud, reflect=True):
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    for _ in range(2):
        r.display(setup, shadows=True, reflect=reflect)
        ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=reflect, uniform_dims=ud)
    pr = want["primary"]
    np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
    assert_bits_equal(r.t.cpu().numpy(), pr["t"], "t")
    assert_bits_equal(r.normal.cpu().numpy(), pr["normal"], "normal")
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
    if reflect:
        np.testing.assert_array_equal(r.active.cpu().numpy(), want["active"])
        np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want["hit_id"])
        assert_bits_equal(r.hit_t.cpu().numpy(), want["hit_t"], "dda t")
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
    return want


def test_all_rays_miss(ugrt, O, torch):
    """The camera is far from the scene and looks past it (the reference takes |t|, so looking AWAY is not enough:
    geometry behind the eye still hits, Q1): no hit, no shadow ray group with work, no secondary ray; every
    stage still runs (empty work lists, zero candidate pairs) and the frame is the oracle's."""
    s = scene(ugrt, "cornell")
    cam = dict(s["cameras"]["B"])
    cam["eye"], cam["look"] = (278, 5000, 278), (279, 5000, 278)  # far above the 556-unit box, looking level
    setup = ugrt.FrameSetup(cam, s["light_camera"], s["shading_light"])
    want = _frame_equals_oracle(ugrt, O, s, setup, 64, 64, (16, 16), (8, 8, 4))
    assert (want["mat_ids"] < 0).all() and want["image"].max() == 0


def test_degenerate_triangles(ugrt, O, torch):
    """Zero-area triangles (repeated vertices, collinear vertices), a triangle of one point, coincident
    duplicates and a sliver 1e-7 wide among the scene's own: the culls must stay conservative through the
    NaN / zero determinants they produce, and the frame must be the oracle's."""
    s0 = scene(ugrt, "cornell")
    verts = np.asarray(s0["verts"], np.float32).reshape(-1, 3)
    lo, hi = verts.min(0), verts.max(0)
    c = (lo + hi) / 2
    extra = np.array([
        c, c, c,                                             # a point
        c, c + (hi - lo) * 0.1, c,                           # repeated vertex
        lo, c, hi,                                           # collinear
        lo + (hi - lo) * [0.2, 0.2, 0.5], lo + (hi - lo) * [0.8, 0.2, 0.5], lo + (hi - lo) * [0.5, 0.2 + 1e-7, 0.5],  # sliver
        verts[0], verts[1], verts[2],                        # duplicate of the first triangle's corners
    ], np.float32)
    v2 = np.concatenate([verts, extra])
    f_extra = (len(verts) + np.arange(len(extra), dtype=np.int32)).reshape(-1, 3)
    s = dict(s0)
    s.update(verts=v2, faces=np.concatenate([np.asarray(s0["faces"], np.int32), f_extra]),
             matidx=np.concatenate([np.asarray(s0["matidx"], np.int32), np.zeros(len(f_extra), np.int32)]))
    setup = setup_for(ugrt, s0, "B")
    want = _frame_equals_oracle(ugrt, O, s, setup, 128, 128, (32, 32), (8, 8, 4))
    assert (want["mat_ids"] >= 0).sum() > 1000


def test_depth_layers_ties_and_grazing_triangles(ugrt, O, torch):
    """What the primary tracer's closest-hit merge and its depth bound must get right.  Camera B looks along +z
    into the box; in front of it stand twelve screen-filling sheets at different depths, listed in shuffled id
    order (a farther sheet is often tested before a nearer one, and whole sheets lie behind the closest hits of
    every ray of a tile: they are the ones the bound drops); every sheet exists twice with identical vertices,
    the copies also shuffled (equal t: the reference's strict "<" keeps the first of the cell list); triangles
    that the rays graze (their plane passes within 1e-3 of the eye, so det is tiny and t is ill-conditioned) lie
    before and behind the sheets; 2000 small random triangles fill the space between.  ids, t (bits), normals,
    shadow flags, the bounce and the image are the oracle's."""
    s0 = scene(ugrt, "cornell")
    rng = np.random.default_rng(20260412)
    eye = np.array([278.0, 273.0, -800.0])
    tris = []

    def sheet(z, wob):
        a = np.array([[-400, -400, z], [1000, -400, z + wob], [-400, 1000, z - wob]], np.float64)
        b = np.array([[1000, 1000, z], [-400, 1000, z - wob], [1000, -400, z + wob]], np.float64)
        return [a, b]

    for k in range(12):
        z = -500.0 + 37.0 * k
        for _ in range(2):  # the coincident copy
            tris += sheet(z, 3.0 * (k % 3))
    for k in range(40):  # grazing: two vertices on a ray from the eye, the third 1e-3 off it
        d = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0])
        d /= np.linalg.norm(d)
        t0, t1 = rng.uniform(50, 900), rng.uniform(50, 900)
        side = np.cross(d, [0.0, 1.0, 0.0])
        tris.append(np.array([eye + t0 * d, eye + t1 * d, eye + 0.5 * (t0 + t1) * d + 40.0 * side + 1e-3 * np.array([0, 1.0, 0])]))
    for k in range(2000):
        c = np.array([rng.uniform(0, 556), rng.uniform(0, 556), rng.uniform(-600, 500)])
        tris.append(c + rng.uniform(-12, 12, (3, 3)))
    order = rng.permutation(len(tris))
    extra = np.concatenate([tris[i] for i in order]).astype(np.float32)
    verts = np.asarray(s0["verts"], np.float32).reshape(-1, 3)
    v2 = np.concatenate([verts, extra])
    f_extra = (len(verts) + np.arange(len(extra), dtype=np.int32)).reshape(-1, 3)
    s = dict(s0)
    s.update(verts=v2, faces=np.concatenate([np.asarray(s0["faces"], np.int32), f_extra]),
             matidx=np.concatenate([np.asarray(s0["matidx"], np.int32),
                                    rng.integers(0, len(s0["mat_list"]), len(f_extra)).astype(np.int32)]))
    setup = setup_for(ugrt, s0, "B")
    want = _frame_equals_oracle(ugrt, O, s, setup, 192, 160, (32, 32), (16, 16, 8))
    ids = want["mat_ids"]
    assert (ids >= 0).sum() > 0.9 * ids.size
    # the work counters show that the bound was exercised: jobs were dropped before their exact tests
    cctx = ugrt.Context(192, 160, light_grid=(32, 32), flags=ugrt.FLAG_COUNT_WORK, uniform_dims=(16, 16, 8))
    cr = ugrt.Renderer(cctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
    cr.display(setup, shadows=True, reflect=False)
    cctx.synchronize()
    st = cctx.stats_primary()
    assert st["rounds"] > 0 and 4 * st["rounds"] < st["jobs"], st
    np.testing.assert_array_equal(cr.intersect_id.cpu().numpy(), ids)


@pytest.mark.parametrize("seed,ntri,W,H", [(1, 600, 128, 96), (2, 6000, 160, 120), (3, 40000, 192, 104), (4, 2500, 64, 64)])
def test_random_soups(ugrt, O, torch, seed, ntri, W, H):
    """Random triangle soups with sizes over three decades (a few that fill the view, many of a pixel or less,
    needles), seen from INSIDE the cloud (triangles on every side of the eye, also behind it: Q1), light inside as
    well: every cull (tile and quadrant boxes, depth bound, beam boxes, bundle boxes of the bounce) sees inputs no
    modelled scene has.  The frame is the oracle's, bit for bit."""
    s0 = scene(ugrt, "cornell")
    rng = np.random.default_rng(7000 + seed)
    size = np.exp(rng.uniform(np.log(0.5), np.log(400.0), ntri))
    size[rng.random(ntri) < 0.01] = 900.0
    c = rng.uniform(-100.0, 656.0, (ntri, 3))
    tri = c[:, None, :] + rng.normal(0.0, 1.0, (ntri, 3, 3)) * size[:, None, None] * 0.5
    needles = rng.random(ntri) < 0.05
    tri[needles, 2] = tri[needles, 0] + (tri[needles, 1] - tri[needles, 0]) * 0.5 + rng.normal(0, 1e-3, (int(needles.sum()), 3))
    verts = tri.reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * ntri, dtype=np.int32).reshape(-1, 3)
    s = dict(s0)
    s.update(verts=verts, faces=faces, matidx=rng.integers(0, len(s0["mat_list"]), ntri).astype(np.int32))
    eye = rng.uniform(150.0, 400.0, 3)
    cam = dict(eye=tuple(eye), look=tuple(eye + rng.normal(0, 1, 3)), up=(0, 1, 0), near=1.0, far=3000.0)
    light = dict(eye=tuple(rng.uniform(100.0, 450.0, 3)), look=tuple(rng.uniform(100.0, 450.0, 3)), up=(0, 0, 1), near=1.0,
                 far=3000.0)
    setup = ugrt.FrameSetup(cam, light, tuple(rng.uniform(100.0, 450.0, 3)))
    want = _frame_equals_oracle(ugrt, O, s, setup, W, H, (32, 32), (16, 16, 8))
    assert (want["mat_ids"] >= 0).sum() > 0.1 * W * H


def test_deferred_chunk_count_and_options(ugrt, O, torch):
    """ugrt_sort_rays without the read-back: the count fetched later equals the synchronous one, and the shadow
    tracer fed UGRT_CHUNKS_ON_DEVICE produces the same flags, in the reference's launch-capped mode and with
    every chunk traced; ugrt_ctx_set_option accepts its key and rejects others."""
    s = scene(ugrt, "hall")
    W, H, lg = 256, 256, (64, 64)
    setup = setup_for(ugrt, s, "ref")
    for flags in (0, ugrt.FLAG_SHADOW_ALL_CHUNKS):
        ctx, r = make(ugrt, s, W, H, lg, flags=flags)
        r.display(setup, shadows=True)  # the renderer defers the count
        ctx.synchronize()
        flags_deferred = r.is_shadowed.cpu().numpy().copy()
        n_deferred = r.num_chunks
        # the same stages with the count on the host
        lvalue, lspan, loffset, _ = ctx.grid_ptrs(ugrt.GRID_SPHERICAL)
        ctx.map_rays_to_light(r.t, r.dir, r.d_map, r.cam_pos, np.float32(np.pi), np.float32(np.pi))
        n_sync = ctx.sort_rays(r.d_map, r.prefix)
        assert n_sync == n_deferred == ctx.sort_rays_chunks() and 0 < n_sync < 0xFFFFFFFF
        r.is_shadowed.zero_()
        ctx.trace_shadow(lvalue, r.d_verts, r.d_faces, lspan, loffset, r.t, r.dir, r.is_shadowed, r.d_map, r.prefix,
                         r.cam_pos, n_sync)
        ctx.synchronize()
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), flags_deferred)
        assert flags_deferred.sum() > 0
    ctx.set_option("dda_rays_per_wave", 64)
    ctx.set_option("dda_rays_per_wave", 0)
    with pytest.raises(ugrt.UgrtError):
        ctx.set_option("dda_rays_per_wave", 65)
    with pytest.raises(ugrt.UgrtError):
        ctx.set_option("no_such_option", 1)


def test_error_paths(ugrt, torch):
    with pytest.raises(ugrt.UgrtError) as e:
        ugrt.Context(250, 256)
    assert e.value.code == ugrt.UGRT_EINVAL
    with pytest.raises(ugrt.UgrtError):
        ugrt.Context(256, 256, rows=(5, 5))
    with pytest.raises(ugrt.UgrtError):
        ugrt.Context(256, 256, light_grid=(127, 128))
    ctx = ugrt.Context(64, 64, light_grid=(16, 16))
    with pytest.raises(ugrt.UgrtError):
        ctx.grid_info(ugrt.GRID_UNIFORM)  # not built
    with pytest.raises(ugrt.UgrtError):
        ctx.grid_build_perspective(None, None, 3)


def test_full_size_bench_workload(ugrt, O, torch):
    """BASELINE configs[2] at full size (1M triangles, 1920x1080, primary + shadow + bounce) on the GPU;
    the oracle checks a band of tile rows bit for bit (per-ray results do not depend on the other rays when
    every shadow chunk is traced) and the whole frame through size-independent properties."""
    s = ugrt.scenes.crash(scale=1.0)
    W, H, lg, ud = 1920, 1080, (128, 128), (128, 128, 64)
    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    setup = setup_for(ugrt, s, "ref")
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    N = W * H
    # --- properties over the whole frame
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
    k = u32(key).astype(np.int64)
    assert (np.diff(k) >= 0).all() and gi.total_refs == len(k) > 10 ** 7
    sp, off, v = u32(span).astype(np.int64), u32(offset).astype(np.int64), u32(value).astype(np.int64)
    assert sp.sum() == gi.total_refs and (off == np.concatenate([[0], np.cumsum(sp)[:-1]])).all()
    same = k[1:] == k[:-1]
    assert (v[1:][same] > v[:-1][same]).all()  # stable: ascending triangle ids inside a cell
    np.testing.assert_array_equal(np.bincount(k, minlength=gi.num_cells), sp)
    dm = u32(r.d_map)
    assert (np.sort(dm[:N]) == np.arange(N)).all() and (np.diff(dm[N:].astype(np.int64)) >= 0).all()
    pf = u32(r.prefix)[:r.num_chunks].astype(np.int64)
    assert pf[0] == 0 and (np.diff(pf) > 0).all() and (np.diff(pf) <= 64).all()
    t = r.t.cpu().numpy()
    d = r.dir.cpu().numpy().reshape(-1, 3)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    hid, act = r.hit_id.cpu().numpy(), r.active.cpu().numpy()
    assert ((hid >= 0) <= (act == 1)).all() and act.sum() > 10 ** 5
    assert ((t > 0) | (t == -1)).all()
    # --- the oracle on a band of 4 tile rows through the middle of the image
    rows = (66, 70)
    want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True, reflect=True, uniform_dims=ud)
    a, b = want["p0"], want["p0"] + want["n"]
    pr = want["primary"]
    np.testing.assert_array_equal(t[a:b].view(np.uint32), pr["t"][a:b].view(np.uint32))
    np.testing.assert_array_equal(r.normal.cpu().numpy()[3 * a:3 * b].view(np.uint32), pr["normal"][3 * a:3 * b].view(np.uint32))
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy()[a:b], want["is_shadowed"][a:b])
    np.testing.assert_array_equal(act[a:b], want["active"][a:b])
    np.testing.assert_array_equal(hid[a:b], want["hit_id"][a:b])
    np.testing.assert_array_equal(r.hit_t.cpu().numpy()[a:b].view(np.uint32), want["hit_t"][a:b].view(np.uint32))
    np.testing.assert_array_equal(r.image.cpu().numpy()[3 * a:3 * b], want["image"][3 * a:3 * b])
    assert (pr["id"][a:b] >= 0).sum() > 1000 and want["is_shadowed"][a:b].sum() > 100


@pytest.mark.parametrize("name,W,H,lg,ud", [("crash", 384, 216, (64, 64), (32, 32, 16)), ("hall", 256, 256, (128, 128), (64, 64, 32))])
def test_batched_light_and_uniform_builds(ugrt, O, torch, name, W, H, lg, ud):
    """ugrt_grid_build_batch_begin / _end: the light grid's and the uniform grid's reference lists are sorted in shared
    launches (one histogram kernel, one kernel per pass level; the list with fewer passes drops out).  Frame 0 builds
    both in the waiting form (nothing is deferred), the later ones in the asynchronous form as a batch; after every
    frame both grids' arrays equal the CPU restatement's element for element, and so does the frame.  Then the same
    with the batch switched off, with ballot ranks and 4096-pair tiles, and with one build of the batch only."""
    s = scene(ugrt, name)
    setup = setup_for(ugrt, s, "ref")
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud, all_chunks=True)
    cx = ugrt.Context(W, H, light_grid=lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=ud)
    r = ugrt.Renderer(cx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True, helper_thread=False)
    for c in (r.ctx, r.aux):
        c.set_option("async_build", 1)
    before = r.aux.get_state("radix_launches")
    per_frame = []
    for k, (batch, rank, items) in enumerate([(True, -1, -1), (True, -1, -1), (True, -1, -1), (False, -1, -1), (True, 0, 8), (True, -1, 16)]):
        r.batch_builds = batch
        r.aux.set_option("sort_rank", rank)
        r.aux.set_option("sort_items", items)
        r.display(setup, shadows=True, reflect=True)
        r.synchronize()
        now = r.aux.get_state("radix_launches")
        per_frame.append(now - before)
        before = now
        what = "frame %d (batch %r)" % (k, batch)
        for which, g in ((ugrt.GRID_SPHERICAL, "lgrid"), (ugrt.GRID_UNIFORM, "ugrid")):
            value, key, span, offset, gi = r.aux.grid_arrays(which)
            np.testing.assert_array_equal(u32(key), want[g]["keys"], err_msg=what)
            np.testing.assert_array_equal(u32(value), want[g]["vals"], err_msg=what)
            np.testing.assert_array_equal(u32(span), want[g]["span"], err_msg=what)
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"], err_msg=what)
        np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want["hit_id"], err_msg=what)
        np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"], err_msg=what)
    # a batch is its passes once (the longer list's) and one histogram kernel; apart, the lists' passes add up and no
    # histogram kernel runs (a build's fill kernel counts the first digit of its keys itself)
    lb, ub = (bits_for_cells(lg[0] * lg[1]) + 7) // 8, (bits_for_cells(ud[0] * ud[1] * ud[2]) + 7) // 8
    assert per_frame[2] == max(lb, ub) + 1 and per_frame[3] == lb + ub, (per_frame, lb, ub)
    # one build inside a batch, and an empty batch
    r.aux.grid_build_batch_begin()
    r.aux.grid_build_uniform(r.d_faces, r.d_verts, r.F, r.bbmin, r.bbmax)
    r.aux.grid_build_batch_end()
    r.aux.grid_build_batch_begin()
    r.aux.grid_build_batch_end()
    r.aux.synchronize()
    value, key, span, offset, gi = r.aux.grid_arrays(ugrt.GRID_UNIFORM)
    np.testing.assert_array_equal(u32(key), want["ugrid"]["keys"])
    np.testing.assert_array_equal(u32(value), want["ugrid"]["vals"])
    with pytest.raises(ugrt.UgrtError):
        r.aux.grid_build_batch_end()  # no batch is open


@pytest.mark.parametrize("name,W,H,bands", [("crash", 384, 216, 2), ("crash", 384, 216, 3), ("hall", 256, 256, 4)])
def test_banded_frame_equals_the_single_context_frame(ugrt, O, torch, name, W, H, bands):
    """BandedRenderer: ONE frame at a time, cut into bands of tile rows on streams of their own, the light grid, the
    uniform grid and the bounce whole on one side context; every band writes its rows of the same arrays.  Several
    frames (the first in the waiting form, then the asynchronous one, no host wait inside a frame): all buffers equal
    the CPU restatement's full frame, bit for bit."""
    s = scene(ugrt, name)
    lg, ud = (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud, all_chunks=True)
    br = ugrt.BandedRenderer(ugrt.Context, W, H, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], bands=bands,
                             light_grid=lg, uniform_dims=ud, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
    for k in range(4):
        br.display(setup, shadows=True, reflect=True)
        if k != 2:  # (frames 2 and 3 are enqueued back to back)
            br.synchronize()
            torch.cuda.synchronize()
            what = "frame %d" % k
            np.testing.assert_array_equal(br.intersect_id.cpu().numpy(), want["mat_ids"], err_msg=what)
            assert_bits_equal(br.t.cpu().numpy(), want["primary"]["t"], "t, " + what)
            assert_bits_equal(br.normal.cpu().numpy(), want["primary"]["normal"].reshape(-1), "normal, " + what)
            np.testing.assert_array_equal(br.is_shadowed.cpu().numpy(), want["is_shadowed"], err_msg=what)
            np.testing.assert_array_equal(br.active.cpu().numpy(), want["active"], err_msg=what)
            np.testing.assert_array_equal(br.hit_id.cpu().numpy(), want["hit_id"], err_msg=what)
            assert_bits_equal(br.hit_t.cpu().numpy(), want["hit_t"], "bounce t, " + what)
            np.testing.assert_array_equal(br.image.cpu().numpy(), want["image"], err_msg=what)


def bits_for_cells(c):
    b = 1
    while (1 << b) < c:
        b += 1
    return b


def test_bench_setting_four_renderers_in_flight(ugrt, O, torch):
    """The setting bench.py times by default, built here piece by piece: 1 M triangles at 1920x1080, FOUR renderers
    (each two contexts on two streams fed by this one host thread: overlap=True, helper_thread=False), builds and
    shadow pass that never wait (async_build), the bounce on 3072 persistent waves of 64 rays, the shadow kernels on 4096,
    UGRT_FLAG_STATIC_GEOMETRY.  Eight steps are dealt round-robin without a synchronisation in between; afterwards
    every renderer's buffers are compared with the CPU restatement on a band of tile rows, bit for bit, and with one
    another over the whole frame."""
    import bench

    s = ugrt.scenes.crash(scale=1.0)
    W, H, lg, ud = 1920, 1080, (128, 128), (128, 128, 64)
    setup = setup_for(ugrt, s, "ref")
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY
    renderers = []
    for i in range(4):
        stream = torch.cuda.Stream() if i else None
        with torch.cuda.stream(stream):
            cx = ugrt.Context(W, H, light_grid=lg, flags=flags, uniform_dims=ud)
            rr = ugrt.Renderer(cx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True,
                               helper_thread=False)
        rr._stream = stream
        rr.aux.set_option("dda_blocks", bench.DDA_WAVES_THROUGHPUT)
        rr.aux.set_option("dda_rays_per_wave", bench.DDA_RPW_THROUGHPUT)
        rr.ctx.set_option("shadow_waves", bench.SHADOW_WAVES_THROUGHPUT)
        for c in (rr.ctx, rr.aux):
            c.set_option("async_build", 1)
        renderers.append(rr)
    for k in range(8):
        rr = renderers[k % 4]
        with torch.cuda.stream(rr._stream):
            rr.display(setup, frame_cnt=1, shadows=True, reflect=True)
    for rr in renderers:
        rr.synchronize()  # (raises if an estimate overflowed: none may, the frames are identical)
    torch.cuda.synchronize()
    rows = (60, 64)
    want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True, reflect=True, uniform_dims=ud)
    a, b = want["p0"], want["p0"] + want["n"]
    pr = want["primary"]
    first = renderers[0]
    for i, rr in enumerate(renderers):
        what = "renderer %d" % i
        np.testing.assert_array_equal(rr.t.cpu().numpy()[a:b].view(np.uint32), pr["t"][a:b].view(np.uint32), err_msg=what)
        np.testing.assert_array_equal(rr.is_shadowed.cpu().numpy()[a:b], want["is_shadowed"][a:b], err_msg=what)
        np.testing.assert_array_equal(rr.intersect_id.cpu().numpy()[a:b], want["mat_ids"][a:b], err_msg=what)
        np.testing.assert_array_equal(rr.hit_id.cpu().numpy()[a:b], want["hit_id"][a:b], err_msg=what)
        np.testing.assert_array_equal(rr.hit_t.cpu().numpy()[a:b].view(np.uint32), want["hit_t"][a:b].view(np.uint32), err_msg=what)
        np.testing.assert_array_equal(rr.image.cpu().numpy()[3 * a:3 * b], want["image"][3 * a:3 * b], err_msg=what)
        if i:
            for n in ("t", "is_shadowed", "intersect_id", "hit_id", "hit_t", "image", "normal", "dir"):
                assert torch.equal(getattr(rr, n).view(torch.uint8), getattr(first, n).view(torch.uint8)), (what, n)
    assert (pr["id"][a:b] >= 0).sum() > 1000 and want["is_shadowed"][a:b].sum() > 100 and want["active"][a:b].sum() > 100


def test_4k_frame_bands_and_oracle(ugrt, O, torch):
    """BASELINE configs[3]: the 1 M-triangle scene at 3840x2160 (129 600 screen cells: 17-bit sort keys, three
    radix passes).  The full frame on one context, the same frame as two half-image bands (the multi-GPU
    split) and the two-stream renderer must agree byte for byte; the oracle checks two tile rows."""
    s = ugrt.scenes.crash(scale=1.0)
    W, H, lg, ud = 3840, 2160, (128, 128), (128, 128, 64)
    setup = setup_for(ugrt, s, "ref")
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS
    ctx, r = make(ugrt, s, W, H, lg, flags=flags, udims=ud)
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    N, nby = W * H, H // 8
    value, key, span, offset, gi = ctx.grid_arrays(ugrt.GRID_PERSPECTIVE)
    k = u32(key).astype(np.int64)
    assert gi.num_cells == 129600 and (np.diff(k) >= 0).all() and k.max() >= 1 << 16
    np.testing.assert_array_equal(np.bincount(k, minlength=gi.num_cells), u32(span).astype(np.int64))
    full = {n: getattr(r, n).cpu().numpy().copy() for n in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id", "image")}
    del value, key, span, offset
    # two bands = what two ranks compute
    for rows in ((0, nby // 2), (nby // 2, nby)):
        bctx, br = make(ugrt, s, W, H, lg, rows=rows, flags=flags, udims=ud)
        br.display(setup, shadows=True, reflect=True)
        bctx.synchronize()
        a, b = bctx.p0, bctx.p0 + bctx.npix
        for n in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id"):
            np.testing.assert_array_equal(getattr(br, n).cpu().numpy()[a:b].view(np.uint32), full[n][a:b].view(np.uint32),
                                          err_msg="band %s %s" % (rows, n))
        np.testing.assert_array_equal(br.image.cpu().numpy()[3 * a:3 * b], full["image"][3 * a:3 * b])
        del br, bctx
    # two streams
    octx = ugrt.Context(W, H, light_grid=lg, flags=flags, uniform_dims=ud)
    orr = ugrt.Renderer(octx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True)
    for _ in range(2):
        orr.display(setup, shadows=True, reflect=True)
    orr.synchronize()
    torch.cuda.synchronize()
    for n in full:
        np.testing.assert_array_equal(getattr(orr, n).cpu().numpy().view(np.uint8), full[n].view(np.uint8), err_msg=n)
    orr.close()
    # the oracle on two tile rows
    rows = (135, 137)
    want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True, reflect=True, uniform_dims=ud)
    a, b = want["p0"], want["p0"] + want["n"]
    np.testing.assert_array_equal(full["t"][a:b].view(np.uint32), want["primary"]["t"][a:b].view(np.uint32))
    np.testing.assert_array_equal(full["is_shadowed"][a:b], want["is_shadowed"][a:b])
    np.testing.assert_array_equal(full["hit_id"][a:b], want["hit_id"][a:b])
    np.testing.assert_array_equal(full["image"][3 * a:3 * b], want["image"][3 * a:3 * b])


def test_gather_path_without_records(ugrt, O, torch):
    """When the caller's vertex/face arrays are not the ones the grids were last built from, the tracers fall
    back to gathering 3 indices + 3 vertices per reference (the reference's own staging); same results."""
    s = scene(ugrt, "hall")
    W, H, lg, ud = 256, 256, (64, 64), (32, 32, 16)
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    verts2, faces2 = r.d_verts.clone(), r.d_faces.clone()
    for name in ("trace_primary", "trace_shadow", "trace_dda"):
        orig = getattr(ctx, name)

        def patched(*a, _orig=orig):
            a = [verts2 if x is r.d_verts else faces2 if x is r.d_faces else x for x in a]
            return _orig(*a)

        setattr(ctx, name, patched)
    setup = setup_for(ugrt, s, "ref")
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud)
    np.testing.assert_array_equal(r.t.cpu().numpy().view(np.uint32), want["primary"]["t"].view(np.uint32))
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
    np.testing.assert_array_equal(r.hit_id.cpu().numpy(), want["hit_id"])
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])


@pytest.mark.parametrize("helper_thread", [True, False])
def test_overlapped_display_equals_sequential(ugrt, O, torch, helper_thread):
    """Renderer(overlap=True) builds the light and uniform grids on a second stream/context; same frame.  With a
    helper thread for the side stream (builds that wait for the device), or from one thread with builds that never
    wait (option async_build)."""
    s = scene(ugrt, "crash")
    W, H, lg, ud = 256, 144, (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    ctx2 = ugrt.Context(W, H, light_grid=lg, uniform_dims=ud)
    r2 = ugrt.Renderer(ctx2, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True,
                       helper_thread=helper_thread)
    for _ in range(4):  # several frames: the streams must stay ordered across frames
        r2.display(setup, shadows=True, reflect=True)
    r2.synchronize()
    torch.cuda.synchronize()
    for name in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id", "image"):
        a, b = getattr(r, name).cpu().numpy(), getattr(r2, name).cpu().numpy()
        np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8), err_msg=name)
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud)
    np.testing.assert_array_equal(r2.image.cpu().numpy(), want["image"])
    r2.close()


@pytest.mark.parametrize("name,cam,W,H,lg", [("hall", "ref", 256, 256, (64, 64)), ("crash", "ref", 256, 144, (128, 128)),
                                             ("cornell", "B", 256, 256, (32, 32))])
@pytest.mark.parametrize("slabs", [2, 4])
def test_z_slabs(ugrt, O, torch, name, cam, W, H, lg, slabs):
    """NUM_SLABS > 1 (main.cu.h:18): projCoordZ, the zMin/zMax loop (frustum_grid.h:221-241), SlabKernel
    (grid_kernel.cu:334), slab keys in both grids, the slab walk of rckernel_alpha (trace_kernel.cu:132-229) with
    its state machine as written, the light kernel over all slabs of a cell (light_kernel.cu:105)."""
    s = scene(ugrt, name)
    setup = setup_for(ugrt, s, cam)
    for all_chunks in (False, True):
        flags = ugrt.FLAG_SHADOW_ALL_CHUNKS if all_chunks else 0
        ctx = ugrt.Context(W, H, light_grid=lg, flags=flags, uniform_dims=(32, 32, 16), slabs=slabs)
        r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
        r.display(setup, frame_cnt=1, shadows=True)
        ctx.synchronize()
        want = O.frame(s, setup, W, H, light_grid=lg, all_chunks=all_chunks, slabs=slabs)
        for which, key in ((ugrt.GRID_PERSPECTIVE, "grid"), (ugrt.GRID_SPHERICAL, "lgrid")):
            # the renderer rebuilt the perspective grid last with the light camera?  no: one build each per frame
            value, keys, span, offset, gi = ctx.grid_arrays(which)
            g = want[key]
            assert gi.num_cells == len(g["span"]) and gi.total_refs == g["R"]
            np.testing.assert_array_equal(u32(keys), g["keys"])
            np.testing.assert_array_equal(u32(value), g["vals"])
            np.testing.assert_array_equal(u32(span), g["span"])
            np.testing.assert_array_equal(u32(offset), g["offset"])
            si = ctx.grid_slabs(which)
            assert si.slabs == slabs
            assert_bits_equal(np.array([si.z_min, si.z_max], np.float32), np.array(g["zrange"], np.float32), "zMin/zMax")
            pz = ctx.wrap_u32(si.d_proj_coord_z, len(g["zmin"]))
            np.testing.assert_array_equal(u32(pz), bits(g["zmin"]))
        pr = want["primary"]
        np.testing.assert_array_equal(r.t.cpu().numpy().view(np.uint32), bits(pr["t"]))
        assert_bits_equal(r.normal.cpu().numpy(), pr["normal"], "normal")
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
        np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
        np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
        assert (pr["id"] >= 0).sum() > 0


@pytest.mark.parametrize("name,nparts", [("hall", 2), ("crash", 3)])
def test_sharded_grid_build_merge(ugrt, O, torch, name, nparts):
    """SURVEY 8f.1 on the device: the light grid and the uniform grid are built in `nparts` windows of the
    triangle list (what the ranks of a node do) and merged by ugrt_grid_merge_shards; keys, values, span, offset
    and cells_used are those of the full build and of the oracle."""
    from ugrt import parallel

    s = scene(ugrt, name)
    W, H, lg, udims = 256, 144, (64, 64), (32, 32, 16)
    ctx, r = make(ugrt, s, W, H, lg, udims=udims)
    setup = setup_for(ugrt, s, "ref")
    lcam = O.cam_from(setup.light_camera, setup.fovy, float(np.float32(W) / np.float32(H)))
    F = r.F
    for which in (ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM):
        def build():
            if which == ugrt.GRID_SPHERICAL:
                ctx.upload_camera(lcam.cc)
                ctx.grid_build_spherical(r.d_faces, r.d_verts, F, float(np.float32(np.pi)), float(np.float32(np.pi)))
            else:
                ctx.grid_build_uniform(r.d_faces, r.d_verts, F, r.bbmin, r.bbmax)
        parts = []
        for k in range(nparts):
            ctx.set_face_window(*parallel.face_window(k, nparts, F))
            build()
            value, key, span, offset, gi = ctx.grid_arrays(which)
            parts.append((key.clone(), value.clone(), span.clone(), gi.total_refs))
        ctx.grid_merge_shards(which, [p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts],
                              [p[3] for p in parts])
        ctx.synchronize()
        value, key, span, offset, gi = ctx.grid_arrays(which)
        got = [u32(x).copy() for x in (key, value, span, offset)] + [gi.total_refs, gi.cells_used]
        ctx.set_face_window(0, -1)
        build()
        ctx.synchronize()
        value, key, span, offset, gi = ctx.grid_arrays(which)
        full = [u32(x) for x in (key, value, span, offset)] + [gi.total_refs, gi.cells_used]
        for a, b in zip(got, full):
            np.testing.assert_array_equal(a, b)
        v3 = np.asarray(s["verts"], np.float32).reshape(-1, 3)
        want = (O.grid_spherical(lcam.cc, s["faces"], s["verts"], *lg) if which == ugrt.GRID_SPHERICAL
                else O.grid_uniform(s["faces"], s["verts"], v3.min(0), v3.max(0), udims))
        np.testing.assert_array_equal(got[0], want["keys"])
        np.testing.assert_array_equal(got[1], want["vals"])
        np.testing.assert_array_equal(got[2], want["span"])
        np.testing.assert_array_equal(got[3], want["offset"])
        assert got[4] == want["R"] and got[5] == want["used"]


def test_full_size_hall_1024(ugrt, O, torch):
    """BASELINE configs[1] at full size: the ~80 k-triangle hall at the reference's own 1024 x 1024 and
    128 x 128 light cells (main.cu.h:10-26), primary + shadow.  1024 is the one width at which the exact bilinear
    fetch of the direction table has the k/256 weights of the hardware filter (ray set-up).  The
    reference's launch rule (strict) is checked against the oracle on the WHOLE frame - which chunks are traced
    depends on all rays - and the all-chunks mode on a band of tile rows."""
    s = ugrt.scenes.hall(scale=1.0)
    W = H = 1024
    lg = (128, 128)
    setup = setup_for(ugrt, s, "ref")
    assert 70000 < s["num_faces"] < 90000
    # strict launch rule, whole frame
    ctx, r = make(ugrt, s, W, H, lg)
    r.display(setup, frame_cnt=1, shadows=True)
    ctx.synchronize()
    want = O.frame(s, setup, W, H, light_grid=lg, all_chunks=False)
    pr = want["primary"]
    np.testing.assert_array_equal(r.t.cpu().numpy().view(np.uint32), bits(pr["t"]))
    assert_bits_equal(r.dir.cpu().numpy(), pr["dir"], "dir")
    assert_bits_equal(r.normal.cpu().numpy(), pr["normal"], "normal")
    for which, key in ((ugrt.GRID_PERSPECTIVE, "grid"), (ugrt.GRID_SPHERICAL, "lgrid")):
        value, keys, span, offset, gi = ctx.grid_arrays(which)
        assert gi.num_cells == 16384 and gi.total_refs == want[key]["R"]
        np.testing.assert_array_equal(u32(keys), want[key]["keys"])
        np.testing.assert_array_equal(u32(value), want[key]["vals"])
        np.testing.assert_array_equal(u32(span), want[key]["span"])
        np.testing.assert_array_equal(u32(offset), want[key]["offset"])
    np.testing.assert_array_equal(u32(r.d_map), want["map"])
    assert r.num_chunks == want["nchunks"]
    np.testing.assert_array_equal(u32(r.prefix)[:r.num_chunks], want["prefix"][:want["nchunks"]])
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"])
    np.testing.assert_array_equal(r.intersect_id.cpu().numpy(), want["mat_ids"])
    np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"])
    assert (pr["id"] >= 0).mean() > 0.5 and want["is_shadowed"].sum() > 1000
    # all chunks, a band of tile rows (a ray's flag does not depend on the other rays then)
    ctx2, r2 = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
    r2.display(setup, frame_cnt=1, shadows=True)
    ctx2.synchronize()
    rows = (60, 68)
    wb = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True)
    a, b = wb["p0"], wb["p0"] + wb["n"]
    np.testing.assert_array_equal(r2.is_shadowed.cpu().numpy()[a:b], wb["is_shadowed"][a:b])
    np.testing.assert_array_equal(r2.image.cpu().numpy()[3 * a:3 * b], wb["image"][3 * a:3 * b])
    # every ray of the strict frame that is shadowed is shadowed with all chunks traced
    assert (r2.is_shadowed.cpu().numpy() >= want["is_shadowed"]).all()


def test_full_size_animated_rebuild(ugrt, O, torch):
    """BASELINE configs[4] at full size: the 1 M-triangle scene at 1920 x 1080, the 800 k-triangle sub-range
    transformed by Model::rotate_bunny(1.81 + 0.05 k) (scene.h:122, transformation_kernel.cu:4-18) before each of
    three frames, all grids rebuilt from the moved vertices.  Vertices bit-equal to the oracle's; the last frame
    against the oracle on a band of tile rows plus the whole-frame properties."""
    import ctypes

    s = ugrt.scenes.crash(scale=1.0)
    W, H, lg, ud = 1920, 1080, (128, 128), (128, 128, 64)
    setup = setup_for(ugrt, s, "ref")
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_STATIC_GEOMETRY  # as bench.py --animate runs it
    ctx, r = make(ugrt, s, W, H, lg, flags=flags, udims=ud)
    off, size = s["animated_offset"], s["animated_size"]
    assert size >= 700000
    r.init_orig_list(size, off)
    verts = np.ascontiguousarray(s["verts"], np.float32).reshape(-1).copy()
    orig = verts[3 * off:3 * (off + size)].copy()
    for k in range(3):
        rot = 1.81 + 0.05 * k
        r.rotate_bunny(rot)
        r.display(setup, shadows=True, reflect=True)
        cr, sr = ctypes.c_float(), ctypes.c_float()
        ugrt.lib.ugrt_rot_cos_sin(rot, ctypes.byref(cr), ctypes.byref(sr))
        O.animate(verts, orig, size, off, cr.value, sr.value)
    ctx.synchronize()
    assert_bits_equal(r.d_verts.cpu().numpy(), verts, "animated vertices")
    N = W * H
    for which in (ugrt.GRID_PERSPECTIVE, ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM):
        value, key, span, offset, gi = ctx.grid_arrays(which)
        k_, v_ = u32(key).astype(np.int64), u32(value).astype(np.int64)
        assert (np.diff(k_) >= 0).all() and gi.total_refs == len(k_)
        same = k_[1:] == k_[:-1]
        assert (v_[1:][same] > v_[:-1][same]).all()
        np.testing.assert_array_equal(np.bincount(k_, minlength=gi.num_cells), u32(span).astype(np.int64))
    dm = u32(r.d_map)
    assert (np.sort(dm[:N]) == np.arange(N)).all() and (np.diff(dm[N:].astype(np.int64)) >= 0).all()
    rows = (64, 68)
    want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, all_chunks=True, reflect=True, uniform_dims=ud, verts=verts)
    a, b = want["p0"], want["p0"] + want["n"]
    np.testing.assert_array_equal(r.t.cpu().numpy()[a:b].view(np.uint32), want["primary"]["t"][a:b].view(np.uint32))
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy()[a:b], want["is_shadowed"][a:b])
    np.testing.assert_array_equal(r.hit_id.cpu().numpy()[a:b], want["hit_id"][a:b])
    np.testing.assert_array_equal(r.hit_t.cpu().numpy()[a:b].view(np.uint32), want["hit_t"][a:b].view(np.uint32))
    np.testing.assert_array_equal(r.image.cpu().numpy()[3 * a:3 * b], want["image"][3 * a:3 * b])
    # the moved geometry is what was traced: the static frame differs from this one
    ctx0, r0 = make(ugrt, s, W, H, lg, flags=flags, udims=ud)
    r0.display(setup, shadows=True, reflect=True)
    ctx0.synchronize()
    assert (r0.image.cpu().numpy()[3 * a:3 * b] != want["image"][3 * a:3 * b]).any()


@pytest.mark.parametrize("where", ["main", "side"])
def test_overlapped_frame_survives_a_failed_frame(ugrt, O, torch, where):
    """A call that raises in one two-stream frame (on the main thread or on the helper thread of the side stream)
    must not leave the helper's result queue one entry behind: the next frames equal the sequential renderer's."""
    s = scene(ugrt, "crash")
    W, H, lg, ud = 256, 144, (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    ctx, r = make(ugrt, s, W, H, lg, udims=ud)
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    ctx2 = ugrt.Context(W, H, light_grid=lg, uniform_dims=ud)
    r2 = ugrt.Renderer(ctx2, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"], overlap=True)
    r2.display(setup, shadows=True, reflect=True)
    target = ctx2 if where == "main" else r2.aux
    name = "trace_primary" if where == "main" else "grid_build_uniform"
    real = getattr(target, name)

    def boom(*a, **k):
        setattr(target, name, real)  # only this frame
        raise RuntimeError("injected failure")

    setattr(target, name, boom)
    with pytest.raises(RuntimeError, match="injected"):
        r2.display(setup, shadows=True, reflect=True)
    r2.synchronize()
    assert r2._done.empty()
    for _ in range(2):
        r2.display(setup, shadows=True, reflect=True)
    r2.synchronize()
    torch.cuda.synchronize()
    for n in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id", "image"):
        np.testing.assert_array_equal(getattr(r, n).cpu().numpy().view(np.uint8), getattr(r2, n).cpu().numpy().view(np.uint8),
                                      err_msg=n)
    r2.close()


def test_async_builds_equal_the_waiting_form(ugrt, O, torch):
    """Option "async_build": no call of the frame waits for the device (the counts the reference reads back stay
    there); grids, flags and the image of later frames are those of the waiting form, ugrt_grid_info.total_refs is
    right after a synchronisation, and a capacity that does not fit is reported by ugrt_ctx_synchronize."""
    s = scene(ugrt, "crash")
    W, H, lg, ud = 256, 144, (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    ctx2, r2 = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    ctx2.set_option("async_build", 1)
    for _ in range(3):  # the first frame waits (no estimate yet), the others do not
        r2.display(setup, shadows=True, reflect=True)
    ctx2.synchronize()
    for which in (ugrt.GRID_PERSPECTIVE, ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM):
        a, b = ctx.grid_arrays(which), ctx2.grid_arrays(which)
        assert a[4].total_refs == b[4].total_refs and a[4].cells_used == b[4].cells_used
        for x, y in zip(a[:4], b[:4]):
            np.testing.assert_array_equal(u32(x), u32(y))
    for n in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id", "image"):
        np.testing.assert_array_equal(getattr(r, n).cpu().numpy().view(np.uint8), getattr(r2, n).cpu().numpy().view(np.uint8),
                                      err_msg=n)
    assert ctx2.stats()[1] == ctx.stats()[1]
    # a scene that suddenly needs far more references than the frame before: reported, then repaired
    big = ugrt.scenes.crash(scale=0.08)
    r3 = ugrt.Renderer(ctx2, big["verts"], big["faces"], big["matidx"], big["mat_list"], big["reflect"])
    r3.display(setup_for(ugrt, big, "ref"), shadows=True, reflect=True)
    with pytest.raises(ugrt.UgrtError, match="asynchronous"):
        ctx2.synchronize()
    r3.display(setup_for(ugrt, big, "ref"), shadows=True, reflect=True)  # waits, sizes exactly
    ctx2.synchronize()
    ctx3, r4 = make(ugrt, big, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    r4.display(setup_for(ugrt, big, "ref"), shadows=True, reflect=True)
    ctx3.synchronize()
    np.testing.assert_array_equal(r3.image.cpu().numpy(), r4.image.cpu().numpy())


@pytest.mark.parametrize("W,H", [(1024, 1024), (640, 360)])
def test_strict_texture_flag(ugrt, O, torch, W, H):
    """UGRT_FLAG_STRICT_TEXTURE (parity unpinned): kernel and oracle share ugrt_tex_linear8, so the frame equals the
    oracle's strict frame bit for bit; at 1024 x 1024 it also equals the default frame, at 640 x 360 it does not."""
    s = scene(ugrt, "hall")
    setup = setup_for(ugrt, s, "ref")
    lg = (64, 64)
    rows = (H // 16 - 4, H // 16 + 4)
    ctx, r = make(ugrt, s, W, H, lg, rows=rows, flags=ugrt.FLAG_STRICT_TEXTURE)
    r.display(setup, shadows=True)
    ctx.synchronize()
    ctx0, r0 = make(ugrt, s, W, H, lg, rows=rows)
    r0.display(setup, shadows=True)
    ctx0.synchronize()
    try:
        want = O.frame(s, setup, W, H, rows=rows, light_grid=lg, strict_texture=True)
    finally:
        O.set_strict_texture(False)
    sl = slice(ctx.p0, ctx.p0 + ctx.npix)
    sl3 = slice(3 * ctx.p0, 3 * (ctx.p0 + ctx.npix))
    assert_bits_equal(r.dir.cpu().numpy()[sl3], want["primary"]["dir"].reshape(-1)[sl3], "dir")
    assert_bits_equal(r.t.cpu().numpy()[sl], want["primary"]["t"][sl], "t")
    np.testing.assert_array_equal(r.is_shadowed.cpu().numpy()[sl], want["is_shadowed"][sl])
    np.testing.assert_array_equal(r.image.cpu().numpy()[sl3], want["image"].reshape(-1)[sl3])
    same = bool(torch.equal(r.dir[sl3].view(torch.int32), r0.dir[sl3].view(torch.int32)))
    assert same == (W == 1024)


def test_ray_sort_on_demand(ugrt, O, torch):
    """Deferred ugrt_sort_rays under FLAG_SHADOW_ALL_CHUNKS: the frame does not sort the ray map (the chunk list only says
    which rays the reference's launch traces - all of them with that flag - and the shadow tracer orders (pixel, cell)
    pairs itself); the sorted map, the chunk starts and the chunk count are produced when they are asked for.  Checked:
    the shadow flags with and without the sort ("ray_sort" 1), the radix launches a frame saves, processData's outputs
    on demand after the frame (and after a second frame that overwrote the map), and that a context without the flag
    always sorts."""
    s = scene(ugrt, "crash")
    W, H, lg = 256, 144, (64, 64)
    setup = setup_for(ugrt, s, "ref")
    want = O.frame(s, setup, W, H, light_grid=lg, all_chunks=True)
    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS)
    launches = {}
    for mode in (1, 0, -1):
        ctx.set_option("ray_sort", mode)
        r.display(setup, frame_cnt=1, shadows=True)  # (buffers sized, estimates taken)
        ctx.synchronize()
        before = ctx.get_state("radix_launches")
        r.display(setup, frame_cnt=1, shadows=True)
        ctx.synchronize()
        launches[mode] = ctx.get_state("radix_launches") - before
        np.testing.assert_array_equal(r.is_shadowed.cpu().numpy(), want["is_shadowed"], err_msg="ray_sort %d" % mode)
        np.testing.assert_array_equal(r.image.cpu().numpy(), want["image"], err_msg="ray_sort %d" % mode)
        if mode != 1:  # nothing sorted the map so far: cells in pixel order
            cells = u32(r._d_map)[W * H:].astype(np.int64)
            assert (np.diff(cells) < 0).any()
        # processData's outputs, on demand
        assert r.num_chunks == want["nchunks"]
        np.testing.assert_array_equal(u32(r.d_map), want["map"])
        np.testing.assert_array_equal(u32(r.prefix)[:r.num_chunks], want["prefix"][:want["nchunks"]])
    passes = (bits_for_cells(lg[0] * lg[1] + 1) + 7) // 8
    assert launches[0] == launches[-1] == launches[1] - (passes + 1), launches
    # without the flag the chunk list decides what is traced: the deferred form sorts at once
    ctx2, r2 = make(ugrt, s, W, H, lg)
    r2.display(setup, frame_cnt=1, shadows=True)
    ctx2.synchronize()
    cells = u32(r2._d_map)[W * H:].astype(np.int64)
    assert (np.diff(cells) >= 0).all()
    want2 = O.frame(s, setup, W, H, light_grid=lg, all_chunks=False)
    np.testing.assert_array_equal(r2.is_shadowed.cpu().numpy(), want2["is_shadowed"])


def test_float_to_int_instructions_equal_the_portable_forms(ugrt):
    """ugrt_f2i / ugrt_f2u / ugrt_floor2i pin the reference's float -> integer conversions down (NaN -> 0, saturating); on
    the device they are v_cvt_i32_f32 / v_cvt_u32_f32 / v_floor_f32 instead of the portable forms' branches.  Every float
    bit pattern, all three functions, on the device."""
    ctx = ugrt.Context(64, 64)
    assert ctx.get_state("f2i_mismatches") == 0


def test_lane_reductions_without_the_lds_crossbar(ugrt):
    """The tracers' box reductions run on DPP controls and gfx950's v_permlane16_swap / v_permlane32_swap (ugrt_packet.h)
    instead of __shfl_xor (a ds_bpermute_b32 per step).  A reduction that misses lanes would give a box too small - wrong
    culls - or, silently, one too large; the library compares them with the __shfl_xor forms on the device."""
    ctx = ugrt.Context(64, 64)
    assert ctx.get_state("lane_reduce_mismatches") == 0


def test_short_reciprocal_equals_the_division_for_every_float(ugrt):
    """The exact triangle tests invert det by v_rcp_f32 and one Newton step instead of the compiler's division sequence
    (ugrt_dev.h d_recip_det).  The claim is that this is the SAME float as 1.0f / det for every det the tests can reach
    (|det| >= 1e-21; 2^124 and above, infinities and NaNs take the division): the library runs all 2^32 bit patterns
    through the function on the device and counts the operands whose result differs by a bit."""
    ctx = ugrt.Context(64, 64)
    assert ctx.get_state("recip_mismatches") == 0


@pytest.mark.parametrize("async_build", [0, 1])
def test_sort_options_through_a_frame(ugrt, O, torch, async_build):
    """The radix sort's forms under a whole frame (three grid builds, ray sort, the shadow tracer's three sorts): ranks by
    LDS atomics / by ballots, tiles of 4096 / 8192 pairs, the library's sort.  Every pass counts the digit of the pass
    that follows and the last workgroup of a pass clears its rows: the same frame several times in a row must keep
    giving the oracle's grids and images (a row left dirty would misplace the next sort's digits)."""
    s = scene(ugrt, "crash")
    W, H, lg, ud = 256, 144, (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s, "ref")
    ctx, r = make(ugrt, s, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    ctx.set_option("async_build", async_build)
    want = O.frame(s, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud, all_chunks=True)
    for k, (rank, items, library) in enumerate([(-1, -1, 0), (0, -1, 0), (-1, 8, 0), (0, 16, 0), (-1, 16, 0), (-1, -1, 1), (-1, -1, 0)]):
        ctx.set_option("sort_rank", rank)
        ctx.set_option("sort_items", items)
        ctx.set_option("sort_library", library)
        r.display(setup, shadows=True, reflect=True)
        ctx.synchronize()
        for n, key in (("intersect_id", "mat_ids"), ("is_shadowed", "is_shadowed"), ("hit_id", "hit_id"), ("image", "image")):
            np.testing.assert_array_equal(getattr(r, n).cpu().numpy(), want[key], err_msg="%s frame %d" % (n, k))
        for which, g in ((ugrt.GRID_PERSPECTIVE, "grid"), (ugrt.GRID_SPHERICAL, "lgrid"), (ugrt.GRID_UNIFORM, "ugrid")):
            value, key, span, offset, gi = ctx.grid_arrays(which)
            np.testing.assert_array_equal(u32(key), want[g]["keys"])
            np.testing.assert_array_equal(u32(value), want[g]["vals"])


def test_async_build_growth_between_launch_size_and_buffer_capacity(ugrt, O, torch):
    """A grid (and the shadow pass's pair list) that grows by about a third from one frame to the next: more than the
    launches of an asynchronous build are sized for (the last need + 25 % + 64 K), less than the grow-only buffers
    hold (1.5 x the first need).  The overflow check must be made against the LAUNCH size: every kernel of the build
    runs one thread per reference of the launch, so a count between the two would leave references unfilled and
    unsorted without any report.  Expected: UGRT_EOVERFLOW at the synchronisation, then a repaired frame; in every
    case the buffers end up equal to a context that only ever built in the waiting form."""
    a, b = ugrt.scenes.crash(scale=0.30), ugrt.scenes.crash(scale=0.405)
    W, H, lg, ud = 320, 200, (64, 64), (32, 32, 16)
    flags = ugrt.FLAG_SHADOW_ALL_CHUNKS
    ctx, ra = make(ugrt, a, W, H, lg, flags=flags, udims=ud)
    ctx.set_option("async_build", 1)
    for _ in range(3):
        ra.display(setup_for(ugrt, a, "ref"), shadows=True, reflect=True)
    ctx.synchronize()
    refs_a = [ctx.grid_info(g).total_refs for g in (ugrt.GRID_PERSPECTIVE, ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM)]
    rb = ugrt.Renderer(ctx, b["verts"], b["faces"], b["matidx"], b["mat_list"], b["reflect"])
    setup_b = setup_for(ugrt, b, "ref")
    rb.display(setup_b, shadows=True, reflect=True)
    raised = False
    try:
        ctx.synchronize()
    except ugrt.UgrtError as e:
        assert "asynchronous" in str(e)
        raised = True
    if raised:
        rb.display(setup_b, shadows=True, reflect=True)  # waits, sizes exactly
        ctx.synchronize()
    wctx, wr = make(ugrt, b, W, H, lg, flags=flags, udims=ud)
    wr.display(setup_b, shadows=True, reflect=True)
    wctx.synchronize()
    refs_b = [wctx.grid_info(g).total_refs for g in (ugrt.GRID_PERSPECTIVE, ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM)]
    # at least one grid lies in the window the old check missed; then the overflow must have been reported
    in_window = [ra_ > 262144 and ra_ * 1.25 + 65536 < rb_ <= ra_ * 1.5 for ra_, rb_ in zip(refs_a, refs_b)]
    assert any(in_window), (refs_a, refs_b)
    assert raised, (refs_a, refs_b)
    for which in (ugrt.GRID_PERSPECTIVE, ugrt.GRID_SPHERICAL, ugrt.GRID_UNIFORM):
        x, y = ctx.grid_arrays(which), wctx.grid_arrays(which)
        assert x[4].total_refs == y[4].total_refs
        for p, q in zip(x[:4], y[:4]):
            np.testing.assert_array_equal(u32(p), u32(q))
    for n in ("t", "is_shadowed", "hit_id", "hit_t", "intersect_id", "image"):
        np.testing.assert_array_equal(getattr(rb, n).cpu().numpy().view(np.uint8), getattr(wr, n).cpu().numpy().view(np.uint8),
                                      err_msg=n)


def test_async_build_meets_a_wide_triangle_it_was_not_built_for(ugrt, O, torch):
    """A grid whose last build had no wide triangle (one that covers every cell) is built asynchronously without
    the merge stage.  When the geometry then brings one (here: a sheet around the whole scene, which the uniform
    grid and the light grid bin to every cell), the build must report it instead of losing it: UGRT_EOVERFLOW at
    the synchronisation, a repaired frame afterwards, equal to a context that never built asynchronously."""
    s0 = scene(ugrt, "crash")
    W, H, lg, ud = 256, 144, (64, 64), (32, 32, 16)
    setup = setup_for(ugrt, s0, "ref")
    verts = np.asarray(s0["verts"], np.float32).reshape(-1, 3)
    lo, hi = verts.min(0), verts.max(0)
    d = hi - lo
    sheet = np.array([lo - 0.01 * d, [hi[0] + 0.01 * d[0], hi[1] + 0.01 * d[1], lo[2] - 0.01 * d[2]],
                      [lo[0] - 0.01 * d[0], hi[1] + 0.01 * d[1], hi[2] + 0.01 * d[2]]], np.float32)
    v2 = np.concatenate([verts, sheet])
    f2 = np.concatenate([np.asarray(s0["faces"], np.int32), np.array([[len(verts), len(verts) + 1, len(verts) + 2]], np.int32)])
    s1 = dict(s0)
    s1.update(verts=v2, faces=f2, matidx=np.concatenate([np.asarray(s0["matidx"], np.int32), np.zeros(1, np.int32)]))
    ctx, r = make(ugrt, s0, W, H, lg, flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, udims=ud)
    ctx.set_option("async_build", 1)
    for _ in range(3):
        r.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    r1 = ugrt.Renderer(ctx, s1["verts"], s1["faces"], s1["matidx"], s1["mat_list"], s1["reflect"])
    r1.display(setup, shadows=True, reflect=True)
    with pytest.raises(ugrt.UgrtError, match="asynchronous"):
        ctx.synchronize()
    for _ in range(3):  # waits once and sizes exactly, then runs asynchronously again (now with the merge stage)
        r1.display(setup, shadows=True, reflect=True)
    ctx.synchronize()
    want = O.frame(s1, setup, W, H, light_grid=lg, reflect=True, uniform_dims=ud, all_chunks=True)
    for n, k in (("intersect_id", "mat_ids"), ("is_shadowed", "is_shadowed"), ("hit_id", "hit_id"), ("image", "image")):
        np.testing.assert_array_equal(getattr(r1, n).cpu().numpy(), want[k], err_msg=n)
    gi = ctx.grid_info(ugrt.GRID_UNIFORM)
    assert gi.total_refs > gi.num_cells  # the sheet is in every cell of the uniform grid
