"""CPU-only checks of the oracle itself: structural invariants, the committed golden fixture,
the uniform-grid DDA against brute force, strict-vs-all shadow chunks, band splitting."""
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hall(ugrt):
    return ugrt.scenes.hall(scale=0.05)


def setup_of(ugrt, s, cam="ref"):
    return ugrt.FrameSetup(s["cameras"][cam], s["light_camera"], s["shading_light"])


def check_grid(g, C):
    keys, vals, span, offset = g["keys"], g["vals"], g["span"], g["offset"]
    assert len(keys) == g["R"] == int(g["sizes"].astype(np.int64).sum())
    assert (np.diff(keys.astype(np.int64)) >= 0).all()
    same = keys[1:] == keys[:-1]
    assert (vals[1:][same] > vals[:-1][same]).all()  # stable sort: ascending triangle ids inside a cell
    assert int(span.astype(np.int64).sum()) == g["R"] and len(span) == C
    np.testing.assert_array_equal(offset, np.concatenate([[0], np.cumsum(span.astype(np.int64))[:-1]]).astype(np.uint32))
    np.testing.assert_array_equal(np.bincount(keys, minlength=C).astype(np.uint32), span)
    np.testing.assert_array_equal(np.bincount(vals, minlength=len(g["sizes"])).astype(np.uint32), g["sizes"])
    assert g["used"] == int((span > 0).sum())


def test_grid_invariants(ugrt, O, hall):
    cam = O.cam_from(hall["cameras"]["ref"], 45.0, 1.0)
    check_grid(O.grid_perspective(cam.cc, hall["faces"], hall["verts"], 32, 32), 1024)
    lcam = O.cam_from(hall["light_camera"], 45.0, 1.0)
    check_grid(O.grid_spherical(lcam.cc, hall["faces"], hall["verts"], 64, 64), 4096)
    v = hall["verts"]
    check_grid(O.grid_uniform(hall["faces"], v, v.min(0), v.max(0), (16, 16, 8)), 2048)


def test_golden_cornell(ugrt, O):
    """tests/golden/cornell256_B.npz was written by tests/golden/make_golden.py from this oracle
    (oracle-generated, NOT reference output): it pins the oracle against accidental change."""
    z = np.load(os.path.join(GOLD, "cornell256_B.npz"))
    s = ugrt.scenes.cornell()
    r = O.frame(s, setup_of(ugrt, s, "B"), 256, 256, light_grid=(128, 128))
    pr = r["primary"]
    np.testing.assert_array_equal(pr["id"], z["id"].astype(np.int32))
    np.testing.assert_array_equal(pr["t"].view(np.uint32), z["t_bits"])
    np.testing.assert_array_equal(r["is_shadowed"].astype(np.uint8), z["shadowed"])
    np.testing.assert_array_equal(r["image"], z["image"])
    assert zlib.crc32(pr["dir"].tobytes()) == int(z["crc_dir"]) and zlib.crc32(pr["normal"].tobytes()) == int(z["crc_normal"])
    assert zlib.crc32(r["map"].tobytes()) == int(z["crc_map"]) and r["nchunks"] == int(z["nchunks"])
    assert r["grid"]["R"] == int(z["R"]) and r["lgrid"]["R"] == int(z["lR"])


def test_tie_break_lowest_triangle_wins(ugrt, O):
    """Cornell: the light quad (ids 2,3) is coplanar with the ceiling (ids 4,5); the strict `<` keeps the
    first ref in the stable-sorted list = the lower id (trace_kernel.cu:38)."""
    s = ugrt.scenes.cornell()
    r = O.frame(s, setup_of(ugrt, s, "B"), 256, 256, shadows=False)
    ids = set(np.unique(r["primary"]["id"]))
    assert {2, 3} <= ids and {4, 5} <= ids


def test_dda_matches_brute_force(ugrt, O, hall):
    r = O.frame(hall, setup_of(ugrt, hall), 128, 128, light_grid=(32, 32), reflect=True, uniform_dims=(24, 24, 12),
                shadows=False)
    N = 128 * 128
    bt, bid = O.brute_nearest(hall["verts"], hall["faces"], r["rays"], r["active"], 0, N, N)
    act = r["active"] == 1
    assert act.sum() > 1000
    same = r["hit_id"][act] == bid[act]
    # a hit exactly on a cell face can belong to the neighbour cell by one ulp: tolerate a handful
    assert same.mean() > 0.999, same.mean()
    both = act & (r["hit_id"] == bid)
    np.testing.assert_array_equal(r["hit_t"][both].view(np.uint32), bt[both].view(np.uint32))
    tests, cells, nact = r["dda_counters"]
    assert nact == act.sum() and cells >= nact and tests > 0


def test_shadow_strict_is_subset_of_all_chunks(ugrt, O, hall):
    st = O.frame(hall, setup_of(ugrt, hall), 128, 128, light_grid=(32, 32))
    al = O.frame(hall, setup_of(ugrt, hall), 128, 128, light_grid=(32, 32), all_chunks=True)
    assert ((st["is_shadowed"] == 1) <= (al["is_shadowed"] == 1)).all()
    # strict: the reference launches nbx*nby = 256 blocks, block b takes chunk b-1 -> chunks >= 255 are dropped
    n, nch = st["n"], st["nchunks"]
    assert nch > 256
    dropped = st["map"][:n][st["prefix"][255]:]
    assert (st["is_shadowed"][dropped] == 0).all()
    assert al["is_shadowed"].sum() >= st["is_shadowed"].sum()


def test_band_union_equals_full_frame(ugrt, O, hall):
    full = O.frame(hall, setup_of(ugrt, hall), 128, 128, light_grid=(32, 32), all_chunks=True)
    img = np.zeros_like(full["image"])
    for rows in ((0, 5), (5, 16)):
        b = O.frame(hall, setup_of(ugrt, hall), 128, 128, rows=rows, light_grid=(32, 32), all_chunks=True)
        a, e = 3 * b["p0"], 3 * (b["p0"] + b["n"])
        img[a:e] = b["image"][a:e]
        assert b["grid"]["R"] <= full["grid"]["R"]
    np.testing.assert_array_equal(img, full["image"])


def test_empty_and_degenerate_inputs(ugrt, O):
    """a scene entirely outside the band, zero-area triangles, a triangle through the eye plane"""
    cam = O.cam_from(dict(eye=(0, 0, 0), look=(0, 0, -1), up=(0, 1, 0), near=0.1, far=100.0), 45.0, 1.0)
    verts = np.array([[0, 0, -5], [0, 0, -5], [0, 0, -5], [-1, -1, 1], [1, -1, -3], [0, 1, -3]], np.float32)
    faces = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    g = O.grid_perspective(cam.cc, faces, verts, 8, 8)
    check_grid(g, 64)
    out = O.trace_primary(cam, 64, 64, g, verts, faces)
    assert set(np.unique(out["id"])) <= {-2, 1}
    g2 = O.grid_perspective(cam.cc, faces[:1], verts, 8, 8, rows=(0, 2))
    assert g2["R"] == 0 and g2["span"].sum() == 0
    out2 = O.trace_primary(cam, 64, 64, g2, verts, faces[:1], rows=(0, 2))
    assert (out2["id"][:2 * 8 * 64] == -2).all() and (out2["t"][:2 * 8 * 64] == -1).all()


def test_slabs_one_equals_the_unslabbed_path(O, ugrt):
    """orc_trace_primary_slabs with one slab is orc_trace_primary; with more slabs the light kernel's flags stay
    those of the unslabbed light grid (a cell's slabs partition its list, light_kernel.cu:105-113)."""
    s = ugrt.scenes.hall(scale=0.05)
    setup = ugrt.FrameSetup(s["cameras"]["ref"], s["light_camera"], s["shading_light"])
    W = H = 128
    a = O.frame(s, setup, W, H, light_grid=(32, 32), all_chunks=True)
    cam = a["cam"]
    g1 = O.grid_perspective(cam.cc, s["faces"], s["verts"], W // 8, H // 8)
    import numpy as np
    o = dict(normal=np.zeros(3 * W * H, np.float32), t=np.zeros(W * H, np.float32), dir=np.zeros(3 * W * H, np.float32),
             shadowed=np.zeros(W * H, np.int32), id=np.zeros(W * H, np.int32))
    O._lib.orc_trace_primary_slabs(O._p(cam.cc), O._p(cam.tex), W, H, W // 8, H // 8, 0, H // 8, 1, O._p(g1["vals"]),
                                   O._p(g1["span"]), O._p(g1["offset"]), O._p(O._f32(s["verts"]).reshape(-1)),
                                   O._p(O._i32(s["faces"]).reshape(-1)), O._p(o["normal"]), O._p(o["t"]), O._p(o["dir"]),
                                   O._p(o["shadowed"]), O._p(o["id"]))
    np.testing.assert_array_equal(o["id"], a["primary"]["id"])
    np.testing.assert_array_equal(o["t"].view(np.uint32), a["primary"]["t"].view(np.uint32))
    b = O.frame(s, setup, W, H, light_grid=(32, 32), all_chunks=True, slabs=4)
    # every triangle sits in exactly one slab of each cell it covers
    assert b["grid"]["R"] == a["grid"]["R"] and b["lgrid"]["R"] == a["lgrid"]["R"]
    assert len(b["grid"]["span"]) == 4 * len(a["grid"]["span"])
    sp4 = b["lgrid"]["span"].reshape(-1, 4).sum(1)
    np.testing.assert_array_equal(sp4, a["lgrid"]["span"])
    # the slab walk accepts a hit only in the slab of its ndc depth and drops accepted rays of a tile that goes on:
    # it never finds MORE hits than the single-slab walk, and the scene keeps some
    hit4, hit1 = b["primary"]["id"] >= 0, a["primary"]["id"] >= 0
    assert hit4.sum() > 0 and not (hit4 & ~hit1).any()


def test_strict_texture_weights(O, ugrt):
    """UGRT_FLAG_STRICT_TEXTURE (parity unpinned: CUDA's documented 8-bit weight rule, ugrt_fmath.h ugrt_tex_linear8):
    at 1024 x 1024 the exact float weights ARE multiples of 1/256, so the rays are bit-identical to the default path;
    at 1920 x 1080 they are not, and the quantised weights give other rays."""
    s = ugrt.scenes.cornell()
    setup = ugrt.FrameSetup(s["cameras"]["B"], s["light_camera"], s["shading_light"])
    try:
        for W, H, same in ((1024, 1024, True), (1920, 1080, False)):
            rows = (H // 16 - 2, H // 16 + 2)
            a = O.frame(s, setup, W, H, rows=rows, shadows=False)["primary"]
            b = O.frame(s, setup, W, H, rows=rows, shadows=False, strict_texture=True)["primary"]
            equal = np.array_equal(a["dir"].view(np.uint32), b["dir"].view(np.uint32))
            assert equal == same, (W, H)
            if not same:  # still the same picture to within the filter's resolution
                assert np.abs(a["dir"] - b["dir"]).max() < 2e-3 and (a["id"] == b["id"]).mean() > 0.98
    finally:
        O.set_strict_texture(False)
