"""ctypes binding of oracle/libugrt_oracle.so + the reference frame sequence on the CPU.

Test infrastructure only (the checker); the product never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libugrt_oracle.so")
REF_OBJDUMP = os.path.join(ORACLE_DIR, "_ref", "ref_objdump")
PI_F = float(np.float32(np.pi))


def build():
    """(Re)build the oracle if its source is newer than the library."""
    src = os.path.join(ORACLE_DIR, "ugrt_oracle.c")
    hdr = os.path.join(ROOT, "include", "ugrt_fmath.h")
    if (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in (src, hdr)):
        subprocess.run(["make", "-C", ORACLE_DIR, "libugrt_oracle.so"], check=True, capture_output=True)


build()
# UGRT_ORACLE_LIB: an instrumented build of the oracle (make -C oracle asan)
_lib = C.CDLL(os.environ.get("UGRT_ORACLE_LIB") or LIB)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


_lib.orc_set_threads.restype = C.c_int
_lib.orc_inclusive_scan.restype = C.c_uint
_lib.orc_cell_boundaries.restype = C.c_uint
_lib.orc_process_rays.restype = C.c_uint
_lib.orc_stable_sort_pairs.restype = C.c_int
_lib.orc_write_ppm.restype = C.c_int
_lib.orc_parse_materials.restype = C.c_int


def set_threads(n):
    return _lib.orc_set_threads(C.c_int(n))


class Cam:
    """Outputs of orc_camera + orc_camcoords + orc_dirtex."""

    def __init__(self, eye, look, up, near, far, fovy=45.0, aspect=1.0):
        self.worldori = np.zeros(4, np.float32)
        self.MV, self.P, self.MVP = (np.zeros(16, np.float32) for _ in range(3))
        self.planes = np.zeros(36, np.float32)
        self.corners = np.zeros(24, np.float32)
        _lib.orc_camera(_p(_f32(eye)), _p(_f32(look)), _p(_f32(up)), C.c_float(near), C.c_float(far),
                        C.c_float(fovy), C.c_float(aspect), _p(self.worldori), _p(self.MV), _p(self.P),
                        _p(self.MVP), _p(self.planes), _p(self.corners))
        self.cc = np.zeros(64, np.float32)
        _lib.orc_camcoords(_p(self.worldori), _p(self.corners), _p(self.MV), _p(self.P), _p(self.MVP), _p(self.cc))
        self.tex = np.zeros(100, np.float32)
        _lib.orc_dirtex(_p(self.cc), _p(self.tex))


def cam_from(params, fovy, aspect):
    return Cam(params["eye"], params["look"], params["up"], params["near"], params["far"], fovy, aspect)


def _finish_grid(rng_fill, sizes, F, C_cells, nkeys):
    scan = np.zeros(F, np.uint32)
    R = _lib.orc_inclusive_scan(_p(sizes), _p(scan), C.c_int(F))
    keys = np.zeros(max(R, 1), np.uint32)
    vals = np.zeros(max(R, 1), np.uint32)
    rng_fill(scan, keys, vals)
    rc = _lib.orc_stable_sort_pairs(_p(keys), _p(vals), C.c_uint(R), C.c_uint(nkeys))
    assert rc == 0, rc
    span = np.zeros(C_cells, np.uint32)
    offset = np.zeros(C_cells, np.uint32)
    used = _lib.orc_cell_boundaries(_p(keys), C.c_uint(R), C.c_uint(C_cells), _p(span), _p(offset))
    return dict(keys=keys[:R], vals=vals[:R], span=span, offset=offset, R=int(R), used=int(used), sizes=sizes,
                scan=scan)


def _slab_fill(rng, zmin, F, nby, slabs, zinit):
    """zMin/zMax host loop + SlabKernel + the fill with slab keys (NUM_SLABS > 1)."""
    zlo, zhi = C.c_float(), C.c_float()
    _lib.orc_zrange(_p(zmin), C.c_int(F), C.c_float(zinit[0]), C.c_float(zinit[1]), C.byref(zlo), C.byref(zhi))
    zlist = np.zeros(F, np.uint32)
    _lib.orc_slab_kernel(_p(zmin), C.c_int(F), C.c_int(slabs), zlo, zhi, _p(zlist))
    fill = lambda scan, k, v: _lib.orc_fill_slabs(_p(rng), _p(scan), _p(zlist), C.c_int(F), C.c_int(nby),
                                                  C.c_int(slabs), _p(k), _p(v))
    return fill, zlist, (zlo.value, zhi.value)


def grid_perspective(cc, faces, verts, nbx, nby, rows=None, slabs=1):
    """FrustumGrid::buildGrid, frustum_grid.h:210-366."""
    faces, verts = _i32(faces).reshape(-1), _f32(verts).reshape(-1)
    F = len(faces) // 3
    lo, hi = rows if rows is not None else (0, nby)
    rng = np.zeros(F * 4, np.int32)
    sizes = np.zeros(F, np.uint32)
    zmin = np.zeros(F, np.float32)
    _lib.orc_persp_ranges(_p(_f32(cc)), _p(faces), _p(verts), C.c_int(F), C.c_int(nbx), C.c_int(nby), C.c_int(lo),
                          C.c_int(hi), _p(rng), _p(sizes), _p(zmin))
    if slabs > 1:
        fill, zlist, zr = _slab_fill(rng, zmin, F, nby, slabs, (2.0, -2.0))
        g = _finish_grid(fill, sizes, F, nbx * nby * slabs, nbx * nby * slabs)
        g["zlist"], g["zrange"] = zlist, zr
    else:
        g = _finish_grid(lambda scan, k, v: _lib.orc_fill_2d(_p(rng), _p(scan), C.c_int(F), C.c_int(nby), _p(k), _p(v)),
                         sizes, F, nbx * nby, nbx * nby)
    g["rng"], g["zmin"] = rng, zmin
    return g


def _apply_window(sizes, window):
    """A shard of the build (multi-GPU, not in the reference): triangles outside [f0, f1) get no references."""
    if window is not None:
        sizes[:window[0]] = 0
        sizes[window[1]:] = 0


def grid_spherical(cc, faces, verts, lnbx, lnby, xM=PI_F, yM=PI_F, slabs=1, window=None):
    """FrustumGrid::buildSphericalGrid, frustum_grid.h:368-532."""
    faces, verts = _i32(faces).reshape(-1), _f32(verts).reshape(-1)
    F = len(faces) // 3
    rng = np.zeros(F * 4, np.int32)
    sizes = np.zeros(F, np.uint32)
    zmin = np.zeros(F, np.float32)
    _lib.orc_sph_ranges(_p(_f32(cc)), _p(faces), _p(verts), C.c_int(F), C.c_int(lnbx), C.c_int(lnby),
                        C.c_float(xM), C.c_float(yM), _p(rng), _p(sizes), _p(zmin))
    _apply_window(sizes, window)
    if slabs > 1:
        fill, zlist, zr = _slab_fill(rng, zmin, F, lnby, slabs, (9999.9, -9999.9))
        g = _finish_grid(fill, sizes, F, lnbx * lnby * slabs, lnbx * lnby * slabs)
        g["zlist"], g["zrange"] = zlist, zr
    else:
        g = _finish_grid(lambda scan, k, v: _lib.orc_fill_2d(_p(rng), _p(scan), C.c_int(F), C.c_int(lnby), _p(k), _p(v)),
                         sizes, F, lnbx * lnby, lnbx * lnby)
    g["rng"], g["zmin"] = rng, zmin
    return g


def uniform_setup(bbmin, bbmax, dims):
    g = np.zeros(12, np.float32)
    _lib.orc_uniform_setup(_p(_f32(bbmin)), _p(_f32(bbmax)), _p(_i32(dims)), _p(g))
    return g


def grid_uniform(faces, verts, bbmin, bbmax, dims, window=None):
    faces, verts = _i32(faces).reshape(-1), _f32(verts).reshape(-1)
    F = len(faces) // 3
    dims = _i32(dims)
    ug = uniform_setup(bbmin, bbmax, dims)
    rng = np.zeros(F * 6, np.int32)
    sizes = np.zeros(F, np.uint32)
    _lib.orc_uniform_ranges(_p(ug), _p(dims), _p(faces), _p(verts), C.c_int(F), _p(rng), _p(sizes))
    _apply_window(sizes, window)
    ncell = int(dims[0]) * int(dims[1]) * int(dims[2])
    g = _finish_grid(lambda scan, k, v: _lib.orc_fill_3d(_p(rng), _p(scan), C.c_int(F), _p(dims), _p(k), _p(v)),
                     sizes, F, ncell, ncell)
    g["ug"], g["dims"] = ug, dims
    return g


def trace_primary(cam, W, H, grid, verts, faces, rows=None, out=None, slabs=1):
    """rckernel_alpha, trace_kernel.cu:84-270."""
    nbx, nby = W // 8, H // 8
    lo, hi = rows if rows is not None else (0, nby)
    N = W * H
    o = out or dict(normal=np.zeros(3 * N, np.float32), t=np.zeros(N, np.float32), dir=np.zeros(3 * N, np.float32),
                    shadowed=np.zeros(N, np.int32), id=np.zeros(N, np.int32))
    if slabs > 1:
        _lib.orc_trace_primary_slabs(_p(cam.cc), _p(cam.tex), C.c_int(W), C.c_int(H), C.c_int(nbx), C.c_int(nby),
                                     C.c_int(lo), C.c_int(hi), C.c_int(slabs), _p(grid["vals"]), _p(grid["span"]),
                                     _p(grid["offset"]), _p(_f32(verts).reshape(-1)), _p(_i32(faces).reshape(-1)),
                                     _p(o["normal"]), _p(o["t"]), _p(o["dir"]), _p(o["shadowed"]), _p(o["id"]))
        o["mt_tests"], o["refs_staged"] = 0, 0
        return o
    cnt = np.zeros(2, np.uint64)
    _lib.orc_trace_primary(_p(cam.cc), _p(cam.tex), C.c_int(W), C.c_int(H), C.c_int(nbx), C.c_int(nby), C.c_int(lo),
                           C.c_int(hi), _p(grid["vals"]), _p(grid["span"]), _p(grid["offset"]),
                           _p(_f32(verts).reshape(-1)), _p(_i32(faces).reshape(-1)), _p(o["normal"]), _p(o["t"]),
                           _p(o["dir"]), _p(o["shadowed"]), _p(o["id"]), _p(cnt))
    o["mt_tests"], o["refs_staged"] = int(cnt[0]), int(cnt[1])
    return o


def map_rays(lcc, t, dirs, cam_pos, lnbx, lnby, p0, n, xM=PI_F, yM=PI_F):
    d_map = np.zeros(2 * n, np.uint32)
    _lib.orc_map_rays(_p(_f32(lcc)), _p(t), _p(dirs), _p(_f32(cam_pos)), C.c_float(xM), C.c_float(yM),
                      C.c_int(lnbx), C.c_int(lnby), C.c_int(p0), C.c_int(n), _p(d_map))
    return d_map


def process_rays(d_map, n, nkeys, cap):
    prefix = np.zeros(cap, np.uint32)
    nchunks = _lib.orc_process_rays(_p(d_map), C.c_int(n), C.c_uint(nkeys), _p(prefix), C.c_uint(cap))
    return prefix, int(nchunks)


def trace_shadow(lcc, lgrid, C_light, verts, faces, t, dirs, is_shadowed, d_map, prefix, cam_pos, nchunks,
                 launch_blocks, n, strict=True, slabs=1):
    cnt = np.zeros(2, np.uint64)
    _lib.orc_trace_shadow_slabs(_p(_f32(lcc)), _p(lgrid["vals"]), _p(_f32(verts).reshape(-1)),
                                _p(_i32(faces).reshape(-1)), _p(lgrid["span"]), _p(lgrid["offset"]), C.c_uint(C_light),
                                C.c_int(slabs), _p(t), _p(dirs), _p(is_shadowed), _p(d_map), _p(prefix),
                                _p(_f32(cam_pos)), C.c_uint(nchunks), C.c_uint(launch_blocks), C.c_int(n),
                                C.c_int(1 if strict else 0), _p(cnt))
    return int(cnt[0]), int(cnt[1])


def shade(cc, light_pos, img, normal, t, dirs, ids, cam_pos, mat_idx, mat_list, p0, n, spot=False, dump=None):
    mat_list = _f32(mat_list).reshape(-1)
    _lib.orc_shade(_p(_f32(cc)), _p(_f32(light_pos)), _p(img), _p(normal), _p(t), _p(dirs), _p(ids),
                   _p(_f32(cam_pos)), _p(_i32(mat_idx)), _p(mat_list), C.c_int(len(mat_list) // 6), C.c_int(p0),
                   C.c_int(n), C.c_int(1 if spot else 0), _p(dump))


def add_shadows(img, is_shadowed, p0, n):
    _lib.orc_add_shadows(_p(img), _p(is_shadowed), C.c_int(p0), C.c_int(n))


def shade_perlin(img, ids, W, p0, n):
    _lib.orc_shade_perlin(_p(img), _p(ids), C.c_int(W), C.c_int(p0), C.c_int(n))


def animate(verts, orig, size, offset, cr, sr):
    _lib.orc_animate(_p(verts), _p(orig), C.c_int(size), C.c_int(offset), C.c_float(cr), C.c_float(sr))


def write_ppm(path, img_hw3):
    a = np.ascontiguousarray(img_hw3, np.uint8)
    return _lib.orc_write_ppm(str(path).encode(), C.c_int(a.shape[1]), C.c_int(a.shape[0]), _p(a))


def parse_materials(path):
    n = _lib.orc_parse_materials(str(path).encode(), None, C.c_int(0))
    if n < 0:
        return None
    out = np.zeros(n * 6, np.float32)
    _lib.orc_parse_materials(str(path).encode(), _p(out), C.c_int(n))
    return out


def reflect_rays(cam_pos, t, dirs, ids, mat_idx, reflect, verts, faces, eps, p0, n, N):
    rays = np.zeros(6 * N, np.float32)
    active = np.zeros(N, np.int32)
    reflect = _f32(reflect)
    _lib.orc_reflect_rays(_p(_f32(cam_pos)), _p(t), _p(dirs), _p(ids), _p(_i32(mat_idx)), _p(reflect),
                          C.c_int(len(reflect)), _p(_f32(verts).reshape(-1)), _p(_i32(faces).reshape(-1)),
                          C.c_float(eps), C.c_int(p0), C.c_int(n), _p(rays), _p(active))
    return rays, active


def trace_dda(ugrid, verts, faces, rays, active, p0, n, N):
    hit_t = np.full(N, -1.0, np.float32)
    hit_id = np.full(N, -2, np.int32)
    cnt = np.zeros(3, np.uint64)
    _lib.orc_trace_dda(_p(ugrid["ug"]), _p(ugrid["dims"]), _p(ugrid["vals"]), _p(ugrid["span"]),
                       _p(ugrid["offset"]), _p(_f32(verts).reshape(-1)), _p(_i32(faces).reshape(-1)), _p(rays),
                       _p(active), C.c_int(p0), C.c_int(n), _p(hit_t), _p(hit_id), _p(cnt))
    return hit_t, hit_id, [int(x) for x in cnt]


def brute_nearest(verts, faces, rays, active, p0, n, N):
    hit_t = np.full(N, -1.0, np.float32)
    hit_id = np.full(N, -2, np.int32)
    faces = _i32(faces).reshape(-1)
    _lib.orc_brute_nearest(_p(_f32(verts).reshape(-1)), _p(faces), C.c_int(len(faces) // 3), _p(rays), _p(active),
                           C.c_int(p0), C.c_int(n), _p(hit_t), _p(hit_id))
    return hit_t, hit_id


def shade_reflect(cc, light_pos, img, normal, t, dirs, ids, cam_pos, mat_idx, mat_list, reflect, verts, faces, rays,
                  active, hit_t, hit_id, p0, n):
    mat_list = _f32(mat_list).reshape(-1)
    _lib.orc_shade_reflect(_p(_f32(cc)), _p(_f32(light_pos)), _p(img), _p(normal), _p(t), _p(dirs), _p(ids),
                           _p(_f32(cam_pos)), _p(_i32(mat_idx)), _p(mat_list), _p(_f32(reflect)),
                           C.c_int(len(mat_list) // 6), _p(_f32(verts).reshape(-1)), _p(_i32(faces).reshape(-1)),
                           _p(rays), _p(active), _p(hit_t), _p(hit_id), C.c_int(p0), C.c_int(n))


def set_strict_texture(on):
    """UGRT_FLAG_STRICT_TEXTURE for the ray set-ups that follow (a process-wide switch of the checker)."""
    _lib.orc_set_strict_texture(C.c_int(1 if on else 0))


def frame(scene, setup, W, H, rows=None, light_grid=(128, 128), all_chunks=False, shadows=True, reflect=False,
          uniform_dims=(64, 64, 32), frame_cnt=1, reflect_eps=1e-3, verts=None, slabs=1, strict_texture=False):
    """display(), main.cu:59-302, on the CPU.  Returns every intermediate array and, in
    r["times"], the seconds spent per stage (used by bench.py's cpu_baseline leg)."""
    import time as _time

    set_strict_texture(strict_texture)

    times = {}

    class _T:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t0 = _time.perf_counter()

        def __exit__(self, *a):
            times[self.name] = times.get(self.name, 0.0) + _time.perf_counter() - self.t0

    verts = _f32(scene["verts"] if verts is None else verts).reshape(-1)
    faces = _i32(scene["faces"]).reshape(-1)
    nbx, nby = W // 8, H // 8
    lo, hi = rows if rows is not None else (0, nby)
    p0, n, N = lo * 8 * W, (hi - lo) * 8 * W, W * H
    aspect = float(np.float32(W) / np.float32(H))
    cam = cam_from(setup.camera, setup.fovy, aspect)
    r = dict(cam=cam, p0=p0, n=n, times=times)
    with _T("build_perspective"):
        r["grid"] = grid_perspective(cam.cc, faces, verts, nbx, nby, (lo, hi), slabs=slabs)
    with _T("trace_primary"):
        r["primary"] = trace_primary(cam, W, H, r["grid"], verts, faces, (lo, hi), slabs=slabs)
    pr = r["primary"]
    cam_pos = cam.worldori[:3].copy()
    is_shadowed = pr["shadowed"].copy()
    cur_cc = cam.cc
    if shadows:
        lcam = cam_from(setup.light_camera, setup.fovy, aspect)
        r["lcam"] = lcam
        cur_cc = lcam.cc
        lx, ly = light_grid
        with _T("map_rays"):
            d_map = map_rays(lcam.cc, pr["t"], pr["dir"], cam_pos, lx, ly, p0, n)
        r["map_unsorted"] = d_map.copy()
        with _T("build_spherical"):
            r["lgrid"] = grid_spherical(lcam.cc, faces, verts, lx, ly, slabs=slabs)
        with _T("sort_rays"):
            prefix, nchunks = process_rays(d_map, n, lx * ly + 1, n // 64 + lx * ly + 2)
        r["map"], r["prefix"], r["nchunks"] = d_map, prefix, nchunks
        with _T("trace_shadow"):
            r["shadow_tests"] = trace_shadow(lcam.cc, r["lgrid"], lx * ly, verts, faces, pr["t"], pr["dir"],
                                             is_shadowed, d_map, prefix, cam_pos, nchunks, nbx * nby, n,
                                             strict=not all_chunks, slabs=slabs)
    r["is_shadowed"] = is_shadowed
    img = np.zeros(3 * N, np.uint8)
    ids = pr["id"].copy()
    if reflect:
        v3 = verts.reshape(-1, 3)
        with _T("reflect_gen"):
            rays, active = reflect_rays(cam_pos, pr["t"], pr["dir"], pr["id"], scene["matidx"], scene["reflect"],
                                        verts, faces, reflect_eps, p0, n, N)
        with _T("build_uniform"):
            r["ugrid"] = grid_uniform(faces, verts, v3.min(0), v3.max(0), uniform_dims)
        with _T("trace_dda"):
            hit_t, hit_id, cnt = trace_dda(r["ugrid"], verts, faces, rays, active, p0, n, N)
        r.update(rays=rays, active=active, hit_t=hit_t, hit_id=hit_id, dda_counters=cnt)
        with _T("shade"):
            shade_reflect(cur_cc, setup.shading_light, img, pr["normal"], pr["t"], pr["dir"], ids, cam_pos,
                          scene["matidx"], scene["mat_list"], scene["reflect"], verts, faces, rays, active, hit_t,
                          hit_id, p0, n)
    else:
        with _T("shade"):
            shade(cur_cc, setup.shading_light, img, pr["normal"], pr["t"], pr["dir"], ids, cam_pos, scene["matidx"],
                  scene["mat_list"], p0, n, spot=frame_cnt >= 2)
    r["image_unshadowed"] = img.copy()
    if shadows:
        add_shadows(img, is_shadowed, p0, n)
    r["image"], r["mat_ids"] = img, ids
    return r
