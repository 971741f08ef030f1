"""Procedural scenes for the BASELINE configs (no .obj ships with the reference).

Every generator writes a real Wavefront ``.obj`` + ``.mtl`` (so the loader path
is exercised) and the reference's positional material file
(``Model::some_material``, scene.h:370-439: per material 3 tokens, 3 floats Ka,
1 token, 3 floats Kd, 11 tokens, 1 texture token or ``NA``).  Seed 20100502
(SURVEY.md section 8d).  Coordinates are written with 6 decimals; both loaders
parse the same text, so the float32 vertex arrays are identical.

  cornell()  -- the public Cornell box data, 16 quads pre-triangulated = 32 tris
  hall()     -- "Sibenik"-class stand-in: room [0,28]x[0,26]x[0,9], 24 columns,
                stepped dais, vaulted ceiling; ~80 k tris, 6 materials
  crash()    -- "crashing"-class stand-in: the room + table + 20 chairs
                (200 000 tris) + icosphere level 7 (327 680) + debris shards
                = 1 000 000 tris; sphere + shards are the animated sub-range
"""
import os

import numpy as np

SEED = 20100502

# main.cu:87-90, :158-164, per_frame_funcs.h:8-10
REF_CAMERA = dict(eye=(3, 15, 5), look=(13, 13, 3), up=(0, 0, 1), near=0.1, far=100.0)
REF_LIGHT_CAMERA = dict(eye=(14, 13, 8), look=(14, 13, 0.0), up=(0, 1, 0), near=0.1, far=100.0)
REF_SHADING_LIGHT = (10.0, 12.0, 6.0)


class Mesh:
    def __init__(self):
        self.v, self.f, self.m = [], [], []
        self.nv = 0

    def add(self, verts, faces, mat):
        verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
        faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
        self.v.append(verts)
        self.f.append(faces + self.nv)
        self.m.append(np.full(len(faces), mat, dtype=np.int32))
        self.nv += len(verts)

    def ntris(self):
        return int(sum(len(f) for f in self.f))

    def arrays(self):
        return (np.concatenate(self.v), np.concatenate(self.f).astype(np.int32), np.concatenate(self.m))


def grid_quad(p0, du, dv, nu, nv):
    """Tessellated parallelogram p0 + s*du + t*dv, nu x nv quads -> 2*nu*nv triangles."""
    p0, du, dv = (np.asarray(a, dtype=np.float64) for a in (p0, du, dv))
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    verts = p0 + S[..., None] * du + T[..., None] * dv
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return verts.reshape(-1, 3), faces


def add_box(mesh, lo, hi, n, mat, faces="xyzXYZ"):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    quads = {
        "x": (lo, ey, ez), "X": (lo + ex, ey, ez), "y": (lo, ex, ez), "Y": (lo + ey, ex, ez),
        "z": (lo, ex, ey), "Z": (lo + ez, ex, ey),
    }
    for k in faces:
        mesh.add(*grid_quad(*quads[k], n, n), mat)


def add_cylinder(mesh, cx, cy, z0, z1, r, nseg, nstack, mat, flutes=0):
    th = np.linspace(0.0, 2 * np.pi, nseg + 1)
    z = np.linspace(z0, z1, nstack + 1)
    TH, Z = np.meshgrid(th, z, indexing="ij")
    rr = r * (1.0 + (0.06 * np.cos(flutes * TH) if flutes else 0.0))
    verts = np.stack([cx + rr * np.cos(TH), cy + rr * np.sin(TH), Z], -1)
    idx = np.arange((nseg + 1) * (nstack + 1)).reshape(nseg + 1, nstack + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    mesh.add(verts.reshape(-1, 3), faces, mat)


def add_vault(mesh, x0, x1, y0, y1, zbase, rise, nu, nv, mat):
    """Barrel vault along x over [y0,y1]."""
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, np.pi, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    verts = np.stack([x0 + S * (x1 - x0), (y0 + y1) / 2 - np.cos(T) * (y1 - y0) / 2, zbase + rise * np.sin(T)], -1)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    mesh.add(verts.reshape(-1, 3), faces, mat)


def icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        ab, bc, ca = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                            np.stack([ab, bc, ca], 1)])
    return v, f


def add_room(mesh, n=40):
    """Room [0,28]x[0,26]x[0,9], walls tessellated n x n.  Materials: 0 floor, 1 walls, 2 ceiling.

    The wall x = 0 (behind the reference camera at (3,15,5)) is left out, like a stage set: the
    reference's primary intersection takes |t| (trace_kernel.cu:35), so a wall
    BEHIND the eye would win every pixel as a ghost hit and hide the scene.
    """
    add_box(mesh, (0, 0, 0), (28, 26, 9), n, 1, faces="XyY")
    add_box(mesh, (0, 0, 0), (28, 26, 9), n, 0, faces="z")
    add_box(mesh, (0, 0, 0), (28, 26, 9), n, 2, faces="Z")


MATERIALS_ROOM = [
    # name, Ka, Kd, reflect
    ("m0_floor", (0.2, 0.2, 0.2), (0.62, 0.58, 0.52), 0.5),
    ("m1_wall", (0.2, 0.2, 0.2), (0.80, 0.78, 0.70), 0.0),
    ("m2_ceiling", (0.2, 0.2, 0.2), (0.85, 0.85, 0.88), 0.0),
    ("m3_column", (0.2, 0.2, 0.2), (0.70, 0.66, 0.60), 0.0),
    ("m4_wood", (0.2, 0.2, 0.2), (0.55, 0.35, 0.20), 0.5),
    ("m5_object", (0.2, 0.2, 0.2), (0.80, 0.25, 0.20), 0.0),
]

MATERIALS_CORNELL = [
    ("c0_white", (0.2, 0.2, 0.2), (0.76, 0.75, 0.50), 0.0),
    ("c1_light", (0.2, 0.2, 0.2), (1.00, 1.00, 1.00), 0.0),
    ("c2_green", (0.2, 0.2, 0.2), (0.15, 0.48, 0.09), 0.0),
    ("c3_red", (0.2, 0.2, 0.2), (0.63, 0.06, 0.04), 0.0),
]

CORNELL_QUADS = [
    # (material, 4 vertices)  -- the public Cornell box data
    (0, [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)]),                 # floor
    (1, [(343, 548.8, 227), (343, 548.8, 332), (213, 548.8, 332), (213, 548.8, 227)]),  # light
    (0, [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)]),      # ceiling
    (0, [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)]),    # back wall
    (2, [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)]),                  # right wall
    (3, [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)]),      # left wall
    (0, [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)]),            # short block
    (0, [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)]),
    (0, [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)]),
    (0, [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)]),
    (0, [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]),
    (0, [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)]),          # tall block
    (0, [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)]),
    (0, [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)]),
    (0, [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)]),
    (0, [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]),
]


def write_scene(outdir, name, verts, faces, matidx, materials):
    """Write <name>.obj, <name>.mtl, <name>.mat; returns their paths."""
    os.makedirs(outdir, exist_ok=True)
    obj, mtl, mat = (os.path.join(outdir, name + e) for e in (".obj", ".mtl", ".mat"))
    with open(mtl, "w") as fp:
        for nm, ka, kd, refl in materials:
            fp.write("newmtl %s\nKa %.6f %.6f %.6f\nKd %.6f %.6f %.6f\nKs 1.000000 1.000000 1.000000\nNs 0\n"
                     "d 1\nr %.6f\nillum 2\n\n" % ((nm,) + tuple(ka) + tuple(kd) + (refl,)))
    with open(mat, "w") as fp:
        for nm, ka, kd, refl in materials:
            fp.write("newmtl %s Ka %.6f %.6f %.6f Kd %.6f %.6f %.6f Ks 1 1 1 Ns 0 d 1 r %.6f map NA\n"
                     % ((nm,) + tuple(ka) + tuple(kd) + (refl,)))
    with open(obj, "w") as fp:
        fp.write("# generated by uniformgrid-raytracing_amd.scenes (seed %d)\nmtllib %s.mtl\n" % (SEED, name))
        fp.write("".join("v %.6f %.6f %.6f\n" % tuple(r) for r in verts))
        # usemtl only when the material changes
        change = np.flatnonzero(np.diff(matidx, prepend=-999))
        bounds = list(change) + [len(faces)]
        f1 = faces + 1
        for a, b in zip(bounds[:-1], bounds[1:]):
            fp.write("usemtl %s\n" % materials[int(matidx[a])][0])
            fp.write("".join("f %d %d %d\n" % tuple(r) for r in f1[a:b]))
    return obj, mtl, mat


def _finish(outdir, name, mesh, materials, extra):
    verts, faces, matidx = mesh.arrays()
    info = dict(name=name, num_faces=len(faces), num_vertices=len(verts), materials=materials)
    info.update(extra)
    if outdir is not None:
        info["obj"], info["mtl"], info["mat"] = write_scene(outdir, name, verts, faces, matidx, materials)
    info["verts"] = verts.astype(np.float32)  # generation values; the loaders parse the 6-decimal text
    info["faces"] = faces
    info["matidx"] = matidx
    info["mat_list"] = np.array([list(ka) + list(kd) for _, ka, kd, _ in materials], dtype=np.float32)
    info["reflect"] = np.array([r for _, _, _, r in materials], dtype=np.float32)
    return info


def cornell(outdir=None):
    mesh = Mesh()
    for m, q in CORNELL_QUADS:
        mesh.add(q, [[0, 1, 2], [0, 2, 3]], m)
    mn, mx = np.array([0, 0, 0.0]), np.array([556, 548.8, 559.2])
    c = (mn + mx) / 2
    cams = {
        # main.cu:114-118 (the reference's commented Cornell camera) and a view from outside the box
        "A": dict(eye=tuple(c), look=(c[0], c[1], mx[2]), up=(0, 1, 0), near=1.0, far=650.0),
        "B": dict(eye=(278, 273, -800), look=(278, 273, 0), up=(0, 1, 0), near=1.0, far=2000.0),
    }
    light_cam = dict(eye=(278, 540, 279.5), look=(278, 0, 279.5), up=(0, 0, 1), near=1.0, far=2000.0)
    return _finish(outdir, "cornell", mesh, MATERIALS_CORNELL,
                   dict(cameras=cams, light_camera=light_cam, shading_light=(278.0, 500.0, 279.5)))


def _hall_static(mesh, wall_n=40, col_seg=32, col_stack=26, vault=(64, 160)):
    add_room(mesh, wall_n)
    k = 0
    for ix in range(12):
        for iy in range(2):  # two rows, the centre aisle stays open for the reference camera
            cx, cy = 3.0 + ix * 2.1, 5.0 + iy * 16.0
            add_cylinder(mesh, cx, cy, 0.0, 7.5, 0.45, col_seg, col_stack, 3, flutes=8)
            k += 1
    for s in range(3):
        add_box(mesh, (20 - s * 0.8, 8 + s * 0.8, s * 0.35), (27 + 0.0, 18 - s * 0.8, (s + 1) * 0.35), 10, 4,
                faces="xyYZX")
    add_vault(mesh, 0.5, 27.5, 1.0, 25.0, 7.5, 1.4, vault[0], vault[1], 2)


def hall(outdir=None, scale=1.0):
    """~80 000 triangles at scale 1 (scale < 1 shrinks the tessellation for CPU-sized tests)."""
    mesh = Mesh()
    s = max(0.05, scale) ** 0.5
    _hall_static(mesh, wall_n=max(2, int(40 * s)), col_seg=max(6, int(32 * s)), col_stack=max(2, int(26 * s)),
                 vault=(max(4, int(64 * s)), max(8, int(160 * s))))
    return _finish(outdir, "hall%dk" % round(mesh.ntris() / 1000), mesh, MATERIALS_ROOM,
                   dict(cameras={"ref": REF_CAMERA}, light_camera=REF_LIGHT_CAMERA,
                        shading_light=REF_SHADING_LIGHT))


def crash(outdir=None, scale=1.0):
    """1 000 000 triangles at scale 1: 200 000 static + icosphere(7) + shards (animated sub-range).

    The animated object lives in the reference's "orig" space (centre (12,11,4.5)),
    which copy_data_transform (transformation_kernel.cu:10-16) maps to (14.5,13,4).
    """
    rng = np.random.default_rng(SEED)
    mesh = Mesh()
    s = max(0.02, scale) ** 0.5
    add_room(mesh, max(2, int(40 * s)))
    n_box = max(1, int(12 * s))
    add_box(mesh, (10, 9, 1.4), (19, 17, 1.6), max(2, int(40 * s)), 4)  # table top
    for lx, ly in ((10.2, 9.2), (18.4, 9.2), (10.2, 16.4), (18.4, 16.4)):
        add_box(mesh, (lx, ly, 0), (lx + 0.4, ly + 0.4, 1.4), n_box, 4, faces="xXyY")
    for i in range(20):
        ang = 2 * np.pi * i / 20
        cx, cy = 14.5 + 7.0 * np.cos(ang), 13.0 + 6.0 * np.sin(ang)
        add_box(mesh, (cx - 0.35, cy - 0.35, 0.75), (cx + 0.35, cy + 0.35, 0.85), n_box, 4)
        add_box(mesh, (cx - 0.35, cy + 0.27, 0.85), (cx + 0.35, cy + 0.35, 1.8), n_box, 4)
        for dx, dy in ((-0.33, -0.33), (0.25, -0.33), (-0.33, 0.25), (0.25, 0.25)):
            add_box(mesh, (cx + dx, cy + dy, 0), (cx + dx + 0.08, cy + dy + 0.08, 0.75), max(1, n_box // 2), 4,
                    faces="xXyY")
    static_target = int(round(200000 * scale / 2)) * 2
    left = static_target - mesh.ntris()
    if left > 0:  # a rug that brings the static part to the exact count: near-square quads, not slivers
        q = left // 2
        nu = max(1, int((q * 11.0 / 9.0) ** 0.5))
        nv = max(1, q // nu)
        mesh.add(*grid_quad((9, 8.5, 0.01), (11, 0, 0), (0, 9, 0), nu, nv), 4)
        rest = q - nu * nv
        if rest > 0:  # fringe: `rest` small quads along one edge
            mesh.add(*grid_quad((9, 8.4, 0.01), (11.0 * rest / max(nu, rest), 0, 0), (0, 0.1, 0), rest, 1), 4)
    n_static_faces, n_static_verts = mesh.ntris(), mesh.nv
    level = 7 if scale >= 1.0 else max(1, int(round(7 + np.log(max(scale, 1e-3)) / np.log(4))))
    sv, sf = icosphere(level)
    mesh.add(sv * (2.0 * 12 / 9) + np.array([12, 11, 4.5]), sf, 5)
    total_target = int(round(1000000 * scale))
    nsh = max(0, total_target - mesh.ntris())
    if nsh:
        ctr = rng.normal(0.0, 2.0 * 12 / 9, size=(nsh, 3)) + np.array([12, 11, 4.5])
        ctr[:, 0] = np.clip(ctr[:, 0], 4.5, 19.5)
        ctr[:, 1] = np.clip(ctr[:, 1], 3.5, 18.5)
        ctr[:, 2] = np.clip(ctr[:, 2], 0.6, 8.4)
        edge = rng.uniform(0.02, 0.1, size=(nsh, 1, 1)) * 12 / 9
        tri = rng.normal(size=(nsh, 3, 3))
        tri /= np.linalg.norm(tri, axis=2, keepdims=True)
        verts = ctr[:, None, :] + tri * edge
        mesh.add(verts.reshape(-1, 3), np.arange(nsh * 3).reshape(-1, 3), 5)
    return _finish(outdir, "crash%dk" % round(mesh.ntris() / 1000), mesh, MATERIALS_ROOM,
                   dict(cameras={"ref": REF_CAMERA}, light_camera=REF_LIGHT_CAMERA,
                        shading_light=REF_SHADING_LIGHT, animated_offset=n_static_verts,
                        animated_size=mesh.nv - n_static_verts, static_faces=n_static_faces))
