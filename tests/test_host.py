"""Host layer of the product (loader, material file, camera, PPM) against the checkers:
the REFERENCE's own obj_parser (oracle/_ref/ref_objdump, built from /root/reference) and
the oracle's restatement of camera.h / per_app_funcs.h / scene.h."""
import os
import subprocess

import numpy as np
import pytest


def ref_dump(O, obj_path, tmp_path):
    if not os.path.exists(O.REF_OBJDUMP):
        pytest.skip("oracle/_ref/ref_objdump not built (reference checkout absent)")
    out = str(tmp_path / "dump.bin")
    subprocess.run([O.REF_OBJDUMP, os.path.basename(obj_path), out], cwd=os.path.dirname(obj_path), check=True,
                   capture_output=True)
    raw = open(out, "rb").read()
    nv, nf, nm = np.frombuffer(raw, np.int32, 3)
    o = 12
    verts = np.frombuffer(raw, np.float32, nv * 3, o); o += nv * 12
    faces = np.frombuffer(raw, np.int32, nf * 3, o); o += nf * 12
    matidx = np.frombuffer(raw, np.int32, nf, o); o += nf * 4
    bbox = np.frombuffer(raw, np.float32, 6, o); o += 24
    mats = np.frombuffer(raw, np.float64, nm * 15, o).reshape(nm, 15)
    return dict(verts=verts, faces=faces, matidx=matidx, bbox=bbox, mats=mats)


@pytest.mark.parametrize("which", ["cornell", "hall", "crash"])
def test_loader_matches_reference_parser(ugrt, O, tmp_path, which):
    gen = {"cornell": lambda d: ugrt.scenes.cornell(d), "hall": lambda d: ugrt.scenes.hall(d, scale=0.05),
           "crash": lambda d: ugrt.scenes.crash(d, scale=0.01)}[which]
    s = gen(str(tmp_path))
    m = ugrt.Model()
    m.some_material(s["mat"])
    m.load_model(s["obj"])
    ref = ref_dump(O, s["obj"], tmp_path)
    assert m.num_faces == s["num_faces"] == len(ref["faces"]) // 3
    np.testing.assert_array_equal(m.h_vertexlist, ref["verts"])
    np.testing.assert_array_equal(m.h_facelist, ref["faces"])
    np.testing.assert_array_equal(m.h_materiallist_index, ref["matidx"])
    mn, mx = m.bounds()
    np.testing.assert_array_equal(np.concatenate([mn, mx]), ref["bbox"])
    np.testing.assert_array_equal(m.h_reflectlist, ref["mats"][:, 9].astype(np.float32))
    # the generator's own arrays are what went into the file
    np.testing.assert_array_equal(m.h_facelist, s["faces"].reshape(-1))
    np.testing.assert_array_equal(m.h_materiallist_index, s["matidx"])
    np.testing.assert_allclose(m.h_vertexlist, s["verts"].reshape(-1), atol=1e-6 * max(1.0, abs(s["verts"]).max()))


def test_loader_obj_grammar_quirks(ugrt, O, tmp_path):
    """negative indices, v/vt/vn forms, quads cut to 3 indices, prefix material lookup, comments."""
    (tmp_path / "q.mtl").write_text("newmtl red\nKd 1 0 0\nr 0.25\nnewmtl redder\nKd 0.5 0 0\nnewmtl blue\nKa 0 0 1\n")
    (tmp_path / "q.obj").write_text(
        "# comment\nmtllib q.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0 0\n"
        "usemtl blue\nf 1 2 3 4\nusemtl red\nf 1/1 2/1 3/1\nf 1//1 2//1 4//1\nusemtl redd\nf -4 -3 -2\n"
        "usemtl nosuch\nf 1/1/1 3/1/1 4/1/1\ng grp\ns off\no obj\n")
    m = ugrt.Model()
    m.load_model(str(tmp_path / "q.obj"))
    ref = ref_dump(O, str(tmp_path / "q.obj"), tmp_path)
    np.testing.assert_array_equal(m.h_facelist, ref["faces"])
    np.testing.assert_array_equal(m.h_materiallist_index, ref["matidx"])
    np.testing.assert_array_equal(m.h_vertexlist, ref["verts"])
    assert list(m.h_materiallist_index) == [2, 0, 0, 1, -1]
    assert list(m.h_facelist.reshape(-1, 3)[3]) == [0, 1, 2]
    np.testing.assert_array_equal(m.h_reflectlist, np.array([0.25, 0, 0], np.float32))


def test_loader_errors(ugrt, tmp_path):
    m = ugrt.Model()
    with pytest.raises(ugrt.UgrtError) as e:
        m.load_model(str(tmp_path / "missing.obj"))
    assert e.value.code == ugrt.UGRT_EIO
    with pytest.raises(ugrt.UgrtError):
        m.some_material(str(tmp_path / "missing.mat"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ugrt.UgrtError):
        m.load_model(str(tmp_path / "bad.obj"))


def test_material_file_matches_oracle(ugrt, O, tmp_path):
    s = ugrt.scenes.hall(str(tmp_path), scale=0.05)
    m = ugrt.Model()
    m.some_material(s["mat"])
    np.testing.assert_array_equal(m.h_materiallist, O.parse_materials(s["mat"]))
    np.testing.assert_array_equal(m.h_materiallist.reshape(-1, 6), s["mat_list"])
    assert m.num_materials == 6


CAMS = [
    dict(eye=(3, 15, 5), look=(13, 13, 3), up=(0, 0, 1), near=0.1, far=100.0),      # main.cu:87-90
    dict(eye=(14, 13, 8), look=(14, 13, 0.0), up=(0, 1, 0), near=0.1, far=100.0),    # main.cu:158-164
    dict(eye=(0, 20, 5), look=(27, 1, 3), up=(0, 0, 1), near=0.1, far=100.0),        # main.cu:82-85
    dict(eye=(278, 273, -800), look=(278, 273, 0), up=(0, 1, 0), near=1.0, far=2000.0),
]


@pytest.mark.parametrize("cam", CAMS)
@pytest.mark.parametrize("aspect", [1.0, 16.0 / 9.0])
def test_camera_matches_oracle(ugrt, O, cam, aspect):
    c = ugrt.renderer.make_camera(cam, 45.0, aspect)
    o = O.cam_from(cam, 45.0, aspect)
    np.testing.assert_array_equal(c.modelview_matrix, o.MV)
    np.testing.assert_array_equal(c.projection_matrix, o.P)
    np.testing.assert_array_equal(c.mvp_matrix, o.MVP)
    np.testing.assert_array_equal(c.frustum_plane_eq.reshape(-1), o.planes)
    np.testing.assert_array_equal(c.frustumcorner.reshape(-1), o.corners)
    np.testing.assert_array_equal(c.camcoords, o.cc)
    np.testing.assert_array_equal(c.direction_table(), o.tex)


def test_camera_geometry(ugrt):
    """MVP maps the four near corners to ndc (+-1, +-1, -1) in the order the kernels assume
    (c0=(+1,-1), c1=(-1,-1), c2=(-1,+1), c3=(+1,+1): SURVEY.md A0)."""
    c = ugrt.renderer.make_camera(CAMS[0], 45.0, 1.0)
    M = c.mvp_matrix.reshape(4, 4).T.astype(np.float64)
    want = [(1, -1), (-1, -1), (-1, 1), (1, 1)]
    for i in range(4):
        p = M @ np.append(c.frustumcorner[i].astype(np.float64), 1.0)
        p = p[:3] / p[3]
        np.testing.assert_allclose(p, [want[i][0], want[i][1], -1.0], atol=2e-3)
    # gluPerspective(45, 1, .1, 100)
    P = c.projection_matrix
    assert abs(P[5] - 1.0 / np.tan(np.radians(22.5))) < 1e-6 and P[11] == -1.0


def test_ppm_matches_oracle(ugrt, O, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(16, 24, 3), dtype=np.uint8)
    img[0, 0] = (0, 9, 10)
    img[0, 1] = (99, 100, 255)
    ugrt.write_ppm(tmp_path / "a.ppm", img)
    assert O.write_ppm(tmp_path / "b.ppm", img) == 0
    a, b = (tmp_path / "a.ppm").read_bytes(), (tmp_path / "b.ppm").read_bytes()
    assert a == b
    assert a.startswith(b"P3\n24 16\n255\n\n0 9 10 99 100 255 ")
    with pytest.raises(ugrt.UgrtError):
        ugrt.write_ppm(tmp_path / "nodir" / "c.ppm", img)


def test_scene_cache_roundtrip(ugrt, tmp_path):
    s = ugrt.scenes.crash(str(tmp_path), scale=0.02)
    m = ugrt.Model()
    m.some_material(s["mat"])
    m.load_model(s["obj"])
    m.save_cache(tmp_path / "scene.bin")
    m2 = ugrt.Model()
    m2.load_cache(tmp_path / "scene.bin")
    for name in ("h_vertexlist", "h_facelist", "h_materiallist_index", "h_materiallist", "h_reflectlist"):
        np.testing.assert_array_equal(getattr(m, name), getattr(m2, name))
    assert m2.num_materials == m.num_materials == 6
    np.testing.assert_array_equal(np.concatenate(m.bounds()), np.concatenate(m2.bounds()))
    # truncated and foreign files are rejected
    raw = (tmp_path / "scene.bin").read_bytes()
    (tmp_path / "cut.bin").write_bytes(raw[: len(raw) // 2])
    (tmp_path / "bad.bin").write_bytes(b"NOTACACHE" + raw[9:])
    for f in ("cut.bin", "bad.bin", "missing.bin"):
        with pytest.raises(ugrt.UgrtError) as e:
            ugrt.Model().load_cache(tmp_path / f)
        assert e.value.code == ugrt.UGRT_EIO


def test_dynamic_directory_frames(ugrt, O, tmp_path):
    """Model::tmp_model (scene.h:70): <dir>/f_<i>.obj replaces the vertices, faces stay."""
    s = ugrt.scenes.hall(str(tmp_path), scale=0.05)
    os.rename(s["obj"], tmp_path / "f_0.obj")
    txt = (tmp_path / "f_0.obj").read_text().splitlines()
    moved = [("v %.6f %.6f %.6f" % tuple(float(x) + 0.25 for x in l.split()[1:])) if l.startswith("v ") else l for l in txt]
    (tmp_path / "f_1.obj").write_text("\n".join(moved) + "\n")
    m = ugrt.Model(frames=2)
    m.load_model(tmp_path / "f_0.obj")
    v0, f0 = m.h_vertexlist.copy(), m.h_facelist.copy()
    m.tmp_model(tmp_path, 1)
    np.testing.assert_allclose(m.h_vertexlist, v0 + np.float32(0.25), atol=1e-5)
    np.testing.assert_array_equal(m.h_facelist, f0)
    with pytest.raises(ugrt.UgrtError):
        m.tmp_model(tmp_path, 7)
