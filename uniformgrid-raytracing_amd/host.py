"""Host-side mirrors of the reference's Model / Camera classes and writePPM.

Nothing here touches the GPU; all of it runs inside libugrt.so's C++ host code.
"""
import ctypes as C

import numpy as np

from . import lib, check, CameraStruct, _f3, _P


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class Model:
    """class Model, scene.h:13-57: some_material(), load_model(), the flat host lists."""

    def __init__(self, frames=1):
        self._h = _P()
        check(lib.ugrt_scene_create(C.byref(self._h)))
        self.num_frames = frames

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ugrt_scene_destroy(self._h)
            self._h = None

    def some_material(self, path):
        check(lib.ugrt_scene_some_material(self._h, str(path).encode()))

    def load_model(self, path):
        check(lib.ugrt_scene_load_model(self._h, str(path).encode()))

    def save_cache(self, path):
        check(lib.ugrt_scene_save_cache(self._h, str(path).encode()))

    def load_cache(self, path):
        check(lib.ugrt_scene_load_cache(self._h, str(path).encode()))

    def tmp_model(self, directory, i):
        check(lib.ugrt_scene_load_frame(self._h, str(directory).encode(), int(i)))

    def _counts(self):
        nv, nf, nm = C.c_int(), C.c_int(), C.c_int()
        check(lib.ugrt_scene_counts(self._h, C.byref(nv), C.byref(nf), C.byref(nm)))
        return nv.value, nf.value, nm.value

    num_vertices = property(lambda s: s._counts()[0])
    num_faces = property(lambda s: s._counts()[1])
    num_materials = property(lambda s: s._counts()[2])

    @property
    def h_vertexlist(self):
        return _view(lib.ugrt_scene_vertexlist(self._h), self.num_vertices * 3, np.float32)

    @property
    def h_facelist(self):
        return _view(lib.ugrt_scene_facelist(self._h), self.num_faces * 3, np.int32)

    @property
    def h_materiallist_index(self):
        return _view(lib.ugrt_scene_materiallist_index(self._h), self.num_faces, np.int32)

    @property
    def h_materiallist(self):
        return _view(lib.ugrt_scene_materiallist(self._h), self.num_materials * 6, np.float32)

    @property
    def h_reflectlist(self):
        n = C.c_int()
        p = lib.ugrt_scene_reflectlist(self._h, C.byref(n))
        return _view(p, n.value, np.float32)

    def bounds(self):
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        check(lib.ugrt_scene_bounds(self._h, mn, mx))
        return np.array(mn[:], dtype=np.float32), np.array(mx[:], dtype=np.float32)


class Camera:
    """class Camera, camera.h:7-47 (GL replaced by libugrt's own float gluPerspective/gluLookAt)."""

    def __init__(self, fovy=45.0, aspect=1.0):
        self.fovy, self.aspect = float(fovy), float(aspect)
        self.c, self.l, self.u = (0, 0, 0), (0, 0, -1), (0, 1, 0)
        self.nearPlane, self.farPlane = 0.1, 100.0
        self.s = CameraStruct()

    def setCameraCenter(self, x, y, z):
        self.c = (x, y, z)

    def setCameraLookAt(self, x, y, z):
        self.l = (x, y, z)

    def setCameraUp(self, x, y, z):
        self.u = (x, y, z)

    def setNearFar(self, n, f):
        self.nearPlane, self.farPlane = n, f

    def adjustCameraAndPosition(self):
        """adjustCameraAndPosition + getGLMatrices + getFrustumProperties, camera.h:86-253."""
        check(lib.ugrt_camera_set(C.byref(self.s), _f3(self.c), _f3(self.l), _f3(self.u), self.nearPlane,
                                  self.farPlane, self.fovy, self.aspect))
        return self

    getGLMatrices = getFrustumProperties = lambda self: self

    def _arr(self, name):
        return np.ctypeslib.as_array(getattr(self.s, name)).astype(np.float32).copy()

    worldori = property(lambda s: s._arr("worldori"))
    modelview_matrix = property(lambda s: s._arr("modelview_matrix"))
    projection_matrix = property(lambda s: s._arr("projection_matrix"))
    mvp_matrix = property(lambda s: s._arr("mvp_matrix"))
    frustum_plane_eq = property(lambda s: s._arr("frustum_plane_eq"))
    frustumcorner = property(lambda s: s._arr("frustumcorner"))
    camcoords = property(lambda s: s._arr("camcoords"))

    def direction_table(self):
        t = (C.c_float * 100)()
        check(lib.ugrt_camera_direction_table(self.s.camcoords, t))
        return np.array(t[:], dtype=np.float32)


def write_ppm(path, rgb):
    """writePPM, per_app_funcs.h:39.  rgb: uint8 array [H, W, 3], row 0 = bottom of the view."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    check(lib.ugrt_write_ppm(str(path).encode(), w, h, a.ctypes.data_as(_P)))
