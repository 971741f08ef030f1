"""uniformgrid-raytracing_amd -- MI355X-native grid ray tracer (host bindings).

The product is ``libugrt.so`` (HIP kernels for gfx950 + C++ host code behind the
C-ABI of ``include/ugrt.h``).  This package is the thin Python host layer over
that ABI: ctypes prototypes, and classes that mirror the reference's host
classes for the render path (``Model`` scene.h:13, ``Camera`` camera.h:7,
``FrustumGrid`` frustum_grid.h:15, ``FrustumTracer`` frustum_tracer.h:14,
``DecisionData`` decision_data.h:7, ``Shader`` shader.h:14) plus ``display()``
(main.cu:59) as :func:`renderer.Renderer.display`.

PyTorch is used only as the device-memory / stream / ``torch.distributed``
plumbing: every device buffer is a torch tensor whose ``data_ptr()`` goes
through the C-ABI.  There is no CPU fallback: importing works without a GPU
(so that symbols can be checked), but every device entry point returns
``UGRT_ENODEV`` and raises :class:`UgrtError` when no HIP device is present, and
a missing ``libugrt.so`` raises at import.

The directory name contains a hyphen (it is the reference's name + ``_amd``);
import it with ``importlib.import_module("uniformgrid-raytracing_amd")`` or via
the ``ugrt`` alias module at the repository root.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UGRT_LIB: an instrumented build of the library (e.g. the AddressSanitizer build of the host translation unit
# that tests/test_sanitizers.py runs); with UGRT_HOST_ONLY=1 such a build may lack the device entry points
LIB_PATH = os.environ.get("UGRT_LIB") or os.path.join(_HERE, "libugrt.so")
_HOST_ONLY = os.environ.get("UGRT_HOST_ONLY", "") == "1"

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libugrt.so is missing at %s: build it with `make -C %s/csrc` "
        "(or __graft_entry__.build()); there is no Python/CPU fallback" % (LIB_PATH, _HERE)
    )

# PyTorch-ROCm bundles its own HIP runtime with the same SONAME as /opt/rocm's (libamdhip64.so.7).
# Device pointers are only valid inside ONE runtime, so torch's copy has to be the one in the
# process: load it before libugrt.so resolves its own dependency.
try:
    if _HOST_ONLY:
        raise ImportError("host-only")
    import torch  # noqa: F401
except ImportError:  # host-only use (loader, camera, PPM) works without torch
    torch = None

lib = C.CDLL(LIB_PATH)

UGRT_OK, UGRT_EINVAL, UGRT_ENODEV, UGRT_EHIP, UGRT_EIO, UGRT_ENOMEM, UGRT_EOVERFLOW = range(7)
FLAG_SHADOW_ALL_CHUNKS = 1
FLAG_COUNT_WORK = 2
FLAG_STATIC_GEOMETRY = 4
FLAG_STRICT_TEXTURE = 8
CHUNKS_ON_DEVICE = 0xFFFFFFFF
GRID_PERSPECTIVE, GRID_SPHERICAL, GRID_UNIFORM = 0, 1, 2
STAGES = [
    "build_count", "build_scan", "build_fill", "build_sort", "build_bounds", "trace_primary", "map_rays",
    "sort_rays", "trace_shadow", "shade", "reflect_gen", "trace_dda", "animate", "worklist", "shadow_cull", "shadow_prep",
]


class UgrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ugrt error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("tile", C.c_int), ("slabs", C.c_int),
        ("light_nbx", C.c_int), ("light_nby", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
        ("flags", C.c_uint), ("uniform_dims", C.c_int * 3),
    ]


class CameraStruct(C.Structure):
    _fields_ = [
        ("worldori", C.c_float * 4), ("modelview_matrix", C.c_float * 16), ("projection_matrix", C.c_float * 16),
        ("mvp_matrix", C.c_float * 16), ("frustum_plane_eq", (C.c_float * 6) * 6),
        ("frustumcorner", (C.c_float * 3) * 8), ("camcoords", C.c_float * 64),
    ]


class GridInfo(C.Structure):
    _fields_ = [
        ("d_triangle_value_list", C.c_void_p), ("d_triangle_key_list", C.c_void_p), ("d_span", C.c_void_p),
        ("d_offset", C.c_void_p), ("total_refs", C.c_uint), ("num_cells", C.c_uint), ("cells_used", C.c_uint),
    ]


class SlabInfo(C.Structure):
    _fields_ = [("slabs", C.c_int), ("d_proj_coord_z", C.c_void_p), ("z_min", C.c_float), ("z_max", C.c_float)]


_P = C.c_void_p
_F3 = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/ugrt.h declares
PROTOTYPES = {
    "ugrt_version": (C.c_int, []),
    "ugrt_last_error": (C.c_char_p, []),
    "ugrt_scene_create": (C.c_int, [C.POINTER(_P)]),
    "ugrt_scene_some_material": (C.c_int, [_P, C.c_char_p]),
    "ugrt_scene_load_model": (C.c_int, [_P, C.c_char_p]),
    "ugrt_scene_load_frame": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "ugrt_scene_counts": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ugrt_scene_vertexlist": (_P, [_P]),
    "ugrt_scene_facelist": (_P, [_P]),
    "ugrt_scene_materiallist_index": (_P, [_P]),
    "ugrt_scene_materiallist": (_P, [_P]),
    "ugrt_scene_reflectlist": (_P, [_P, C.POINTER(C.c_int)]),
    "ugrt_scene_save_cache": (C.c_int, [_P, C.c_char_p]),
    "ugrt_scene_load_cache": (C.c_int, [_P, C.c_char_p]),
    "ugrt_scene_bounds": (C.c_int, [_P, _F3, _F3]),
    "ugrt_scene_destroy": (None, [_P]),
    "ugrt_camera_set": (C.c_int, [C.POINTER(CameraStruct), _F3, _F3, _F3, C.c_float, C.c_float, C.c_float, C.c_float]),
    "ugrt_camera_direction_table": (C.c_int, [_F3, _F3]),
    "ugrt_write_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _P]),
    "ugrt_rot_cos_sin": (C.c_int, [C.c_float, _F3, _F3]),
    "ugrt_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(Config)]),
    "ugrt_ctx_set_stream": (C.c_int, [_P, _P]),
    "ugrt_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "ugrt_ctx_get_state": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_longlong)]),
    "ugrt_ctx_synchronize": (C.c_int, [_P]),
    "ugrt_ctx_destroy": (None, [_P]),
    "ugrt_upload_camera": (C.c_int, [_P, _F3]),
    "ugrt_set_light_position": (C.c_int, [_P, _F3]),
    "ugrt_grid_build_perspective": (C.c_int, [_P, _P, _P, C.c_int]),
    "ugrt_grid_build_spherical": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, C.c_float]),
    "ugrt_grid_build_uniform": (C.c_int, [_P, _P, _P, C.c_int, _F3, _F3]),
    "ugrt_grid_build_batch_begin": (C.c_int, [_P]),
    "ugrt_grid_build_batch_end": (C.c_int, [_P]),
    "ugrt_grid_get_info": (C.c_int, [_P, C.c_int, C.POINTER(GridInfo)]),
    "ugrt_grid_get_slabs": (C.c_int, [_P, C.c_int, C.POINTER(SlabInfo)]),
    "ugrt_ctx_set_face_window": (C.c_int, [_P, C.c_int, C.c_int]),
    "ugrt_grid_merge_shards": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                         C.POINTER(C.c_uint)]),
    "ugrt_geometry_changed": (C.c_int, [_P]),
    "ugrt_sort_pairs": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_int]),
    "ugrt_trace_primary": (C.c_int, [_P] * 11),
    "ugrt_map_rays_to_light": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_float]),
    "ugrt_sort_rays": (C.c_int, [_P, _P, _P, C.c_uint, C.POINTER(C.c_uint)]),
    "ugrt_sort_rays_chunks": (C.c_int, [_P, C.POINTER(C.c_uint)]),
    "ugrt_trace_shadow": (C.c_int, [_P] * 12 + [C.c_uint]),
    "ugrt_shade_simple": (C.c_int, [_P] * 9 + [C.c_int]),
    "ugrt_shade_spotlight": (C.c_int, [_P] * 9 + [C.c_int, _P]),
    "ugrt_shade_add_shadows": (C.c_int, [_P, _P, _P]),
    "ugrt_shade_perlin": (C.c_int, [_P] * 6),
    "ugrt_reflect_rays": (C.c_int, [_P] * 7 + [C.c_int, _P, _P, C.c_float, _P, _P]),
    "ugrt_trace_dda": (C.c_int, [_P] * 10),
    "ugrt_shade_reflect": (C.c_int, [_P] * 10 + [C.c_int] + [_P] * 6),
    "ugrt_animate": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float]),
    "ugrt_prof_enable": (C.c_int, [_P, C.c_int]),
    "ugrt_prof_reset": (C.c_int, [_P]),
    "ugrt_prof_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ugrt_stats_get": (C.c_int, [_P, C.POINTER(C.c_ulonglong)]),
    "ugrt_stats_dda": (C.c_int, [_P, C.POINTER(C.c_ulonglong), C.c_int]),
    "ugrt_stats_dda_split": (C.c_int, [_P, C.POINTER(C.c_uint)]),
    "ugrt_stats_primary": (C.c_int, [_P, C.POINTER(C.c_ulonglong), C.c_int]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    if _HOST_ONLY and not hasattr(lib, _name):
        continue
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc):
    if rc != 0:
        raise UgrtError(rc, lib.ugrt_last_error().decode("utf-8", "replace"))


def _f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


from .host import Model, Camera, write_ppm  # noqa: E402
from .device import Context  # noqa: E402
from . import scenes  # noqa: E402
from .renderer import Renderer, FrameSetup, BandedRenderer  # noqa: E402
