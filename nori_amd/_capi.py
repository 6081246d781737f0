"""ctypes mirror of include/nori_hip.h (structs, enums, prototypes).

Only declarations live here; the product loads ``libnori_hip.so`` through
:func:`load_hip` and fails loudly if it is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# ---------------------------------------------------------------- enums
BSDF_DIFFUSE, BSDF_MIRROR, BSDF_DIELECTRIC, BSDF_MICROFACET = range(4)
BSDF_NAMES = {"diffuse": 0, "mirror": 1, "dielectric": 2, "microfacet": 3}
(INTEGRATOR_NORMALS, INTEGRATOR_AO, INTEGRATOR_SIMPLE, INTEGRATOR_WHITTED,
 INTEGRATOR_PATH_MATS, INTEGRATOR_PATH_EMS, INTEGRATOR_PATH_MIS) = range(7)
INTEGRATOR_NAMES = {"normals": 0, "ao": 1, "simple": 2, "whitted": 3,
                    "path_mats": 4, "path_ems": 5, "path_mis": 6}
RFILTER_NAMES = {"gaussian": 0, "mitchell": 1, "tent": 2, "box": 3}
WARP_NAMES = {"square": 0, "tent": 1, "disk": 2, "uniform_sphere": 3,
              "uniform_hemisphere": 4, "cosine_hemisphere": 5, "beckmann": 6}
SEED_PER_SAMPLE, SEED_NORI_BLOCK = 0, 1
MEASURE_UNKNOWN, MEASURE_SOLID_ANGLE, MEASURE_DISCRETE = 0, 1, 2
NO_HIT = 0xFFFFFFFF
TILE_SIZE = 16

STATUS = {0: "NORI_OK", -1: "NORI_ERR_INVALID_ARGUMENT", -2: "NORI_ERR_NO_DEVICE",
          -3: "NORI_ERR_OUT_OF_MEMORY", -4: "NORI_ERR_NOT_READY",
          -5: "NORI_ERR_UNSUPPORTED", -6: "NORI_ERR_INTERNAL"}


class NoriError(RuntimeError):
    """Python face of NoriException (include/nori/common.h:135-140)."""


# -------------------------------------------------------------- structs
class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", C.c_float * 3), ("alpha", C.c_float),
                ("int_ior", C.c_float), ("ext_ior", C.c_float), ("ks", C.c_float)]


class MeshDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
                ("bsdf", BsdfDesc), ("is_emitter", C.c_int32), ("radiance", C.c_float * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fov", C.c_float),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("to_world", C.c_float * 16)]


class RFilterDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("radius", C.c_float), ("stddev", C.c_float),
                ("B", C.c_float), ("C", C.c_float)]


class IntegratorDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", C.c_float * 3), ("energy", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("meshes", C.POINTER(MeshDesc)),
                ("camera", CameraDesc), ("rfilter", RFilterDesc),
                ("integrator", IntegratorDesc), ("sample_count", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
                ("tile_mod", C.c_uint32), ("tile_rem", C.c_uint32),
                ("seed_mode", C.c_int32), ("count_traversal", C.c_int32), ("time_kernels", C.c_int32),
                ("stream", C.c_void_p)]


class RenderStats(C.Structure):
    _fields_ = [("n_camera_samples", C.c_uint64), ("n_closest_rays", C.c_uint64),
                ("n_shadow_rays", C.c_uint64), ("n_node_tests", C.c_uint64),
                ("n_tri_tests", C.c_uint64), ("n_invalid", C.c_uint64),
                ("kernel_ms", C.c_float), ("n_workgroups", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("trace_ms", C.c_float), ("shade_ms", C.c_float), ("film_ms", C.c_float), ("n_trace_launches", C.c_uint32), ("engine", C.c_uint32),
                ("trace_cus", C.c_uint32), ("tail_ms", C.c_float), ("tail_cus", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AccelInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32),
                ("max_depth", C.c_uint32), ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("total_bytes", C.c_uint64), ("build_ms", C.c_float), ("sah_cost", C.c_float),
                ("node_children", C.c_uint32), ("node_records_32b", C.c_uint32), ("built_on_device", C.c_uint32), ("n_references", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


# numpy dtypes of the per-query records (nori_ray: 32 B, nori_intersection: 104 B)
import numpy as np  # noqa: E402

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("mint", "<f4"), ("maxt", "<f4")])
ITS_DTYPE = np.dtype([("p", "<f4", 3), ("t", "<f4"), ("uv", "<f4", 2),
                      ("sh_s", "<f4", 3), ("sh_t", "<f4", 3), ("sh_n", "<f4", 3),
                      ("geo_s", "<f4", 3), ("geo_t", "<f4", 3), ("geo_n", "<f4", 3),
                      ("mesh", "<u4"), ("tri", "<u4")])
assert RAY_DTYPE.itemsize == 32 and ITS_DTYPE.itemsize == 104

# --------------------------------------------------------- prototypes
_P = C.c_void_p
_F = C.POINTER(C.c_float)

HIP_ABI_VERSION = 7      # NORI_HIP_ABI_VERSION of the include/nori_hip.h these ctypes structs mirror

#: every symbol include/nori_hip.h declares -> (restype, argtypes)
HIP_PROTOTYPES = {
    "nori_hip_abi_version": (C.c_int, []),
    "nori_hip_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "nori_hip_destroy": (None, [_P]),
    "nori_hip_last_error": (C.c_char_p, [_P]),
    "nori_hip_upload_scene": (C.c_int, [_P, C.POINTER(SceneDesc)]),
    "nori_hip_build_accel": (C.c_int, [_P, C.c_int]),
    "nori_hip_accel_info": (C.c_int, [_P, C.POINTER(AccelInfo)]),
    "nori_hip_debug_excursions": (C.c_int, [_P, _P, C.c_int]),
    "nori_hip_set_option": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    "nori_hip_get_option": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_size_t]),
    "nori_hip_border_size": (C.c_int, [_P]),
    "nori_hip_intersect": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    "nori_hip_intersect_device": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _P]),
    "nori_hip_sample_rays": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "nori_hip_li": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
    "nori_hip_bsdf_sample": (C.c_int, [_P, C.POINTER(BsdfDesc), _P, _P, C.c_size_t, _P, _P, _P, _P]),
    "nori_hip_bsdf_eval": (C.c_int, [_P, C.POINTER(BsdfDesc), _P, _P, C.c_size_t, _P]),
    "nori_hip_bsdf_pdf": (C.c_int, [_P, C.POINTER(BsdfDesc), _P, _P, C.c_size_t, _P]),
    "nori_hip_warp": (C.c_int, [_P, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "nori_hip_warp_pdf": (C.c_int, [_P, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "nori_hip_pcg32_floats": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_uint32, _P]),
    "nori_hip_pcg32_floats_at": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_size_t, C.c_uint32, _P]),
    "nori_hip_splat": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "nori_hip_render": (C.c_int, [_P, C.POINTER(RenderParams), _P, C.POINTER(RenderStats)]),
    "nori_hip_render_host": (C.c_int, [_P, C.POINTER(RenderParams), _P, C.POINTER(RenderStats)]),
    "nori_hip_develop": (C.c_int, [_P, _P, _P, _P]),
    "nori_hip_block_acc_floats": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "nori_hip_render_block_rows": (C.c_int, [_P, C.POINTER(RenderParams), C.c_uint32, C.c_uint32, _P, C.POINTER(RenderStats)]),
    "nori_hip_resolve_blocks": (C.c_int, [_P, _P, _P, _P]),
    "nori_hip_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "nori_hip_group_destroy": (None, [_P]),
    "nori_hip_group_size": (C.c_int, [_P]),
    "nori_hip_group_ctx": (_P, [_P, C.c_int]),
    "nori_hip_group_last_error": (C.c_char_p, [_P]),
    "nori_hip_group_transport": (C.c_char_p, [_P]),
    "nori_hip_group_warning": (C.c_char_p, [_P]),
    "nori_hip_group_engines": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int]),
    "nori_hip_group_upload_scene": (C.c_int, [_P, C.POINTER(SceneDesc), C.c_int]),
    "nori_hip_group_render_host": (C.c_int, [_P, C.POINTER(RenderParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(RenderStats), C.POINTER(C.c_float)]),
}


def bind(lib, prototypes):
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None


def hip_library_path() -> str:
    return os.environ.get("NORI_HIP_LIBRARY", os.path.join(LIB_DIR, "libnori_hip.so"))


def load_hip():
    """Load libnori_hip.so (built by __graft_entry__.build()).  Raises if absent."""
    global _hip
    if _hip is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise NoriError(f"{path} not found: build it with `python __graft_entry__.py` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = bind(C.CDLL(path, mode=C.RTLD_GLOBAL), HIP_PROTOTYPES)
        # the structs above mirror include/nori_hip.h of ONE version: a library built from another would write past them
        if lib.nori_hip_abi_version() != HIP_ABI_VERSION:
            raise NoriError(f"{path} reports ABI version {lib.nori_hip_abi_version()}, these bindings are written for {HIP_ABI_VERSION} (include/nori_hip.h): rebuild")
        _hip = lib
    return _hip


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
