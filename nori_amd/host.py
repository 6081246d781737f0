"""Python face of libnori_host.so (include/nori_host.h): load unmodified Nori
XML scene / test files through the C++ host and get them as `Scene` objects.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _capi as capi
from ._capi import NoriError
from .scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene

DEFER_TESTS, QUIET = 1, 2
_P = C.c_void_p


class TestInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("significance_level", C.c_float), ("sample_count", C.c_int32),
                ("test_count", C.c_int32), ("resolution", C.c_int32), ("min_exp_frequency", C.c_int32),
                ("n_angles", C.c_uint32), ("n_references", C.c_uint32), ("n_bsdfs", C.c_uint32),
                ("n_scenes", C.c_uint32), ("angles", C.POINTER(C.c_float)), ("references", C.POINTER(C.c_float))]


HOST_PROTOTYPES = {
    "nori_host_load_xml": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "nori_host_free": (None, [_P]),
    "nori_host_last_error": (C.c_char_p, []),
    "nori_host_root_type": (C.c_int, [_P]),
    "nori_host_root_string": (C.c_char_p, [_P]),
    "nori_host_scene_desc": (C.c_int, [_P, C.POINTER(capi.SceneDesc)]),
    "nori_host_test_info_get": (C.c_int, [_P, C.POINTER(TestInfo)]),
    "nori_host_test_bsdf": (C.c_int, [_P, C.c_uint32, C.POINTER(capi.BsdfDesc)]),
    "nori_host_test_scene_desc": (C.c_int, [_P, C.c_uint32, C.POINTER(capi.SceneDesc)]),
    "nori_host_test_run": (C.c_int, [_P]),
    "nori_host_render": (C.c_int, [_P, _P, C.POINTER(capi.RenderStats)]),
    "nori_host_save_images": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "nori_host_load_exr": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nori_host_free_buffer": (None, [_P]),
}

_host = None


def load_host():
    global _host
    if _host is None:
        path = os.environ.get("NORI_HOST_LIBRARY", os.path.join(capi.LIB_DIR, "libnori_host.so"))
        if not os.path.exists(path):
            raise NoriError(f"{path} not found: run `python __graft_entry__.py`")
        capi.load_hip()   # dependency, same directory
        _host = capi.bind(C.CDLL(path, mode=C.RTLD_GLOBAL), HOST_PROTOTYPES)
    return _host


_BSDF_TYPES = {v: k for k, v in capi.BSDF_NAMES.items()}
_INTEGRATORS = {v: k for k, v in capi.INTEGRATOR_NAMES.items()}
_RFILTERS = {v: k for k, v in capi.RFILTER_NAMES.items()}


def bsdf_from_desc(d: capi.BsdfDesc) -> Bsdf:
    return Bsdf(_BSDF_TYPES[d.type], tuple(d.albedo), d.alpha, d.int_ior, d.ext_ior)


def scene_from_desc(d: capi.SceneDesc) -> Scene:
    """Deep-copies a borrowed nori_scene_desc into numpy-backed Python objects."""
    meshes = []
    for i in range(d.n_meshes):
        m = d.meshes[i]
        nv, nf = m.n_vertices, m.n_triangles
        pos = np.ctypeslib.as_array(m.positions, (nv, 3)).copy() if nv else np.zeros((0, 3), np.float32)
        idx = np.ctypeslib.as_array(m.indices, (nf, 3)).copy() if nf else np.zeros((0, 3), np.uint32)
        nrm = np.ctypeslib.as_array(m.normals, (nv, 3)).copy() if m.normals else None
        uv = np.ctypeslib.as_array(m.texcoords, (nv, 2)).copy() if m.texcoords else None
        b = bsdf_from_desc(m.bsdf)
        if b.type != "microfacet":
            # keep reference defaults for fields the plugin does not own
            b = Bsdf(b.type, b.albedo if b.type == "diffuse" else (0.5, 0.5, 0.5),
                     0.1, m.bsdf.int_ior if b.type == "dielectric" else 1.5046,
                     m.bsdf.ext_ior if b.type == "dielectric" else 1.000277)
        meshes.append(Mesh(pos, idx, nrm, uv, b, tuple(m.radiance) if m.is_emitter else None, f"mesh{i}"))
    c = d.camera
    cam = Camera(c.width, c.height, c.fov, c.near_clip, c.far_clip, np.array(list(c.to_world), np.float32).reshape(4, 4))
    rf = RFilter(_RFILTERS[d.rfilter.type], d.rfilter.radius, d.rfilter.stddev, d.rfilter.B, d.rfilter.C)
    it = Integrator(_INTEGRATORS[d.integrator.type], tuple(d.integrator.position), tuple(d.integrator.energy))
    return Scene(meshes, cam, rf, it, d.sample_count)


@dataclass
class TestFile:
    """Contents of a <test type="ttest|chi2test"> file."""
    kind: str
    significance_level: float
    sample_count: int
    angles: List[float] = field(default_factory=list)
    references: List[float] = field(default_factory=list)
    bsdfs: List[Bsdf] = field(default_factory=list)
    scenes: List[Scene] = field(default_factory=list)
    test_count: int = 5
    resolution: int = 10
    min_exp_frequency: int = 5


class HostRoot:
    """Owner of a parsed XML root (NoriObject graph held by the C++ host)."""

    def __init__(self, path: str, defer_tests: bool = True, quiet: bool = True):
        self._lib = load_host()
        h = _P()
        flags = (DEFER_TESTS if defer_tests else 0) | (QUIET if quiet else 0)
        rc = self._lib.nori_host_load_xml(os.fsencode(path), flags, C.byref(h))
        if rc != 0:
            raise NoriError(self._lib.nori_host_last_error().decode())
        self._h = h
        self.path = path

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nori_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def class_type(self) -> int:
        return self._lib.nori_host_root_type(self._h)

    def __str__(self):
        return self._lib.nori_host_root_string(self._h).decode()

    def scene(self) -> Scene:
        d = capi.SceneDesc()
        if self._lib.nori_host_scene_desc(self._h, C.byref(d)) != 0:
            raise NoriError("root is not a <scene>")
        return scene_from_desc(d)

    def test(self) -> TestFile:
        info = TestInfo()
        if self._lib.nori_host_test_info_get(self._h, C.byref(info)) != 0:
            raise NoriError("root is not a <test>")
        t = TestFile("ttest" if info.kind == 0 else "chi2test", info.significance_level, info.sample_count,
                     [info.angles[i] for i in range(info.n_angles)], [info.references[i] for i in range(info.n_references)],
                     test_count=info.test_count, resolution=info.resolution, min_exp_frequency=info.min_exp_frequency)
        for i in range(info.n_bsdfs):
            b = capi.BsdfDesc()
            self._lib.nori_host_test_bsdf(self._h, i, C.byref(b))
            t.bsdfs.append(bsdf_from_desc(b))
        for i in range(info.n_scenes):
            d = capi.SceneDesc()
            if self._lib.nori_host_test_scene_desc(self._h, i, C.byref(d)) != 0:
                raise NoriError(self._lib.nori_host_last_error().decode())
            t.scenes.append(scene_from_desc(d))
        return t

    def run_test(self) -> bool:
        """Run the <test> on the GPU (C++ host + device twins). True = all passed."""
        rc = self._lib.nori_host_test_run(self._h)
        if rc < 0:
            raise NoriError(self._lib.nori_host_last_error().decode())
        return rc == 0

    def render(self):
        sc = self.scene()
        rgbw = np.zeros(sc.frame_shape(), np.float32)
        st = capi.RenderStats()
        rc = self._lib.nori_host_render(self._h, capi.ptr(rgbw), C.byref(st))
        if rc != 0:
            raise NoriError(self._lib.nori_host_last_error().decode())
        return rgbw, st.as_dict()


def load_xml(path: str) -> Scene:
    """loadFromXML (src/parser.cpp:16) + flatten, for a <scene> file."""
    r = HostRoot(path)
    try:
        return r.scene()
    finally:
        r.close()


def save_images(basename: str, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, np.float32)
    rc = load_host().nori_host_save_images(os.fsencode(basename), capi.ptr(rgb), rgb.shape[1], rgb.shape[0])
    if rc != 0:
        raise NoriError(load_host().nori_host_last_error().decode())


def load_exr(path: str) -> np.ndarray:
    lib = load_host()
    p, w, h = C.POINTER(C.c_float)(), C.c_int(), C.c_int()
    if lib.nori_host_load_exr(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)) != 0:
        raise NoriError(lib.nori_host_last_error().decode())
    out = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.nori_host_free_buffer(p)
    return out
