"""Flattened scene (`nori_scene_desc`) held in numpy arrays.

This is the data format on the caller's side of the C ABI: what the C++ host
(`libnori_host.so`, which parses Nori's XML/OBJ) hands over, what the `.npz`
fixtures under tests/golden/ store, and what `Renderer.upload` consumes.
Field names follow the reference's plugin parameters (src/diffuse.cpp:19,
src/microfacet.cpp:17-36, src/perspective.cpp:22-39, src/rfilter.cpp).
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _capi as capi


@dataclass
class Bsdf:
    type: str = "diffuse"
    albedo: tuple = (0.5, 0.5, 0.5)        # diffuse albedo / microfacet kd
    alpha: float = 0.1
    int_ior: float = 1.5046
    ext_ior: float = 1.000277

    @property
    def ks(self) -> float:                  # src/microfacet.cpp:35
        return float(np.float32(1.0) - np.float32(max(np.float32(x) for x in self.albedo)))

    def desc(self) -> capi.BsdfDesc:
        d = capi.BsdfDesc()
        d.type = capi.BSDF_NAMES[self.type]
        d.albedo[:] = [float(x) for x in self.albedo]
        d.alpha, d.int_ior, d.ext_ior = self.alpha, self.int_ior, self.ext_ior
        d.ks = self.ks if self.type == "microfacet" else 0.0
        return d


@dataclass
class Mesh:
    positions: np.ndarray                   # (nV, 3) float32, world space
    indices: np.ndarray                     # (nF, 3) uint32
    normals: Optional[np.ndarray] = None    # (nV, 3) float32
    texcoords: Optional[np.ndarray] = None  # (nV, 2) float32
    bsdf: Bsdf = field(default_factory=Bsdf)
    radiance: Optional[tuple] = None        # area emitter if not None
    name: str = ""

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(self.indices, dtype=np.uint32).reshape(-1, 3)
        if self.normals is not None:
            self.normals = np.ascontiguousarray(self.normals, dtype=np.float32).reshape(-1, 3)
        if self.texcoords is not None:
            self.texcoords = np.ascontiguousarray(self.texcoords, dtype=np.float32).reshape(-1, 2)


@dataclass
class Camera:
    width: int = 1280
    height: int = 720
    fov: float = 30.0
    near_clip: float = 1e-4
    far_clip: float = 1e4
    to_world: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))


@dataclass
class RFilter:
    type: str = "gaussian"
    radius: float = 2.0
    stddev: float = 0.5
    B: float = 1.0 / 3.0
    C: float = 1.0 / 3.0


@dataclass
class Integrator:
    type: str = "path_mis"
    position: tuple = (0.0, 0.0, 0.0)
    energy: tuple = (0.0, 0.0, 0.0)


@dataclass
class Scene:
    meshes: List[Mesh] = field(default_factory=list)
    camera: Camera = field(default_factory=Camera)
    rfilter: RFilter = field(default_factory=RFilter)
    integrator: Integrator = field(default_factory=Integrator)
    sample_count: int = 1

    # ------------------------------------------------------------ C view
    def c_desc(self):
        """(SceneDesc, keepalive): the ctypes struct plus the objects its pointers borrow."""
        n = len(self.meshes)
        arr = (capi.MeshDesc * max(n, 1))()
        keep = [arr]
        fp = C.POINTER(C.c_float)
        for i, m in enumerate(self.meshes):
            d = arr[i]
            d.n_vertices, d.n_triangles = m.positions.shape[0], m.indices.shape[0]
            d.positions = m.positions.ctypes.data_as(fp)
            d.normals = m.normals.ctypes.data_as(fp) if m.normals is not None else None
            d.texcoords = m.texcoords.ctypes.data_as(fp) if m.texcoords is not None else None
            d.indices = m.indices.ctypes.data_as(C.POINTER(C.c_uint32))
            d.bsdf = m.bsdf.desc()
            d.is_emitter = 1 if m.radiance is not None else 0
            d.radiance[:] = [float(x) for x in (m.radiance or (0, 0, 0))]
            keep.append(m)
        s = capi.SceneDesc()
        s.n_meshes = n
        s.meshes = C.cast(arr, C.POINTER(capi.MeshDesc))
        c = self.camera
        s.camera.width, s.camera.height = int(c.width), int(c.height)
        s.camera.fov, s.camera.near_clip, s.camera.far_clip = c.fov, c.near_clip, c.far_clip
        s.camera.to_world[:] = [float(x) for x in np.asarray(c.to_world, dtype=np.float32).reshape(16)]
        f = self.rfilter
        s.rfilter.type = capi.RFILTER_NAMES[f.type]
        s.rfilter.radius, s.rfilter.stddev, s.rfilter.B, s.rfilter.C = f.radius, f.stddev, f.B, f.C
        it = self.integrator
        s.integrator.type = capi.INTEGRATOR_NAMES[it.type]
        s.integrator.position[:] = [float(x) for x in it.position]
        s.integrator.energy[:] = [float(x) for x in it.energy]
        s.sample_count = int(self.sample_count)
        return s, keep

    @property
    def n_triangles(self) -> int:
        return int(sum(m.indices.shape[0] for m in self.meshes))

    @property
    def border(self) -> int:
        f = self.rfilter
        r = 1.0 if f.type == "tent" else (0.5 if f.type == "box" else f.radius)
        return int(np.ceil(np.float32(r) - np.float32(0.5)))

    def frame_shape(self):
        b = self.border
        return (self.camera.height + 2 * b, self.camera.width + 2 * b, 4)

    # --------------------------------------------------------- (de)serialise
    def geometry_digest(self) -> str:
        """sha256 over everything a fixture stores as arrays (camera transform, per mesh V / F / N / UV)."""
        import hashlib
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(self.camera.to_world, dtype=np.float32).tobytes())
        for m in self.meshes:
            for a in (m.positions, m.indices, m.normals, m.texcoords):
                h.update(b"-" if a is None else np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    def save_npz(self, path: str, base: Optional[str] = None) -> None:
        """`base`: name of a fixture in the same directory with the same arrays (several shipped scenes differ in the integrator,
        the sample count or one BSDF only): the file then holds the parameters and the digest of the arrays, not the arrays."""
        meta = {
            "camera": {"width": self.camera.width, "height": self.camera.height, "fov": self.camera.fov,
                       "near_clip": self.camera.near_clip, "far_clip": self.camera.far_clip},
            "rfilter": vars(self.rfilter), "integrator": {"type": self.integrator.type,
                                                          "position": list(self.integrator.position),
                                                          "energy": list(self.integrator.energy)},
            "sample_count": self.sample_count,
            "meshes": [{"name": m.name, "bsdf": {"type": m.bsdf.type, "albedo": list(map(float, m.bsdf.albedo)),
                                                 "alpha": m.bsdf.alpha, "int_ior": m.bsdf.int_ior,
                                                 "ext_ior": m.bsdf.ext_ior},
                        "radiance": list(map(float, m.radiance)) if m.radiance is not None else None,
                        "normals": m.normals is not None, "texcoords": m.texcoords is not None}
                       for m in self.meshes],
        }
        if base is not None:
            meta["base"], meta["geometry_digest"] = base, self.geometry_digest()
            np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
            return
        arrays = {"to_world": np.asarray(self.camera.to_world, dtype=np.float32)}
        for i, m in enumerate(self.meshes):
            arrays[f"m{i}_V"] = m.positions
            arrays[f"m{i}_F"] = m.indices
            if m.normals is not None:
                arrays[f"m{i}_N"] = m.normals
            if m.texcoords is not None:
                arrays[f"m{i}_UV"] = m.texcoords
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(path, **arrays)

    @staticmethod
    def load_npz(path: str) -> "Scene":
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        if "base" in meta:      # parameters of this scene over the arrays of another fixture (save_npz)
            import os
            sc = Scene.load_npz(os.path.join(os.path.dirname(os.path.abspath(path)), meta["base"] + ".npz"))
            assert len(sc.meshes) == len(meta["meshes"]), "variant fixture does not match its base"
            cam = sc.camera.to_world
            sc.camera = Camera(to_world=cam, **meta["camera"])
            sc.rfilter = RFilter(**meta["rfilter"])
            it = meta["integrator"]
            sc.integrator = Integrator(it["type"], tuple(it["position"]), tuple(it["energy"]))
            sc.sample_count = meta["sample_count"]
            for m, mm in zip(sc.meshes, meta["meshes"]):
                b = mm["bsdf"]
                m.bsdf = Bsdf(b["type"], tuple(b["albedo"]), b["alpha"], b["int_ior"], b["ext_ior"])
                m.radiance = tuple(mm["radiance"]) if mm["radiance"] is not None else None
                m.name = mm["name"]
            assert sc.geometry_digest() == meta["geometry_digest"], "variant fixture: the base fixture's arrays have changed"
            return sc
        sc = Scene()
        sc.camera = Camera(to_world=z["to_world"].astype(np.float32), **meta["camera"])
        sc.rfilter = RFilter(**meta["rfilter"])
        it = meta["integrator"]
        sc.integrator = Integrator(it["type"], tuple(it["position"]), tuple(it["energy"]))
        sc.sample_count = meta["sample_count"]
        for i, mm in enumerate(meta["meshes"]):
            b = mm["bsdf"]
            sc.meshes.append(Mesh(z[f"m{i}_V"], z[f"m{i}_F"],
                                  z[f"m{i}_N"] if mm["normals"] else None,
                                  z[f"m{i}_UV"] if mm["texcoords"] else None,
                                  Bsdf(b["type"], tuple(b["albedo"]), b["alpha"], b["int_ior"], b["ext_ior"]),
                                  tuple(mm["radiance"]) if mm["radiance"] is not None else None, mm["name"]))
        return sc
