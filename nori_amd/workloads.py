"""The BASELINE.json configurations as named, seeded workloads (SURVEY.md 8(d) "Configs as concrete inputs").

Every workload is a `Scene` (nori_amd.scene) plus how a node of GPUs shards it.  Geometry comes either from
the fixtures under tests/golden/ (the reference's own scenes, flattened by tools/make_goldens.py -- the GPU
box has no /root/reference) or from the generators below (numpy, fixed seeds: the same triangles on every
run and every rank).

  c1-bunny-normals     scenes/pa1/bunny.xml, `normals`, 512 x 512, 1 spp                (plumbing; CPU reference path)
  c2-ao-icosphere      ambient occlusion, diffuse, 1024 x 1024, 64 spp on a procedurally displaced icosphere
                       (327,680 triangles, stand-in for the absent ajax.obj) over a floor: BVH + intersect only
  pa4-cbox-path_mis    (= c3, the headline) geometry of scenes/pa4/cbox/cbox-distributed.xml, path_mis,
                       1024 x 1024, 256 spp
  c4-table-mis         scenes/pa5/table/table_mis.xml (microfacet + dielectric), 2048 x 2048, 1024 spp, tile split
  c5-terrain-10m       10,004,450-triangle fractal terrain (2237^2 grid cells x 2, value noise, seed 5) under one
                       area-light quad, path_mis, 1024 x 1024, 512 spp, sample split
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@dataclass
class Workload:
    name: str
    scene: Scene
    split: str              # how N GPUs shard it: "tile" | "sample"
    config: str             # which BASELINE.json config it is
    generator: str          # where the triangles come from


# ------------------------------------------------------------------ generators
def lookat(origin, target, up):
    """The <lookat> transform of src/parser.cpp:266-288 (columns left, newUp, dir, origin)."""
    o, t, u = (np.asarray(v, dtype=np.float32) for v in (origin, target, up))
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u / np.linalg.norm(u), d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    new_up /= np.linalg.norm(new_up)
    m = np.eye(4, dtype=np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m.astype(np.float32)


def quad(p0, p1, p2, p3):
    return np.array([p0, p1, p2, p3], dtype=np.float32), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)


def icosphere(subdiv: int):
    """Unit icosphere, 20 * 4^subdiv triangles, vectorised subdivision (subdiv 7 -> 327,680 triangles)."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
                  (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
                  (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    for _ in range(subdiv):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(e, axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + np.asarray(inv).reshape(-1).reshape(3, -1)                # midpoint ids of edges ab, bc, ca per face
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        ab, bc, ca = m[0], m[1], m[2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v.astype(np.float32), f.astype(np.uint32)


def terrain(n_tris: int, seed: int = 5):
    """(n+1)^2 grid over [-1,1]^2, a few octaves of value noise as height; 2 n^2 triangles."""
    n = max(1, int(round((n_tris / 2) ** 0.5)))
    rng = np.random.default_rng(seed)
    h = np.zeros((n + 1, n + 1), dtype=np.float32)
    amp, cells = 0.25, 4
    while cells <= n and amp > 1e-4:
        g = rng.uniform(-1, 1, (cells + 1, cells + 1)).astype(np.float32)
        x = np.linspace(0, cells, n + 1, dtype=np.float32)
        i = np.minimum(x.astype(np.int32), cells - 1)
        t = x - i
        t = t * t * (3 - 2 * t)
        gx = g[:, i] * (1 - t) + g[:, i + 1] * t
        h += amp * (gx[i, :] * (1 - t[:, None]) + gx[i + 1, :] * t[:, None])
        amp *= 0.5
        cells *= 2
    xs = np.linspace(-1, 1, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    pos = np.stack([X, h, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32), indexing="xy")
    a = (i * (n + 1) + j).ravel()
    b, c, d = a + 1, a + (n + 1), a + (n + 2)
    idx = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)]).astype(np.uint32)
    return pos, idx


def terrain_scene(n_tris: int, size: int, spp: int, integrator: str = "path_mis") -> Scene:
    pos, idx = terrain(n_tris)
    meshes = [Mesh(pos, idx, bsdf=Bsdf("diffuse", (0.6, 0.55, 0.5)), name="terrain")]
    v, f = quad((-0.5, 1.5, -0.5), (0.5, 1.5, -0.5), (0.5, 1.5, 0.5), (-0.5, 1.5, 0.5))
    meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(15.0, 15.0, 15.0), name="light"))
    cam = Camera(size, size, 45.0, to_world=lookat((0.0, 1.2, 2.2), (0, 0, 0), (0, 1, 0)))
    return Scene(meshes, cam, RFilter(), Integrator(integrator), spp)


def ao_scene(size: int, spp: int, subdiv: int = 7) -> Scene:
    p, f = icosphere(subdiv)
    p = (p * (1 + 0.03 * np.sin(7 * p[:, :1]) * np.cos(5 * p[:, 1:2]))).astype(np.float32)
    fl, ff = quad((-4, -1.1, -4), (-4, -1.1, 4), (4, -1.1, 4), (4, -1.1, -4))
    cam = Camera(size, size, 35.0, to_world=lookat((0, 0.6, 4), (0, -0.1, 0), (0, 1, 0)))
    return Scene([Mesh(p, f, None, bsdf=Bsdf("diffuse"), name="blob"), Mesh(fl, ff, bsdf=Bsdf("diffuse"), name="floor")],
                 cam, RFilter(), Integrator("ao"), spp)


# ------------------------------------------------------------------ registry
def _golden(name, width=None, height=None, spp=None, integrator=None) -> Scene:
    sc = Scene.load_npz(os.path.join(GOLDEN, name + ".npz"))
    if width:
        sc.camera.width, sc.camera.height = int(width), int(height)
    if spp:
        sc.sample_count = int(spp)
    if integrator:
        sc.integrator.type = integrator
    return sc


NAMES = ("c1-bunny-normals", "c2-ao-icosphere", "pa4-cbox-path_mis", "c4-table-mis", "c5-terrain-10m")
ALIASES = {"c1": "c1-bunny-normals", "c2": "c2-ao-icosphere", "c3": "pa4-cbox-path_mis", "c3-cbox-path_mis": "pa4-cbox-path_mis",
           "c4": "c4-table-mis", "c5": "c5-terrain-10m"}


def load(name: str, width: int | None = None, height: int | None = None, spp: int | None = None,
         triangles: int | None = None) -> Workload:
    """The named workload at its BASELINE.json size, or at the overrides given (tests use small ones)."""
    name = ALIASES.get(name, name)
    if name == "c1-bunny-normals":
        sc = _golden("pa1-bunny", width or 512, height or 512, spp or 1, "normals")
        return Workload(name, sc, "tile", "configs[0]: pa1 bunny, normals, 512x512, 1 spp", "tests/golden/pa1-bunny.npz (scenes/pa1/bunny.xml)")
    if name == "c2-ao-icosphere":
        sc = ao_scene(width or 1024, spp or 64)
        if height:
            sc.camera.height = int(height)
        return Workload(name, sc, "tile", "configs[1]: AO, diffuse, 1024x1024, 64 spp (BVH + intersect, one bounce)",
                        "icosphere(7) displaced by 0.03 sin(7x) cos(5y) + floor quad; stand-in for ajax.obj")
    if name == "pa4-cbox-path_mis":
        sc = _golden("pa4-cbox-path_mis", width or 1024, height or 1024, spp or 256)
        return Workload(name, sc, "tile", "configs[2]: pa4 Cornell box, path_mis, 1024x1024, 256 spp",
                        "tests/golden/pa4-cbox-path_mis.npz (scenes/pa4/cbox/cbox-distributed.xml)")
    if name == "c4-table-mis":
        sc = _golden("pa5-table_mis", width or 2048, height or 2048, spp or 1024)
        return Workload(name, sc, "tile", "configs[3]: pa5 table (microfacet + dielectric), path_mis, 2048x2048, 1024 spp, tile split",
                        "tests/golden/pa5-table_mis.npz (scenes/pa5/table/table_mis.xml)")
    if name == "c5-terrain-10m":
        sc = terrain_scene(triangles or 10_000_000, width or 1024, spp or 512)
        if height:
            sc.camera.height = int(height)
        return Workload(name, sc, "sample", "configs[4]: synthetic 10M-triangle scene, path_mis, 512 spp, sample split",
                        "terrain(): 2237^2-cell height field, 5 octaves of value noise, seed 5, + area-light quad")
    if os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        sc = _golden(name, width, height, spp)
        return Workload(name, sc, "tile", "shipped scene " + name, f"tests/golden/{name}.npz")
    raise ValueError(f"unknown workload {name!r}; known: {', '.join(NAMES)}")
