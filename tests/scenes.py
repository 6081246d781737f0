"""Procedural test scenes (numpy only).  Seeds are fixed; nothing is read from
/root/reference at run time, so these also work on the GPU box."""
from __future__ import annotations

import numpy as np

from nori_amd.scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene


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
    v = np.array([p0, p1, p2, p3], dtype=np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    return v, f


def icosphere(subdiv=2, radius=1.0, center=(0, 0, 0), with_normals=True):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    n = np.array(v, dtype=np.float32)
    pos = (n * radius + np.asarray(center, dtype=np.float32)).astype(np.float32)
    return pos, np.array(f, dtype=np.uint32), (n if with_normals else None)


def triangle_soup(n=500, seed=1, extent=1.0, size=0.25):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    v = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32).reshape(-1, 3)
    f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return v, f


def random_rays(n, seed=2, extent=2.5, target_extent=1.0):
    """Rays from a shell around the origin aimed at random points of the scene."""
    from nori_amd._capi import RAY_DTYPE
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(0.2, extent, (n, 1))
    t = rng.uniform(-target_extent, target_extent, (n, 3))
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays["o"], rays["d"] = o.astype(np.float32), d.astype(np.float32)
    rays["mint"], rays["maxt"] = 1e-4, np.inf
    return rays


def cornell_box(width=64, height=64, spp=4, integrator="path_mis", sphere_subdiv=2, sphere_bsdfs=None,
                rfilter=None):
    """A Cornell-box-like scene: 5 axis-aligned walls, two spheres, one area light."""
    m = []
    white, red, green = Bsdf("diffuse", (0.725, 0.71, 0.68)), Bsdf("diffuse", (0.63, 0.065, 0.05)), Bsdf("diffuse", (0.161, 0.133, 0.427))
    # floor, ceiling, back
    v, f = quad((-1, 0, -1), (-1, 0, 1), (1, 0, 1), (1, 0, -1)); m.append(Mesh(v, f, bsdf=white, name="floor"))
    v, f = quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)); m.append(Mesh(v, f, bsdf=white, name="ceiling"))
    v, f = quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)); m.append(Mesh(v, f, bsdf=white, name="back"))
    v, f = quad((-1, 0, -1), (-1, 2, -1), (-1, 2, 1), (-1, 0, 1)); m.append(Mesh(v, f, bsdf=red, name="left"))
    v, f = quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)); m.append(Mesh(v, f, bsdf=green, name="right"))
    sb = sphere_bsdfs or [Bsdf("diffuse"), Bsdf("diffuse")]
    p, f, n = icosphere(sphere_subdiv, 0.35, (-0.45, 0.35, -0.3)); m.append(Mesh(p, f, n, bsdf=sb[0], name="sphere1"))
    p, f, n = icosphere(sphere_subdiv, 0.35, (0.45, 0.35, 0.3)); m.append(Mesh(p, f, n, bsdf=sb[1], name="sphere2"))
    v, f = quad((-0.25, 1.98, -0.25), (0.25, 1.98, -0.25), (0.25, 1.98, 0.25), (-0.25, 1.98, 0.25))
    m.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(20.0, 20.0, 20.0), name="light"))
    cam = Camera(width, height, 40.0, to_world=lookat((0, 1, 4.2), (0, 1, 0), (0, 1, 0)))
    return Scene(m, cam, rfilter or RFilter(), Integrator(integrator), spp)


def soup_scene(n=500, seed=1, width=32, height=32, integrator="normals"):
    v, f = triangle_soup(n, seed)
    cam = Camera(width, height, 45.0, to_world=lookat((0, 0, 4), (0, 0, 0), (0, 1, 0)))
    return Scene([Mesh(v, f, name="soup")], cam, RFilter(), Integrator(integrator), 1)
