"""BASELINE.json configs[4] analogue: a synthetic multi-million-triangle scene (fractal terrain under an
area light), BVH built on the device, path_mis through the wavefront engine.  Unlike the Cornell box
the tree (64 B nodes + 48 B triangles, ~1 GB at 10 M triangles) is far larger than L2 / Infinity Cache,
so traversal really streams from HBM.

    python tests/stress_large.py [--tris 10000000] [--size 512] [--spp 16] [--check 256]

Prints one JSON line.  `--check K` compares K camera rays against the oracle's brute-force scan
(test infrastructure; bit-exact hit records expected)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nori_amd.scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene  # noqa: E402
from tests.scenes import lookat, quad  # noqa: E402


def terrain(n_tris: int, seed: int = 5):
    """(n+1)^2 grid over [-1,1]^2 with a few octaves of value noise as height; 2 n^2 triangles."""
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
        gx = g[:, i] * (1 - t) + g[:, i + 1] * t              # (cells+1, n+1)
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


def make_scene(n_tris: int, size: int, spp: int, integrator: str = "path_mis") -> Scene:
    pos, idx = terrain(n_tris)
    meshes = [Mesh(pos, idx, bsdf=Bsdf("diffuse", (0.6, 0.55, 0.5)), name="terrain")]
    v, f = quad((-0.5, 1.5, -0.5), (0.5, 1.5, -0.5), (0.5, 1.5, 0.5), (-0.5, 1.5, 0.5))
    meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(15.0, 15.0, 15.0), name="light"))
    cam = Camera(size, size, 45.0, to_world=lookat((0.0, 1.2, 2.2), (0, 0, 0), (0, 1, 0)))
    return Scene(meshes, cam, RFilter(), Integrator(integrator), spp)


def run(n_tris=10_000_000, size=512, spp=16, check=0, builder=1, engine="auto", reps=2, device=0):
    import torch
    from nori_amd.render import Renderer

    t0 = time.time()
    sc = make_scene(n_tris, size, spp)
    t_scene = time.time() - t0
    r = Renderer(device)
    t0 = time.time()
    r.upload(sc, builder=builder)
    t_upload = time.time() - t0
    info = r.accel_info()
    r.set_option("engine", engine)
    frame = torch.zeros(r.frame_shape(), device=f"cuda:{device}")
    best = None
    for _ in range(reps):
        frame.zero_()
        st = r.render_into(frame)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    frame.zero_()
    cst = r.render_into(frame, count_traversal=True)
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    out = {
        "triangles": info["n_triangles"], "nodes": info["n_nodes"], "max_depth": info["max_depth"],
        "accel_bytes": info["total_bytes"], "build_ms": round(info["build_ms"], 2), "builder": builder,
        "scene_gen_s": round(t_scene, 2), "upload_s": round(t_upload, 2), "size": size, "spp": spp, "engine": engine,
        "kernel_ms": round(best["kernel_ms"], 3), "rays": rays, "mrays_per_s": round(rays / best["kernel_ms"] / 1e3, 1),
        "node_tests_per_ray": round(cst["n_node_tests"] / max(rays, 1), 2),
        "tri_tests_per_ray": round(cst["n_tri_tests"] / max(rays, 1), 2),
        "traversal_bytes_per_ray": round((cst["n_node_tests"] * 64 + cst["n_tri_tests"] * 48) / max(rays, 1), 1),
        "mean_w": float(frame[..., 3].mean().item()),
        "finite": bool(torch.isfinite(frame).all().item()),
    }
    if check:
        from tests.backends import Oracle
        o = Oracle(sc)                      # brute force (no set_accel)
        rays_c = r.sample_rays(np.random.default_rng(11).uniform(0, size, (check, 2)).astype(np.float32))
        a, b = o.intersect(rays_c), r.intersect(rays_c)
        out["checked_rays"] = int(check)
        out["hits"] = int((a["mesh"] != 0xFFFFFFFF).sum())
        out["bit_exact"] = bool(all(np.array_equal(a[k], b[k]) for k in a.dtype.names))
    r.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--builder", type=int, default=1)
    ap.add_argument("--engine", default="auto")
    a = ap.parse_args()
    print(json.dumps(run(a.tris, a.size, a.spp, a.check, a.builder, a.engine)))
