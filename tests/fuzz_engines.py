"""Randomised consistency sweep of the two render engines (megakernel, wavefront): random small scenes --
triangle soups and spheres with random BSDFs, one or more area lights, every integrator, every filter,
ragged frame sizes, LBVH or SAH trees, small path budgets that force batching and wf_finish early or
never -- must give the same ray counts and the same frame (float summation order aside).

    python tests/fuzz_engines.py [--seconds 60] [--seed 0]

`tests/test_gpu_wavefront.py::test_fuzz_engines_short` runs a few rounds of it."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nori_amd.scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene  # noqa: E402
from tests import scenes  # noqa: E402

INTEGRATORS = ["normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis"]
FILTERS = ["gaussian", "mitchell", "tent", "box"]


def random_bsdf(rng):
    k = rng.integers(0, 4)
    if k == 0:
        return Bsdf("diffuse", tuple(rng.uniform(0.1, 0.9, 3)))
    if k == 1:
        return Bsdf("mirror")
    if k == 2:
        return Bsdf("dielectric")
    return Bsdf("microfacet", tuple(rng.uniform(0.05, 0.6, 3)), float(rng.uniform(0.05, 0.6)))


def random_scene(rng):
    meshes = []
    for _ in range(rng.integers(1, 4)):
        if rng.random() < 0.5:
            v, f = scenes.triangle_soup(int(rng.integers(1, 400)), int(rng.integers(1 << 30)), 1.0, float(rng.uniform(0.05, 0.5)))
            meshes.append(Mesh(v, f, bsdf=random_bsdf(rng), name="soup"))
        else:
            p, f, n = scenes.icosphere(int(rng.integers(0, 3)), float(rng.uniform(0.2, 0.7)), tuple(rng.uniform(-0.6, 0.6, 3)), bool(rng.random() < 0.7))
            meshes.append(Mesh(p, f, n, bsdf=random_bsdf(rng), name="sphere"))
    v, f = scenes.quad((-3, -1.2, -3), (-3, -1.2, 3), (3, -1.2, 3), (3, -1.2, -3))
    meshes.append(Mesh(v, f, bsdf=random_bsdf(rng), name="floor"))
    for _ in range(rng.integers(1, 3)):
        c = rng.uniform(-1.5, 1.5, 3); c[1] = rng.uniform(1.2, 2.5); s = rng.uniform(0.2, 0.8)
        v, f = scenes.quad(c + [-s, 0, -s], c + [s, 0, -s], c + [s, 0, s], c + [-s, 0, s])
        meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=tuple(rng.uniform(2, 20, 3)), name="light"))
    w, h = int(rng.integers(9, 90)), int(rng.integers(9, 70))
    cam = Camera(w, h, float(rng.uniform(25, 70)), to_world=scenes.lookat(tuple(rng.uniform(-1, 1, 3) + [0, 0.5, 4]), (0, 0, 0), (0, 1, 0)))
    integ = Integrator(INTEGRATORS[int(rng.integers(0, len(INTEGRATORS)))])
    integ.position, integ.energy = (0.3, 1.5, 0.5), (30.0, 25.0, 20.0)
    return Scene(meshes, cam, RFilter(FILTERS[int(rng.integers(0, 4))]), integ, int(rng.integers(1, 9)))


def one_round(seed, renderer_cls, oracle=False):
    rng = np.random.default_rng(seed)
    sc = random_scene(rng)
    builder = int(rng.integers(0, 3))      # host SAH, device radix tree, device PLOC
    mk = renderer_cls(0).upload(sc, builder=builder)
    wf = renderer_cls(0).upload(sc, builder=builder)
    wf.set_option("engine", "wavefront")
    wf.set_option("wavefront_paths", int(rng.choice([256, 256 * 7, 1 << 14, 1 << 28])))        # the pool ...
    wf.set_option("wavefront_samples", int(rng.choice([256, 256 * 7, 1 << 14, 1 << 28])))      # ... and the batch: smaller, equal or bigger (regeneration)
    env = {"NORI_HIP_WF_FINISH_PATHS": str(int(rng.choice([256, 4096, 1 << 19]))), "NORI_HIP_WF_SYNC_EVERY": str(int(rng.choice([1, 2, 6]))),
           "NORI_HIP_WF_NO_ASM_LOOP": "1"}      # counters compared on the megakernel's tree form, the 64-B nodes (the 32-B records test a little more)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        a, sa = mk.render_host(count_traversal=True)
        b, sb = wf.render_host(count_traversal=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        mk.close(); wf.close()
    what = f"seed {seed}: {sc.integrator.type}, {sc.rfilter.type}, {sc.camera.width}x{sc.camera.height}x{sc.sample_count}, builder {builder}, {env}"
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests", "n_invalid"):
        assert sa[k] == sb[k], (what, k, sa[k], sb[k])
    assert np.isfinite(a).all() and np.isfinite(b).all(), what
    np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5, err_msg=what)
    if oracle:
        # against the CPU restatement of the reference: same random numbers, so only libm ulps (and the few
        # decisions they flip) separate the frames -- DESIGN.md section 5
        from nori_amd.render import develop_host
        from tests.backends import Oracle
        o = Oracle(sc, use_bvh=True)
        ref, so = o.render_host()
        border = o.border
        o.close()
        for k in ("n_closest_rays", "n_shadow_rays"):
            assert abs(int(so[k]) - int(sb[k])) <= 3e-3 * so[k] + 3, (what, k, so[k], sb[k])
        np.testing.assert_allclose(b[..., 3], ref[..., 3], rtol=1e-5, atol=1e-6, err_msg=what)
        x, y = develop_host(ref, border), develop_host(b, border)
        rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-2)
        assert (rel < 1e-3).mean() > 0.95, (what, float((rel < 1e-3).mean()))
        assert abs(float(x.mean()) - float(y.mean())) < 1e-2 * max(float(x.mean()), 1e-3), what
    return int(sa["n_closest_rays"] + sa["n_shadow_rays"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--oracle", action="store_true", help="also compare every frame with the CPU oracle's render")
    a = ap.parse_args()
    from nori_amd.render import Renderer
    t0, n, rays = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        rays += one_round(a.seed + n, Renderer, a.oracle)
        n += 1
    print(f"fuzz_engines: {n} rounds from seed {a.seed}, {rays} rays, both engines agree" + (" with each other and with the oracle" if a.oracle else ""))


if __name__ == "__main__":
    main()
