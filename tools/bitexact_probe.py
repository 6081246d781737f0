"""How much of the device path is bit-identical to the oracle?  (GPU box)  python tools/bitexact_probe.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from nori_amd.render import Renderer
from nori_amd.scene import Bsdf, Scene
from tests import scenes
from tests.backends import Oracle

r = Renderer(0).upload(scenes.soup_scene(4))
rng = np.random.default_rng(5)
n = 1_000_000
s = rng.uniform(0, 1, (n, 2)).astype(np.float32)
for name, param in [("tent", 0), ("disk", 0), ("uniform_sphere", 0), ("uniform_hemisphere", 0), ("cosine_hemisphere", 0), ("beckmann", 0.3), ("beckmann", 0.05)]:
    a, b = Oracle.warp(name, s, param), r.warp(name, s, param)
    pa, pb = Oracle.warp_pdf(name, a, param), r.warp_pdf(name, a, param)
    print(f"warp {name} {param}: values differ {int((a != b).sum())}, pdf differ {int((pa != pb).sum())} of {n}")
wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
wo = rng.normal(size=(n, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
for b in [Bsdf("diffuse", (0.2, 0.5, 0.7)), Bsdf("mirror"), Bsdf("dielectric"), Bsdf("microfacet", (0.1, 0.2, 0.15), 0.1, 1.5), Bsdf("microfacet", (0.4, 0.2, 0.3), 0.6, 1.8, 1.3)]:
    o = Oracle.bsdf_sample(b, wi, s); g = r.bsdf_sample(b, wi, s)
    print(f"bsdf {b.type}: sample wo differ {int((o[0] != g[0]).any(1).sum())} weight {int((o[1] != g[1]).any(1).sum())} eta {int((o[2] != g[2]).sum())} measure {int((o[3] != g[3]).sum())}; "
          f"eval differ {int((Oracle.bsdf_eval(b, wi, wo) != r.bsdf_eval(b, wi, wo)).any(1).sum())} pdf differ {int((Oracle.bsdf_pdf(b, wi, wo) != r.bsdf_pdf(b, wi, wo)).sum())}")
r.close()
for integ in ["normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis"]:
    sb = [Bsdf("mirror"), Bsdf("dielectric")] if integ in ("whitted", "path_mis") else [Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")]
    sc = scenes.cornell_box(16, 16, 1, integ, sphere_bsdfs=sb)
    sc.integrator.position, sc.integrator.energy = (0, 1.5, 0.5), (30, 30, 30)
    r, o = Renderer(0).upload(sc), Oracle(sc, use_bvh=True)
    m = 200000
    rays = o.sample_rays(rng.uniform(0, 16, (m, 2)).astype(np.float32))
    ss = rng.integers(0, 2 ** 62, m, dtype=np.uint64); sq = rng.integers(0, 2 ** 62, m, dtype=np.uint64)
    a, b = o.li(rays, ss, sq), r.li(rays, ss, sq)
    bad = (a != b).any(1)
    print(f"li {integ}: {int(bad.sum())} of {m} paths differ; max rel {np.abs(a - b).max() / max(a.max(), 1e-9):.2e}")
    r.close()
for name in ["pa4-cbox-path_mis", "pa5-table_mis", "pa5-cbox_mis", "pa5-veach_mis"]:
    sc = Scene.load_npz(f"tests/golden/{name}.npz")
    sc.camera.width, sc.camera.height, sc.sample_count = 200, 150, 4
    r, o = Renderer(0).upload(sc), Oracle(sc, use_bvh=True)
    m = 200000
    rays = o.sample_rays(rng.uniform(0, 150, (m, 2)).astype(np.float32))
    ss = rng.integers(0, 2 ** 62, m, dtype=np.uint64); sq = rng.integers(0, 2 ** 62, m, dtype=np.uint64)
    a, b = o.li(rays, ss, sq), r.li(rays, ss, sq)
    bad = (a != b).any(1)
    print(f"li scene {name}: {int(bad.sum())} of {m} paths differ")
    A, sa = o.render_host(); B, sb_ = r.render_host()
    print(f"   render: rays oracle {sa['n_closest_rays']}+{sa['n_shadow_rays']} device {sb_['n_closest_rays']}+{sb_['n_shadow_rays']}; max abs frame diff {np.abs(A - B).max():.3e}")
    r.close()
