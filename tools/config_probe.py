"""One-GPU measurements of the BASELINE.json configurations other than the headline one.
C2: ambient occlusion on a ~330 K-triangle displaced icosphere, 1024^2, 64 spp (intersect-dominated, one bounce).
C4: pa5 table (microfacet + dielectric), path_mis, 2048^2, 1024 spp, this GPU's eighth of the tiles."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nori_amd.render import Renderer
from nori_amd.scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene
from tests import scenes

def run(name, sc, **kw):
    r = Renderer(0).upload(sc)
    f = torch.zeros(r.frame_shape(), device="cuda")
    best = None
    for _ in range(2):
        f.zero_(); st = r.render_into(f, **kw)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    info = r.accel_info()
    print(f"{name}: {info['n_triangles']} tris, build {info['build_ms']:.1f} ms, {best['kernel_ms']:.1f} ms, {rays/1e6:.0f} M rays, "
          f"{rays / best['kernel_ms'] / 1e3:.0f} Mrays/s, finite {bool(torch.isfinite(f).all())}")
    r.close()

if "c2" in sys.argv[1:] or len(sys.argv) == 1:
    p, f, n = scenes.icosphere(7, 1.0)
    rng = np.random.default_rng(1)
    p = (p * (1 + 0.03 * np.sin(7 * p[:, :1]) * np.cos(5 * p[:, 1:2]))).astype(np.float32)
    fl, ff = scenes.quad((-4, -1.1, -4), (-4, -1.1, 4), (4, -1.1, 4), (4, -1.1, -4))
    cam = Camera(1024, 1024, 35.0, to_world=scenes.lookat((0, 0.6, 4), (0, -0.1, 0), (0, 1, 0)))
    sc = Scene([Mesh(p, f, None, bsdf=Bsdf("diffuse"), name="blob"), Mesh(fl, ff, bsdf=Bsdf("diffuse"), name="floor")], cam, RFilter(), Integrator("ao"), 64)
    run("C2 ao", sc)
if "c4" in sys.argv[1:] or len(sys.argv) == 1:
    sc = Scene.load_npz("tests/golden/pa5-table_mis.npz")
    sc.camera.width, sc.camera.height, sc.sample_count = 2048, 2048, 1024
    run("C4 table, 1/8 of the tiles", sc, tile_mod=8, tile_rem=3)
