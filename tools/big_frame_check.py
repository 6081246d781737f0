import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nori_amd.render import Renderer
from nori_amd.scene import Scene
sc = Scene.load_npz("tests/golden/pa5-cbox_mis.npz")
sc.camera.width, sc.camera.height, sc.sample_count = 1920, 1080, 300
r = Renderer(0).upload(sc)
f = torch.zeros(r.frame_shape(), device="cuda")
for i in range(2):
    f.zero_(); st = r.render_into(f)
rays = st["n_closest_rays"] + st["n_shadow_rays"]
print("1080p x 300 spp pa5 cbox (mirror + dielectric):", round(st["kernel_ms"], 1), "ms", round(rays / st["kernel_ms"] / 1e3), "Mrays/s", "samples", st["n_camera_samples"], "invalid", st["n_invalid"], "finite", bool(torch.isfinite(f).all()))
a = f.clone()
# same render in 5 sample slices and 3 tile slices: sums must agree
f.zero_()
for b, c in ((0, 100), (100, 1), (101, 199)):
    r.render_into(f, spp_begin=b, spp_count=c)
print("sample slices max rel diff", float(((f - a).abs() / a.abs().clamp_min(1e-3)).max()))
f.zero_()
for k in range(3):
    r.render_into(f, tile_mod=3, tile_rem=k)
print("tile slices max rel diff", float(((f - a).abs() / a.abs().clamp_min(1e-3)).max()))
w = a[..., 3]
b = r.border
print("weights interior min/max", float(w[b + 2:-b - 2, b + 2:-b - 2].min()), float(w.max()))
