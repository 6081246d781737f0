"""cost of film_order = reference on the headline workload (GPU box)"""
import sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
sc = workloads.load("pa4-cbox-path_mis").scene
r = Renderer(0).upload(sc)
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
for order in ("fast", "reference", "fast", "reference"):
    r.set_option("film_order", order)
    f.zero_(); st = r.render_into(f, time_kernels=True)
    print(order, round(st["kernel_ms"], 1), "ms, film", round(st["film_ms"], 2), "ms")
