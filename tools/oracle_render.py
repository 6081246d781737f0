"""Renders a named test scene with the CPU oracle and saves the RGBW frame (.npy) -- run by the tests in a subprocess with
NORI_ORACLE_LIBRARY=oracle/liboracle_glibc.so to get the render of the oracle build that calls the host's libm.
   python tools/oracle_render.py <cornell|table|cbox> <width> <height> <spp> <integrator> <out.npy>"""
import sys
sys.path.insert(0, ".")
import numpy as np
from nori_amd import workloads
from nori_amd.scene import Bsdf
from tests import scenes
from tests.backends import Oracle
kind, w, h, spp, integ, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
if kind == "cornell":
    sc = scenes.cornell_box(w, h, spp, integ, sphere_bsdfs=[Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("dielectric")])
elif kind == "table":
    sc = workloads.load("c4-table-mis", w, h, spp).scene
else:
    sc = workloads.load("pa4-cbox-path_mis", w, h, spp).scene
rgbw, st = Oracle(sc, use_bvh=True).render_host()
np.save(out, rgbw)
print(st["n_closest_rays"], st["n_shadow_rays"])
