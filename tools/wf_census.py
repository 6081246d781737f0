"""wf_extend lane census on the headline workload: NORI_HIP_CENSUS=1 python tools/wf_census.py"""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("NORI_HIP_CENSUS", "1")
import torch
from nori_amd.render import Renderer
from nori_amd.scene import Scene
sc = Scene.load_npz("tests/golden/pa4-cbox-path_mis.npz")
r = Renderer(0).upload(sc)
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
st = r.render_into(f, count_traversal=True)
print({k: st[k] for k in ("n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests")})
