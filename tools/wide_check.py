"""GPU: wide (BVH4, quantised) nodes against the oracle's brute force and against the BVH2 layout."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from nori_amd.render import Renderer
from tests import fuzz_intersect, scenes
from tests.backends import Oracle

class WideRenderer(Renderer):
    def upload(self, sc, build=True, builder=0):
        self.set_option("accel_layout", "bvh4q")
        return super().upload(sc, build, builder)     # both builders -> wide nodes

t0, hits, n = time.time(), 0, 0
while time.time() - t0 < float(os.environ.get("FUZZ_SECONDS", 30)):
    hits += fuzz_intersect.one_round(5000 + n, WideRenderer, n_rays=8000); n += 1
print(f"wide fuzz: {n} rounds, {hits} hits bit-identical, {fuzz_intersect.TOLERATED[0]} tolerated")
for name in ("pa5-table_mis", "pa4-cbox-path_mis"):
    from nori_amd.scene import Scene
    sc = Scene.load_npz(f"tests/golden/{name}.npz"); sc.camera.width, sc.camera.height, sc.sample_count = 256, 192, 8
    a = Renderer(0); a.set_option("accel_layout", "bvh2"); a.upload(sc); a.set_option("engine", "wavefront")
    b = Renderer(0); b.set_option("accel_layout", "bvh4q"); b.upload(sc); b.set_option("engine", "wavefront")
    A, sa = a.render_host(count_traversal=True); B, sb = b.render_host(count_traversal=True)
    print(name, "bvh2 vs bvh4q frames equal:", np.array_equal(A, B), "rays", sa["n_closest_rays"] + sa["n_shadow_rays"], sb["n_closest_rays"] + sb["n_shadow_rays"],
          "node tests", sa["n_node_tests"], sb["n_node_tests"], "tri tests", sa["n_tri_tests"], sb["n_tri_tests"], b.accel_info())
    a.close(); b.close()
