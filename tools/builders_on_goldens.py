"""Every shipped scene (tests/golden/*.npz) through the host's and the device's builder: node / triangle tests of a counted render, the time of
a timed one (wavefront engine), the build.   python tools/builders_on_goldens.py [width height spp]   -> profiles/r6_25_builders_on_goldens.txt"""
import glob, os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from nori_amd.render import Renderer
from nori_amd.scene import Scene

W, H, SPP = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 512, 64)
for f in sorted(glob.glob(os.path.join("tests", "golden", "*.npz"))):
    try:
        sc = Scene.load_npz(f)
    except Exception as e:      # (fixtures that are not scenes)
        continue
    sc.camera.width, sc.camera.height, sc.sample_count = W, H, SPP
    row = []
    for builder in (0, 2):
        r = Renderer(0).upload(sc, builder=builder)
        r.set_option("engine", "wavefront")
        info = r.accel_info()
        frame = torch.zeros(r.frame_shape(), device="cuda")
        st = r.render_into(frame, count_traversal=True)
        best = 1e30
        for _ in range(3):
            frame.zero_(); t = r.render_into(frame, time_kernels=True); best = min(best, t["trace_ms"])
        row.append((info, st, best))
        r.close()
    (ih, sh, th), (idv, sd, td) = row
    rays = sh["n_closest_rays"] + sh["n_shadow_rays"]
    assert rays == sd["n_closest_rays"] + sd["n_shadow_rays"]
    print(f"{os.path.basename(f)[:-4]:28s} {ih['n_triangles']:8d} tris | host: build {ih['build_ms']:6.1f} ms depth {ih['max_depth']:2d} node {sh['n_node_tests'] / rays:6.2f} tri {sh['n_tri_tests'] / rays:5.2f} trace {th:7.3f} ms"
          f" | device: build {idv['build_ms']:6.1f} ms refs {idv['n_references']:8d} depth {idv['max_depth']:2d} node {sd['n_node_tests'] / rays:6.2f} tri {sd['n_tri_tests'] / rays:5.2f} trace {td:7.3f} ms"
          f" | trace device / host {td / th:5.3f}", flush=True)
