import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
sc = workloads.load(os.environ.get("WORKLOAD", "pa4-cbox-path_mis"), spp=int(os.environ["SPP"]) if "SPP" in os.environ else None).scene
r = Renderer(0).upload(sc, builder=int(os.environ.get("BUILDER", 2)))
print(r.accel_info())
r.set_option("engine", os.environ.get("ENGINE", "wavefront"))
if "PATHS" in os.environ: r.set_option("wavefront_paths", int(os.environ["PATHS"]))          # the pool: paths in flight
if "SAMPLES" in os.environ: r.set_option("wavefront_samples", int(os.environ["SAMPLES"]))    # camera samples per batch
f = torch.zeros(r.frame_shape(), device="cuda")
for i in range(int(os.environ.get("REPS", 3))):
    f.zero_(); st = r.render_into(f, tile_mod=int(os.environ.get("TILE_MOD", 1)), count_traversal=bool(int(os.environ.get("COUNT", 0))),
                            time_kernels=bool(int(os.environ.get("TIMEK", 0))))
    if int(os.environ.get("COUNT", 0)): print({k: st[k] for k in ("n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests")})
    if int(os.environ.get("TIMEK", 0)): print("trace", round(st["trace_ms"], 2), "shade", round(st["shade_ms"], 2), "film", round(st["film_ms"], 2), end="  ")
    if int(os.environ.get("HASH", 0)): import hashlib; print("frame", hashlib.sha1(f.cpu().numpy().tobytes()).hexdigest()[:12], "rays", st["n_closest_rays"] + st["n_shadow_rays"], end="  ")
    print(round(st["kernel_ms"], 1), "ms", round((st["n_closest_rays"] + st["n_shadow_rays"]) / st["kernel_ms"] / 1e3, 1), "Mrays/s")
