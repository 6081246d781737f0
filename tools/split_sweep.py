"""The CU split of the wavefront engine (wavefront.hip, wavefront_render) swept in ONE process on one box:
    SPLITS="0 32 48 64 80 96" WORKLOAD=pa4-cbox-path_mis REPS=3 python tools/split_sweep.py
prints ms per frame (best and median of REPS) per number of CUs given to the shading side, the per-class HIP-event sums, and
whether the frame has the bits of the unsplit frame."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from nori_amd.render import Renderer
from nori_amd import workloads
wl = os.environ.get("WORKLOAD", "pa4-cbox-path_mis")
sc = workloads.load(wl, spp=int(os.environ["SPP"]) if "SPP" in os.environ else None).scene
r = Renderer(0).upload(sc, builder=int(os.environ.get("BUILDER", 0)))
r.set_option("engine", "wavefront")
if "PATHS" in os.environ: r.set_option("wavefront_paths", int(os.environ["PATHS"]))
f = torch.zeros(r.frame_shape(), device="cuda")
reps = int(os.environ.get("REPS", 3))
ref = None
for cus in [int(x) for x in os.environ.get("SPLITS", "0 32 48 64 80 96").split()]:
    os.environ["NORI_HIP_WF_SPLIT_CUS"] = str(cus)
    ms = []
    for i in range(reps):
        f.zero_(); st = r.render_into(f, time_kernels=(i == reps - 1)); ms.append(st["kernel_ms"])
    frame = f.cpu().numpy()
    if ref is None: ref = frame
    rays = st["n_closest_rays"] + st["n_shadow_rays"]
    print(f"{wl} shading CUs {cus:3d} (trace on {st['trace_cus']}): best {min(ms):7.2f} median {sorted(ms)[len(ms) // 2]:7.2f} ms  {rays / min(ms) / 1e3:8.1f} Mrays/s | "
          f"timed run {ms[-1]:7.2f}: trace {st['trace_ms']:.2f} shade {st['shade_ms']:.2f} film {st['film_ms']:.2f} | frame {'identical' if np.array_equal(frame, ref) else 'DIFFERS max %g' % np.abs(frame - ref).max()}", flush=True)
