"""Reordering the paths between bounces through a radix-sorted index, measured on one box in one process -- the experiment of round 5
(profiles/r5_02_sort_ceiling.txt).  The knob it drives (NORI_HIP_WF_SORT, wf_sort.hip) exists in commit f64423d only: the result was negative
(wf_extend slower with the sort cost excluded) and the code was removed again.  Kept as the record of what was run:
    WORKLOAD=... SPP=... CONFIGS="off 1,3 1,5 1,7 2,5 3,4" python tools/sort_probe.py"""
import os, sys, re, subprocess
sys.path.insert(0, ".")
import numpy as np, torch
from nori_amd.render import Renderer
from nori_amd import workloads
wl = os.environ.get("WORKLOAD", "pa4-cbox-path_mis")
sc = workloads.load(wl, spp=int(os.environ["SPP"]) if "SPP" in os.environ else None).scene
r = Renderer(0).upload(sc, builder=int(os.environ.get("BUILDER", 0)))
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
reps = int(os.environ.get("REPS", 2))
ref = None
for cfg in os.environ.get("CONFIGS", "default off 1,3 1,5 1,7 2,5 3,4").split():
    os.environ.pop("NORI_HIP_WF_SORT", None); os.environ.pop("NORI_HIP_WF_SYNC_EVERY", None)
    if cfg != "default": os.environ["NORI_HIP_WF_SYNC_EVERY"] = "1"
    if cfg not in ("default", "off"): os.environ["NORI_HIP_WF_SORT"] = cfg
    best = None
    for i in range(reps):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    frame = f.cpu().numpy()
    if ref is None: ref = frame
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    print(f"{wl} sort {cfg:8s}: frame {best['kernel_ms']:8.2f} ms | trace {best['trace_ms']:8.2f} shade {best['shade_ms']:7.2f} film {best['film_ms']:6.2f} | {rays / best['kernel_ms'] / 1e3:8.1f} Mrays/s | "
          f"frame {'identical' if np.array_equal(frame, ref) else 'DIFFERS max %g' % np.abs(frame - ref).max()} rays {rays}", flush=True)
