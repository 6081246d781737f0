"""Traversal (vector ALU) beside shading (HBM) on the SAME CUs: two pipes, the traversal kernels of both on one plain stream, shading and film on
another (NORI_HIP_WF_OVERLAP2), wf_extend in 512-thread workgroups, three per CU = 6 waves per SIMD, so that one wf_shade workgroup per CU fits
beside it.  One process, alternating, frames compared bit for bit.   WORKLOAD=pa4-cbox-path_mis python tools/overlap2_probe.py"""
import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
wl = os.environ.get("WORKLOAD", "pa4-cbox-path_mis")
sc = workloads.load(wl, spp=int(os.environ["SPP"]) if "SPP" in os.environ else None).scene
r = Renderer(0).upload(sc)
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
ref = None
KEYS = ("EXTEND_BLOCK", "EXTEND_WGS_PER_CU", "OVERLAP2", "PIPES")
def run(label, **env):
    global ref
    for k in KEYS: os.environ.pop("NORI_HIP_WF_" + k, None)
    for k, v in env.items(): os.environ["NORI_HIP_WF_" + k] = str(v)
    best = None
    for _ in range(3):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    same = "" if ref is None else ("frame identical" if torch.equal(f, ref) else "frame differs (two pipes add a pixel's samples in another order)" if torch.allclose(f, ref, rtol=1e-4, atol=1e-5) else "FRAME WRONG")
    if ref is None: ref = f.clone()
    print(f"{wl} {label:52s}: frame {best['kernel_ms']:8.2f} ms | trace {best['trace_ms']:8.2f} shade {best['shade_ms']:8.2f} film {best['film_ms']:6.2f} | {rays / best['kernel_ms'] / 1e3:8.1f} Mrays/s | {same}", flush=True)
for k in range(2):
    run("shipped: one pipe, 1024 x 2")
    run("two streams, 1024 x 2 (no room beside it)", OVERLAP2=1)
    run("two streams, 512 x 3 + shade beside", OVERLAP2=1, EXTEND_BLOCK=512)
    run("two streams, 512 x 2 + shade beside", OVERLAP2=1, EXTEND_BLOCK=512, EXTEND_WGS_PER_CU=2)
    run("two streams, 1024 x 1 + shade beside", OVERLAP2=1, EXTEND_WGS_PER_CU=1)
    run("one pipe, 512 x 3", EXTEND_BLOCK=512)
