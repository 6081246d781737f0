"""wf_extend in workgroups of 512 threads at 6 waves per SIMD with the ray's plane coefficients kept in registers (NORI_HIP_WF_EXTEND_BLOCK=512)
against the shipped 1024-thread workgroups at 8 waves per SIMD; one process, alternating, frames compared bit for bit.
    WORKLOAD=pa4-cbox-path_mis python tools/block_probe.py
(The variant -- wf_extend<..., BLOCK = 512> with the coefficients made at the refill -- measured slower and is not in the tree: it is the
working copy of the commit before "wf_extend at 6 waves per SIMD ... removed"; profiles/r5_16_*.)"""
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
def run(label, **env):
    global ref
    for k in ("NORI_HIP_WF_EXTEND_BLOCK", "NORI_HIP_WF_EXTEND_WGS_PER_CU"): os.environ.pop(k, None)
    for k, v in env.items(): os.environ["NORI_HIP_WF_" + k] = str(v)
    best = None
    for _ in range(3):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    same = "" if ref is None else ("frame identical" if torch.equal(f, ref) else "FRAME DIFFERS")
    if ref is None: ref = f.clone()
    print(f"{wl} {label:34s}: frame {best['kernel_ms']:8.2f} ms | trace {best['trace_ms']:8.2f} shade {best['shade_ms']:8.2f} film {best['film_ms']:6.2f} | {rays / best['kernel_ms'] / 1e3:8.1f} Mrays/s | {same}", flush=True)
for k in range(2):
    run("1024 threads, 8 waves/SIMD")
    run("512 threads x 3, coefficients kept", EXTEND_BLOCK=512)
    run("512 threads x 2", EXTEND_BLOCK=512, EXTEND_WGS_PER_CU=2)
