"""The tail of a batch (wf_finish) on a few CUs, and in its persistent form -- how long does it take there?  One process, one box;
every configuration's frame is compared bit for bit with the first.  Knobs are read by wavefront_render at every call.
    WORKLOAD=c4-table-mis SPP=128 python tools/tail_probe.py"""
import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
wl = os.environ.get("WORKLOAD", "c4-table-mis")
sc = workloads.load(wl, spp=int(os.environ.get("SPP", 128))).scene
r = Renderer(0).upload(sc)
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
KEYS = ("NORI_HIP_WF_FINISH_PATHS", "NORI_HIP_WF_TAIL_CUS", "NORI_HIP_WF_STATIC_FIRST")
def run(label, **env):
    for k in KEYS: os.environ.pop(k, None)
    for k, v in env.items(): os.environ["NORI_HIP_WF_" + k] = str(v)
    best = None
    for _ in range(int(os.environ.get("REPS", 2))):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["n_closest_rays"] + best["n_shadow_rays"]
    same = "" if run.ref is None else ("frame identical" if torch.equal(f, run.ref) else "FRAME DIFFERS")
    if run.ref is None: run.ref = f.clone()
    print(f"{wl} {label:38s}: frame {best['kernel_ms']:8.2f} ms | trace {best['trace_ms']:8.2f} shade+tail {best['shade_ms']:8.2f} tail beside {best['tail_ms']:7.2f} film {best['film_ms']:6.2f} | {rays / best['kernel_ms'] / 1e3:8.1f} Mrays/s | {same} rays {rays}", flush=True)
run.ref = None
run("tails on the bulk's stream", TAIL_CUS=0)
for fp in (524288, 1048576, 2097152, 4194304, 262144):
    for cus in (32, 64):
        run(f"beside, {cus} CUs, finish from {fp} paths", TAIL_CUS=cus, FINISH_PATHS=fp)
run("tails on the bulk's stream", TAIL_CUS=0)
