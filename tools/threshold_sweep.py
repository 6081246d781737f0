"""wf_extend's voting thresholds (refill / leaf / repeat; wavefront.hip reads the environment per render call) swept on one
workload in one process: WORKLOAD, SPP, CONFIGS="r/l/p r/l/p ..." (default: a grid around the shipped 32 / 16 / 24).
Prints trace / shade ms of two timed renders per setting."""
import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads

sc = workloads.load(os.environ.get("WORKLOAD", "c5-terrain-10m"), spp=int(os.environ.get("SPP", 128))).scene
r = Renderer(0).upload(sc, builder=int(os.environ.get("BUILDER", 0)))
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
default = "32/16/24 40/16/24 48/16/24 32/24/24 32/32/24 32/16/32 32/16/40 40/24/32 48/32/40 24/12/16 32/16/24"
for cfg in os.environ.get("CONFIGS", default).split():
    refill, leaf, rep = cfg.split("/")
    os.environ.update(NORI_HIP_WF_REFILL=refill, NORI_HIP_WF_LEAF=leaf, NORI_HIP_WF_INNER_REPEAT=rep)
    out = []
    for i in range(3):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if i: out.append((round(st["trace_ms"], 2), round(st["shade_ms"], 2), round(st["kernel_ms"], 1)))
    print(f"refill {refill:>2} leaf {leaf:>2} repeat {rep:>2}   trace / shade / frame ms: " + "   ".join("%.2f / %.2f / %.1f" % o for o in out), flush=True)
