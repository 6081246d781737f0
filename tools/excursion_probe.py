"""Renders the golden scenes (every integrator the reference's tests pin, every shipped material) through
libnori_hip_count.so -- the product sources built with -DNORI_COUNT_EXCURSIONS -- and prints, per scene, how many operands
left the domain on which exact_rcp / exact_div / exact_sqrt (rt_types.h) are verified bit-identical to the IEEE
operations, and how many results the full-range fallback recomputed.  One JSON line per scene.
   NORI_HIP_LIBRARY=nori_amd/lib/libnori_hip_count.so python tools/excursion_probe.py [spp] [workloads]
`workloads`: the BASELINE.json configurations at their full geometry and frame size instead (headline, C2, C4, C5; wavefront engine)."""
import json, os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
from tests import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
jobs = [(f"cornell_box/{integ}", scenes.cornell_box(96, 96, spp, integ)) for integ in ("normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis")]
jobs += [("pa4-cbox-path_mis", workloads.load("pa4-cbox-path_mis", 128, 128, spp).scene), ("pa5-table_mis", workloads.load("c4-table-mis", 128, 128, spp).scene)]
engines = ("megakernel", "wavefront")
if len(sys.argv) > 2 and sys.argv[2] == "workloads":
    jobs = [(n, None) for n in ("pa4-cbox-path_mis", "c2-ao-icosphere", "c4-table-mis", "c5-terrain-10m")]
    engines = ("wavefront",)
for engine in engines:
    for name, sc in jobs:
        if sc is None:
            sc = workloads.load(name, spp=spp).scene      # built one at a time: the terrain is 10 M triangles
        r = Renderer(0).upload(sc)
        r.set_option("engine", engine)
        r.excursions(reset=True)
        frame = torch.zeros(r.frame_shape(), dtype=torch.float32, device="cuda:0")
        st = r.render_into(frame)
        torch.cuda.synchronize()
        ex = r.excursions()
        print(json.dumps({"scene": name, "engine": engine, "rays": int(st["n_closest_rays"] + st["n_shadow_rays"]), "rcp_out_of_domain": ex[0],
                          "div_out_of_domain": ex[1], "sqrt_out_of_domain": ex[2], "fallbacks": ex[3], "invalid_samples": int(st.get("n_invalid", 0))}), flush=True)
        r.close()
