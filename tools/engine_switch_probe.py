"""Where the "auto" engine switch belongs: megakernel against wavefront engine on the headline scene at job sizes around the rule's
2^24 camera samples per call (nori_hip.hip).  python tools/engine_switch_probe.py  ->  one line per size."""
import sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
import os
JOBS = [("pa4-cbox-path_mis", w, s) for w, s in ((64, 4), (64, 16), (128, 16), (256, 4), (256, 8), (256, 16), (512, 8), (512, 16), (512, 32), (512, 64), (1024, 32), (1024, 64))]
JOBS += [("c1-bunny-normals", 512, 1), ("c1-bunny-normals", 1024, 1), ("c2-ao-icosphere", 256, 4), ("c2-ao-icosphere", 512, 4)]
for name, width, spp in JOBS:
    sc = workloads.load(name, width, width, spp).scene
    out = []
    for engine in ("megakernel", "wavefront"):
        r = Renderer(0).upload(sc); r.set_option("engine", engine)
        f = torch.zeros(r.frame_shape(), device="cuda")
        best = None
        for i in range(4):
            f.zero_(); st = r.render_into(f)
            if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
        out.append((engine, best["kernel_ms"], (best["n_closest_rays"] + best["n_shadow_rays"]) / best["kernel_ms"] / 1e3))
        r.close()
    samples = width * width * spp
    print(f"{name} {width}x{width}x{spp} = 2^{samples.bit_length() - 1} samples: " + ", ".join(f"{e} {ms:.2f} ms ({mr:.0f} Mrays/s)" for e, ms, mr in out)
          + f" -> {'wavefront' if out[1][1] < out[0][1] else 'megakernel'} (rule: {'wavefront' if samples >= (1 << 24) else 'megakernel'})", flush=True)
