"""What a reordering of the paths between bounces is worth to wf_extend, counted on the CPU before any kernel is written: the
wave-level model of tests/emu/emu_wavesim.h (the real per-lane traversal under the kernel's voting rules) over a sample of the
tiles of the FULL-resolution headline frame -- the coherence of a pass depends on the footprint of a tile and on the samples per
pixel, so the frame is sampled, not shrunk -- with the paths of the passes after the first sorted inside windows of W paths by a key.
    python tools/sort_sim.py [tile_stride] [spp] [levels] [workload]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from nori_amd import workloads
from tests.backends import Emu, emu_lib
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wl = sys.argv[4] if len(sys.argv) > 4 else "pa4-cbox-path_mis"
sc = workloads.load(wl, spp=spp).scene
e = Emu(sc); lib = emu_lib()
lib.emu_wave_sim.restype = C.c_int
COST = dict(node=51, leaf=100, refill=130, trip=50)
def run(name, window=0, kind=0, bits=0, refill=32, leaf=16, inner=24):
    pol = (C.c_int * 12)(refill, leaf, inner, 0, 1024, 0, 0, 0, stride, window, kind, bits)
    out = np.zeros((nl, 16), np.uint64)
    n = lib.emu_wave_sim(e._h, C.c_uint32(spp), C.c_uint32(nl), pol, out.ctypes.data_as(C.c_void_p))
    rows = []
    for k in range(min(n, nl)):
        rays, trips, ns, nlan, ls, llan, rf = [float(x) for x in out[k][:7]]
        cost = ns * COST["node"] + ls * COST["leaf"] + rf * COST["refill"] + trips * COST["trip"]
        rows.append((rays, ns, nlan, ls, llan, rf, cost))
    tot = np.array(rows).sum(axis=0); later = np.array(rows[1:]).sum(axis=0)
    def fmt(r): return f"rays {r[0]:.3g} node steps/ray {r[1] / r[0]:.4f} ({r[2] / max(r[1], 1):4.1f} lanes) leaf {r[3] / r[0]:.4f} ({r[4] / max(r[3], 1):4.1f}) refills {r[5] / r[0]:.4f} wave-instr/ray {r[6] / r[0]:6.2f}"
    print(f"{name:40s} first: {fmt(rows[0])}\n{'':40s} later: {fmt(later)}", flush=True)
    return later[6] / later[0]
base = run("generation order (shipped)")
import os
CASES = (("octant, window 256", dict(window=256, kind=1)),
                 ("octant(A), window 4096", dict(window=4096, kind=3, bits=0)),
                 ("cell 2b + octant, window 4096", dict(window=4096, kind=2, bits=2)),
                 ("cell 3b + octant, window 65536", dict(window=65536, kind=2, bits=3)),
                 ("octant + cell 4b, window 65536", dict(window=65536, kind=3, bits=4)),
                 ("shadow bit + cell 3b + octant, 65536", dict(window=65536, kind=4, bits=3)),
                 ("cell 3b + dir 5b, window 65536", dict(window=65536, kind=5, bits=3)),
                 ("octa dir 8b + cell 3b, window 65536", dict(window=65536, kind=6, bits=3)),
                 ("cell 3b + octa dir 8b, window 65536", dict(window=65536, kind=7, bits=3)),
                 ("octa dir 8b + cell 3b, window 1M", dict(window=1 << 20, kind=6, bits=3)),
                 ("cell 4b + octa dir 8b, window 1M", dict(window=1 << 20, kind=7, bits=4)),
                 ("cell 5b + octant, global", dict(window=1 << 30, kind=2, bits=5)),
                 ("cell 5b + octa dir 8b, global", dict(window=1 << 30, kind=7, bits=5)))
for name, kw in CASES:
    c = run(name, **kw); print(f"      -> later passes {100 * (c / base - 1):+.1f} % wave instructions per ray")
