"""Spatial splits in the host builder (scene_prep.cpp, build_tree_spatial) counted on the CPU before any GPU time is spent: the tree's
SAH cost and size, and what wf_extend's wave model (tests/emu/emu_wavesim.h: the real per-lane traversal under the kernel's voting
rules) executes per ray -- node steps, triangle steps, lanes -- for a small frame of the workload.
    python tools/sbvh_probe.py [workload] [width] [spp] [budgets...]"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import numpy as np
from nori_amd import workloads
from tests.backends import Emu, emu_lib
wl = sys.argv[1] if len(sys.argv) > 1 else "pa4-cbox-path_mis"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 96
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
budgets = [float(x) for x in sys.argv[4:]] or [0.0, 0.1, 0.3, 1.0]
lib = emu_lib(); lib.emu_wave_sim.restype = C.c_int
COST = dict(node=51, leaf=100, refill=130, trip=50)
base = None
for b in budgets:
    os.environ["NORI_HIP_SBVH"] = str(b)
    sc = workloads.load(wl, width=w, height=w, spp=spp).scene
    e = Emu(sc)
    info = e.accel_info() if hasattr(e, "accel_info") else {}
    nl = 6
    pol = (C.c_int * 12)(32, 16, 24, 0, 1024, 0, 0, 0, 1, 0, 0, 0)
    out = np.zeros((nl, 16), np.uint64)
    n = lib.emu_wave_sim(e._h, C.c_uint32(spp), C.c_uint32(nl), pol, out.ctypes.data_as(C.c_void_p))
    tot = out[:min(n, nl)].sum(axis=0).astype(np.float64)
    rays, trips, ns, nlan, ls, llan, rf, rfl, lns, lls = tot[:10]
    cost = (ns * COST["node"] + ls * COST["leaf"] + rf * COST["refill"] + trips * COST["trip"]) / rays
    if base is None: base = cost
    print(f"{wl} budget {b:4.2f}: {info} | per ray: lane node steps {lns / rays:6.2f} lane triangle-pair steps {lls / rays:5.2f} | wave node steps {ns / rays:.4f} ({nlan / max(ns, 1):4.1f} lanes) "
          f"leaf {ls / rays:.4f} ({llan / max(ls, 1):4.1f}) | VALU wave-instr per ray {cost:6.2f} ({100 * (cost / base - 1):+.1f} %)", flush=True)
