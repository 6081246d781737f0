"""Scheduling policies of wf_extend compared on the CPU: the wave-level model of tests/emu/emu_wavesim.h counts node steps,
leaf steps, refills and trips per ray for the passes of a small frame.  python tools/wave_sim.py [width] [spp] [levels]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from nori_amd import workloads
from tests.backends import Emu, emu_lib
w = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sc = workloads.load("pa4-cbox-path_mis", width=w, height=w, spp=spp).scene
e = Emu(sc); lib = emu_lib()
lib.emu_wave_sim.restype = C.c_int
COST = dict(node=51, leaf=100, refill=130, trip=50, pend=60)      # VALU instructions per wave-level event (ISA of the shipped kernel: the hand-written node loop on 32-B records; a trip includes the 24 of the ray's plane coefficients)
def run(name, refill=32, leaf=16, inner=24, postpone=0, chunk=1024, sort=0, pend=0, pretest=0):
    pol = (C.c_int * 12)(refill, leaf, inner, postpone, chunk, sort, pend, pretest, 1, 0, 0, 0)      # [8..11]: tile stride, sort window, key kind, cell bits (emu_wave_sim)
    out = np.zeros((nl, 16), np.uint64)
    n = lib.emu_wave_sim(e._h, C.c_uint32(spp), C.c_uint32(nl), pol, out.ctypes.data_as(C.c_void_p))
    tot = out[:min(n, nl)].sum(axis=0).astype(np.float64)
    rays, trips, ns, nlan, ls, llan, rf, rfl, lns, lls, nidle, nleafw, pr, prl, pre, prel = tot
    cost = ns * COST["node"] + (ls + pre) * COST["leaf"] + rf * COST["refill"] + trips * COST["trip"] + pr * COST["pend"]
    print(f"{name:34s} per ray: node steps {ns / rays:.4f} ({nlan / max(ns, 1):4.1f} lanes) leaf {ls / rays:.4f} ({llan / max(ls, 1):4.1f}) refills {rf / rays:.4f} trips {trips / rays:.4f}"
          f" pre-tests {pre / rays:.4f} | in node steps: {nidle / max(ns, 1):4.1f} lanes idle, {nleafw / max(ns, 1):4.1f} wait at a leaf | lane-node/ray {lns / rays:.2f} lane-leaf/ray {lls / rays:.2f} | VALU wave-instr per ray {cost / rays:6.2f}", flush=True)
    return cost / rays
base = run("shipped (32/16/24)")
for kw in (dict(pretest=1), dict(pretest=1, leaf=8), dict(pretest=1, refill=24), dict(pend=4), dict(pend=8), dict(pend=16), dict(pend=8, refill=40), dict(pend=8, refill=24), dict(pend=12, refill=32, leaf=12), dict(refill=24), dict(refill=40), dict(leaf=8), dict(leaf=24), dict(inner=16), dict(inner=32), dict(postpone=1), dict(postpone=1, leaf=24), dict(postpone=1, leaf=32),
           dict(postpone=1, leaf=24, refill=24), dict(sort=1), dict(sort=1, postpone=1, leaf=24)):
    c = run(str(kw), **kw); print(f"      -> {100 * (c / base - 1):+.1f} %")
