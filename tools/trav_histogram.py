"""Which nodes and which leaf pair records do the rays of a workload visit?  CPU harness (emulated device code built with
-DNORI_TRAV_HISTOGRAM, see rt_trace.h); prints how much of the traffic the hottest records carry -- the sizing argument
for what wf_extend keeps in LDS.
    g++ -O2 -std=c++17 -fPIC -ffp-contract=off -pthread -shared -DNORI_TRAV_HISTOGRAM=1 -o /tmp/libnori_emu_hist.so \
        tests/emu/emu.cpp nori_amd/csrc/device/scene_prep.cpp tools/trav_hist_impl.cpp
    python tools/trav_histogram.py /tmp/libnori_emu_hist.so [workload] [width] [spp]
Visits of records that are cached in the harness's LDS image (links with kTopBit) are not counted: with NORI_EMU_TOP_IMAGE=0 the
histogram is of all visits, with NORI_EMU_TOP_NODES=n of what an image of n nodes leaves to memory; NORI_EMU_NODEQ=0 walks the 64-B
nodes instead of the 32-B records."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import numpy as np
lib_path = sys.argv[1]
from tests import backends
backends._emu = None
_orig = backends._make
backends._make = lambda d, n: lib_path if n == "libnori_emu.so" else _orig(d, n)
from nori_amd import workloads
from tests.backends import Emu
wl = sys.argv[2] if len(sys.argv) > 2 else "pa4-cbox-path_mis"
w = int(sys.argv[3]) if len(sys.argv) > 3 else 128
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sc = workloads.load(wl, width=w, height=w, spp=spp).scene
e = Emu(sc)
info = e.accel_info()
_, st = e.render_host(count_traversal=True)
lib = C.CDLL(lib_path)
nr = st["n_closest_rays"] + st["n_shadow_rays"]
print(info, "rays", nr, "node tests/ray", st["n_node_tests"] / nr, "tri tests/ray", st["n_tri_tests"] / nr)
for kind, name, n in ((0, "nodes", info["n_nodes"] + 1), (1, "pairs", 1 << 20)):
    h = np.zeros(n, np.uint64)
    lib.nori_trav_hist_get(kind, h.ctypes.data_as(C.c_void_p), C.c_uint32(n))
    tot = h.sum(); order = np.argsort(h)[::-1]; cum = np.cumsum(h[order]) / max(tot, 1)
    print(f"{name}: {int((h > 0).sum())} visited, {int(tot)} visits ({tot / nr:.2f} per ray)")
    for k in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 256, 512, 1024):
        if k <= len(cum): print(f"   hottest {k:5d}: {cum[k - 1]:.3f} of the visits")
    print("   hottest ids:", [(int(i), round(float(h[i]) / nr, 3)) for i in order[:24]])
