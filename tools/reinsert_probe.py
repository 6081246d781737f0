"""CPU probe of the device builder's round-6 stages (triangle splitting up front, parallel re-insertion): the builder's steps run as loops in
the emulation harness (tests/emu/emu_builder.h), hits against the oracle's brute force and node / triangle tests per ray of a small render.
   python tools/reinsert_probe.py cbox|table|ico|terrain [terrain_triangles]
What profiles/r6_20_split_cpu_counts.txt was made with (its rows for Karras & Aila's spare-area priority came from the same script before the
priority was changed to the box's volume)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from nori_amd import workloads
from tests import scenes
from tests.backends import Emu, Oracle


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def probe(name, sc, variants, check_hits=True):
    rays = None
    for label, env in variants:
        env2 = dict(env, NORI_HIP_ACCEL_LAYOUT="bvh2")
        t0 = time.time()
        e = with_env(env2, lambda: Emu(sc))
        tb = time.time() - t0
        info = e.accel_info()
        _, st = e.render_host(count_traversal=True)
        nr = st["n_closest_rays"] + st["n_shadow_rays"]
        ok = ""
        if check_hits:
            if rays is None:
                rays = scenes.random_rays(4000, seed=11)
                ref = Oracle(sc).intersect(rays)
            got = e.intersect(rays)
            ok = "hits==brute force" if all(np.array_equal(ref[k], got[k], equal_nan=ref[k].dtype.kind == "f") for k in ref.dtype.names) else "HITS DIFFER"
        print(f"{name:10s} {label:34s} refs {info['n_references']:8d} nodes {info['n_nodes']:8d} depth {info['max_depth']:3d} sah {info['sah_cost']:7.3f} "
              f"node/ray {st['n_node_tests'] / nr:6.2f} tri/ray {st['n_tri_tests'] / nr:5.2f}  build {tb:5.1f}s {ok}", flush=True)
        e.close()


P = {"NORI_EMU_BUILDER": "ploc"}
V = [("host SAH + SBVH + reinsert", {"NORI_EMU_BUILDER": "sah"}),
     ("PLOC + sweeps only", dict(P, NORI_HIP_REINSERT_ITERS="0", NORI_HIP_SPLIT_BUDGET="0")),
     ("PLOC, no splitting", dict(P, NORI_HIP_SPLIT_BUDGET="0")),
     ("PLOC as shipped", dict(P))]
for it, stride in ((8, 1), (16, 1), (64, 8)):
    V.append((f"  re-insertion {it} rounds stride {stride}", dict(P, NORI_HIP_SPLIT_BUDGET="0", NORI_HIP_REINSERT_ITERS=str(it), NORI_HIP_REINSERT_STRIDE=str(stride))))
for scale, inside in ((0, 0), (4, 0), (4, 4), (1.5, 4)):
    V.append((f"  splitting scale {scale} inside {inside}", dict(P, NORI_HIP_SPLIT_SCALE=str(scale), NORI_HIP_SPLIT_INSIDE=str(inside))))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cbox"
    if which == "cbox": probe("cbox", workloads.load("pa4-cbox-path_mis", width=64, height=64, spp=4).scene, V)
    if which == "table": probe("table", workloads.load("c4-table-mis", width=64, height=64, spp=4).scene, V)
    if which == "ico": probe("ico", workloads.load("c2-ao-icosphere", width=64, height=64, spp=4).scene, V, check_hits=False)
    if which == "terrain": probe("terrain", workloads.load("c5-terrain-10m", width=48, height=48, spp=4, triangles=int(sys.argv[2]) if len(sys.argv) > 2 else 200000).scene, V, check_hits=False)
