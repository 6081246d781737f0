"""CPU probe of the device BVH builders (their steps run in the emulation harness, tests/emu/emu_builder.h):
hits against the oracle's brute force and node / triangle tests per ray of a small render, per builder.
   python tools/builder_probe.py [terrain_triangles]"""
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


def probe(name, sc, check_hits=True):
    rays = None
    for label, env in (("host SAH", {"NORI_EMU_BUILDER": "sah"}), ("radix tree", {"NORI_EMU_BUILDER": "lbvh"}),
                       ("PLOC r=8", {"NORI_EMU_BUILDER": "ploc", "NORI_HIP_PLOC_RADIUS": "8"}),
                       ("PLOC r=16", {"NORI_EMU_BUILDER": "ploc", "NORI_HIP_PLOC_RADIUS": "16"}),
                       ("PLOC r=32", {"NORI_EMU_BUILDER": "ploc", "NORI_HIP_PLOC_RADIUS": "32"})):
        for layout in ("bvh2", "bvh4q"):
            env2 = dict(env, NORI_HIP_ACCEL_LAYOUT=layout)
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
            print(f"{name:14s} {label:10s} {layout:5s} nodes {info['n_nodes']:8d} depth {info['max_depth']:3d} sah {info['sah_cost']:7.2f} "
                  f"node tests/ray {st['n_node_tests'] / nr:6.2f} tri tests/ray {st['n_tri_tests'] / nr:5.2f}  build {tb:5.1f}s {ok}", flush=True)
            e.close()


if __name__ == "__main__":
    n_terrain = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    probe("cbox", workloads.load("pa4-cbox-path_mis", width=48, height=48, spp=4).scene)
    probe("terrain", workloads.load("c5-terrain-10m", width=48, height=48, spp=4, triangles=n_terrain).scene, check_hits=n_terrain <= 50000)
