"""The BASELINE.json configurations THEMSELVES in the driver-run GPU suite -- `workloads.load("c2" | "c4" | "c5")` at their full
geometry (327,680 / 10,004,450 triangles, 2048 x 2048 frame), a few samples per pixel so that the CPU oracle finishes in seconds --
not smaller stand-ins: wavefront engine with the production node loop (32-B records + hand-written loop for the BVH2 trees, wide
nodes for the terrain) against Oracle(use_bvh=True).  Bar: ray counts EQUAL (bit-identical paths), frame within the SURVEY 8(d)
image contract.  Semantics held: Accel::rayIntersect (src/accel.cpp:23-43), renderBlock / render (src/main.cpp:27-56)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.backends import Oracle
from tests.test_gpu_parity import assert_image_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,spp,triangles,size,layout", [
    ("c2-ao-icosphere", 4, 327_682, 1024, 2),          # BASELINE configs[1]: BVH2, 32-B records
    ("c4-table-mis", 2, None, 2048, 2),                # configs[3]: tree deeper than the LDS stack, microfacet + dielectric
    ("c5-terrain-10m", 2, 9_999_394, 1024, 4),        # configs[4]: 10 M triangles, wide nodes (667 MB tree)
])
def test_baseline_config_at_full_geometry_matches_the_oracle(name, spp, triangles, size, layout):
    from nori_amd import workloads
    from nori_amd.render import Renderer
    wl = workloads.load(name, spp=spp)
    sc = wl.scene
    assert (sc.camera.width, sc.camera.height) == (size, size)
    r = Renderer(0).upload(sc)                         # the builder bench.py uses
    r.set_option("engine", "wavefront")
    info = r.accel_info()
    if triangles is not None:
        assert info["n_triangles"] == triangles
    assert info["node_children"] == layout
    if layout == 2:
        assert info["node_records_32b"] == 1           # the hand-written loop's tree form
    B, sb = r.render_host()
    assert sb["engine"] == 1 and sb["n_invalid"] == 0
    o = Oracle(sc, use_bvh=True)
    A, sa = o.render_host(threads=os.cpu_count() or 1)
    assert sb["n_camera_samples"] == sa["n_camera_samples"] == size * size * spp
    for k in ("n_closest_rays", "n_shadow_rays"):
        assert int(sa[k]) == int(sb[k]), (name, k, sa[k], sb[k])
    assert_image_parity(A, B, r.border, f"{name} {size}x{size}x{spp}, {info['n_triangles']} triangles")
    r.close(); o.close()


def test_device_builder_at_ten_million_triangles_gives_the_host_trees_hits():
    """builder = auto on the 10 M-triangle terrain: above 2^22 triangles it is the device's PLOC builder with its treelet sweeps on
    (lbvh.hip: a wave per treelet, tree data handed over at agent scope) -- the path no small scene takes.  A valid tree gives the
    scan's answers, and so does the host's SAH tree (checked against the oracle above): every field of the intersection record equal
    for a grid of vertical rays that reaches every cell of the height field (each triangle is hit by at least one ray or shares its
    leaf with one that is), for slanted rays, for shadow queries; and the frame of a render has the same bits, ray for ray."""
    from nori_amd import workloads
    from nori_amd._capi import RAY_DTYPE
    from nori_amd.render import Renderer
    from tests.test_gpu_parity import ITS_FIELDS
    sc = workloads.load("c5-terrain-10m", spp=1).scene
    host = Renderer(0).upload(sc, builder=0)
    dev = Renderer(0).upload(sc, builder=2)
    ih, idv = host.accel_info(), dev.accel_info()
    assert idv["n_triangles"] == ih["n_triangles"] == 9_999_394 and idv["node_children"] == ih["node_children"] == 4
    assert (idv["n_nodes"], idv["sah_cost"]) != (ih["n_nodes"], ih["sah_cost"])      # two builders, two trees
    assert idv["build_ms"] < 0.25 * ih["build_ms"], (idv["build_ms"], ih["build_ms"])
    n = 3200                                              # 10.2 M vertical rays over the 2236 x 2236 cells: ~2 per triangle
    xs = (np.arange(n, dtype=np.float32) + 0.5) / n * 2.0 - 1.0
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    rays = np.zeros(n * n, dtype=RAY_DTYPE)
    rays["o"] = np.stack([X.ravel(), np.full(n * n, 1.4, np.float32), Z.ravel()], axis=1)
    rays["d"] = np.array([0.0, -1.0, 0.0], np.float32)
    rays["mint"], rays["maxt"] = 1e-4, np.inf
    rng = np.random.default_rng(7)
    slanted = np.zeros(1 << 20, dtype=RAY_DTYPE)
    slanted["o"] = np.stack([rng.uniform(-1, 1, 1 << 20), rng.uniform(0.3, 1.0, 1 << 20), rng.uniform(-1, 1, 1 << 20)], axis=1).astype(np.float32)
    target = np.stack([rng.uniform(-1, 1, 1 << 20), np.zeros(1 << 20), rng.uniform(-1, 1, 1 << 20)], axis=1)
    d = target - slanted["o"].astype(np.float64)
    slanted["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    slanted["mint"], slanted["maxt"] = 1e-4, np.inf
    for batch in (rays, slanted):
        a, b = host.intersect(batch), dev.intersect(batch)
        assert (a["tri"] != 0xffffffff).mean() > 0.5
        for k in ITS_FIELDS:
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(host.intersect(batch, True)["mesh"], dev.intersect(batch, True)["mesh"])
    hit = np.unique(host.intersect(rays)["tri"])
    assert hit.size > 0.8 * idv["n_triangles"], hit.size          # the grid reaches (nearly) every triangle
    for r in (host, dev):
        r.set_option("engine", "wavefront")
    A, sa = host.render_host()
    B, sb = dev.render_host()
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays"):
        assert sa[k] == sb[k], k
    assert np.array_equal(A, B)
    host.close(); dev.close()


def test_shading_arithmetic_stays_inside_its_verified_domain_on_the_baseline_configs():
    """exact_rcp / exact_div / exact_sqrt (rt_types.h) carry no fallback: outside their verified domains IEEE's inf / 0 may come
    out as NaN, and a sample the reference would count with value 0 would be DROPPED (src/block.cpp:63-67).  The golden scenes
    are checked in tests/test_gpu_parity.py; here the workloads the numbers are quoted on -- headline, C2, C4 at 2048^2, C5 at
    10 M triangles, 32 samples per pixel each -- run through the counting build: zero operands outside the domains, zero NaN /
    infinite results, zero dropped samples."""
    lib = os.path.join(ROOT, "nori_amd", "lib", "libnori_hip_count.so")
    assert os.path.exists(lib), "libnori_hip_count.so missing: __graft_entry__.build() makes it"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "excursion_probe.py"), "32", "workloads"], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, NORI_HIP_LIBRARY=lib), timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert [row["scene"] for row in rows] == ["pa4-cbox-path_mis", "c2-ao-icosphere", "c4-table-mis", "c5-terrain-10m"]
    for row in rows:
        print("[excursions]", row)
        assert row["rays"] > 32 * 1024 * 1024 and row["engine"] == "wavefront"
        assert (row["rcp_out_of_domain"], row["div_out_of_domain"], row["sqrt_out_of_domain"], row["fallbacks"], row["invalid_samples"]) == (0, 0, 0, 0, 0), row
