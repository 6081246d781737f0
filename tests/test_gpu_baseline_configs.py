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
