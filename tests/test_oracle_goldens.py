"""Pin the CPU oracle against every golden the reference's own tests hold for
this path: the Student-t scene tests, the microfacet t-test and
chi^2 test, and the warptest chi^2 cases -- plus the published PCG32 known
answers.  Fixtures: tests/golden/tests/*.json (tools/make_goldens.py)."""
import numpy as np
import pytest

from nori_amd.scene import Bsdf
from tests import stat_harness as sh
from tests.backends import Oracle


def _check(results):
    bad = [(i, info) for i, (ok, info) in enumerate(results) if not ok]
    assert not bad, bad


@pytest.mark.parametrize("name", ["pa4-test-mesh-furnace", "pa4-test-mesh", "pa5-test-furnace", "pa5-test-direct"])
def test_scene_ttests(name):
    """scenes/pa4/tests/test-mesh-furnace.xml:16 (whitted: 1.5, 1.8), test-mesh.xml:4-5,
    scenes/pa5/tests/test-furnace.xml:17 (1/(1-a)), test-direct.xml:4-7 (polygonal lights)."""
    meta = sh.load_test(name)
    _check(sh.run_ttest_scenes(lambda sc: Oracle(sc), meta))


def test_microfacet_ttest():
    """scenes/pa5/tests/ttest-microfacet.xml:4-5: five incidence angles, five albedos."""
    _check(sh.run_ttest_bsdf(Oracle, sh.load_test("pa5-ttest-microfacet")))


def test_microfacet_chi2():
    """scenes/pa5/tests/chi2test-microfacet.xml:5-24: three (alpha, IOR, kd) configurations x 5 directions."""
    _check(sh.run_chi2test(Oracle, sh.load_test("pa5-chi2test-microfacet")))


@pytest.mark.parametrize("name,param", [("square", 0), ("tent", 0), ("disk", 0), ("uniform_sphere", 0),
                                        ("uniform_hemisphere", 0), ("cosine_hemisphere", 0),
                                        ("beckmann", 0.05), ("beckmann", 0.3), ("beckmann", 0.8)])
def test_warptest(name, param):
    """src/warptest.cpp:79-82,109-215 (CLI names and recipe)."""
    ok, info = sh.run_warptest(Oracle, name, param)
    assert ok, info


@pytest.mark.parametrize("alpha,kd", [(0.05, 0.0), (0.3, 0.5), (0.8, 0.9)])
def test_warptest_microfacet_brdf(alpha, kd):
    """warptest microfacet_brdf mode: create_microfacet_bsdf(alpha, kd, angle 0), src/warptest.cpp:289-301."""
    ok, info = sh.run_warptest(Oracle, "microfacet_brdf", bsdf=Bsdf("microfacet", (kd, kd, kd), alpha),
                               wi=np.float32([0, 0, 1]))
    assert ok, info


def test_pcg32_known_answers():
    """ext/pcg32 is un-vendored; anchor the restated generator on the published PCG
    reference output (pcg32-demo of the PCG C library, seed (42, 54), round 1):
    0xa15c02b7 0x7b47f409 0xba1d3330 0x83d2f293 0xbfa4784b 0xcbed606e."""
    expect = np.array([0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e], np.uint32)
    assert np.array_equal(Oracle.pcg32_uints(42, 54, False, 6), expect)
    # default-constructed state (0x853c49e6748fea9b, 0xda3e39cb94b95bdb): first output of pcg32-global-demo
    assert Oracle.pcg32_uints(0, 0, True, 1)[0] == 0x152ca78d
    f = Oracle.pcg32_floats(np.array([42], np.uint64), np.array([54], np.uint64), 6)[0]
    assert np.array_equal(f, ((expect >> 9) | 0x3f800000).view(np.float32) - np.float32(1))
    assert (f >= 0).all() and (f < 1).all()


def test_fresnel_limits():
    """src/common.cpp:259-288: normal incidence ((n1-n2)/(n1+n2))^2, TIR -> 1, equal IOR -> 0."""
    r0 = ((1.5 - 1.0) / (1.5 + 1.0)) ** 2
    assert abs(Oracle.fresnel(1.0, 1.0, 1.5) - r0) < 1e-6
    assert abs(Oracle.fresnel(-1.0, 1.0, 1.5) - r0) < 1e-6
    assert Oracle.fresnel(-0.1, 1.0, 1.5) == 1.0
    assert Oracle.fresnel(0.3, 1.3, 1.3) == 0.0


def test_filter_tables():
    """src/block.cpp:18-27 + src/rfilter.cpp: closed forms at the tabulated positions."""
    from nori_amd.scene import RFilter
    from tests import scenes
    sc = scenes.soup_scene(1)
    x = 2.0 * np.arange(32) / 32
    sc.rfilter = RFilter("gaussian")
    g = Oracle(sc).filter_table()
    np.testing.assert_allclose(g[:32], np.maximum(0, np.exp(-2 * x * x) - np.exp(-8.0)), rtol=1e-5, atol=1e-7)
    assert g[32] == 0
    sc.rfilter = RFilter("tent")
    np.testing.assert_allclose(Oracle(sc).filter_table()[:32], 1 - np.arange(32) / 32, rtol=1e-6)
    sc.rfilter = RFilter("box")
    o = Oracle(sc)
    assert (o.filter_table()[:32] == 1).all() and o.border == 0
    sc.rfilter = RFilter("mitchell")
    m = Oracle(sc).filter_table()
    assert abs(m[0] - (6 - 2 / 3) / 6) < 1e-6 and abs(m[16] - (1 / 18)) < 1e-6


def test_specified_libm_vs_host_libm_renders(tmp_path):
    """The oracle evaluates sin / cos / log / exp by a pinned specification (oracle_libm.h) instead of calling the host's
    libm as the reference does; the device implements the same specification.  The independent witness: the SAME oracle
    built with the host libm's functions (oracle/liboracle_glibc.so) renders the same images within the image contract of
    SURVEY 8(d) -- in practice to ~1e-9, with equal ray counts -- so the specification stands in for what the reference
    computes, and a mistake shared by its two implementations could not hide behind their bit-equality."""
    import os
    import subprocess
    import sys
    import numpy as np
    from nori_amd.render import develop_host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "liboracle_glibc.so"], check=True, capture_output=True)
    for kind, w, h, spp in (("cornell", 64, 48, 8), ("table", 48, 48, 4)):
        frames, rays = [], []
        for lib in (None, os.path.join(root, "oracle", "liboracle_glibc.so")):
            env = dict(os.environ)
            env.pop("NORI_ORACLE_LIBRARY", None)
            if lib:
                env["NORI_ORACLE_LIBRARY"] = lib
            out = tmp_path / f"{kind}_{'glibc' if lib else 'spec'}.npy"
            p = subprocess.run([sys.executable, os.path.join(root, "tools", "oracle_render.py"), kind, str(w), str(h), str(spp), "path_mis", str(out)],
                               capture_output=True, text=True, cwd=root, env=env, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            frames.append(np.load(out)); rays.append(p.stdout.split())
        a, b = develop_host(frames[0], 2), develop_host(frames[1], 2)
        rel = (np.abs(a - b) / np.maximum(np.abs(a), 1e-2)).max(axis=-1)
        assert (rel <= 1e-3).mean() >= 0.999 and rel.mean() <= 1e-4, (kind, float(rel.mean()))
        assert rays[0] == rays[1], kind
