"""The reference's own goldens executed THROUGH THE GPU PATH (SURVEY.md §8d
parity check 2): every <test> XML of scenes/pa4/tests and scenes/pa5/tests and
the warptest cases, with BSDF::sample/pdf, Warp::* and Integrator::Li running in
the HIP kernels.  Also runs them through the C++ host's own <test> objects."""
import os

import numpy as np
import pytest

from nori_amd.scene import Bsdf
from tests import scenes, stat_harness as sh

pytestmark = pytest.mark.gpu


def _check(results):
    bad = [(i, info) for i, (ok, info) in enumerate(results) if not ok]
    assert not bad, bad


@pytest.fixture(scope="module")
def util(renderer_factory):
    return renderer_factory(scenes.soup_scene(2))


@pytest.mark.parametrize("name", ["pa4-test-mesh-furnace", "pa4-test-mesh", "pa5-test-furnace", "pa5-test-direct"])
def test_scene_ttests_on_device(renderer_factory, name):
    _check(sh.run_ttest_scenes(lambda sc: renderer_factory(sc), sh.load_test(name)))


def test_microfacet_ttest_on_device(util):
    _check(sh.run_ttest_bsdf(util, sh.load_test("pa5-ttest-microfacet")))


def test_microfacet_chi2_on_device(util):
    _check(sh.run_chi2test(util, sh.load_test("pa5-chi2test-microfacet")))


@pytest.mark.parametrize("name,param", [("square", 0), ("tent", 0), ("disk", 0), ("uniform_sphere", 0),
                                        ("uniform_hemisphere", 0), ("cosine_hemisphere", 0),
                                        ("beckmann", 0.05), ("beckmann", 0.3), ("beckmann", 0.8)])
def test_warptest_on_device(util, name, param):
    ok, info = sh.run_warptest(util, name, param)
    assert ok, info


def test_warptest_microfacet_brdf_on_device(util):
    ok, info = sh.run_warptest(util, "microfacet_brdf", bsdf=Bsdf("microfacet", (0.5, 0.5, 0.5), 0.3), wi=np.float32([0, 0, 1]))
    assert ok, info


def test_host_test_objects_and_cli(tmp_path):
    """C++ host: synthesize a <test> XML + OBJ, run it via nori_host_test_run and the `nori` / `warptest` CLIs."""
    import subprocess
    from nori_amd import host, _capi
    (tmp_path / "box.obj").write_text(
        "v -1 -1 -1\nv 1 -1 -1\nv 1 1 -1\nv -1 1 -1\nv -1 -1 1\nv 1 -1 1\nv 1 1 1\nv -1 1 1\n"
        "f 1 2 3 4\nf 8 7 6 5\nf 1 5 6 2\nf 2 6 7 3\nf 3 7 8 4\nf 5 1 4 8\n")
    (tmp_path / "furnace.xml").write_text("""<?xml version="1.0"?>
<test type="ttest">
  <string name="references" value="2, 2"/>
  <integer name="sampleCount" value="50000"/>
  <scene><integrator type="path_mis"/>
    <camera type="perspective"><float name="fov" value="10"/><integer name="width" value="1"/><integer name="height" value="1"/></camera>
    <mesh type="obj"><string name="filename" value="box.obj"/>
      <bsdf type="diffuse"><color name="albedo" value="0.5, 0.5, 0.5"/></bsdf>
      <emitter type="area"><color name="radiance" value="1, 1, 1"/></emitter></mesh></scene>
  <scene><integrator type="path_mats"/>
    <camera type="perspective"><float name="fov" value="10"/><integer name="width" value="1"/><integer name="height" value="1"/></camera>
    <mesh type="obj"><string name="filename" value="box.obj"/>
      <bsdf type="diffuse"><color name="albedo" value="0.5, 0.5, 0.5"/></bsdf>
      <emitter type="area"><color name="radiance" value="1, 1, 1"/></emitter></mesh></scene>
</test>""")
    r = host.HostRoot(str(tmp_path / "furnace.xml"))
    assert r.run_test()
    r.close()
    exe = os.path.join(_capi.LIB_DIR, "nori")
    p = subprocess.run([exe, str(tmp_path / "furnace.xml"), "--no-gui"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Passed 2/2 tests." in p.stdout
    wt = os.path.join(_capi.LIB_DIR, "warptest")
    for args in (["cosine_hemisphere"], ["beckmann", "0.3"], ["microfacet_brdf", "0.3", "0.5"]):
        p = subprocess.run([wt] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
    # a scene render through the CLI writes EXR + PNG next to the XML
    (tmp_path / "scene.xml").write_text("""<scene><integrator type="path_mis"/>
    <sampler type="independent"><integer name="sampleCount" value="8"/></sampler>
    <camera type="perspective"><float name="fov" value="60"/><integer name="width" value="48"/><integer name="height" value="32"/>
      <transform name="toWorld"><lookat origin="0,0,0.5" target="0,0,-1" up="0,1,0"/></transform></camera>
    <mesh type="obj"><string name="filename" value="box.obj"/>
      <bsdf type="diffuse"><color name="albedo" value="0.5, 0.5, 0.5"/></bsdf>
      <emitter type="area"><color name="radiance" value="1, 1, 1"/></emitter></mesh></scene>""")
    p = subprocess.run([exe, str(tmp_path / "scene.xml")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    img = host.load_exr(str(tmp_path / "scene.exr"))
    assert img.shape == (32, 48, 3)
    # furnace: radiance 1/(1-a) = 2 everywhere
    assert abs(img.mean() - 2.0) < 0.05
    assert (tmp_path / "scene.png").stat().st_size > 100
