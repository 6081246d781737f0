"""CPU-side checks of the C++ host (XML / OBJ / plugin parameters / EXR) and of
the C-ABI library: it loads and exports every symbol include/*.h declares
(no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from nori_amd import NoriError, _capi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scenes"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(nori_(?:hip|host)_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    lib = _capi.load_hip()
    names = _declared("nori_hip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_capi.HIP_PROTOTYPES)
    # one version of header, library and bindings (the structs the library fills carry no size field)
    import re
    declared = int(re.search(r"#define NORI_HIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "nori_hip.h")).read()).group(1))
    assert lib.nori_hip_abi_version() == declared == _capi.HIP_ABI_VERSION


def test_host_library_exports_every_declared_symbol():
    lib = host.load_host()
    names = _declared("nori_host.h")
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(host.HOST_PROTOTYPES)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nori_amd.render import Renderer
    with pytest.raises(NoriError, match="NO_DEVICE"):
        Renderer(0)


def test_cli_gpus_without_gpu_fails_at_no_device(tmp_path):
    """`nori scene.xml --gpus 2` parses the scene, flattens it and asks the library for two GPUs: on a box without any the
    failure is the library's "no HIP device", through NoriException -- not a crash, not a CPU fallback."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nori_amd import _capi
    (tmp_path / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s.xml").write_text("""<scene><integrator type="normals"/><sampler type="independent"><integer name="sampleCount" value="1"/></sampler>
      <camera type="perspective"><integer name="width" value="16"/><integer name="height" value="16"/></camera>
      <mesh type="obj"><string name="filename" value="tri.obj"/></mesh></scene>""")
    exe = os.path.join(_capi.LIB_DIR, "nori")
    for extra in (["--gpus", "2"], ["--gpus", "2", "--split", "sample", "--merge", "reduce"]):
        p = subprocess.run([exe, str(tmp_path / "s.xml")] + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode != 0 and "no HIP device available" in p.stdout + p.stderr, p.stdout + p.stderr
    p = subprocess.run([exe, str(tmp_path / "s.xml"), "--gpus", "0"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "positive integer" in p.stdout + p.stderr


def _write_scene(tmp_path, body, obj=True):
    if obj:
        (tmp_path / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\n"
                                          "f 1/1/1 2/2/1 4/4/1 3/3/1\n")
    p = tmp_path / "s.xml"
    p.write_text(body)
    return str(p)


def test_parser_defaults_quads_transforms(tmp_path):
    path = _write_scene(tmp_path, """<?xml version='1.0' encoding='utf-8'?>
<!-- a comment -->
<scene>
  <integrator type="simple"><point name="position" value="1, 2, 3"/><color name="energy" value="4 5 6"/></integrator>
  <camera type="perspective">
    <transform name="toWorld"><scale value="-1,1,1"/><lookat target="0, 0, 1" origin="0, 0, 0" up="0, 1, 0"/></transform>
  </camera>
  <mesh type="obj"><string name="filename" value="tri.obj"/>
    <transform name="toWorld"><scale value="2,2,2"/><translate value="1,0,0"/></transform></mesh>
  <mesh type="obj"><string name="filename" value="tri.obj"/><bsdf type="microfacet"><color name="kd" value="0.1,0.2,0.3"/></bsdf>
    <emitter type="area"><color name="radiance" value="1,2,3"/></emitter></mesh>
</scene>""")
    sc = host.load_xml(path)
    # defaults: 1280x720, fov 30, gaussian r=2 sd=.5, independent 1 spp, diffuse 0.5 (perspective.cpp:24-36, mesh.cpp:23-29)
    assert (sc.camera.width, sc.camera.height, sc.camera.fov) == (1280, 720, 30.0)
    assert sc.rfilter.type == "gaussian" and sc.rfilter.radius == 2.0 and sc.sample_count == 1
    assert sc.integrator.type == "simple" and sc.integrator.position == (1.0, 2.0, 3.0) and sc.integrator.energy == (4.0, 5.0, 6.0)
    m0, m1 = sc.meshes
    assert m0.bsdf.type == "diffuse" and m0.bsdf.albedo == (0.5, 0.5, 0.5) and m0.radiance is None
    # quad -> 2 triangles (0,1,2) (3,0,2) over de-duplicated vertices (obj.cpp:63-91)
    assert m0.indices.tolist() == [[0, 1, 2], [3, 0, 2]]
    # toWorld = translate * scale (each op left-multiplies, parser.cpp:238-262)
    np.testing.assert_allclose(m0.positions, [[1, 0, 0], [3, 0, 0], [3, 2, 0], [1, 2, 0]])
    np.testing.assert_allclose(m0.normals, [[0, 0, 1]] * 4)
    assert m0.texcoords.shape == (4, 2)
    assert m1.bsdf.type == "microfacet" and m1.radiance == (1.0, 2.0, 3.0)
    assert abs(m1.bsdf.ks - 0.7) < 1e-6 and m1.bsdf.alpha == pytest.approx(0.1)
    # lookat then scale(-1,1,1): left = up x dir = (1,0,0) -> mirrored
    np.testing.assert_allclose(sc.camera.to_world, np.diag([-1, 1, 1, 1]), atol=1e-7)


@pytest.mark.parametrize("body,msg", [
    ("<scene><integrator type='normals'/></scene>", "No camera was specified"),
    ("<scene><camera type='perspective'/></scene>", "No integrator was specified"),
    ("<scene><integrator type='nope'/></scene>", "could not be found"),
    ("<scene><integrator type='normals' foo='1'/></scene>", "unexpected attribute"),
    ("<scene><float name='x'/></scene>", "missing attribute"),
    ("<scene><bogus/></scene>", "unexpected tag"),
    ("<float name='x' value='1'/>", "must be a Nori object"),
    ("<scene><translate value='1,1,1'/></scene>", "transform nodes can only contain"),
    ("<scene><integrator type='diffuse'/></scene>", "Unexpectedly constructed an object"),
    ("<scene><integrator type='normals'><float name='a' value='1x'/></integrator></scene>", "Could not parse floating point"),
    ("<scene><integrator type='simple'/></scene>", "Property 'position' is missing"),
    ("<scene><integrator type='simple'><float name='position' value='1'/></integrator></scene>", "wrong type"),
    ("<scene><mesh type='obj'><string name='filename' value='missing.obj'/></mesh></scene>", "Unable to open OBJ file"),
    ("<scene><integrator type='normals'/><integrator type='ao'/></scene>", "only be one integrator"),
    ("<scene><mesh type='obj'><string name='filename' value='tri.obj'/><bsdf type='diffuse'/><bsdf type='mirror'/></mesh></scene>", "multiple BSDF"),
    ("<scene><integrator type='normals'></scene>", "closing tag"),
])
def test_parser_errors(tmp_path, body, msg):
    path = _write_scene(tmp_path, body)
    with pytest.raises(NoriError, match=msg):
        host.load_xml(path)


def test_exr_files_read_by_an_independent_reader(tmp_path):
    """The EXR bytes the host writes (src/bitmap.cpp:69-96 in the reference: RGB float channels, comments = "Generated by Nori")
    are parsed by a second reader written from the file-layout specification (tests/exr_reader.py): header fields, required
    attributes, alphabetical channel list, an offset table that points exactly at its scan line blocks, and the pixels."""
    from tests.exr_reader import read_exr_rgb
    rng = np.random.default_rng(4)
    for shape in ((17, 23, 3), (1, 1, 3), (40, 3, 3)):
        img = rng.uniform(-2, 50, shape).astype(np.float32)
        img[0, 0] = [np.inf, 0.0, -0.0]
        host.save_images(str(tmp_path / "w"), img)
        hdr, rgb = read_exr_rgb(str(tmp_path / "w.exr"))
        assert np.array_equal(rgb, img) and np.array_equal(np.signbit(rgb), np.signbit(img))
        assert hdr["comments"] == "Generated by Nori" and hdr["compression"] == 0 and hdr["lineOrder"] == 0
        assert hdr["dataWindow"] == hdr["displayWindow"] == (0, 0, shape[1] - 1, shape[0] - 1)
        assert [c[:2] for c in hdr["channels"]] == [("B", 2), ("G", 2), ("R", 2)]
        assert hdr["pixelAspectRatio"] == 1.0 and hdr["screenWindowWidth"] == 1.0 and hdr["screenWindowCenter"] == (0.0, 0.0)


def test_exr_png_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 4, (17, 23, 3)).astype(np.float32)
    host.save_images(str(tmp_path / "a"), img)
    back = host.load_exr(str(tmp_path / "a.exr"))
    assert np.array_equal(back, img)
    raw = open(tmp_path / "a.exr", "rb").read()
    assert raw[:4] == b"\x76\x2f\x31\x01" and b"Generated by Nori" in raw
    png = open(tmp_path / "a.png", "rb").read()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    # decode the PNG (zlib) and check the sRGB curve of common.cpp:166-180
    import struct, zlib
    pos, idat = 8, b""
    while pos < len(png):
        ln, tp = struct.unpack(">I4s", png[pos:pos + 8])
        if tp == b"IDAT":
            idat += png[pos + 8:pos + 8 + ln]
        pos += 12 + ln
    px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(17, 1 + 23 * 3)[:, 1:].reshape(17, 23, 3)
    v = img.astype(np.float64)
    srgb = np.where(v <= 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055)
    assert np.abs(px.astype(int) - np.clip(255 * srgb, 0, 255).astype(int)).max() <= 1


@needs_ref
def test_shipped_scenes_load_unchanged():
    import glob
    ok = 0
    for f in sorted(glob.glob(REF + "/**/*.xml", recursive=True)):
        if "ajax" in f:       # ajax.obj is not shipped with the reference
            with pytest.raises(NoriError, match="Unable to open OBJ file"):
                host.HostRoot(f)
            continue
        r = host.HostRoot(f)
        assert r.class_type in (0, 9)
        (r.scene() if r.class_type == 0 else r.test())
        r.close()
        ok += 1
    assert ok >= 19


@needs_ref
def test_golden_fixtures_are_current():
    """tests/golden/*.npz equal what the host produces from the reference files now."""
    from nori_amd.scene import Scene
    sc = host.load_xml(REF + "/pa5/cbox/cbox_mis.xml")
    g = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "pa5-cbox_mis.npz"))
    assert len(sc.meshes) == len(g.meshes) == 6
    for a, b in zip(sc.meshes, g.meshes):
        assert np.array_equal(a.positions, b.positions) and np.array_equal(a.indices, b.indices)
        assert a.bsdf.type == b.bsdf.type and a.radiance == b.radiance
    assert np.array_equal(sc.camera.to_world, g.camera.to_world) and sc.sample_count == g.sample_count == 256


def test_statistics_match_scipy():
    """nori/hypothesis.h CDFs (C++) vs scipy, via a tiny compiled probe."""
    import subprocess, tempfile
    from scipy import stats
    src = r'''#include <nori/hypothesis.h>
#include <cstdio>
int main(){ for (double t : {-3.0,-0.5,0.0,1.2,4.0}) for (int d : {3,30,99999}) printf("%.12g\n", hypothesis::students_t_cdf(t,d));
for (double x : {0.5,5.0,50.0,300.0}) for (int d : {1,10,199}) printf("%.12g\n", hypothesis::chi2_cdf(x,d)); }'''
    with tempfile.TemporaryDirectory() as td:
        open(td + "/p.cpp", "w").write(src)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "nori_amd/csrc/host"), td + "/p.cpp", "-o", td + "/p"], check=True)
        got = [float(x) for x in subprocess.run([td + "/p"], capture_output=True, text=True, check=True).stdout.split()]
    want = [stats.t.cdf(t, d) for t in (-3.0, -0.5, 0.0, 1.2, 4.0) for d in (3, 30, 99999)]
    want += [stats.chi2.cdf(x, d) for x in (0.5, 5.0, 50.0, 300.0) for d in (1, 10, 199)]
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-12)


def _obj_reference(text):
    """The loader's semantics stated plainly (src/obj.cpp:42-93): serial scan, quads as (0,1,2) (3,0,2),
    corners de-duplicated by their (p, t, n) index triple in first-use order."""
    pos, uv, nrm, keys, ids, faces = [], [], [], {}, [], []
    for line in text.split("\n"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            pos.append([np.float32(x) for x in tok[1:4]])
        elif tok[0] == "vt":
            uv.append([np.float32(x) for x in tok[1:3]])
        elif tok[0] == "vn":
            nrm.append([np.float32(x) for x in tok[1:4]])
        elif tok[0] == "f":
            c = []
            for t in tok[1:5]:
                parts = t.split("/")
                c.append((int(parts[0]), int(parts[1]) if len(parts) > 1 and parts[1] else -1, int(parts[2]) if len(parts) > 2 and parts[2] else -1))
            order = c[:3] + ([c[3], c[0], c[2]] if len(c) == 4 else [])
            for k in order:
                if k not in keys:
                    keys[k] = len(ids)
                    ids.append(k)
                faces.append(keys[k])
    P = np.array([pos[k[0] - 1] for k in ids], np.float32)
    T = np.array([uv[k[1] - 1] for k in ids], np.float32) if uv else None
    N = np.array([nrm[k[2] - 1] for k in ids], np.float32) if nrm else None
    return P, T, N, np.array(faces, np.uint32).reshape(-1, 3)


_SCENE_WITH_MESH = """<scene><integrator type='normals'/><camera type='perspective'/>
<mesh type='obj'><string name='filename' value='%s'/></mesh></scene>"""


@pytest.mark.parametrize("flavour", ["p", "p/t/n", "p//n"])
def test_obj_loader_parallel_slices_match_serial_semantics(tmp_path, flavour):
    """A file large enough (> 4 MB per worker) to be cut into several slices that are scanned in
    parallel: same vertices, same first-use order, same faces as a serial reader."""
    rng = np.random.default_rng(7)
    nv, nf = 60000, 220000
    lines = ["# generated", "o thing"]
    P = rng.uniform(-3, 3, (nv, 3)).astype(np.float32)
    lines += ["v %.9g %.9g %.9g" % tuple(p) for p in P]
    lines += ["vt %.7g %.7g" % tuple(t) for t in rng.uniform(0, 1, (nv // 2, 2))]
    N = rng.normal(size=(nv // 3, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    lines += ["vn %.9g %.9g %.9g" % tuple(n) for n in N.astype(np.float32)]
    for i in range(nf):
        k = 4 if i % 5 == 0 else 3
        a = rng.integers(1, nv + 1, k)
        if flavour == "p":
            lines.append("f " + " ".join(str(x) for x in a) + ("  " if i % 7 == 0 else ""))
        elif flavour == "p/t/n":
            lines.append("f " + " ".join("%d/%d/%d" % (x, 1 + x % (nv // 2), 1 + x % (nv // 3)) for x in a))
        else:
            lines.append("f " + " ".join("%d//%d" % (x, 1 + (x * 7) % (nv // 3)) for x in a) + "\r")
        if i % 50000 == 0:
            lines += ["s off", "usemtl none", ""]
    text = "\n".join(lines)          # no trailing newline on purpose
    if flavour == "p":
        text = "\n".join(l for l in text.split("\n") if not l.startswith(("vt", "vn")))
    if flavour == "p//n":       # a file WITH vt records whose corners carry no t index is an error in the reference too
        text = "\n".join(l for l in text.split("\n") if not l.startswith("vt"))
    text = text + ("\n# pad" * 1) + ("\n#" + "x" * 120) * (1 + (9 << 20) // 122 if len(text) < (9 << 20) else 1)
    (tmp_path / "big.obj").write_text(text)
    assert len(text) > (8 << 20)
    sc = host.load_xml(_write_scene(tmp_path, _SCENE_WITH_MESH % "big.obj", obj=False))
    m = sc.meshes[0]
    Pr, Tr, Nr, Fr = _obj_reference(text)
    assert np.array_equal(m.indices, Fr)
    assert np.array_equal(m.positions, Pr)
    if flavour == "p":
        assert m.normals is None and m.texcoords is None
    else:
        np.testing.assert_allclose(m.normals, Nr, rtol=0, atol=2e-7)      # renormalised by toWorld (identity)
        if flavour == "p/t/n":
            assert np.array_equal(m.texcoords, Tr)


@pytest.mark.parametrize("face,msg", [("f 1/1/1/1 2 3", "Invalid vertex data"), ("f 1 x 3", "Could not parse integer value"),
                                      ("f 1 2", "Could not parse integer value"), ("f 1 2 9", "range")])
def test_obj_loader_errors(tmp_path, face, msg):
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n" + face + "\n")
    with pytest.raises(Exception, match=msg):
        host.load_xml(_write_scene(tmp_path, _SCENE_WITH_MESH % "bad.obj", obj=False))


def test_committed_bench_line_follows_the_contract():
    """The latest committed default bench line (profiles/r*_bench_default.json, written by `python bench.py` on the GPU box in the round's
    profile set) carries every field the driver and the judge read, with consistent values; since round 6 its tree is device-built."""
    import glob
    import json
    import re
    root = os.path.dirname(os.path.dirname(__file__))
    files = [f for f in glob.glob(os.path.join(root, "profiles", "r*_bench_default.json")) if re.match(r"r\d+_\d+_bench_default\.json$", os.path.basename(f))]
    path = sorted(files, key=lambda f: tuple(int(x) for x in re.match(r"r(\d+)_(\d+)_", os.path.basename(f)).groups()))[-1]
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "accel"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert d["config"]["workload"] == "pa4-cbox-path_mis" and (d["config"]["width"], d["config"]["height"], d["config"]["spp"]) == (1024, 1024, 256)
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bound_evidence", "hbm_measured_frac"):
        assert k in r, k
    assert r["bound"] in ("valu", "hbm") and r["unit"] in ("Tops/s", "GB/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_ms"] < d["ms_per_step"] and r["traffic"] is not None and r["traffic"] > 0      # the counters of the line's own build were on file
    assert r["bound_evidence"]["source"].startswith("profiles/") and os.path.exists(os.path.join(root, r["bound_evidence"]["source"]))
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0
    assert d["parity"]["ok"] is True and d["parity"]["rays_cpu"] == d["parity"]["rays_gpu"]
    assert d["accel"]["builder"] == "ploc" and d["accel"]["built_on_device"] == 1 and d["accel"]["n_references"] >= d["accel"]["n_triangles"]
