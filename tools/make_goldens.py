#!/usr/bin/env python
"""Compile the reference's scene and test XML files into flattened fixtures.

/root/reference does not exist on the GPU box, so everything the `-m gpu`
tests, smoke() and bench.py need from it is generated HERE by loading the
unmodified XML/OBJ files through our own C++ host (libnori_host.so: XML parser,
OBJ loader, plugin constructors) and saving the resulting `nori_scene_desc`
as .npz (nori_amd.scene.Scene.save_npz).  No reference file is copied.

    python tools/make_goldens.py [/root/reference]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nori_amd import host  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

SCENES = {
    # fixture name: (xml relative to scenes/, overrides)
    "pa1-bunny": ("pa1/bunny.xml", {}),
    "pa4-cbox-distributed": ("pa4/cbox/cbox-distributed.xml", {}),
    # BASELINE config 3: geometry of cbox-distributed.xml, path_mis, 1024x1024, 256 spp
    "pa4-cbox-path_mis": ("pa4/cbox/cbox-distributed.xml", {"integrator": "path_mis", "width": 1024, "height": 1024, "spp": 256}),
    "pa4-cbox-whitted": ("pa4/cbox/cbox-whitted.xml", {}),
    "pa4-motto-dielectric": ("pa4/motto/motto-dielectric.xml", {}),
    "pa5-cbox_mis": ("pa5/cbox/cbox_mis.xml", {}),
    "pa5-table_mis": ("pa5/table/table_mis.xml", {}),
    "pa5-veach_mis": ("pa5/veach_mi/veach_mis.xml", {}),
    # the other loadable shipped scenes: the arrays of a fixture above under another integrator / sample count / BSDF --
    # stored as parameters + the digest of the arrays (Scene.save_npz(base=...))
    "pa4-motto-diffuse": ("pa4/motto/motto-diffuse.xml", {"variant": True}),
    "pa5-cbox_ems": ("pa5/cbox/cbox_ems.xml", {"variant": True}),
    "pa5-cbox_mats": ("pa5/cbox/cbox_mats.xml", {"variant": True}),
    "pa5-table_ems": ("pa5/table/table_ems.xml", {"variant": True}),
    "pa5-table_mats": ("pa5/table/table_mats.xml", {"variant": True}),
    "pa5-veach_ems": ("pa5/veach_mi/veach_ems.xml", {"variant": True}),
    "pa5-veach_mats": ("pa5/veach_mi/veach_mats.xml", {"variant": True}),
}
TESTS = ["pa4/tests/test-mesh-furnace.xml", "pa4/tests/test-mesh.xml", "pa5/tests/test-furnace.xml",
         "pa5/tests/test-direct.xml", "pa5/tests/ttest-microfacet.xml", "pa5/tests/chi2test-microfacet.xml"]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    os.makedirs(os.path.join(OUT, "tests"), exist_ok=True)
    full = {}      # digest of a scene's arrays -> the fixture that stores them
    for name, (xml, ov) in SCENES.items():
        sc = host.load_xml(os.path.join(ref, "scenes", xml))
        if "integrator" in ov:
            sc.integrator.type = ov["integrator"]
        sc.camera.width = ov.get("width", sc.camera.width)
        sc.camera.height = ov.get("height", sc.camera.height)
        sc.sample_count = ov.get("spp", sc.sample_count)
        for i, m in enumerate(sc.meshes):
            m.name = f"{name}:{i}"
        path = os.path.join(OUT, name + ".npz")
        digest = sc.geometry_digest()
        base = full.get(digest) if ov.get("variant") else None
        sc.save_npz(path, base=base)
        if base is None:
            full[digest] = name
        print(f"{name}: {sc.n_triangles} tris, {os.path.getsize(path) / 1024:.0f} KiB" + (f" (arrays of {base})" if base else ""))
    for xml in TESTS:
        r = host.HostRoot(os.path.join(ref, "scenes", xml))
        t = r.test()
        base = xml.replace("/tests/", "-").replace("/", "-").replace(".xml", "")
        meta = {"source": "scenes/" + xml, "kind": t.kind, "significance_level": t.significance_level,
                "sample_count": t.sample_count, "angles": t.angles, "references": t.references,
                "test_count": t.test_count, "resolution": t.resolution, "min_exp_frequency": t.min_exp_frequency,
                "bsdfs": [{"type": b.type, "albedo": list(map(float, b.albedo)), "alpha": b.alpha,
                           "int_ior": b.int_ior, "ext_ior": b.ext_ior} for b in t.bsdfs],
                "scenes": []}
        for i, sc in enumerate(t.scenes):
            fn = f"{base}__scene{i}.npz"
            sc.save_npz(os.path.join(OUT, "tests", fn))
            meta["scenes"].append(fn)
        with open(os.path.join(OUT, "tests", base + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        print(f"{base}: {t.kind}, {len(t.bsdfs)} bsdfs, {len(t.scenes)} scenes")
        r.close()


if __name__ == "__main__":
    main()
