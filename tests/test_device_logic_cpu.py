"""CPU checks of the DEVICE code's per-lane logic (tests/emu: the same rt_*.h
headers the HIP kernels are built from, compiled with g++) against the oracle.
The real kernels are checked by tests/test_gpu_parity.py on the GPU box."""
import os

import numpy as np
import pytest

from nori_amd.scene import Bsdf
from tests import scenes
from tests.backends import Emu, Oracle

ITS_FIELDS = ["p", "t", "uv", "sh_s", "sh_t", "sh_n", "geo_s", "geo_t", "geo_n", "mesh", "tri"]


def _assert_its_equal(a, b):
    for k in ITS_FIELDS:
        assert np.array_equal(a[k], b[k]), f"intersection field {k} differs"


@pytest.mark.parametrize("n_tris,seed", [(1, 3), (7, 4), (300, 5), (5000, 6)])
def test_bvh_traversal_equals_brute_force_soup(n_tris, seed):
    sc = scenes.soup_scene(n_tris, seed)
    rays = scenes.random_rays(20000 if n_tris < 1000 else 4000, seed=seed + 10)
    o, e = Oracle(sc), Emu(sc)
    _assert_its_equal(o.intersect(rays), e.intersect(rays))
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])


def test_bvh_traversal_cornell_axis_aligned_and_ties():
    """Axis-aligned walls (flat boxes), shared edges and bounded segments."""
    sc = scenes.cornell_box(16, 16, 1)
    rays = scenes.random_rays(30000, seed=7, extent=0.9)
    rays["o"] += np.float32([0, 1, 0])
    # axis-aligned directions incl. exact zeros, and rays lying in wall planes
    rays["d"][:3000] = np.float32([0, -1, 0])
    rays["d"][3000:6000] = np.float32([1, 0, 0])
    rays["o"][6000:7000, 1] = 0.0
    rays["d"][6000:7000, 1] = 0.0
    rays["d"][6000:7000] /= np.linalg.norm(rays["d"][6000:7000], axis=1, keepdims=True)
    rays["maxt"][7000:9000] = np.random.default_rng(1).uniform(0.1, 2.0, 2000).astype(np.float32)
    o, e = Oracle(sc), Emu(sc)
    _assert_its_equal(o.intersect(rays), e.intersect(rays))
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])


def test_duplicate_triangles_tie_rule():
    """Coincident triangles: the linear scan keeps the LAST one (mesh.cpp:75 accepts t <= maxt)."""
    from nori_amd.scene import Mesh, Scene
    v = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    f = np.uint32([[0, 1, 2]])
    sc = scenes.soup_scene(1)
    sc.meshes = [Mesh(v, f), Mesh(v, np.uint32([[0, 1, 2], [0, 1, 2]])), Mesh(v + np.float32([0, 0, 1]), f)]
    rays = scenes.random_rays(2000, seed=3, extent=3.0, target_extent=0.6)
    o, e = Oracle(sc), Emu(sc)
    a, b = o.intersect(rays), e.intersect(rays)
    _assert_its_equal(a, b)
    hit0 = a["mesh"] != 0xFFFFFFFF
    assert hit0.any() and set(np.unique(a["mesh"][hit0])) <= {1, 2}
    assert (a["tri"][a["mesh"] == 1] == 1).all()


def _mixed_soup(seed, n_small=3000, n_big=60, coincident=True):
    """Many small triangles and a few that span the scene (the inputs spatial splits are for), some of them twice."""
    from nori_amd.scene import Mesh
    v1, f1 = scenes.triangle_soup(n_small, seed, size=0.06)
    v2, f2 = scenes.triangle_soup(n_big, seed + 1, extent=0.3, size=1.2)
    sc = scenes.soup_scene(1)
    sc.meshes = [Mesh(v1, f1, name="small"), Mesh(v2, f2, name="big")]
    if coincident:
        sc.meshes.append(Mesh(v2[:30], f2[:10], name="big again"))      # equal t on a duplicated triangle: the larger index wins
    return sc


@pytest.mark.parametrize("seed,margin,budget", [(21, 1.0, 1.0), (22, 0.99, 0.3), (23, 1.0, 4.0)])
def test_spatial_splits_give_the_scans_answers(monkeypatch, seed, margin, budget):
    """build_tree_spatial (scene_prep.cpp): a triangle cut by a spatial split hangs in several leaves.  A leaf step tests the whole
    triangle whichever leaf it is reached through (src/mesh.cpp:39-76: same operands, same t, u, v), and the scan's tie rule (equal t:
    the larger index wins, src/accel.cpp:30-40) replaces a hit by itself -- so every field of the record is the scan's, for closest
    hits and shadow queries, on 64-B nodes and on the 32-B records, also with coincident triangles in different meshes."""
    sc = _mixed_soup(seed)
    rays = scenes.random_rays(20000, seed=seed + 10)
    o = Oracle(sc)
    monkeypatch.setenv("NORI_HIP_SBVH", "0")
    plain = Emu(sc).accel_info()
    monkeypatch.setenv("NORI_HIP_SBVH", str(budget)); monkeypatch.setenv("NORI_HIP_SBVH_MARGIN", str(margin))
    e = Emu(sc)
    info = e.accel_info()
    assert info["n_leaves"] > plain["n_leaves"] and info["total_bytes"] > plain["total_bytes"]      # references were duplicated
    assert info["sah_cost"] < plain["sah_cost"]
    a = o.intersect(rays)
    _assert_its_equal(a, e.intersect(rays))
    assert (a["mesh"] == 2).any() and not (a["mesh"] == 1)[np.isin(a["tri"], np.arange(10))].any()       # the second copy wins its ties
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])
    monkeypatch.setenv("NORI_EMU_NODEQ", "0")       # ... and on the 64-B nodes
    _assert_its_equal(a, Emu(sc).intersect(rays))


def test_spatial_splits_on_wide_nodes_and_in_renders(monkeypatch):
    """The same tree collapsed into wide (BVH4) nodes, and whole renders: the frame and the ray counts of the tree without spatial
    splits (the hits are the scan's either way, so every path is the same path)."""
    monkeypatch.setenv("NORI_HIP_SBVH", "1.0"); monkeypatch.setenv("NORI_HIP_SBVH_MARGIN", "1.0")
    sc = _mixed_soup(31, 2000, 40)
    rays = scenes.random_rays(8000, seed=41)
    e = _with_layout("bvh4q", lambda: Emu(sc))
    assert e.accel_info()["node_children"] == 4
    _assert_its_equal(Oracle(sc).intersect(rays), e.intersect(rays))
    from tests.scenes import cornell_box
    from nori_amd.scene import Mesh
    cb = cornell_box(24, 24, 2, "path_mis")
    v, f = scenes.triangle_soup(12, 5, extent=0.3, size=0.9)
    cb.meshes.append(Mesh((v * 0.4 + np.float32([0, 1.0, 0])).astype(np.float32), f, bsdf=Bsdf("diffuse", (0.4, 0.5, 0.6)), name="shards"))
    with_splits, sa = Emu(cb).render_host()
    monkeypatch.setenv("NORI_HIP_SBVH", "0")
    without, sb = Emu(cb).render_host()
    assert np.array_equal(with_splits, without)
    assert sa["n_closest_rays"] == sb["n_closest_rays"] and sa["n_shadow_rays"] == sb["n_shadow_rays"]


def _check_bvh2_invariants(e, n_refs_min):
    """A flattened BVH2 tree (64-B nodes, rt_types.h): every inner node is reached exactly once from the root, a child's box lies
    inside its parent's (centre / half-extent: [c - h, c + h]), every leaf link covers its own range of pair records and the ranges
    tile [0, n_pairs) without overlap.  Returns the number of leaves."""
    n64 = e.node_records()[0]
    n = n64.shape[0]
    links = n64[:, 12:14].copy().view(np.uint32).astype(np.int64)
    links[links >= 2 ** 31] -= 2 ** 32
    # q0 = (lc.x, lc.y, rc.x, rc.y), q1 = (lc.z, rc.z, lh.z, rh.z), q2 = (lh.x, lh.y, rh.x, rh.y)
    c = np.stack([np.stack([n64[:, 0], n64[:, 1], n64[:, 4]], 1), np.stack([n64[:, 2], n64[:, 3], n64[:, 5]], 1)], 1).astype(np.float64)
    h = np.stack([np.stack([n64[:, 8], n64[:, 9], n64[:, 6]], 1), np.stack([n64[:, 10], n64[:, 11], n64[:, 7]], 1)], 1).astype(np.float64)
    lo, hi = c - h, c + h
    seen = np.zeros(n, np.int32); ranges = []
    stack = [(0, None, None)]
    while stack:
        i, plo, phi = stack.pop()
        seen[i] += 1
        for k in (0, 1):
            if plo is not None:                                 # this node's two boxes lie inside the box its parent holds for it
                assert (lo[i, k] >= plo - 1e-6 * (1 + np.abs(plo))).all() and (hi[i, k] <= phi + 1e-6 * (1 + np.abs(phi))).all(), i
            link = int(links[i, k])
            if link >= 0:
                stack.append((link, lo[i, k], hi[i, k]))
            else:
                code = ~link
                ranges.append((code >> 3, (code & 7) + 1))
    assert (seen == 1).all(), "an inner node is unreachable or reached twice"
    ranges.sort()
    pos = ranges[0][0]
    assert pos in (0, 1)
    for first, cnt in ranges:
        assert first == pos, "leaf ranges overlap or leave a gap"
        pos += cnt
    assert 2 * (pos - ranges[0][0]) >= n_refs_min
    return len(ranges)


def test_tree_invariants_after_spatial_splits_and_reinsertion(monkeypatch):
    """The structure itself, not only what rays see of it: inner nodes reachable exactly once, children inside their parents' boxes,
    leaf ranges tiling the pair records -- for the top-down tree, with spatial splits forced, and after re-insertion."""
    from nori_amd.scene import Scene
    table = Scene.load_npz(os.path.join(os.path.dirname(__file__), "golden", "pa5-table_mis.npz"))
    table.camera.width = table.camera.height = 16; table.sample_count = 1
    for sc, n_tris in ((_objects_on_planes(), 5006), (_mixed_soup(23, coincident=False), 3060), (table, 22766)):
        counts = {}
        for name, env in (("plain", {"NORI_HIP_SBVH": "0", "NORI_HIP_REINSERT": "0"}), ("splits", {"NORI_HIP_SBVH": "1.0", "NORI_HIP_SBVH_MARGIN": "1.0", "NORI_HIP_REINSERT": "0"}),
                          ("default", {})):
            for k in ("NORI_HIP_SBVH", "NORI_HIP_SBVH_MARGIN", "NORI_HIP_REINSERT"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            e = Emu(sc)
            info = e.accel_info()
            leaves = _check_bvh2_invariants(e, n_tris)
            assert leaves == info["n_leaves"] == info["n_nodes"] + 1
            counts[name] = info
        assert counts["splits"]["n_leaves"] > counts["plain"]["n_leaves"]
        assert counts["default"]["sah_cost"] <= counts["plain"]["sah_cost"]


@pytest.mark.parametrize("builder,env", [("lbvh", {}), ("ploc", {"NORI_HIP_REINSERT_ITERS": "0"}), ("ploc", {}), ("ploc", {"NORI_HIP_REINSERT_ITERS": "12", "NORI_HIP_REINSERT_STRIDE": "1"}),
                                         ("ploc", {"NORI_HIP_SPLIT_BUDGET": "1.0", "NORI_HIP_SPLIT_SCALE": "0", "NORI_HIP_SPLIT_INSIDE": "0"})])
def test_tree_invariants_of_the_device_builders_trees(builder, env):
    """The same structural check on what the DEVICE builders' steps emit (lbvh_steps.h run as loops, tests/emu/emu_builder.h): radix tree,
    PLOC + sweeps, and after parallel re-insertion -- every winner's move relinks five nodes while others move elsewhere in the same
    iteration; a torn tree would leave a node unreachable, reached twice, or a leaf range uncovered."""
    from nori_amd.scene import Scene
    table = Scene.load_npz(os.path.join(os.path.dirname(__file__), "golden", "pa5-table_mis.npz"))
    table.camera.width = table.camera.height = 16; table.sample_count = 1
    for sc, n_tris in ((_objects_on_planes(), 5006), (_mixed_soup(23, coincident=False), 3060), (table, 22766)):
        e = _with_env(dict(env, NORI_EMU_BUILDER=builder, NORI_HIP_ACCEL_LAYOUT="bvh2"), lambda: Emu(sc))
        leaves = _check_bvh2_invariants(e, n_tris)
        assert leaves == e.accel_info()["n_nodes"] + 1


def _objects_on_planes():
    """Small objects between planes many times their size -- the pa5 table scene's proportions."""
    from nori_amd.scene import Mesh
    v1, f1 = scenes.triangle_soup(5000, 51, extent=0.5, size=0.05)
    sc = scenes.soup_scene(1)
    sc.meshes = [Mesh(v1, f1, name="objects")]
    for k, (y, e) in enumerate(((-0.3, 8.0), (2.5, 12.0), (-6.0, 20.0))):
        qv, qf = scenes.quad((-e, y, -e), (-e, y, e), (e, y, e), (e, y, -e))
        sc.meshes.append(Mesh(qv, qf, name=f"plane{k}"))
    return sc


def test_tree_optimised_by_reinsertion_gives_the_scans_answers(monkeypatch):
    """optimize_tree_reinsertion (scene_prep.cpp): inner nodes taken out and their subtrees re-inserted where the SAH cost grows least --
    kept when the cost drops by 2 %, which it does when small objects stand on planes many times their size (the top-down builder's
    first splits bin the centres of the whole scene).  Leaves are untouched, inner boxes are unions: a valid tree, the scan's answers
    field by field; no deeper than the tree it came from (or 15 levels); a tree it cannot improve by 2 % is left as it was."""
    sc = _objects_on_planes()
    rays = scenes.random_rays(20000, seed=61, extent=1.5, target_extent=0.3)
    rays["d"][:2000] = np.float32([0, -1, 0])
    o = Oracle(sc)
    monkeypatch.setenv("NORI_HIP_REINSERT", "0")
    plain = Emu(sc).accel_info()
    monkeypatch.delenv("NORI_HIP_REINSERT")
    e = Emu(sc)
    info = e.accel_info()
    assert info["sah_cost"] < 0.95 * plain["sah_cost"], (info["sah_cost"], plain["sah_cost"])
    assert info["n_nodes"] == plain["n_nodes"] and info["n_leaves"] == plain["n_leaves"] and info["total_bytes"] == plain["total_bytes"]
    assert info["max_depth"] <= max(plain["max_depth"], 15)
    a = o.intersect(rays)
    assert (a["mesh"] != 0xFFFFFFFF).mean() > 0.5
    _assert_its_equal(a, e.intersect(rays))
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])
    wide = _with_layout("bvh4q", lambda: Emu(sc))
    _assert_its_equal(a, wide.intersect(rays))
    # the reference's Cornell box (the headline scene): 0.1 % to gain -- below the threshold, the tree is the top-down builder's
    from nori_amd.scene import Scene
    cb = Scene.load_npz(os.path.join(os.path.dirname(__file__), "golden", "pa4-cbox-path_mis.npz"))
    cb.camera.width = cb.camera.height = 16; cb.sample_count = 1
    kept = Emu(cb).accel_info()
    monkeypatch.setenv("NORI_HIP_REINSERT", "0")
    assert Emu(cb).accel_info()["sah_cost"] == kept["sah_cost"]


def test_empty_scene():
    from nori_amd.scene import Scene
    sc = scenes.soup_scene(1)
    sc.meshes = []
    rays = scenes.random_rays(64)
    assert (Emu(sc).intersect(rays)["mesh"] == 0xFFFFFFFF).all()
    assert (Oracle(sc).intersect(rays)["mesh"] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("integ", ["normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis"])
def test_li_bitwise(integ):
    """Same seeds -> same radiance, bit for bit (both sides use glibc libm here)."""
    sb = [Bsdf("mirror"), Bsdf("dielectric")] if integ in ("whitted", "path_mis") else \
        [Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")]
    sc = scenes.cornell_box(16, 16, 1, integ, sphere_bsdfs=sb)
    sc.integrator.position, sc.integrator.energy = (0, 1.5, 0.5), (30, 30, 30)
    o, e = Oracle(sc, use_bvh=True), Emu(sc)
    rng = np.random.default_rng(11)
    ps = rng.uniform(0, 16, (4000, 2)).astype(np.float32)
    rays = o.sample_rays(ps)
    assert np.array_equal(rays, e.sample_rays(ps))
    ss = rng.integers(0, 2 ** 62, 4000, dtype=np.uint64)
    sq = rng.integers(0, 2 ** 62, 4000, dtype=np.uint64)
    a, b = o.li(rays, ss, sq), e.li(rays, ss, sq)
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and a.max() > 0


@pytest.mark.parametrize("integ,rf", [("path_mis", "gaussian"), ("path_ems", "mitchell"), ("whitted", "tent"), ("normals", "box")])
def test_render_matches_oracle(integ, rf):
    from nori_amd.scene import RFilter
    sc = scenes.cornell_box(40, 24, 4, integ, rfilter=RFilter(rf))   # 40x24: ragged 16-px tiles and 32-px blocks
    o, e = Oracle(sc, use_bvh=True), Emu(sc)
    A, sa = o.render_host()
    B, sb = e.render_host()
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_invalid"):
        assert sa[k] == sb[k], k
    assert sa["n_camera_samples"] == 40 * 24 * 4
    # identical per-sample weights and radiance; only the float summation order differs
    np.testing.assert_allclose(B, A, rtol=2e-5, atol=1e-6)


def test_render_tile_and_sample_split_sum_to_whole():
    sc = scenes.cornell_box(40, 24, 6, "path_mis")
    e = Emu(sc)
    whole, _ = e.render_host()
    parts = sum(e.render_host(tile_mod=3, tile_rem=r)[0] for r in range(3))
    np.testing.assert_allclose(parts, whole, rtol=1e-5, atol=1e-6)
    parts = e.render_host(spp_count=2, spp_begin=0)[0] + e.render_host(spp_count=4, spp_begin=2)[0]
    np.testing.assert_allclose(parts, whole, rtol=1e-5, atol=1e-6)
    o = Oracle(sc, use_bvh=True)
    oparts = sum(o.render_host(tile_mod=3, tile_rem=r)[0] for r in range(3))
    np.testing.assert_allclose(oparts, whole, rtol=2e-5, atol=1e-6)


def test_bsdf_and_warp_twins_bitwise():
    rng = np.random.default_rng(5)
    n = 20000
    s = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    for name, param in [("square", 0), ("tent", 0), ("disk", 0), ("uniform_sphere", 0), ("uniform_hemisphere", 0),
                        ("cosine_hemisphere", 0), ("beckmann", 0.3)]:
        a, b = Oracle.warp(name, s, param), Emu.warp(name, s, param)
        assert np.array_equal(a, b), name
        assert np.array_equal(Oracle.warp_pdf(name, a, param), Emu.warp_pdf(name, a, param)), name
    wi = rng.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(n, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    for b in [Bsdf("diffuse", (0.2, 0.5, 0.7)), Bsdf("mirror"), Bsdf("dielectric"), Bsdf("dielectric", int_ior=1.33, ext_ior=1.0),
              Bsdf("microfacet", (0.1, 0.2, 0.15), 0.1, 1.5), Bsdf("microfacet", (0.4, 0.2, 0.3), 0.6, 1.8, 1.3)]:
        for x, y in zip(Oracle.bsdf_sample(b, wi, s), Emu.bsdf_sample(b, wi, s)):
            assert np.array_equal(x, y, equal_nan=True), b
        assert np.array_equal(Oracle.bsdf_eval(b, wi, wo), Emu.bsdf_eval(b, wi, wo), equal_nan=True)
        assert np.array_equal(Oracle.bsdf_pdf(b, wi, wo), Emu.bsdf_pdf(b, wi, wo), equal_nan=True)
    st = rng.integers(0, 2 ** 63, 100, dtype=np.uint64)
    assert np.array_equal(Oracle.pcg32_floats(st, st[::-1].copy(), 64), Emu.pcg32_floats(st, st[::-1].copy(), 64))


def test_fuzz_intersect_on_emulated_device_code():
    """tests/fuzz_intersect.py (random scene shapes incl. slivers, duplicates, coincident centroids, big
    and tiny scales; special rays) through the device's traversal code compiled for the CPU + the SAH
    builder, against the oracle's linear scan: bit-identical intersection records."""
    from tests import fuzz_intersect
    from tests.backends import Emu

    class EmuRenderer:
        def __init__(self, device):
            self.e = None

        def upload(self, sc, builder=0):
            self.e = Emu(sc)
            return self

        def intersect(self, rays, shadow=False):
            return self.e.intersect(rays, shadow)

        def close(self):
            self.e.close()

    hits = sum(fuzz_intersect.one_round(seed, EmuRenderer, n_rays=5000) for seed in list(range(60)) + [542])
    assert hits > 10000


def test_packed_triangle_pair_and_node_tests_equal_the_scalar_statements():
    """rt_trace.h: tri_pair_test (two Moeller-Trumbore tests on packed f32) and slab_two (both child boxes)
    give, bit for bit, what the scalar tri_test / the slab test of bbox.h give -- random triangles incl.
    collinear and padding ones, zero direction components, flat boxes, three coordinate scales."""
    from tests.backends import emu_lib
    lib = emu_lib()
    assert lib.emu_packed_vs_scalar(400000, 12345) == 0
    assert lib.emu_packed_vs_scalar(100000, 999) == 0


@pytest.mark.parametrize("integ", ["normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis"])
def test_wavefront_records_walk_equals_the_direct_path_loop(integ):
    """wf_records.h: a path walked the way the wavefront engine does -- shadow ray, continuation ray, the answers
    through the 16-B hit record, the path through the flag / state records every vertex -- gives bit for bit
    the radiance of the direct per-lane loop (the megakernel's), for every integrator and BSDF."""
    sb = [Bsdf("mirror"), Bsdf("dielectric")] if integ in ("whitted", "path_mis") else [Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")]
    sc = scenes.cornell_box(32, 32, 1, integ, sphere_bsdfs=sb)
    sc.integrator.position, sc.integrator.energy = (0, 1.5, 0.5), (30, 30, 30)
    e = Emu(sc)
    n = 6000
    ps = np.random.default_rng(5).uniform(0, 32, (n, 2)).astype(np.float32)
    rays = e.sample_rays(ps)
    ss = np.arange(n, dtype=np.uint64) * 7 + 1
    sq = np.arange(n, dtype=np.uint64) + 11
    a = e.li(rays, ss, sq)
    b = np.zeros((n, 3), np.float32)
    from tests.backends import ptr
    assert e.lib.emu_li_records(e._h, ptr(rays), n, ptr(ss), ptr(sq), ptr(b)) == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a.sum(1) > 0).mean() > 0.01          # the comparison is not vacuous (path_mats: only light hits count)
    e.close()


def test_specified_transcendentals():
    """sin / cos / log / exp of the render path follow ONE specification (binary64 arithmetic without FMA, one
    rounding to binary32), implemented twice -- rt_math.h for the device, oracle_libm.h for the
    oracle.  The two agree bit for bit, and both equal the correctly rounded value (numpy float64 -> float32) on
    every one of 4 M arguments per function over the ranges the path uses."""
    from tests.backends import Emu, Oracle
    rng = np.random.default_rng(3)
    n = 4_000_000
    args = {
        "sin": rng.uniform(0, 2 * np.pi, n).astype(np.float32), "cos": rng.uniform(0, 2 * np.pi, n).astype(np.float32),
        "log": np.concatenate([1.0 - rng.uniform(0, 1, n // 2), rng.uniform(1e-30, 1e30, n // 2)]).astype(np.float32),
        "exp": np.concatenate([-rng.exponential(8.0, n // 2), rng.uniform(-104, 88, n // 2)]).astype(np.float32),
    }
    exact = {"sin": np.sin, "cos": np.cos, "log": np.log, "exp": np.exp}
    for op, x in args.items():
        x = x[np.isfinite(x) & ((x > 0) | (op != "log"))]
        a, b = Oracle.libm(op, x), Emu.libm(op, x)
        assert np.array_equal(a, b), op
        with np.errstate(over="ignore", under="ignore"):
            ref = exact[op](x.astype(np.float64)).astype(np.float32)
        assert int((a != ref).sum()) == 0, (op, int((a != ref).sum()))
    edge = np.float32([0.0, 1.0, np.inf, 1e-45, 3.4e38])
    assert np.array_equal(Oracle.libm("log", edge), np.log(edge.astype(np.float64)).astype(np.float32))
    assert np.array_equal(Oracle.libm("exp", np.float32([-200, -104.5, 0, 89, 1e9])), np.float32([0, 0, 1, np.inf, np.inf]))
    neg = rng.uniform(-50, 50, 100000).astype(np.float32)          # sphericalDirection takes any angle
    assert np.array_equal(Oracle.libm("sin", neg), np.sin(neg.astype(np.float64)).astype(np.float32))


def test_emulated_device_paths_equal_the_oracle_bitwise():
    """With the transcendental functions pinned, the per-lane device code (compiled for the CPU) and the oracle
    produce the same radiance for every path, bit for bit -- the CPU-side twin of tests/test_gpu_parity.py."""
    from nori_amd.scene import Bsdf
    from tests import scenes
    from tests.backends import Emu, Oracle
    rng = np.random.default_rng(11)
    for integ, sb in [("path_mis", [Bsdf("mirror"), Bsdf("dielectric")]), ("path_ems", [Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")]),
                      ("whitted", [Bsdf("mirror"), Bsdf("dielectric")])]:
        sc = scenes.cornell_box(16, 16, 1, integ, sphere_bsdfs=sb)
        e, o = Emu(sc), Oracle(sc, use_bvh=True)
        n = 20000
        rays = o.sample_rays(rng.uniform(0, 16, (n, 2)).astype(np.float32))
        ss = rng.integers(0, 2 ** 62, n, dtype=np.uint64); sq = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        assert np.array_equal(o.li(rays, ss, sq), e.li(rays, ss, sq)), integ


def _with_layout(layout, fn):
    import os
    old = os.environ.get("NORI_HIP_ACCEL_LAYOUT")
    os.environ["NORI_HIP_ACCEL_LAYOUT"] = layout
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("NORI_HIP_ACCEL_LAYOUT", None)
        else:
            os.environ["NORI_HIP_ACCEL_LAYOUT"] = old


@pytest.mark.parametrize("n_tris,seed", [(1, 3), (2, 4), (5, 5), (9, 8), (300, 6), (20000, 7)])
def test_wide_nodes_same_hits_as_brute_force(n_tris, seed):
    """WIDE nodes (BVH4, child boxes quantised to 8 bits; rt_types.h, rt_trace.h trav_wide_step) through the emulated
    device code: the hit records of closest-hit and any-hit queries are bit-identical to the oracle's linear scan
    (src/accel.cpp:30-40) -- a quantised box may only ever be larger than the box it stands for."""
    sc = scenes.soup_scene(n_tris, seed)
    rays = scenes.random_rays(30000, seed=seed + 10)
    rays["d"][:500, 0] = 0.0                                  # zero direction components: the 2^60 reciprocal clamp
    rays["d"][500:1000] = np.float32([0, -1, 0])
    e = _with_layout("bvh4q", lambda: Emu(sc))
    o = Oracle(sc)
    info = e.accel_info()
    assert info["node_children"] == (4 if n_tris > 4 else 2) or n_tris <= 8
    a, b = o.intersect(rays), e.intersect(rays)
    for k in a.dtype.names:
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), k
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])


def test_wide_nodes_fuzz_and_render():
    """The fuzzer's scene shapes (degenerate and numerically collinear triangles -> the all-hit subtree, duplicates,
    sheets, scales 1e-3 .. 1e3) through wide nodes, and a whole render: same frame, same ray counts as the BVH2 layout."""
    from tests import fuzz_intersect

    class R:
        def __init__(self, dev):
            pass

        def upload(self, sc, builder=0):
            self.e = _with_layout("bvh4q", lambda: Emu(sc))
            return self

        def intersect(self, rays, shadow=False):
            return self.e.intersect(rays, shadow)

        def close(self):
            self.e.close()

    fuzz_intersect.TOLERATED[0] = 0
    saved, fuzz_intersect.BUILDERS = fuzz_intersect.BUILDERS, (0, 1)      # the harness builds one tree whatever the id
    try:
        hits = sum(fuzz_intersect.one_round(seed, R, n_rays=3000) for seed in range(7000, 7040))
    finally:
        fuzz_intersect.BUILDERS = saved
    assert hits > 20000 and fuzz_intersect.TOLERATED[0] == 0
    sc = scenes.cornell_box(40, 24, 4, "path_mis", sphere_bsdfs=[Bsdf("mirror"), Bsdf("dielectric")])
    a, sa = _with_layout("bvh2", lambda: Emu(sc)).render_host(count_traversal=True)
    w = _with_layout("bvh4q", lambda: Emu(sc))
    assert w.accel_info()["node_children"] == 4
    b, sb = w.render_host(count_traversal=True)
    assert np.array_equal(a, b)
    assert (sa["n_closest_rays"], sa["n_shadow_rays"]) == (sb["n_closest_rays"], sb["n_shadow_rays"])
    assert sb["n_node_tests"] < 0.75 * sa["n_node_tests"]          # four boxes per node record: far fewer node visits


def _with_env(env, fn):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
@pytest.mark.parametrize("layout", ["bvh2", "bvh4q"])
@pytest.mark.parametrize("n_tris,seed", [(1, 3), (4, 9), (5, 5), (6, 2), (37, 8), (300, 6), (20000, 7)])
def test_device_builders_same_hits_as_brute_force(builder, layout, n_tris, seed):
    """The steps of the DEVICE builders (lbvh_steps.h: Morton keys, radix tree / PLOC clustering, left-to-right leaf order,
    leaf collapse, pair / node / wide-node emission) run as loops on the CPU (tests/emu/emu_builder.h): the trees they emit
    give hit records bit-identical to the oracle's linear scan (src/accel.cpp:30-40)."""
    sc = scenes.soup_scene(n_tris, seed)
    rays = scenes.random_rays(20000, seed=seed + 10)
    rays["d"][:300, 0] = 0.0
    e = _with_env({"NORI_EMU_BUILDER": builder, "NORI_HIP_ACCEL_LAYOUT": layout, "NORI_HIP_PLOC_RADIUS": "3" if n_tris < 100 else "16"}, lambda: Emu(sc))
    o = Oracle(sc)
    a, b = o.intersect(rays), e.intersect(rays)
    for k in a.dtype.names:
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), k
    assert np.array_equal(o.intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
def test_device_builders_fuzz(builder):
    """The fuzzer's scene shapes (degenerate and numerically collinear triangles -> unbounded boxes, which PLOC must keep out of
    the spatial clusters; duplicates, sheets, scales 1e-3 .. 1e3) through the device builders' steps, both node layouts."""
    from tests import fuzz_intersect

    for layout in ("bvh2", "bvh4q"):
        class R:
            def __init__(self, dev):
                pass

            def upload(self, sc, builder_id=0, **kw):
                self.e = _with_env({"NORI_EMU_BUILDER": builder, "NORI_HIP_ACCEL_LAYOUT": layout}, lambda: Emu(sc))
                return self

            def intersect(self, rays, shadow=False):
                return self.e.intersect(rays, shadow)

            def close(self):
                self.e.close()

        fuzz_intersect.TOLERATED[0] = 0
        saved, fuzz_intersect.BUILDERS = fuzz_intersect.BUILDERS, (0,)      # the harness builds the tree NORI_EMU_BUILDER names
        try:
            hits = sum(fuzz_intersect.one_round(seed, R, n_rays=2000) for seed in range(8000, 8030))
        finally:
            fuzz_intersect.BUILDERS = saved
        assert hits > 2500 and fuzz_intersect.TOLERATED[0] == 0


def test_ploc_trees_cost_less_than_radix_trees():
    """What PLOC is for: on the Cornell box and on a patch of terrain its trees need fewer node and triangle tests per ray than
    the radix tree over the same Morton order (and stay within 10 % of the host SAH builder's)."""
    from nori_amd import workloads
    for name, kw in (("pa4-cbox-path_mis", {}), ("c5-terrain-10m", {"triangles": 20000})):
        sc = workloads.load(name, width=32, height=32, spp=2, **kw).scene
        cost = {}
        for b in ("sah", "lbvh", "ploc"):
            e = _with_env({"NORI_EMU_BUILDER": b, "NORI_HIP_ACCEL_LAYOUT": "bvh4q"}, lambda: Emu(sc))
            _, st = e.render_host(count_traversal=True)
            cost[b] = (st["n_node_tests"] + 2.0 * st["n_tri_tests"]) / (st["n_closest_rays"] + st["n_shadow_rays"])
            e.close()
        assert cost["ploc"] < cost["lbvh"], (name, cost)
        assert cost["ploc"] < 1.10 * cost["sah"], (name, cost)


def test_treelet_sweeps_improve_the_ploc_tree():
    """Treelet restructuring after PLOC (lbvh_steps.h: the optimal topology of every <= 7-leaf treelet, bottom-up): fewer node
    tests per ray than the clustering alone, on the Cornell box no more than the host's SAH tree needs, and the same hits as the
    brute-force scan however many sweeps ran."""
    from nori_amd import workloads
    sc = workloads.load("pa4-cbox-path_mis", width=32, height=32, spp=2).scene
    rays = scenes.random_rays(4000, seed=21)
    ref = Oracle(sc).intersect(rays)
    nodes = {}
    for sweeps in (0, 1, 2):
        e = _with_env({"NORI_EMU_BUILDER": "ploc", "NORI_HIP_ACCEL_LAYOUT": "bvh2", "NORI_HIP_TREELET_SWEEPS": str(sweeps), "NORI_HIP_REINSERT_ITERS": "0"}, lambda: Emu(sc))
        got = e.intersect(rays)
        assert all(np.array_equal(ref[k], got[k], equal_nan=ref[k].dtype.kind == "f") for k in ref.dtype.names), sweeps
        _, st = e.render_host(count_traversal=True)
        nodes[sweeps] = st["n_node_tests"] / (st["n_closest_rays"] + st["n_shadow_rays"])
        e.close()
    e = _with_env({"NORI_EMU_BUILDER": "sah", "NORI_HIP_ACCEL_LAYOUT": "bvh2"}, lambda: Emu(sc))
    _, st = e.render_host(count_traversal=True)
    sah = st["n_node_tests"] / (st["n_closest_rays"] + st["n_shadow_rays"])
    assert nodes[1] < 0.95 * nodes[0] and nodes[2] <= nodes[1] * 1.01, nodes
    assert nodes[2] < 1.02 * sah, (nodes, sah)


def test_split_parts_cover_the_triangle():
    """split_emit (lbvh_steps.h) on single triangles: whatever the shape -- slivers, axis-parallel, tiny in a huge scene box, far from the origin --
    and however many cuts, every point of the triangle lies in the box of at least one part (the parts are the triangle clipped to cells
    that tile its box; boxes evaluated in binary64, rounded outwards, padded as the whole triangle is -- a sliver's parts keep the sliver's
    wide pad) and no part reaches beyond the triangle's own padded box."""
    from tests.backends import emu_lib, ptr
    lib = emu_lib()
    rng = np.random.default_rng(5)
    w = rng.uniform(0, 1, (4000, 3)); w /= w.sum(1, keepdims=True)
    w[:3] = np.eye(3); w[3:6] = [[0.5, 0.5, 0], [0, 0.5, 0.5], [0.5, 0, 0.5]]      # the corners and edge midpoints too
    n_cut = 0
    for trial in range(400):
        scale = float(rng.choice([1e-3, 1.0, 1.0, 50.0, 1e4]))
        centre = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.0, 1.0, 100.0]))
        tri = centre + rng.uniform(-1, 1, (3, 3)) * scale * rng.choice([1.0, 1.0, 1e-3], 3)      # thin along some axes
        if trial % 7 == 0:
            tri[:, rng.integers(0, 3)] = centre[0]                                              # axis-parallel: a flat box
        tri = tri.astype(np.float32)
        lo, hi = tri.min(0), tri.max(0)
        grow = float(rng.choice([0.0, 3.0, 1000.0])) * (hi - lo).max()
        smin, smax = (lo - grow * rng.uniform(0, 1, 3)).astype(np.float32), (hi + grow * rng.uniform(0, 1, 3)).astype(np.float32)
        pad = np.float32(2e-6 * np.linalg.norm(smax - smin) + 1e-30)
        cuts = int(rng.choice([1, 2, 3, 7, 15, 63]))
        boxes = np.zeros((64, 6), np.float32); keys = np.zeros(64, np.uint64)
        n = lib.emu_split_parts(ptr(np.ascontiguousarray(tri)), cuts, pad, ptr(smin), ptr(smax), ptr(boxes), ptr(keys), 64)
        assert 1 <= n <= cuts + 1, (trial, n, cuts)
        b = boxes[:n].astype(np.float64)
        pts = w @ tri.astype(np.float64)
        eps = 1e-12 * (1.0 + np.abs(pts))[:, None, :]      # (the points are binary64 combinations of the vertices: their own rounding, 1e-16 relative)
        inside = ((pts[:, None, :] >= b[None, :, :3] - eps) & (pts[:, None, :] <= b[None, :, 3:] + eps)).all(2).any(1)
        assert inside.all(), (trial, cuts, n, tri, pts[~inside][:3])
        whole = np.zeros((1, 6), np.float32); k1 = np.zeros(1, np.uint64)
        assert lib.emu_split_parts(ptr(np.ascontiguousarray(tri)), 0, pad, ptr(smin), ptr(smax), ptr(whole), ptr(k1), 1) == 1
        assert (b[:, :3] >= whole[0, :3].astype(np.float64) - 1e-6 * scale).all() and (b[:, 3:] <= whole[0, 3:].astype(np.float64) + 1e-6 * scale).all(), trial
        n_cut += 1 if n > 1 else 0
    assert n_cut > 200


def test_triangle_splitting_in_front_of_the_device_builder():
    """References (lbvh_steps.h): a triangle whose box is several times the scene's typical one AND holds other geometry enters the tree
    as several parts.  On the pa5 table scene that takes a third of the triangle tests per ray without adding node tests; the Cornell box
    (tilted faces of blocks in an empty room), a patch of terrain and a soup of like-sized triangles are left exactly as they were -- one
    reference per triangle, the same tree; hits are the brute-force scan's either way, and with the criteria switched off and every
    triangle cut as often as the cap allows (parts of one triangle in neighbouring leaves, in the same leaf, flat and sliver parts)."""
    from nori_amd import workloads
    for name, kw, cut in (("c4-table-mis", {}, True), ("pa4-cbox-path_mis", {}, False), ("c5-terrain-10m", {"triangles": 20000}, False)):
        sc = workloads.load(name, width=32, height=32, spp=2, **kw).scene
        rays = scenes.random_rays(3000, seed=35)
        ref = Oracle(sc).intersect(rays)
        got = {}
        for label, env in (("off", {"NORI_HIP_SPLIT_BUDGET": "0"}), ("on", {}), ("forced", {"NORI_HIP_SPLIT_BUDGET": "2.0", "NORI_HIP_SPLIT_SCALE": "0", "NORI_HIP_SPLIT_INSIDE": "0", "NORI_HIP_SPLIT_CAP": "7"})):
            e = _with_env(dict(env, NORI_EMU_BUILDER="ploc", NORI_HIP_ACCEL_LAYOUT="bvh2"), lambda: Emu(sc))
            hit = e.intersect(rays)
            assert all(np.array_equal(ref[k], hit[k], equal_nan=ref[k].dtype.kind == "f") for k in ref.dtype.names), (name, label)
            info = e.accel_info()
            _, st = e.render_host(count_traversal=True)
            nr = st["n_closest_rays"] + st["n_shadow_rays"]
            got[label] = (info["n_references"], info["n_nodes"], st["n_node_tests"] / nr, st["n_tri_tests"] / nr)
            e.close()
        assert got["off"][0] == info["n_triangles"] and got["forced"][0] > 1.5 * info["n_triangles"], (name, got)
        if cut:
            assert info["n_triangles"] < got["on"][0] <= 1.3 * info["n_triangles"], (name, got)
            assert got["on"][3] < 0.72 * got["off"][3] and got["on"][2] < 1.0 * got["off"][2], (name, got)
        else:
            assert got["on"] == got["off"], (name, got)
    for seed in (3, 4):
        sc = scenes.soup_scene(400, seed)
        rays = scenes.random_rays(5000, seed=seed + 50)
        ref = Oracle(sc).intersect(rays)
        for layout in ("bvh2", "bvh4q"):
            e = _with_env({"NORI_EMU_BUILDER": "ploc", "NORI_HIP_ACCEL_LAYOUT": layout, "NORI_HIP_SPLIT_BUDGET": "3.0", "NORI_HIP_SPLIT_SCALE": "0", "NORI_HIP_SPLIT_INSIDE": "0"}, lambda: Emu(sc))
            assert e.accel_info()["n_references"] > 2 * 400
            hit = e.intersect(rays)
            assert all(np.array_equal(ref[k], hit[k], equal_nan=ref[k].dtype.kind == "f") for k in ref.dtype.names), (seed, layout)
            assert np.array_equal(Oracle(sc).intersect(rays, True)["mesh"], e.intersect(rays, True)["mesh"])
            e.close()


@pytest.mark.parametrize("stride", [1, 3])
def test_reinsertion_improves_the_ploc_tree(stride):
    """Parallel re-insertion after PLOC and its sweeps (lbvh_steps.h: every candidate searches the same tree for the place where the
    inner nodes' areas shrink most, marks what its move touches, the winners move, the tree is refitted): fewer node tests per ray than
    without, on the Cornell box, the pa5 table and a patch of terrain; the same hits as the brute-force scan after every number of
    iterations -- a move that tore the tree would lose triangles -- also with only every third slot a candidate per iteration."""
    from nori_amd import workloads
    for name, kw, need in (("pa4-cbox-path_mis", {}, 0.985), ("c4-table-mis", {}, 0.97), ("c5-terrain-10m", {"triangles": 20000}, 0.97)):
        sc = workloads.load(name, width=32, height=32, spp=2, **kw).scene
        rays = scenes.random_rays(3000, seed=33)
        ref = Oracle(sc).intersect(rays)
        nodes = {}
        for iters in (0, 1, 8):
            e = _with_env({"NORI_EMU_BUILDER": "ploc", "NORI_HIP_ACCEL_LAYOUT": "bvh2", "NORI_HIP_REINSERT_ITERS": str(iters * stride), "NORI_HIP_REINSERT_STRIDE": str(stride)},
                          lambda: Emu(sc))
            got = e.intersect(rays)
            assert all(np.array_equal(ref[k], got[k], equal_nan=ref[k].dtype.kind == "f") for k in ref.dtype.names), (name, iters)
            _, st = e.render_host(count_traversal=True)
            nodes[iters] = st["n_node_tests"] / (st["n_closest_rays"] + st["n_shadow_rays"])
            e.close()
        assert nodes[8] < need * nodes[0] and nodes[8] < nodes[1] * 1.005, (name, nodes)


# ---------------------------------------------------------------- 32-B node records (rt_nodeq.h) and wf_extend's leaf step

@pytest.mark.parametrize("scene", ["soup", "cornell", "far_and_flat"])
def test_node_records_32b_never_reject_a_box_the_ray_crosses(scene):
    """The 16-bit planes + per-axis slack of the 32-B record against the exact slab test of the stored boxes, every ray against
    every node -- including rays that start far outside the grid, axis-parallel rays and rays lying in the planes of flat boxes."""
    if scene == "soup":
        sc = scenes.soup_scene(3000, 11)
        rays = scenes.random_rays(300, seed=5)
    elif scene == "cornell":
        sc = scenes.cornell_box(16, 16, 1)
        rays = scenes.random_rays(300, seed=6, extent=0.9)
        rays["o"] += np.float32([0, 1, 0])
        rays["d"][:40] = np.float32([0, -1, 0]); rays["d"][40:80] = np.float32([1, 0, 0]); rays["d"][80:120] = np.float32([0, 0, -1])
        rays["o"][120:160, 1] = 0.0; rays["d"][120:160, 1] = 0.0      # in the floor's plane
        rays["d"][120:160] /= np.linalg.norm(rays["d"][120:160], axis=1, keepdims=True)
    else:
        sc = scenes.soup_scene(500, 12)
        rays = scenes.random_rays(300, seed=7, extent=4000.0, target_extent=1.0)      # origins thousands of scene sizes away
        rays["maxt"][:100] = 1e30
    e = Emu(sc)
    assert e.accel_info()["node_records_32b"] == 1
    res = e.nodeq_check(rays)
    assert res is not None
    bad, exact, rec = res
    assert bad == 0, f"{bad} boxes rejected by the 32-B record that the ray crosses"
    assert exact > 0 and rec >= exact
    assert rec <= 1.25 * exact + 50, "the records accept far more boxes than the exact test: planes or slack too loose"


def test_trees_with_unbounded_boxes_have_no_32b_records():
    """Numerically collinear triangles hang under the root in boxes of +-3e38 (rt_types.h): such a tree keeps its 64-B nodes only."""
    from nori_amd.scene import Mesh
    sc = scenes.soup_scene(200, 13)
    v = np.float32([[0, 0, 0], [1, 1, 1], [2, 2, 2.0000002], [0.5, 0.2, 0.1]])
    sc.meshes = list(sc.meshes) + [Mesh(v, np.uint32([[0, 1, 2], [0, 1, 3]]))]
    e, o = Emu(sc), Oracle(sc)
    assert e.accel_info()["node_records_32b"] == 0
    rays = scenes.random_rays(3000, seed=9)
    _assert_its_equal(o.intersect(rays), e.intersect(rays))


def test_leaf_step_with_selects_equals_the_branching_one(monkeypatch):
    """wf_extend's form of the leaf step (hit update as selects, no mesh id kept) is what the harness walks by default; the
    branching form (batch twins, megakernel) must find the same intersections."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, '.'); import numpy as np\n"
            "from tests import scenes\nfrom tests.backends import Emu\n"
            "sc = scenes.cornell_box(16, 16, 1); rays = scenes.random_rays(20000, seed=3, extent=0.9); rays['o'] += np.float32([0, 1, 0])\n"
            "e = Emu(sc); a = e.intersect(rays); b = e.intersect(rays, True)\n"
            "np.save(sys.argv[1], np.concatenate([a['t'], a['uv'].ravel(), a['tri'].astype(np.float32), a['mesh'].astype(np.float32), b['mesh'].astype(np.float32)]))\n")
    outs = []
    for sel in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"leaf_select_{sel}_{os.getpid()}.npy")
        env = dict(os.environ, NORI_EMU_LEAF_SELECT=sel)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        outs.append(np.load(path)); os.remove(path)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("scene", ["soup", "cornell"])
def test_node_records_32b_enclose_the_64b_boxes_read_independently(scene):
    """The record layout decoded here in numpy, from its description alone (rt_nodeq.h: dwords 0 / 1 = x low / high planes, left child
    in bits 0-15, right child in bits 16-31; 2 / 3 = y; 4 / 5 = z; 6 / 7 = the links; plane = mn + q scale) against the 64-B node
    (rt_types.h: q0 = left centre x y, right centre x y; q1 = centres z, half extents z; q2 = half extents x y; q3 = links): every
    box of the 32-B form contains the 64-B box, is no more than a grid step (plus float rounding) wider per side, and the links agree."""
    sc = scenes.soup_scene(4000, 21) if scene == "soup" else scenes.cornell_box(16, 16, 1)
    rec = Emu(sc).node_records()
    assert rec is not None
    n64, n32, mn, scale = rec
    n64 = n64.astype(np.float64); mn = mn.astype(np.float64); scale = scale.astype(np.float64)
    cx = np.stack([n64[:, 0], n64[:, 2]], 1); cy = np.stack([n64[:, 1], n64[:, 3]], 1); cz = n64[:, 4:6]
    hx = np.stack([n64[:, 8], n64[:, 10]], 1); hy = np.stack([n64[:, 9], n64[:, 11]], 1); hz = n64[:, 6:8]
    assert np.array_equal(n32[:, 6:8], rec[0][:, 12:14].view(np.uint32))                   # links
    for a, (c, h) in enumerate(((cx, hx), (cy, hy), (cz, hz))):
        lo_q = np.stack([n32[:, 2 * a] & 0xffff, n32[:, 2 * a] >> 16], 1).astype(np.float64)
        hi_q = np.stack([n32[:, 2 * a + 1] & 0xffff, n32[:, 2 * a + 1] >> 16], 1).astype(np.float64)
        p_lo, p_hi = mn[a] + lo_q * scale[a], mn[a] + hi_q * scale[a]
        box_lo, box_hi = c - h, c + h
        inside_grid_lo, inside_grid_hi = box_lo >= mn[a], box_hi <= mn[a] + 65535.0 * scale[a]
        assert (p_lo[inside_grid_lo] <= box_lo[inside_grid_lo]).all() and (p_hi[inside_grid_hi] >= box_hi[inside_grid_hi]).all()
        slack = scale[a] * (1.0 + 1e-6) + 1e-6 * np.abs(c)
        assert (box_lo - p_lo <= slack).all() and (p_hi - box_hi <= slack).all(), "planes more than a grid step off the box"
        assert (lo_q <= hi_q).all()
