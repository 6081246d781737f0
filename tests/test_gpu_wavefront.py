"""The wavefront engine (wavefront.hip) against the megakernel and the oracle:
same pcg32 streams and the same per-lane path code, so every camera sample
carries the same radiance -- frames may differ by float summation order only,
ray counts must be identical."""
import os

import numpy as np
import pytest

from nori_amd.scene import Bsdf, RFilter, Scene
from tests import scenes
from tests.backends import Oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
WALK_64B = {"NORI_HIP_WF_NO_ASM_LOOP": 1}      # wf_extend on the 64-B nodes (the compiler's loop), the megakernel's tree form


def _budget(wf, n):
    """Batches of n camera samples on a pool of n paths: every batch starts all its samples in its first pass (the shrinking
    schedule).  wavefront_paths alone bounds the POOL: a batch bigger than it regenerates (test_regeneration_*)."""
    wf.set_option("wavefront_paths", n)
    wf.set_option("wavefront_samples", n)


def _pair(renderer_factory, sc):
    mk = renderer_factory(sc)
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    return mk, wf


@pytest.mark.parametrize("integ", ["normals", "ao", "simple", "whitted", "path_mats", "path_ems", "path_mis"])
def test_wavefront_equals_megakernel(renderer_factory, integ):
    sb = [Bsdf("mirror"), Bsdf("dielectric")] if integ in ("whitted", "path_mis") else \
        [Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")]
    sc = scenes.cornell_box(72, 40, 6, integ, sphere_bsdfs=sb)
    sc.integrator.position, sc.integrator.energy = (0, 1.5, 0.5), (30, 30, 30)
    mk, wf = _pair(renderer_factory, sc)
    a, sa = mk.render_host(count_traversal=True)
    # (the megakernel walks the 64-B nodes: node / triangle tests are compared on that tree form; the wavefront engine's own
    # form, the 32-B records, is counted in test_traversal_counters_are_those_of_the_timed_tree_form)
    b, sb_ = _with_env(WALK_64B, lambda: wf.render_host(count_traversal=True))
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests", "n_invalid"):
        assert sa[k] == sb_[k], (k, sa[k], sb_[k])
    np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rf", ["gaussian", "mitchell", "tent", "box"])
def test_wavefront_filters_and_ragged_frames(renderer_factory, rf):
    sc = scenes.cornell_box(33, 17, 5, "path_mis", rfilter=RFilter(rf))
    mk, wf = _pair(renderer_factory, sc)
    a, _ = mk.render_host()
    b, _ = wf.render_host()
    np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5)


def test_wavefront_batching_and_splits(renderer_factory):
    """Small path budgets force sample batches and tile batches; tile / sample splits add up."""
    sc = scenes.cornell_box(72, 40, 12, "path_mis")
    mk, wf = _pair(renderer_factory, sc)
    whole, st = mk.render_host()
    for budget in (256 * 15 * 5, 256 * 4, 256):        # 5 spp per batch; 4 tiles per batch; 1 tile per batch
        _budget(wf, budget)
        b, sb = wf.render_host()
        np.testing.assert_allclose(b, whole, rtol=1e-4, atol=1e-5)
        assert sb["n_closest_rays"] == st["n_closest_rays"]
    _budget(wf, 1 << 20)
    parts = sum(wf.render_host(tile_mod=4, tile_rem=k)[0] for k in range(4))
    np.testing.assert_allclose(parts, whole, rtol=1e-4, atol=1e-5)
    parts = wf.render_host(spp_count=5, spp_begin=0)[0] + wf.render_host(spp_count=7, spp_begin=5)[0]
    np.testing.assert_allclose(parts, whole, rtol=1e-4, atol=1e-5)


def test_wavefront_matches_oracle_on_reference_scene(renderer_factory):
    sc = Scene.load_npz(os.path.join(GOLDEN, "pa5-cbox_mis.npz"))     # mirror + dielectric spheres
    sc.camera.width, sc.camera.height, sc.sample_count = 96, 72, 8
    r = renderer_factory(sc)
    r.set_option("engine", "wavefront")
    B, sb = r.render_host()
    A, sa = Oracle(sc, use_bvh=True).render_host()
    assert sb["n_camera_samples"] == sa["n_camera_samples"] == 96 * 72 * 8
    from tests.test_gpu_parity import assert_image_parity
    assert_image_parity(A, B, r.border, "pa5-cbox_mis 96x72x8 wavefront")


def test_wavefront_empty_scene_and_options(renderer_factory):
    from nori_amd import NoriError
    sc = scenes.soup_scene(3)
    sc.meshes = []
    r = renderer_factory(sc)
    r.set_option("engine", "wavefront")
    rgbw, st = r.render_host(spp_count=2)
    assert st["n_camera_samples"] == 32 * 32 * 2 and (rgbw[..., :3] == 0).all() and rgbw[..., 3].max() > 0
    with pytest.raises(NoriError):
        r.set_option("engine", "nope")
    with pytest.raises(NoriError):
        r.set_option("bogus", 1)


@pytest.mark.gpu
def test_gpu_large_scene_lbvh_both_engines():
    """configs[4] in small: 2 M-triangle terrain, BVH built on the device, both engines; camera-ray hits
    bit-identical to the oracle's brute-force scan, engines agree on every ray count."""
    from tests.stress_large import run
    a = run(n_tris=2_000_000, size=128, spp=4, check=96, builder=1, engine="wavefront", reps=1)
    b = run(n_tris=2_000_000, size=128, spp=4, check=0, builder=1, engine="megakernel", reps=1)
    assert a["bit_exact"] and a["hits"] > 30 and a["finite"] and b["finite"]
    assert a["max_depth"] < 64 and a["build_ms"] < 2000
    assert a["rays"] == b["rays"] and a["node_tests_per_ray"] == b["node_tests_per_ray"]
    assert abs(a["mean_w"] - b["mean_w"]) < 1e-6


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_wavefront_schedule_knobs_do_not_change_the_frame(renderer_factory):
    """wf_finish on/off, readback cadence, chunking, thresholds and pipes only reschedule the same
    per-sample work: every camera sample carries the same radiance, and the film adds the samples of a
    pixel in sample order, so the frames are bit-identical."""
    sc = scenes.cornell_box(96, 64, 9, "path_mis", sphere_bsdfs=[Bsdf("mirror"), Bsdf("dielectric")])
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    ref, _ = wf.render_host()                                                      # the production kernels (hand-written node loop)
    # (counters on ONE tree form: wf_finish walks the 64-B nodes, the counting wf_extend would walk the 32-B records -- a knob that
    # moves paths between the two would move node tests with them)
    refc, sr = _with_env(WALK_64B, lambda: wf.render_host(count_traversal=True))
    assert np.array_equal(refc, ref)
    for env in ({"NORI_HIP_WF_FINISH": 0}, {"NORI_HIP_WF_FINISH_PATHS": 256}, {"NORI_HIP_WF_SYNC_EVERY": 1},
                {"NORI_HIP_WF_STATIC": 0, "NORI_HIP_WF_DYNDIV": 1}, {"NORI_HIP_WF_REFILL": 1, "NORI_HIP_WF_LEAF": 64},
                {"NORI_HIP_WF_PIPES": 2}, {"NORI_HIP_WF_EXTEND_WGS_PER_CU": 1}):
        b, _ = _with_env(env, lambda: wf.render_host())
        assert np.array_equal(b, ref), env
        b, sb = _with_env({**env, **WALK_64B}, lambda: wf.render_host(count_traversal=True))
        for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests"):
            assert sr[k] == sb[k], (env, k)
        assert np.array_equal(b, ref), env


def test_cus_split_between_traversal_and_shading_give_the_same_frame(renderer_factory):
    """The device's CUs shared out between the traversal stream and the shading / film stream (wavefront_render, "split"): two
    pipes, each half of the tiles, the kernels of the two kinds on disjoint CUs.  Same samples, same order per pixel: the frame
    has the bits of the one-stream frame and the ray counts are equal; the stats say how many CUs the ray queries had."""
    sc = scenes.cornell_box(160, 128, 12, "path_mis", sphere_bsdfs=[Bsdf("mirror"), Bsdf("dielectric")])
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    ref, sr = wf.render_host()
    assert sr["trace_cus"] > 0
    for cus in (8, 64, 128):
        b, sb = _with_env({"NORI_HIP_WF_SPLIT_CUS": cus, "NORI_HIP_WF_SPLIT_MIN": 0}, lambda: wf.render_host())
        assert sb["trace_cus"] == sr["trace_cus"] - cus, (cus, sb["trace_cus"], sr["trace_cus"])
        assert np.array_equal(b, ref), cus
        for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays"):
            assert sr[k] == sb[k], (cus, k)
    # several batches per pipe (a small path budget), and the tail kernel off.  (The film adds a pixel's samples batch by batch:
    # the frame to compare with is the one of two pipes with the same budget on ONE stream.)
    _budget(wf, 1 << 16)
    ref2, sr2 = _with_env({"NORI_HIP_WF_PIPES": 2}, lambda: wf.render_host())
    for env in ({}, {"NORI_HIP_WF_FINISH": 0}, {"NORI_HIP_WF_SYNC_EVERY": 1}):
        b, sb = _with_env({"NORI_HIP_WF_SPLIT_CUS": 48, "NORI_HIP_WF_SPLIT_MIN": 0, **env}, lambda: wf.render_host())
        assert np.array_equal(b, ref2), env
        assert sr2["n_closest_rays"] == sb["n_closest_rays"] and sr2["n_shadow_rays"] == sb["n_shadow_rays"], env


def test_tail_of_a_batch_beside_the_next_batch_gives_the_same_frame(renderer_factory):
    """Tail overlap (wavefront_render): in a call of several batches the last live paths of batch k are copied out of the state
    pool and walked by the persistent wf_finish on a stream that owns a few CUs while batch k + 1 runs on the caller's stream; the film gather of batch k
    follows that of batch k - 1 and precedes that of batch k + 1 as before.  Same samples, same order per pixel: the bits of the
    frame rendered with the tails on the bulk's stream, equal ray counts, and the stats say which CUs did what."""
    sc = scenes.cornell_box(160, 128, 12, "path_mis", sphere_bsdfs=[Bsdf("mirror"), Bsdf("dielectric")])
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    for budget in (160 * 128 * 5, 256 * 12, 1 << 20):      # 3 sample batches of all tiles; tile batches; one batch (nothing to overlap)
        _budget(wf, budget)
        ref, sr = _with_env({"NORI_HIP_WF_TAIL_CUS": 0}, lambda: wf.render_host())
        assert sr["tail_cus"] == 0 and sr["tail_ms"] == 0.0
        for cus in (8, 32, 64):
            for env in ({}, {"NORI_HIP_WF_FINISH_PATHS": 256}, {"NORI_HIP_WF_SYNC_EVERY": 1}):
                b, sb = _with_env({"NORI_HIP_WF_TAIL_CUS": cus, **env}, lambda: wf.render_host())
                assert np.array_equal(b, ref), (budget, cus, env)
                for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays"):
                    assert sr[k] == sb[k], (budget, cus, k)
                if budget < (1 << 20):
                    assert sb["tail_cus"] == cus and sb["trace_cus"] == sr["trace_cus"], (budget, cus, sb["tail_cus"], sb["trace_cus"])
                else:
                    assert sb["tail_cus"] == 0 and sb["trace_cus"] == sr["trace_cus"]
    # a tree deeper than wf_finish's LDS stack: the side launch spills into its own columns
    sc = scenes.soup_scene(30000, seed=3, width=80, height=56, integrator="path_mis")
    sc.sample_count = 6
    from nori_amd.scene import Mesh
    v, f = scenes.quad((-3, 3, -3), (3, 3, -3), (3, 3, 3), (-3, 3, 3))
    sc.meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(5.0, 5.0, 5.0), name="light"))
    wf = renderer_factory(sc, builder=1)
    wf.set_option("engine", "wavefront")
    assert wf.accel_info()["max_depth"] + 1 > 16
    _budget(wf, 80 * 56 * 2)
    ref, sr = _with_env({"NORI_HIP_WF_TAIL_CUS": 0}, lambda: wf.render_host())
    b, sb = _with_env({"NORI_HIP_WF_TAIL_CUS": 16}, lambda: wf.render_host())
    assert np.array_equal(b, ref) and sb["tail_cus"] == 16
    assert sr["n_closest_rays"] == sb["n_closest_rays"] and sr["n_shadow_rays"] == sb["n_shadow_rays"]


def test_regeneration_gives_the_bits_of_the_shrinking_schedule(renderer_factory):
    """A batch bigger than the pool (wavefront_paths < its camera samples) starts its samples pass by pass: a pass works on the
    stored survivors plus as many new camera samples as fill the pool (wavefront.hip, pass_shape), neither kernel stores the first
    vertex of a new path.  Which pass a sample starts in is not observable -- pcg32 streams are per (pixel, sample), radiance goes to
    the film store at the sample's own index, the film adds a pixel's samples batch by batch --: the frame and the ray counts of
    the same batches on a pool that holds a whole batch, bit for bit, for any pool size and readback cadence."""
    sc = scenes.cornell_box(160, 128, 12, "path_mis", sphere_bsdfs=[Bsdf("mirror"), Bsdf("dielectric")])
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    n_samples = 160 * 128 * 12
    ref, sr = wf.render_host()                                # one batch, every sample started in the first pass
    # (node / triangle tests are compared on ONE tree form: wf_finish walks the 64-B nodes, and how many paths it ends depends on the pool)
    ref64, sr64 = _with_env(WALK_64B, lambda: wf.render_host(count_traversal=True))
    assert sr["n_camera_samples"] == n_samples and np.array_equal(ref64, ref)
    wf.set_option("wavefront_samples", 1 << 30)               # the batch stays the frame whatever the pool
    for pool in (256, 1000, 256 * 37, 1 << 16, n_samples - 256):
        wf.set_option("wavefront_paths", pool)
        for env in ({}, {"NORI_HIP_WF_SYNC_EVERY": 1}, {"NORI_HIP_WF_FINISH": 0}, {"NORI_HIP_WF_FORCE_MIXED": 1}, WALK_64B):
            b, sb = _with_env(env, lambda: wf.render_host(count_traversal=env is WALK_64B))
            assert np.array_equal(b, ref), (pool, env)
            for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_invalid") + (("n_node_tests", "n_tri_tests") if env is WALK_64B else ()):
                assert sr64[k] == sb[k], (pool, env, k, sr64[k], sb[k])
    # every pass after the first traced as one that may hold new paths (both wf_extend launches, the kMixed wf_shade) on a batch that fits the pool: same bits
    wf.set_option("wavefront_paths", 1 << 29)
    b, sb = _with_env({"NORI_HIP_WF_FORCE_MIXED": 1}, lambda: wf.render_host())
    assert np.array_equal(b, ref) and sb["n_shadow_rays"] == sr["n_shadow_rays"]
    # several batches (3 sample batches of all tiles; tile batches), each regenerating, tails beside the next batch or not
    for batch in (160 * 128 * 5, 256 * 12):
        _budget(wf, batch)
        ref2, sr2 = _with_env({"NORI_HIP_WF_TAIL_CUS": 0}, lambda: wf.render_host())
        np.testing.assert_allclose(ref2, ref, rtol=1e-4, atol=1e-5)
        for pool in (256, batch // 3):
            wf.set_option("wavefront_paths", pool)
            for cus in (0, 32):
                b, sb = _with_env({"NORI_HIP_WF_TAIL_CUS": cus}, lambda: wf.render_host())
                assert np.array_equal(b, ref2), (batch, pool, cus)
                assert sb["tail_cus"] == cus
                for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays"):
                    assert sr2[k] == sb[k], (batch, pool, cus, k)
    # reference film order: the whole frame is one batch whatever the pool and whatever wavefront_samples says
    wf.set_option("wavefront_samples", 0)
    wf.set_option("wavefront_paths", 1 << 29)
    wf.set_option("film_order", "reference")
    fr, _ = wf.render_host()
    wf.set_option("wavefront_paths", 256 * 9)
    assert np.array_equal(wf.render_host()[0], fr)


def test_regeneration_on_wide_and_deep_trees(renderer_factory):
    """The same through the other traversal kernels: wide (BVH4) nodes, and a tree deeper than the LDS stack (spill columns)."""
    sc = scenes.soup_scene(30000, seed=3, width=80, height=56, integrator="path_mis")
    sc.sample_count = 6
    from nori_amd.scene import Mesh
    v, f = scenes.quad((-3, 3, -3), (3, 3, -3), (3, 3, 3), (-3, 3, 3))
    sc.meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(5.0, 5.0, 5.0), name="light"))
    for layout, builder in (("bvh4q", 0), ("bvh2", 1)):
        from nori_amd.render import Renderer
        wf = Renderer(0); wf.set_option("accel_layout", layout); wf.upload(sc, builder=builder); wf.set_option("engine", "wavefront")
        assert wf.accel_info()["max_depth"] + 1 > 16 or layout == "bvh4q"
        ref, sr = wf.render_host()
        wf.set_option("wavefront_samples", 1 << 30)
        for pool in (256 * 3, 80 * 56 * 2):
            wf.set_option("wavefront_paths", pool)
            b, sb = wf.render_host()
            assert np.array_equal(b, ref), (layout, pool)
            assert sr["n_closest_rays"] == sb["n_closest_rays"] and sr["n_shadow_rays"] == sb["n_shadow_rays"], (layout, pool)
        wf.close()


def test_out_of_memory_for_the_path_state_means_smaller_batches(renderer_factory):
    """The number of paths in flight is bounded by what hipMemGetInfo reports free -- a snapshot another context may have spent by the
    time the pool is allocated.  The render call then gives back what it holds and halves the POOL until the state fits
    (nori_hip.hip) instead of failing: the batch -- hence the frame, bit for bit -- stays what it was (the paths of the batch
    start pass by pass on the smaller pool), and OUT_OF_MEMORY only when nothing fits."""
    import torch
    sc = scenes.cornell_box(1024, 1024, 48, "path_mis")          # 50 M paths x 236 B = 11.9 GB of state + sample store in one batch
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free - (7 << 30)), dtype=torch.uint8, device="cuda")      # leave 7 GB: a quarter of the frame per batch fits
    try:
        frame = torch.zeros(wf.frame_shape(), device="cuda")
        st = _with_env({"NORI_HIP_WF_IGNORE_FREE": 1}, lambda: wf.render_into(frame))
        assert st["n_camera_samples"] == 1024 * 1024 * 48
        got = frame.cpu().numpy()
    finally:
        del hog
        torch.cuda.empty_cache()
    wf2 = renderer_factory(sc)
    wf2.set_option("engine", "wavefront")
    ref, sr = wf2.render_host()                      # one batch, all its samples started in the first pass (12 GB of state)
    assert np.array_equal(got, ref)
    assert sr["n_closest_rays"] == st["n_closest_rays"] and sr["n_shadow_rays"] == st["n_shadow_rays"]


def test_wavefront_deep_tree_spills_the_stack(renderer_factory):
    """LBVH over a triangle soup is deeper than 16: with NORI_HIP_WF_STACK=16 wf_extend keeps 16 stack
    entries in LDS and the rest in its global spill columns; same frame and counts as the megakernel
    (64-entry LDS stack)."""
    sc = scenes.soup_scene(30000, seed=3, width=80, height=56, integrator="path_mis")
    sc.sample_count = 4
    from nori_amd.scene import Mesh
    v, f = scenes.quad((-3, 3, -3), (3, 3, -3), (3, 3, 3), (-3, 3, 3))
    sc.meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(5.0, 5.0, 5.0), name="light"))
    mk = renderer_factory(sc, builder=1)
    wf = renderer_factory(sc, builder=1)
    wf.set_option("engine", "wavefront")
    assert mk.accel_info()["max_depth"] + 1 > 16
    a, sa = mk.render_host(count_traversal=True)
    for stack in (16, 24):
        b, sb = _with_env({"NORI_HIP_WF_STACK": stack, **WALK_64B}, lambda: wf.render_host(count_traversal=True))
        for k in ("n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests"):
            assert sa[k] == sb[k], (stack, k)
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5)


def test_traversal_counters_are_those_of_the_timed_tree_form(renderer_factory):
    """count_traversal on the wavefront engine walks what the timed kernel walks: the 32-B node records (16-bit planes on one
    grid -- conservative supersets of the 64-B boxes, rt_nodeq.h) through trav_inner_step_q, the C++ statement of the hand-written
    loop.  Both forms test conservative supersets of the exact boxes (the 64-B one widens its far side by 6e-7 relative, the 32-B
    one snaps to a 16-bit grid), so the counters differ a little either way (+0.3 % node tests, +1.6 % triangle tests
    against the exact boxes on the Cornell box); rays and frame are the same bits."""
    sc = scenes.cornell_box(96, 64, 8, "path_mis", sphere_bsdfs=[Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("dielectric")])
    wf = renderer_factory(sc)
    wf.set_option("engine", "wavefront")
    assert wf.accel_info()["node_records_32b"] == 1
    plain, _ = wf.render_host()
    q, sq = wf.render_host(count_traversal=True)
    e, se = _with_env(WALK_64B, lambda: wf.render_host(count_traversal=True))
    assert np.array_equal(q, plain) and np.array_equal(e, plain)
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays"):
        assert sq[k] == se[k]
    assert abs(sq["n_node_tests"] / se["n_node_tests"] - 1.0) <= 0.03, (se["n_node_tests"], sq["n_node_tests"])
    assert abs(sq["n_tri_tests"] / se["n_tri_tests"] - 1.0) <= 0.08, (se["n_tri_tests"], sq["n_tri_tests"])
    assert (sq["n_node_tests"], sq["n_tri_tests"]) != (se["n_node_tests"], se["n_tri_tests"])      # two different conservative supersets of the same boxes


def test_kernel_class_timing(renderer_factory):
    import torch
    sc = scenes.cornell_box(128, 128, 16, "path_mis")
    for engine in ("megakernel", "wavefront"):
        r = renderer_factory(sc)
        r.set_option("engine", engine)
        frame = torch.zeros(r.frame_shape(), device="cuda")
        st = r.render_into(frame, time_kernels=True)
        assert st["trace_ms"] > 0 and st["film_ms"] > 0 and st["n_trace_launches"] >= 1
        assert (st["shade_ms"] > 0) == (engine == "wavefront")
        assert st["trace_ms"] + st["shade_ms"] + st["film_ms"] <= st["kernel_ms"] * 1.02 + 0.05
        st0 = r.render_into(frame)
        assert st0["trace_ms"] == 0 and st0["n_trace_launches"] == 0


@pytest.mark.parametrize("name", ["pa1-bunny", "pa4-cbox-distributed", "pa4-cbox-path_mis", "pa4-cbox-whitted",
                                  "pa4-motto-dielectric", "pa4-motto-diffuse", "pa5-cbox_mis", "pa5-cbox_ems", "pa5-cbox_mats",
                                  "pa5-table_mis", "pa5-table_ems", "pa5-table_mats", "pa5-veach_mis", "pa5-veach_ems", "pa5-veach_mats"])
def test_reference_scenes_both_engines_and_oracle(renderer_factory, name):
    """Every shipped scene whose meshes ship with the reference (the ajax scenes' OBJ does not): geometry, materials, camera,
    integrator as in the XML (tools/make_goldens.py; scene x integrator pairs that share their arrays share one fixture's; reduced resolution
    and sample count): both engines trace the same rays and produce the same frame, and the frame
    agrees with the CPU oracle within the SURVEY 8(d) image contract (>= 99.9 % of pixels within 1e-3,
    mean relative error <= 1e-4)."""
    sc = Scene.load_npz(os.path.join(GOLDEN, name + ".npz"))
    sc.camera.width, sc.camera.height = sc.camera.width // 4, sc.camera.height // 4
    sc.sample_count = 4
    mk, wf = _pair(renderer_factory, sc)
    a, sa = mk.render_host(count_traversal=True)
    b, sb = _with_env(WALK_64B, lambda: wf.render_host(count_traversal=True))
    for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_node_tests", "n_tri_tests", "n_invalid"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5)
    o = Oracle(sc, use_bvh=True)
    ref, so = o.render_host()
    for k in ("n_closest_rays", "n_shadow_rays"):
        assert int(so[k]) == int(sb[k]), k              # bit-identical paths (tests/test_gpu_parity.py): equal counts
    from tests.test_gpu_parity import assert_image_parity
    assert_image_parity(ref, b, wf.border, name)


def test_fuzz_engines_short():
    """A few rounds of tests/fuzz_engines.py: random scenes / materials / integrators / filters / path budgets,
    megakernel against wavefront."""
    from nori_amd.render import Renderer
    from tests import fuzz_engines
    for seed in range(2000, 2012):
        fuzz_engines.one_round(seed, Renderer, oracle=seed % 2 == 0)


def test_contexts_own_their_buffers():
    """Two contexts in one process (here on the same GPU): the wavefront engine's state pool, its streams and the film
    store belong to the context (include/nori_hip.h), so interleaved renders do not disturb each other and destroying
    one context leaves the other's buffers alone."""
    from nori_amd.render import Renderer
    sc1 = scenes.cornell_box(96, 64, 6, "path_mis")
    sc2 = scenes.cornell_box(48, 80, 5, "path_ems", sphere_bsdfs=[Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("diffuse")])
    a, b = Renderer(0).upload(sc1), Renderer(0).upload(sc2)
    for r in (a, b):
        r.set_option("engine", "wavefront")
    A1, _ = a.render_host()
    B1, _ = b.render_host()
    A2, _ = a.render_host()
    assert np.array_equal(A1, A2)
    _budget(b, 256 * 7)          # b regrows / rebatches its own pool; a is untouched
    B2, _ = b.render_host()
    np.testing.assert_allclose(B2, B1, rtol=1e-4, atol=1e-5)
    assert np.array_equal(a.render_host()[0], A1)
    a.close()                                           # frees a's pool, film store, streams -- not b's
    B3, _ = b.render_host()
    assert np.array_equal(B3, B2)
    c = Renderer(0).upload(sc1)
    c.set_option("engine", "wavefront")
    assert np.array_equal(c.render_host()[0], A1)
    b.close(); c.close()


@pytest.mark.parametrize("members,split,merge", [(1, "tile", "reduce"), (2, "tile", "reduce"), (2, "tile", "gather"), (3, "sample", "reduce"), (4, "tile", "gather")])
def test_device_group_shares_and_merges(members, split, merge):
    """nori_hip_group_* (what `nori scene.xml --gpus N` calls): one context + one host thread per group member, every
    member renders its share into its own frame, one merge on the first member.  On a one-GPU box the members are the same
    device listed several times -- the whole path (threads, shares, pack / add kernels, the merge) runs, only the transport
    is a device copy instead of RCCL.  The merged frame equals ONE render of the whole frame up to summation order."""
    from nori_amd.render import DeviceGroup, Renderer
    sc = scenes.cornell_box(64, 40, 12, "path_mis")
    whole, st = Renderer(0).upload(sc).render_host()
    g = DeviceGroup([0] * members).upload(sc)
    assert g.size == members and g.transport == "copy"
    got, gst, merge_ms = g.render_host(split, merge)
    assert gst["n_camera_samples"] == st["n_camera_samples"]
    assert gst["n_closest_rays"] == st["n_closest_rays"] and gst["n_shadow_rays"] == st["n_shadow_rays"]
    np.testing.assert_allclose(got, whole, rtol=2e-5, atol=1e-6)
    again, _, _ = g.render_host(split, merge)            # buffers are reused, frames cleared
    assert np.array_equal(again, got)
    g.close()


def test_device_group_errors_and_one_rank_rccl(tmp_path):
    """A device the node does not have is reported as such (`--gpus 2` on a one-GPU box fails at "device 1 not found", not
    earlier and not later); the gather merge refuses shares that are not whole tile columns; and RCCL itself -- loaded
    with dlopen, single-process communicators -- runs its reduce with a group of ONE real device."""
    import subprocess
    import torch
    from nori_amd import NoriError, _capi
    from nori_amd.render import DeviceGroup, Renderer
    n_dev = torch.cuda.device_count()
    with pytest.raises(NoriError, match=f"device {n_dev} not found"):
        DeviceGroup(list(range(n_dev + 1)))
    sc = scenes.cornell_box(48, 32, 4, "path_mis")        # 3 tile columns
    g = DeviceGroup([0, 0]).upload(sc)
    with pytest.raises(NoriError, match="gather merge needs"):
        g.render_host("tile", "gather")
    with pytest.raises(NoriError, match="gather merge needs"):
        g.render_host("sample", "gather")
    g.close()
    os.environ["NORI_GROUP_TRANSPORT"] = "rccl"
    try:
        g = DeviceGroup([0]).upload(sc)
        assert g.transport == "rccl"
        got, _, _ = g.render_host("tile", "reduce")
    finally:
        del os.environ["NORI_GROUP_TRANSPORT"]
    np.testing.assert_allclose(got, Renderer(0).upload(sc).render_host()[0], rtol=2e-5, atol=1e-6)
    assert g.warning == "" and g.engines() in ([0], [1])      # one entry per member
    g.close()
    # film_order = reference on some members only is refused (the frame would have no defined order)
    g = DeviceGroup([0, 0]).upload(sc)
    lib = _capi.load_hip()
    assert lib.nori_hip_set_option(lib.nori_hip_group_ctx(g._h, 1), b"film_order", b"reference") == 0
    with pytest.raises(NoriError, match="film_order = reference on some devices of the group only"):
        g.render_host("tile", "reduce")
    g.set_option("film_order", "fast")
    g.render_host("sample", "reduce")
    assert len(g.engines()) == 2
    g.close()
    # options read back as set_option takes them
    r = Renderer(0).upload(sc)
    lib = _capi.load_hip()
    import ctypes as C
    for key, value in (("engine", "wavefront"), ("film_order", "reference"), ("accel_layout", "bvh4q"), ("wavefront_paths", "4096"), ("wavefront_samples", "65536")):
        r.set_option(key, value)
        buf = C.create_string_buffer(32)
        assert lib.nori_hip_get_option(r._h, key.encode(), buf, 32) == 0 and buf.value.decode() == value
    assert lib.nori_hip_get_option(r._h, b"engine", buf, 2) != 0 and lib.nori_hip_get_option(r._h, b"nope", buf, 32) != 0
    r.close()
    # the CLI: --gpus beyond what the node has
    (tmp_path / "box.obj").write_text("v -1 -1 -1\nv 1 -1 -1\nv 1 1 -1\nv -1 1 -1\nv -1 -1 1\nv 1 -1 1\nv 1 1 1\nv -1 1 1\n"
                                      "f 1 2 3 4\nf 8 7 6 5\nf 1 5 6 2\nf 2 6 7 3\nf 3 7 8 4\nf 5 1 4 8\n")
    (tmp_path / "scene.xml").write_text("""<scene><integrator type="path_mis"/>
    <sampler type="independent"><integer name="sampleCount" value="8"/></sampler>
    <camera type="perspective"><float name="fov" value="60"/><integer name="width" value="48"/><integer name="height" value="32"/>
      <transform name="toWorld"><lookat origin="0,0,0.5" target="0,0,-1" up="0,1,0"/></transform></camera>
    <mesh type="obj"><string name="filename" value="box.obj"/>
      <bsdf type="diffuse"><color name="albedo" value="0.5, 0.5, 0.5"/></bsdf>
      <emitter type="area"><color name="radiance" value="1, 1, 1"/></emitter></mesh></scene>""")
    exe = os.path.join(_capi.LIB_DIR, "nori")
    p = subprocess.run([exe, str(tmp_path / "scene.xml"), "--gpus", str(n_dev + 1)], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and f"device {n_dev} not found" in p.stdout + p.stderr
    p = subprocess.run([exe, str(tmp_path / "scene.xml"), "--gpus", "1", "--split", "sample"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "1 GPUs, sample split" in p.stdout, p.stdout + p.stderr
    # --film-order reaches every context: one device or a group, the frame in reference order has the same bits
    from nori_amd import host
    frames = []
    for extra in ([], ["--gpus", "1"]):
        p = subprocess.run([exe, str(tmp_path / "scene.xml"), "--film-order", "reference"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        frames.append(host.load_exr(str(tmp_path / "scene.exr")).copy())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    p = subprocess.run([exe, str(tmp_path / "scene.xml"), "--film-order", "sideways"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--film-order expects" in p.stdout + p.stderr


@pytest.mark.parametrize("split,merge", [("tile", "reduce"), ("tile", "gather"), ("sample", "reduce")])
def test_device_group_over_rccl_on_distinct_devices(split, merge):
    """The real multi-GPU path: distinct devices, RCCL transport (communicators proved by the self-check of
    nori_hip_group_create), ncclReduce / grouped ncclSend + ncclRecv merges.  Needs a node with >= 2 GPUs: skipped on the
    one-GPU boxes this suite usually runs on (there the same code runs with peer copies and with a one-rank RCCL group)."""
    import torch
    from nori_amd.render import DeviceGroup, Renderer
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one GPU: RCCL with more than one rank cannot run here")
    n = 2 if n_dev < 4 else 4
    sc = scenes.cornell_box(64 * n, 48, 16, "path_mis")        # tile columns divisible by n
    whole, st = Renderer(0).upload(sc).render_host()
    g = DeviceGroup(list(range(n))).upload(sc)
    assert g.transport == "rccl" and g.warning == ""
    got, gst, merge_ms = g.render_host(split, merge)
    assert gst["n_closest_rays"] == st["n_closest_rays"] and gst["n_shadow_rays"] == st["n_shadow_rays"]
    np.testing.assert_allclose(got, whole, rtol=2e-5, atol=1e-6)
    assert np.array_equal(g.render_host(split, merge)[0], got)
    g.close()


@pytest.mark.parametrize("builder,layout", [(0, "bvh2"), (1, "bvh2"), (2, "bvh2"), (0, "bvh4q"), (2, "bvh4q")])
def test_lds_image_of_hot_records_changes_nothing(builder, layout):
    """wf_extend walks the hottest nodes and leaf pair records from an image in LDS (rt_top.h: built once per acceleration
    structure by a greedy walk from the root, links rewritten to LDS slots).  With the image switched off every link is a
    memory link: the frame -- same paths, same film order -- must be the same bits, for every builder and both node layouts."""
    from nori_amd.render import Renderer
    sc = scenes.cornell_box(96, 64, 6, "path_mis", sphere_bsdfs=[Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("dielectric")])
    frames = []
    for off in (False, True):
        if off:
            os.environ["NORI_HIP_NO_TOP_IMAGE"] = "1"
        try:
            r = Renderer(0); r.set_option("accel_layout", layout); r.upload(sc, builder=builder); r.set_option("engine", "wavefront")
            frames.append(r.render_host())
            r.close()
        finally:
            os.environ.pop("NORI_HIP_NO_TOP_IMAGE", None)
    (a, sa), (b, sb) = frames
    assert np.array_equal(a, b)
    assert sa["n_closest_rays"] == sb["n_closest_rays"] and sa["n_shadow_rays"] == sb["n_shadow_rays"]


@pytest.mark.parametrize("builder,deep", [(0, False), (1, True), (3, True)])
def test_hand_written_node_loop_equals_the_compilers(builder, deep):
    """wf_extend's default node loop walks the tree's 32-B records (rt_nodeq.h: 16-bit planes on one grid, conservative) in a block
    of gfx950 assembly; NORI_HIP_WF_NO_ASM_LOOP=1 walks the 64-B nodes with the compiler's loop.  The two visit different sets of
    nodes, but the triangles a ray hits -- and with them every path, the ray counts and the frame -- must be the same bits.  Shallow
    tree (stack in LDS), deep trees of the device builders (LDS stack + global column), with and without the LDS image, and with a
    tiny image (most nodes from memory)."""
    from nori_amd.render import Renderer
    if deep:
        sc = scenes.soup_scene(30000, seed=3, width=80, height=56, integrator="path_mis")
        sc.sample_count = 4
        from nori_amd.scene import Mesh
        v, f = scenes.quad((-3, 3, -3), (3, 3, -3), (3, 3, 3), (-3, 3, 3))
        sc.meshes.append(Mesh(v, f, bsdf=Bsdf("diffuse", (0, 0, 0)), radiance=(5.0, 5.0, 5.0), name="light"))
    else:
        sc = scenes.cornell_box(96, 64, 6, "path_mis", sphere_bsdfs=[Bsdf("microfacet", (0.2, 0.3, 0.1), 0.2), Bsdf("dielectric")])
    frames = {}
    for name, env in (("asm", {}), ("compiler", {"NORI_HIP_WF_NO_ASM_LOOP": 1}), ("asm, no image", {"NORI_HIP_NO_TOP_IMAGE": 1}),
                      ("asm, 3 cached nodes", {"NORI_HIP_TOP_NODES": 3})):
        def run():
            r = Renderer(0); r.set_option("accel_layout", "bvh2"); r.upload(sc, builder=builder); r.set_option("engine", "wavefront")
            info = r.accel_info()
            out = r.render_host()
            r.close()
            return info, out
        info, frames[name] = _with_env(env, run)
        assert info["node_records_32b"] == 1 and info["node_children"] == 2
        assert (info["max_depth"] + 1 > 16) == deep
    ref, sr = frames["compiler"]
    for name, (f, s) in frames.items():
        assert s["n_closest_rays"] == sr["n_closest_rays"] and s["n_shadow_rays"] == sr["n_shadow_rays"], name
        assert np.array_equal(f, ref), name


def test_tree_with_unbounded_boxes_takes_the_64_byte_nodes():
    """A numerically collinear triangle hangs under the root in an unbounded box: no 32-B records for that tree, and the frame
    equals the megakernel's."""
    from nori_amd.render import Renderer
    from nori_amd.scene import Mesh
    sc = scenes.cornell_box(64, 48, 4, "path_mis")
    a, b = np.float32([0.1, 0.2, 0.3]), np.float32([0.7, 0.9, -0.2])
    sc.meshes = list(sc.meshes) + [Mesh(np.float32([a, b, 0.5 * (a + b)]), np.uint32([[0, 1, 2]]))]
    out = {}
    for engine in ("megakernel", "wavefront"):
        r = Renderer(0); r.upload(sc); r.set_option("engine", engine)
        assert r.accel_info()["node_records_32b"] == 0
        out[engine] = r.render_host()
        r.close()
    np.testing.assert_allclose(out["wavefront"][0], out["megakernel"][0], rtol=1e-4, atol=1e-5)      # (the engines add a pixel's samples in different orders)
    for k in ("n_closest_rays", "n_shadow_rays"):
        assert out["megakernel"][1][k] == out["wavefront"][1][k]


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["megakernel", "wavefront"])
def test_reference_order_frame_has_the_same_bits_for_any_number_of_shares(engine):
    """film_order = reference shared out by rows of 32x32 blocks (nori_hip_render_block_rows): every share writes the
    accumulators of its own blocks, the arrays are summed -- in any order: they are disjoint -- and the blocks added into the
    frame in BlockGenerator's order.  The frame must have the bits of the one-device frame (which tests/test_gpu_parity.py
    pins, bit for bit, to the single-threaded oracle) for every number of shares, more shares than rows included."""
    import torch
    from nori_amd import dist as ndist
    from nori_amd.render import Renderer
    sc = scenes.cornell_box(112, 72, 5, "path_mis")          # 4 x 3 blocks, the last column / row clipped; 7 x 5 tiles
    r = Renderer(0).upload(sc)
    r.set_option("film_order", "reference"); r.set_option("engine", engine)
    whole, st_whole = r.render_host()
    assert r.block_rows() == 3
    for world in (1, 2, 3, 5):
        accs, rays = [], 0
        for rank in range(world):
            acc = torch.zeros(r.block_acc_floats(), dtype=torch.float32, device="cuda:0")
            r0, rn = ndist.block_rows(rank, world, r.block_rows())
            st = r.render_block_rows_into(acc, r0, rn)
            rays += st["n_closest_rays"] + st["n_shadow_rays"]
            accs.append(acc)
        total = torch.zeros_like(accs[0])
        for k in reversed(range(world)):                      # an order of its own
            total += accs[k]
        frame = torch.zeros(r.frame_shape(), dtype=torch.float32, device="cuda:0")
        r.resolve_blocks(total, frame)
        got = frame.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), (world, float(np.abs(got - whole).max()))
        assert rays == st_whole["n_closest_rays"] + st_whole["n_shadow_rays"]
    # the fast film refuses the call, and so does a tile share
    r.set_option("film_order", "fast")
    from nori_amd import NoriError
    with pytest.raises(NoriError, match="needs film_order = reference"):
        r.render_block_rows_into(torch.zeros(r.block_acc_floats(), dtype=torch.float32, device="cuda:0"), 0, 1)
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("members", [2, 3, 4])
def test_device_group_in_reference_order_gives_the_one_device_bits(members):
    """The C++ group with film_order = reference on its contexts: block rows per member, the accumulators merged, the blocks
    added in the reference's order on the first device -- same bits as one device, whatever `split` says."""
    from nori_amd.render import DeviceGroup, Renderer
    sc = scenes.cornell_box(80, 96, 3, "path_mis")           # 3 block columns (the last clipped), 3 block rows
    r = Renderer(0).upload(sc)
    r.set_option("film_order", "reference")
    whole, st_whole = r.render_host()
    r.close()
    g = DeviceGroup([0] * members).upload(sc)
    g.set_option("film_order", "reference")
    for split in ("tile", "sample"):
        got, st, ms = g.render_host(split, "reduce")
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), (members, split, float(np.abs(got - whole).max()))
        for k in ("n_camera_samples", "n_closest_rays", "n_shadow_rays", "n_invalid"):
            assert st[k] == st_whole[k], (k, st[k], st_whole[k])
    assert len(g.engines()) == members
    g.close()


@pytest.mark.gpu
def test_reference_order_over_torch_distributed_ranks():
    """nori_amd.dist.render_distributed_reference through a launcher-started one-rank RCCL group (what a rank of an N-GPU run
    executes): the merged frame has the one-device bits."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.getcwd())
        import numpy as np, torch, torch.distributed as dist
        from nori_amd import dist as ndist
        from nori_amd.render import Renderer
        from tests import scenes
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
        sc = scenes.cornell_box(96, 64, 4, "path_mis")
        r = Renderer(0).upload(sc)
        r.set_option("film_order", "reference")
        whole, _ = r.render_host()
        frame = torch.zeros(r.frame_shape(), dtype=torch.float32, device="cuda:0")
        ms = []
        ndist.render_distributed_reference(r, frame, sc.sample_count, dist.get_rank(), dist.get_world_size(), merge_ms=ms)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), whole.view(np.uint32))
        assert len(ms) == 1
        side = torch.cuda.Stream()          # a caller's non-blocking stream carries every step of the call
        frame2 = torch.full(r.frame_shape(), 7.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        ndist.render_distributed_reference(r, frame2, sc.sample_count, dist.get_rank(), dist.get_world_size(), stream=side)
        side.synchronize()
        assert np.array_equal(frame2.cpu().numpy().view(np.uint32), whole.view(np.uint32))
        dist.destroy_process_group()
        print("REFERENCE-ORDER-RANKS-OK")
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "REFERENCE-ORDER-RANKS-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
