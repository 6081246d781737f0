"""N > 1 path on the CPU, through bench.py itself: `bench.py --emulate` runs the product's sharding / merge /
reporting code (nori_amd.dist, the JSON line) on gloo ranks with the emulated device headers as the per-rank
renderer -- the same code path `bench.py --gpus N` takes on a GPU node, minus the GPU.  The merged frame on
rank 0 must equal the single-process render up to float summation order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SIZE = dict(width=64, height=48, spp=3)


def _whole():
    from nori_amd import workloads
    from tests.backends import Emu
    sc = workloads.load("pa4-cbox-path_mis", SIZE["width"], SIZE["height"], SIZE["spp"]).scene
    return Emu(sc).render_host()


@pytest.fixture(scope="module")
def whole():
    return _whole()


@pytest.mark.parametrize("world,split,merge", [(1, "tile", "reduce"), (1, "tile", "gather"), (2, "tile", "reduce"), (2, "tile", "gather"), (2, "sample", "reduce"), (3, "tile", "reduce"),
                                               (3, "sample", "reduce"), (4, "tile", "gather"), (8, "tile", "reduce"), (8, "sample", "reduce")])
def test_device_group_driver_on_cpu_ranks(whole, world, split, merge):
    """The driver behind `nori scene.xml --gpus N` / nori_hip_group_render_host (group.hip): one thread per rank rendering its
    share (group_merge.h), then the merge on rank 0 -- with the emulated device code as every rank's renderer.  Equal to ONE
    render of the whole frame up to float summation order; equal ray counts."""
    from nori_amd import workloads
    from tests.backends import Emu
    sc = workloads.load("pa4-cbox-path_mis", SIZE["width"], SIZE["height"], SIZE["spp"]).scene
    got = Emu(sc).group_render_host(world, split, merge)
    assert got is not None
    ref, st = whole
    np.testing.assert_allclose(got[0], ref, rtol=2e-5, atol=1e-6)
    assert got[1]["n_closest_rays"] == st["n_closest_rays"] and got[1]["n_shadow_rays"] == st["n_shadow_rays"]


def test_device_group_gather_rules_and_wide_frames():
    """The gather merge needs whole tile columns per rank; eight ranks on a 128-px-wide frame (8 tile columns), and a border
    so wide that a rank's own strips overlap (16 world < 16 + 2 border): every column is packed once."""
    from nori_amd import workloads
    from nori_amd.scene import RFilter
    from tests import scenes
    from tests.backends import Emu
    sc = workloads.load("pa4-cbox-path_mis", 64, 48, 2).scene          # 4 tile columns
    assert Emu(sc).group_render_host(3, "tile", "gather") is None      # 4 % 3 != 0
    assert Emu(sc).group_render_host(2, "sample", "gather") is None    # gather is a tile-split merge
    for world, width, radius in ((8, 128, 2.0), (2, 64, 8.4), (4, 64, 6.0)):
        sc = scenes.cornell_box(width, 20, 2, "path_mis", rfilter=RFilter("gaussian", radius=radius, stddev=radius / 4))
        e = Emu(sc)
        ref, st = e.render_host()
        for merge in ("gather", "reduce"):
            got = e.group_render_host(world, "tile", merge)
            np.testing.assert_allclose(got[0], ref, rtol=2e-5, atol=1e-6, err_msg=f"{world} ranks, {merge}, radius {radius}")
            assert got[1]["n_closest_rays"] == st["n_closest_rays"]


@pytest.mark.parametrize("world,split,merge", [(2, "tile", "reduce"), (2, "tile", "gather"), (2, "sample", "reduce"),
                                               (3, "tile", "reduce"), (4, "tile", "gather"), (8, "tile", "reduce")])
def test_bench_spawns_ranks_and_merges(tmp_path, whole, world, split, merge):
    """`python bench.py --gpus N` starts its own N ranks (no external launcher) and rank 0 holds the merged frame."""
    frame = tmp_path / "frame.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, BENCH, "--gpus", str(world), "--emulate", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--workload", "pa4-cbox-path_mis", "--width", str(SIZE["width"]), "--height", str(SIZE["height"]), "--spp", str(SIZE["spp"]),
           "--split", split, "--merge", merge, "--dump-frame", str(frame)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    ref, st = whole
    assert out["n_gpus"] == world and out["config"]["workload"] == "pa4-cbox-path_mis"
    assert out["config"]["rays_per_step"] == st["n_closest_rays"] + st["n_shadow_rays"]
    assert out["config"]["parallelism"] == f"{split}-split x{world} + RCCL {merge}"
    assert 0.0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["bound"] in ("valu", "hbm")
    np.testing.assert_allclose(np.load(frame), ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("world,split,merge", [(2, "tile", "gather"), (3, "sample", "reduce")])
def test_bench_line_of_a_multi_rank_run_is_complete(world, split, merge):
    """A line for N > 1 carries what a line for N = 1 carries -- roofline, cpu_baseline (rank 0's host cores) and a parity block,
    here on the frame MERGED from all ranks -- plus what every rank saw (its wall clock, its kernels, its time in the merge)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, BENCH, "--gpus", str(world), "--emulate", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0.5",
           "--workload", "pa4-cbox-path_mis", "--width", "64", "--height", "32", "--spp", "4", "--split", split, "--merge", merge]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == world and out["scaling"] == "strong"
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and "every 1. tile" in out["cpu_baseline"]["sample"]
    par = out["parity"]
    assert par["ok"] and par["rays_cpu"] == par["rays_gpu"] and par["frac_within_1e-3"] >= 0.999 and f"{world} ranks" in par["frame"]
    assert [r["rank"] for r in out["ranks"]] == list(range(world))
    assert all(r["ms_per_step"] > 0 and r["merge_ms"] >= 0 and r["rays_per_step"] > 0 for r in out["ranks"])
    assert sum(r["rays_per_step"] for r in out["ranks"]) == out["config"]["rays_per_step"]


def test_bench_without_gpu_fails_at_no_gpu():
    """On a box without a GPU the multi-GPU entry gets as far as 'no GPU' -- not an assert, not a launcher error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert "no GPU" in p.stdout + p.stderr
    assert "AssertionError" not in p.stderr


def test_gather_needs_divisible_columns():
    from nori_amd.dist import column_strips, gather_frame
    import torch
    x, valid = column_strips(1, 2, 4, 2, 64 + 4)
    assert x.tolist()[:3] == [16, 17, 18] and len(x) == 2 * 20 and bool(valid.all())
    x, valid = column_strips(0, 1, 3, 2, 40 + 4)       # 40-px-wide image: the third tile column is clipped; one rank's strips overlap
    assert int(valid.sum()) == 40 + 4 and sorted(x[valid].tolist()) == list(range(44))      # every frame column packed exactly once
    with pytest.raises(ValueError):
        gather_frame(torch.zeros(8, 52, 4), 0, 2, 3, 2)


def test_gather_packs_every_column_once():
    """A rank's own column strips overlap when 16 world < 16 + 2 border -- always for a process group of ONE rank (what
    `bench.py --gpus 1` under a launcher creates): the halo columns must not be added twice."""
    import tempfile
    import torch
    import torch.distributed as dist
    from nori_amd.dist import column_strips, gather_frame
    for world, tiles_x, border in ((1, 2, 2), (1, 5, 2), (2, 4, 9), (3, 6, 20)):
        cols = tiles_x * 16 + 2 * border
        seen = torch.zeros(cols, dtype=torch.int64)
        for r in range(world):
            x, valid = column_strips(r, world, tiles_x, border, cols)
            seen.index_add_(0, x, valid.to(torch.int64))
        assert int(seen.min()) >= 1                      # every column some rank touched is packed ...
        x0, v0 = column_strips(0, world, tiles_x, border, cols)
        assert len(set(x0[v0].tolist())) == int(v0.sum())      # ... and by one rank at most once
    with tempfile.TemporaryDirectory() as d:
        dist.init_process_group("gloo", init_method=f"file://{d}/store", rank=0, world_size=1)
        try:
            frame = torch.ones(7, 2 * 16 + 4, 4)
            gather_frame(frame, 0, 1, 2, 2)
            assert torch.equal(frame, torch.ones(7, 36, 4))
        finally:
            dist.destroy_process_group()


def test_shard_partitions():
    from nori_amd.dist import shard
    for world in (1, 2, 3, 8):
        for spp in (1, 7, 256):
            parts = [shard("sample", r, world, spp) for r in range(world)]
            assert sum(p["spp_count"] for p in parts) == spp
            pos = 0
            for p in parts:
                assert p["spp_begin"] == pos
                pos += p["spp_count"]
            tiles = [shard("tile", r, world, spp) for r in range(world)]
            assert [t["tile_rem"] for t in tiles] == list(range(world)) and all(t["tile_mod"] == world for t in tiles)


@pytest.mark.parametrize("size", [(1024, 1024), (112, 72), (33, 31), (16, 200), (800, 600)])
@pytest.mark.parametrize("world", [1, 2, 3, 8, 40])
def test_block_row_shares_of_the_reference_order_film(size, world):
    """film_order = reference is shared out by rows of 32x32 blocks (a block's samples are added consecutively): the rows of the
    ranks are contiguous, disjoint and cover the frame; rendered as tiles they are contiguous tile ranges that partition the frame's
    tiles; the C++ group (group_merge.h, film.h -- through the CPU harness) and nori_amd.dist hand out the same rows."""
    import ctypes as C
    from nori_amd import dist as ndist
    from tests.backends import emu_lib
    w, h = size
    lib = emu_lib()
    rows_total = (h + 31) // 32
    tiles = ((w + 15) // 16) * ((h + 15) // 16)
    next_row, next_tile = 0, 0
    for rank in range(world):
        out = (C.c_uint32 * 4)()
        assert lib.emu_group_block_rows(rank, world, w, h, out) == 0
        r0, rn, t0, tn = (int(v) for v in out)
        assert (r0, rn) == ndist.block_rows(rank, world, rows_total)
        assert r0 == next_row and t0 == next_tile
        assert rn in (rows_total // world, rows_total // world + 1)
        # a row of blocks = two rows of tiles (one where the frame ends inside it)
        assert tn == (min(2 * (r0 + rn), (h + 15) // 16) - min(2 * r0, (h + 15) // 16)) * ((w + 15) // 16)
        next_row, next_tile = r0 + rn, t0 + tn
    assert next_row == rows_total and next_tile == tiles


_REFERENCE_RANK = r'''
import os, sys
sys.path.insert(0, os.environ["NORI_REPO"])
import numpy as np, torch, torch.distributed as dist
from nori_amd import dist as ndist

class BlockFilm:
    """Stand-in for a Renderer in reference order: a block's accumulator is a fixed pseudo-random array (what it holds does not matter
    here, that every rank computes the SAME array for a block does), resolve_blocks adds the blocks covering a pixel in a fixed order."""
    def __init__(self, w, h, border):
        self.w, self.h, self.b = w, h, border
        self.bx, self.by = (w + 31) // 32, (h + 31) // 32
        self.side = 32 + 2 * border
    def block_rows(self): return self.by
    def block_acc_floats(self): return self.bx * self.by * self.side * self.side * 4
    def frame_shape(self): return (self.h + 2 * self.b, self.w + 2 * self.b, 4)
    def render_block_rows_into(self, acc, r0, rn, spp_count=None):
        a = acc.view(self.by, self.bx, self.side, self.side, 4)
        for by in range(r0, r0 + rn):
            for bx in range(self.bx):
                g = torch.Generator().manual_seed(1000 * by + bx)
                a[by, bx] = torch.rand(self.side, self.side, 4, generator=g) * (10.0 ** ((by + bx) % 5 - 2))      # magnitudes differ: order matters
        return {"rows": rn}
    def resolve_blocks(self, acc, frame):
        a = acc.view(self.by, self.bx, self.side, self.side, 4)
        order = sorted(((by, bx) for by in range(self.by) for bx in range(self.bx)), key=lambda t: (t[0] * 7 + t[1] * 3) % 11 * 100 + t[0] * 10 + t[1])
        for by, bx in order:                              # a fixed order that is not raster order
            hh, ww = min(32, self.h - 32 * by) + 2 * self.b, min(32, self.w - 32 * bx) + 2 * self.b
            frame[32 * by:32 * by + hh, 32 * bx:32 * bx + ww] += a[by, bx, :hh, :ww]
        return frame

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="file://" + os.environ["NORI_STORE"], rank=rank, world_size=world)
film = BlockFilm(100, 150, 2)                            # 4 x 5 blocks, clipped at both ends; 5 rows over 2 or 3 ranks
frame = torch.zeros(film.frame_shape())
ms = []
st = ndist.render_distributed_reference(film, frame, 4, rank, world, merge_ms=ms)
assert st["rows"] == ndist.block_rows(rank, world, film.block_rows())[1] and len(ms) == 1
if rank == 0:
    whole_acc = torch.zeros(film.block_acc_floats())
    film.render_block_rows_into(whole_acc, 0, film.block_rows())
    whole = film.resolve_blocks(whole_acc, torch.zeros(film.frame_shape()))
    assert torch.equal(frame.view(torch.int32), whole.view(torch.int32)), float((frame - whole).abs().max())
    assert float(whole.abs().sum()) > 0
    print("REFERENCE-ORDER-GLOO-OK")
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_reference_order_merge_over_gloo_ranks(tmp_path, world):
    """nori_amd.dist.render_distributed_reference with world_size 2 and 3 (gloo, CPU tensors, a stand-in film): every rank fills the
    accumulators of its block rows, one reduce of the disjoint arrays, rank 0 adds the blocks in its fixed order -- the merged frame has
    the bits of the frame one rank computes alone, although the blocks' magnitudes differ by 10^4 (a sum of per-rank FRAMES would not)."""
    script = tmp_path / "rank.py"
    script.write_text(_REFERENCE_RANK)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), NORI_STORE=str(tmp_path / "store"), NORI_REPO=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o + e
    assert "REFERENCE-ORDER-GLOO-OK" in outs[0][0]
