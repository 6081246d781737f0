"""N > 1 path on the CPU: two processes, gloo backend, the tile split and the
sample split + SUM-reduce of nori_amd.dist, with the emulated device code as
the per-rank renderer.  The result on rank 0 must equal the single-process
render up to float summation order."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from nori_amd import dist as ndist
from tests import scenes
from tests.backends import Emu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = scenes.cornell_box(40, 24, 6, "path_mis")
emu = Emu(sc)
def render_fn(frame, **kw):
    rgbw, st = emu.render_host(**kw)
    frame += torch.from_numpy(rgbw)
    return st
for mode in ("tile", "sample"):
    frame = torch.zeros(emu.frame_shape(), dtype=torch.float32)
    st = ndist.render_distributed(render_fn, frame, mode, sc.sample_count, rank, world)
    rays = torch.tensor([float(st["n_camera_samples"])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        np.save(os.path.join({out!r}, mode + ".npy"), frame.numpy())
        np.save(os.path.join({out!r}, mode + "_cam.npy"), rays.numpy())
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_two_process_split_and_reduce(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + os.getpid() % 1000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    from tests import scenes
    from tests.backends import Emu
    sc = scenes.cornell_box(40, 24, 6, "path_mis")
    whole, st = Emu(sc).render_host()
    for mode in ("tile", "sample"):
        got = np.load(tmp_path / f"{mode}.npy")
        np.testing.assert_allclose(got, whole, rtol=2e-5, atol=1e-6, err_msg=mode)
        assert np.load(tmp_path / f"{mode}_cam.npy")[0] == st["n_camera_samples"]


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
