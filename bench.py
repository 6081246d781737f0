#!/usr/bin/env python
"""bench.py -- Mrays/s of the hot path (BVH traversal + Li path loop + tile splat).

One "step" = one complete render pass of the workload (every pixel, every
sample) through libnori_hip.  N GPUs (one process per GPU under
torch.distributed.run) split the 16x16 tiles of the frame round-robin and rank 0
receives the RCCL sum-reduce of the RGBW frame (ImageBlock::put(ImageBlock&)
across GPUs).  Total work is fixed as N grows -> "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) extended with
`roofline` (dominant kernel: render_kernel) and `cpu_baseline` (the oracle,
timed on the host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_workload(name: str, width: int, height: int, spp: int):
    from nori_amd.scene import Scene
    from tests import scenes
    golden = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    if os.path.exists(golden):
        sc = Scene.load_npz(golden)
    elif name == "procedural-cornell":
        sc = scenes.cornell_box(width, height, spp, "path_mis", sphere_subdiv=4)
    else:
        raise SystemExit(f"unknown workload {name}")
    sc.camera.width, sc.camera.height, sc.sample_count = width, height, spp
    return sc


def algorithmic_bytes(stats: dict, info: dict, tile_w: int, n_tiles: int, engine: str) -> dict:
    """Algorithmic bytes of one render pass:
      traversal  N_node * 64 + N_tri * 48                     (both engines; from the COUNT pass)
      surface    96 B per closest-hit ray: the triangle's pre-gathered shading record (3 positions +
                 3 normals as 16-B records)
      film       24 B written + 24 B read per camera sample (sample store), plus one
                 read-modify-write of every tile accumulator and one frame read-modify-write
      state      wavefront only (dense ping-pong path state, wavefront.hip): per path vertex
                 (= closest-hit ray) wf_extend reads 36 B + writes the 16-B hit, wf_shade reads 80 B
                 and the surviving path is written back as 80 B; per shadow ray another 64 B
                 (direction + emitter sample, written and read once each); the first vertex of
                 every path is recomputed, not stored (-180 B per camera sample):
                 212 B * closest + 64 B * shadow - 180 B * camera samples.
                 0 for the megakernel, whose paths live in registers."""
    trav = stats["n_node_tests"] * info["node_bytes"] + stats["n_tri_tests"] * info["tri_bytes"]
    surface = stats["n_closest_rays"] * 96
    film = stats["n_camera_samples"] * 48 + n_tiles * 2 * 16 * tile_w * tile_w
    state = 0
    if engine == "wavefront":
        state = stats["n_closest_rays"] * 212 + stats["n_shadow_rays"] * 64 - stats["n_camera_samples"] * 180
    return {"traversal": int(trav), "surface": int(surface), "film": int(film), "state": int(state),
            "total": int(trav + surface + film + state)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("NORI_BENCH_WORKLOAD", "pa4-cbox-path_mis"))
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--engine", default="auto", choices=["auto", "megakernel", "wavefront"])
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if not os.path.exists(os.path.join(ROOT, "tests", "golden", f"{args.workload}.npz")):
        args.workload = "procedural-cornell"
    sc = load_workload(args.workload, args.width, args.height, args.spp)

    from nori_amd.render import Renderer
    r = Renderer(local_rank).upload(sc)
    tiles = ((args.width + 15) // 16) * ((args.height + 15) // 16)
    my_tiles = (tiles - rank + world - 1) // world
    engine = args.engine
    if engine == "auto":      # the library's rule (nori_hip.h, nori_hip_set_option)
        engine = "wavefront" if my_tiles * 256 * args.spp >= (1 << 24) else "megakernel"
    r.set_option("engine", engine)
    info = r.accel_info()
    frame = torch.zeros(r.frame_shape(), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    from nori_amd import dist as ndist

    def step(want_stats=False, count=False, time_kernels=False):
        # tile split over ranks + one RCCL SUM-reduce of the RGBW frame to rank 0
        return ndist.render_distributed(r.render_into, frame, "tile", args.spp, rank, world, stream=stream,
                                        want_stats=want_stats, count_traversal=count, time_kernels=time_kernels)

    # one instrumented pass: traversal counters for the roofline (untimed)
    counted = step(want_stats=True, count=True)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, trace_ms, shade_ms, film_ms, trace_launches = [], [], [], [], 0
    last = None
    for _ in range(args.steps):
        # stats: HIP-event times on the launch stream -- whole pass and per kernel class
        last = step(want_stats=True, time_kernels=True)
        kernel_ms.append(last["kernel_ms"]); trace_ms.append(last["trace_ms"]); shade_ms.append(last["shade_ms"])
        film_ms.append(last["film_ms"]); trace_launches = last["n_trace_launches"]
    barrier()
    dt = time.perf_counter() - t0

    rays_local = float(last["n_closest_rays"] + last["n_shadow_rays"])
    t = torch.tensor([dt, rays_local], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_total = float(tmax[0]), float(tsum[1])
    else:
        rays_total = rays_local

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        mrays = rays_total / (ms_per_step * 1e-3) / 1e6
        # roofline of the dominant kernel on this rank: the ray-query kernel (wf_extend, all its launches
        # of one pass; render_kernel for the megakernel, which also contains the shading)
        tile_w = 16 + 2 * r.border
        parts = algorithmic_bytes(counted, info, tile_w, my_tiles, engine)
        k_ms, t_ms = float(np.mean(kernel_ms)), float(np.mean(trace_ms))
        if engine == "wavefront":
            dom_name = "wf_extend (all launches of one render pass)"
            dom_bytes = parts["traversal"] + counted["n_closest_rays"] * 52 + counted["n_shadow_rays"] * 16 - counted["n_camera_samples"] * 36
        else:
            dom_name = "render_kernel (all launches of one render pass)"
            dom_bytes = parts["traversal"] + parts["surface"] + counted["n_camera_samples"] * 24
        achieved = dom_bytes / (t_ms * 1e-3) / 1e9
        traffic = measured_traffic(args, sc, engine)
        compulsory = info["total_bytes"] + parts["state"] + parts["film"]       # each byte that has to cross HBM at least once
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1024x1024 256spp",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "integrator": sc.integrator.type, "width": args.width,
                       "height": args.height, "spp": args.spp, "triangles": info["n_triangles"],
                       "parallelism": f"tile-split x{world} + RCCL reduce" if world > 1 else "single GPU",
                       "rays_per_step": int(rays_total), "seed_mode": "per_sample", "engine": engine},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic.get("dominant_kernel_bytes") if traffic else None,
                         "kernel_ms": round(t_ms, 3), "launches": int(trace_launches), "algorithmic_bytes": int(dom_bytes),
                         "node_tests": int(counted["n_node_tests"]), "tri_tests": int(counted["n_tri_tests"]),
                         "note": "algorithmic bytes per SURVEY.md 8(d): N_node*64 + N_tri*48 + ray/hit records; "
                                 "this scene's BVH (%.1f MB) is L2-resident, so the node/triangle bytes are served by L2 and "
                                 "'achieved' can exceed the HBM peak -- see 'pass' for what HBM has to carry" % (info["total_bytes"] / 1e6)},
            "pass": {"kernel_ms": round(k_ms, 3), "trace_ms": round(t_ms, 3), "shade_ms": round(float(np.mean(shade_ms)), 3),
                     "film_ms": round(float(np.mean(film_ms)), 3), "launches": int(last["n_workgroups"]) if engine == "wavefront" else None,
                     "algorithmic_bytes_parts": parts,
                     "hbm_compulsory_bytes": int(compulsory),
                     "hbm_compulsory_gbs": round(compulsory / (k_ms * 1e-3) / 1e9, 2),
                     "hbm_compulsory_frac": round(compulsory / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "hbm_measured_bytes": traffic.get("hbm_bytes_per_launch") if traffic else None},
            "accel": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in info.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sc, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, sc, engine):
    """HBM bytes per launch of render_kernel from the PMC passes (FETCH_SIZE, WRITE_SIZE; separate
    rocprofv3 --pmc runs, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  PMC counters cannot be read
    from inside this process, so the figure comes from the committed summary of a profiled run of this
    same command (profiles/*_traffic.json); null if there is none for this workload."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    key = f"{args.workload}:{args.width}x{args.height}:{args.spp}:{engine}"
    return t.get(key)


def usable_cores() -> int:
    """Host cores this process may really use: CPU affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / p_ + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sc, args):
    """The oracle (kind 'port': CPU restatement with a SAH BVH) on all host cores,
    on a bounded sample: the same frame at reduced spp, sized for ~cpu_seconds."""
    from tests.backends import Oracle
    o = Oracle(sc, use_bvh=True)
    cores = usable_cores()
    _, st = o.render_host(spp_count=1, threads=cores)
    per_spp = max(st["kernel_ms"] * 1e-3, 1e-3)
    spp = int(max(1, min(args.spp, args.cpu_seconds / per_spp)))
    _, st = o.render_host(spp_count=spp, threads=cores)
    rays = st["n_closest_rays"] + st["n_shadow_rays"]
    sec = st["kernel_ms"] * 1e-3
    return {"value": round(rays / sec / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{args.width}x{args.height} at {spp} spp of {args.spp} ({rays} rays in {sec:.1f} s, "
                      f"std::thread x{cores}, SAH BVH, per-sample seeding)"}


if __name__ == "__main__":
    main()
