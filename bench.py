#!/usr/bin/env python
"""bench.py -- Mrays/s of the hot path (Accel traversal + Integrator::Li path loop + ImageBlock splat).

One "step" = one complete render pass of the workload (every pixel, every sample) through libnori_hip
(C ABI, include/nori_hip.h).  `--workload` names a BASELINE.json configuration (nori_amd/workloads.py;
default: configs[2], the pa4 Cornell box, path_mis, 1024 x 1024, 256 spp).

N GPUs = one process per GPU.  `python bench.py --gpus N` spawns the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) unless it is
already running under such a launcher (WORLD_SIZE set).  The path shards with no data-path collective:
  --split tile     16x16 tiles round-robin over the ranks (ImageBlock tiles; default for C1-C4)
  --split sample   every rank renders the whole frame with its share of the samples per pixel (C5)
and ends with ONE exchange of the RGBW frame to rank 0 -- ImageBlock::put(ImageBlock&) (src/block.cpp:93-102)
across GPUs: `--merge reduce` (RCCL sum-reduce, either split) or `--merge gather` (RCCL gather of each rank's
tile columns + halo, tile split only).  Total work is fixed as N grows -> "scaling": "strong".

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      dominant kernel (the ray-query kernel, all its launches of one pass), timed live with HIP events on
                the launch stream.  bound "valu": scenes whose BVH is cache-resident are VALU-issue bound
                (profiles/*_counters.json: SQ_INSTS_VALU x the kernel's cycles per instruction vs elapsed cycles);
                achieved = algorithmic f32 vector operations (52 per two-box node test, 54 per triangle test, 9 per
                ray) / time against 78.6 Tops/s (256 CUs x 4 SIMD x 32 lanes x 2.4 GHz, non-FMA);
                beside it `flop_frac` (the same operations against the 157.3 TFLOP/s f32 peak) and `issued_frac` (the
                VALU instructions the shipped code issues per unit against the issue rate).  bound "hbm": trees beyond
                L2 + Infinity Cache; achieved = algorithmic bytes / time against 8 TB/s.
                `traffic` = PMC-measured HBM bytes of that kernel per pass, from the committed counter summary
                whose device-source hash matches this build (else null), stamped with its source.
  parity        the GPU render of exactly the sample range the CPU baseline rendered, compared with it
                (fraction of pixels within 1e-3, mean relative error, ray counts)
  cpu_baseline  the oracle (CPU restatement, `-O3 -march=native -ffp-contract=off`, SAH BVH, std::thread over
                32x32 blocks) on the host cores, on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TOPS = 78.6          # 256 CUs x 4 SIMD-32 x 2.4 GHz: one non-FMA f32 lane-operation per lane per clock
OPS_NODE, OPS_TRI, OPS_RAY = 52, 54, 9      # algorithmic f32 vector ops per two-box node test / triangle test / ray setup
OPS_NODE_WIDE = 103                          # ... per WIDE node test (four quantised boxes: 21 setup + 6 selects + 48 planes + 16 min/max + 12 compares)
# the same work as the kernel ISSUES it (rt_trace.h): VALU instructions per unit of the shipped code
ISSUED_NODE, ISSUED_TRI, ISSUED_RAY = 30, 37.5, 9      # slab_two: 6 v_pk + 12 v_fma + 4 min3/max3 + 2 v_mul + 6 v_cmp; tri_pair_test: ~75 per PAIR of triangles
ISSUED_NODE_32B = 40                          # the hand-written loop on 32-B records: 6 v_bfi + 12 v_cvt (SDWA) + 12 v_fma + 4 min3/max3 + 6 v_cmp
ISSUED_NODE_WIDE = 85
VALU_ISSUE_SLOTS = 256 * 4 * 2.4e9 / 2.4 * 64          # lane-instructions per second at the measured best case of 2.4 cycles per wave64 instruction
FLOP_PEAK_TFLOPS = 157.3                               # MI355X_MICROARCH.md: f32 vector peak (v_pk_fma_f32: 2 lanes x 2 flops)
CACHE_RESIDENT_BYTES = 256 << 20             # a BVH below this lives in L2 + Infinity Cache: HBM is not its bound


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("NORI_BENCH_WORKLOAD", "pa4-cbox-path_mis"))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--triangles", type=int, default=None, help="c5 only: triangle count of the generated terrain")
    ap.add_argument("--split", default="auto", choices=["auto", "tile", "sample"])
    ap.add_argument("--merge", default="reduce", choices=["reduce", "gather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--engine", default="auto", choices=["auto", "megakernel", "wavefront"])
    ap.add_argument("--builder", default="auto", choices=["host", "lbvh", "auto", "ploc"],
                    help="nori_accel_builder: binned SAH with spatial splits on the host, or on the device: radix tree, PLOC + treelet sweeps + "
                         "parallel re-insertion; auto (default) = the device's builder (accel.builder / build_ms are in the line)")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY (tests/test_distributed_cpu.py): run the same sharding / merge / reporting code on CPU "
                         "ranks (gloo) with the emulated device headers standing in for the GPU; never a fallback")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the merged RGBW frame of the last step (.npy)")
    return ap.parse_args(argv)


def self_spawn(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------- renderers
class _EmuRenderer:
    """--emulate: tests/backends.Emu behind the Renderer surface bench.py uses (CPU tensors)."""

    def __init__(self, scene):
        from tests.backends import Emu
        self._e = Emu(scene)
        self.scene = scene
        self.border = self._e.border

    def frame_shape(self):
        return self._e.frame_shape()

    def accel_info(self):
        return self._e.accel_info()

    def set_option(self, k, v):
        pass

    def render_into(self, frame, spp_count=None, spp_begin=0, tile_mod=1, tile_rem=0, count_traversal=False, stream=None,
                    want_stats=True, time_kernels=False):
        import torch
        rgbw, st = self._e.render_host(spp_count=spp_count, spp_begin=spp_begin, tile_mod=tile_mod, tile_rem=tile_rem,
                                       count_traversal=count_traversal)
        frame += torch.from_numpy(rgbw)
        return st


def device_source_sha() -> str:
    """Hash of the device sources: ties a committed counter summary to the build it was measured on."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "nori_amd", "csrc", "device", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    m = re.search(r"^HIP_FLAGS = (.*)$", open(os.path.join(ROOT, "__graft_entry__.py")).read(), re.M)      # the build flags are part of the build
    h.update((m.group(1) if m else "").encode())
    return h.hexdigest()[:16]


def measured_counters(workload: str, engine: str):
    """The newest profiles/*_counters.json (tools/summarize_profile.py) for this workload / engine that was
    collected on THIS build of the device code; None otherwise (a stale figure is worse than none)."""
    sha = device_source_sha()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("device_source_sha") == sha and d.get("workload") == workload and d.get("engine") == engine:
            best = (f, d)
    return best


def bound_evidence(workload: str):
    """What binds wf_extend on this workload, MEASURED on this build (tools/collect_bound_evidence.sh -> tools/bound_evidence.py ->
    profiles/*_bound_evidence.json): VALU busy from the dynamic instruction mix (an interval), and what one more load, 16 / 32 more
    v_mov or 64 idle cycles per node step cost.  None if no file of this build names the workload."""
    sha = device_source_sha()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bound_evidence.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("device_source_sha") != sha:
            continue
        for cfg in d.get("configs", {}).values():
            if cfg.get("workload") == workload:
                best = (f, cfg)
    return best


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not args.emulate and not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU (torch.cuda.is_available() is False); the hot path has no CPU fallback")
    if world > 1 or "RANK" in os.environ:      # under a launcher: a process group even for one rank (exercises RCCL)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.emulate:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.emulate:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    from nori_amd import dist as ndist
    from nori_amd import workloads
    wl = workloads.load(args.workload, args.width, args.height, args.spp, args.triangles)
    sc = wl.scene
    width, height, spp = sc.camera.width, sc.camera.height, sc.sample_count
    split = wl.split if args.split == "auto" else args.split
    if args.merge == "gather" and split != "tile":
        raise SystemExit("bench.py: --merge gather needs --split tile (a sample split merges by sum-reduce)")

    if args.emulate:
        r = _EmuRenderer(sc)
    else:
        from nori_amd.render import Renderer
        r = Renderer(local_rank).upload(sc, builder={"host": 0, "lbvh": 1, "auto": 2, "ploc": 3}[args.builder])
    tiles_x, tiles_y = (width + 15) // 16, (height + 15) // 16
    tiles = tiles_x * tiles_y
    shard = ndist.shard(split, rank, world, spp)
    my_tiles = (tiles - shard["tile_rem"] + shard["tile_mod"] - 1) // shard["tile_mod"]
    r.set_option("engine", args.engine)      # a request: the library reports what actually rendered (nori_render_stats.engine)
    info = r.accel_info()
    frame = torch.zeros(r.frame_shape(), dtype=torch.float32, device=dev)
    stream = None if args.emulate else torch.cuda.current_stream(dev)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        if not args.emulate:
            torch.cuda.synchronize(dev)

    merge_ms = []      # per step: this rank's time in the exchange (its wait for the slowest rank included)

    def step(want_stats=False, count=False, time_kernels=False, spp_count=None):
        # this rank's share of the frame, then ONE exchange: ImageBlock::put(ImageBlock&) across GPUs
        return ndist.render_distributed(r.render_into, frame, split, spp if spp_count is None else spp_count, rank, world, merge=args.merge,
                                        tiles_x=tiles_x, border=r.border, merge_ms=merge_ms, stream=stream, want_stats=want_stats,
                                        count_traversal=count, time_kernels=time_kernels)

    # one instrumented pass: traversal counters for the roofline (untimed)
    counted = step(want_stats=True, count=True)
    if "engine" in counted:
        engine = {0: "megakernel", 1: "wavefront", 2: "block-serial"}[int(counted["engine"])]
    else:                      # --emulate: the CPU harness walks paths one by one, sized like the library's rule would
        engine = args.engine if args.engine != "auto" else ("wavefront" if my_tiles * 256 * shard["spp_count"] >= (1 << (24 if sc.integrator.type == "normals" else 19)) else "megakernel")
    for _ in range(args.warmup):
        step()
    barrier()
    del merge_ms[:]
    t0 = time.perf_counter()
    kernel_ms, trace_ms, shade_ms, film_ms, trace_launches = [], [], [], [], 0
    last = None
    for _ in range(args.steps):
        # stats: HIP-event times on the launch stream -- whole pass and per kernel class
        last = step(want_stats=True, time_kernels=True)
        kernel_ms.append(last["kernel_ms"]); trace_ms.append(last.get("trace_ms", 0.0)); shade_ms.append(last.get("shade_ms", 0.0))
        film_ms.append(last.get("film_ms", 0.0)); trace_launches = last.get("n_trace_launches", 0)
    barrier()
    dt = time.perf_counter() - t0

    rays_local = float(last["n_closest_rays"] + last["n_shadow_rays"])
    t = torch.tensor([dt, rays_local], dtype=torch.float64, device=dev)
    # what every rank saw, for reading a scaling record without a re-run: its own wall clock per step, the HIP-event time of its
    # share's kernels, and its time in the merge (which contains its wait for the slowest rank)
    mine = torch.tensor([dt / args.steps * 1e3, float(np.mean(kernel_ms)), float(np.mean(merge_ms)) if merge_ms else 0.0, rays_local],
                        dtype=torch.float64, device=dev)
    per_rank = [mine]
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_total = float(tmax[0]), float(tsum[1])
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    else:
        rays_total = rays_local

    if rank == 0:
        if args.dump_frame:
            np.save(args.dump_frame, frame.cpu().numpy())
        ms_per_step = dt / args.steps * 1e3
        mrays = rays_total / (ms_per_step * 1e-3) / 1e6
        k_ms, t_ms = float(np.mean(kernel_ms)), float(np.mean(trace_ms))
        if t_ms <= 0.0:
            t_ms = k_ms
        rays_c = counted["n_closest_rays"] + counted["n_shadow_rays"]
        # (the wavefront engine walks a BVH2 tree that has 32-B node records with its hand-written loop, whatever the tree's depth)
        node_loop_32b = engine == "wavefront" and info.get("node_children", 2) == 2 and info.get("node_records_32b", 0) == 1
        node_record_bytes = 32 if node_loop_32b else int(info["node_bytes"])
        trav_bytes = counted["n_node_tests"] * node_record_bytes + counted["n_tri_tests"] * info["tri_bytes"]
        if engine == "wavefront":
            dom_name = "wf_extend (all launches of one render pass)"
            # the path record as wf_extend sees it (wf_records.h): per closest-hit ray 12 B origin + 16 B direction-with-flags read and
            # the 16-B hit record written; per shadow ray 16 B (its direction, maxt) + the 16-B continuation direction read again
            # after it; the first vertex is recomputed, not read (-28 B per camera sample)
            rec_bytes = counted["n_closest_rays"] * 44 + counted["n_shadow_rays"] * 32 - counted["n_camera_samples"] * 28
        else:
            dom_name = "render_kernel (all launches of one render pass)"
            rec_bytes = counted["n_closest_rays"] * 96 + counted["n_camera_samples"] * 24
        ctr = measured_counters(wl.name, engine)
        cache_resident = info["total_bytes"] <= CACHE_RESIDENT_BYTES
        wide = info.get("node_children", 2) == 4
        ops_node = OPS_NODE_WIDE if wide else OPS_NODE
        ops = counted["n_node_tests"] * ops_node + counted["n_tri_tests"] * OPS_TRI + rays_c * OPS_RAY
        hbm_alg = (trav_bytes + rec_bytes) / (t_ms * 1e-3) / 1e9               # GB/s of algorithmic bytes
        valu_ach = ops / (t_ms * 1e-3) / 1e12                                   # Tops/s of algorithmic vector operations
        fr_hbm, fr_valu = hbm_alg / HBM_PEAK_GBS, valu_ach / VALU_PEAK_TOPS
        cd = ctr[1].get("dominant_kernel", {}) if ctr else {}
        # HBM bytes of the dominant kernel from the counters.  FETCH_SIZE counts a wide stream at half its bytes (the guide's doubling)
        # but record gathers 1 : 1 (tools/ubench_fetch.hip, profiles/r5_03_fetch_write_size_calibration.txt): the traversal kernels'
        # reads are gathers of node / leaf records plus the path state they stream -- known bytes, which the counter saw once
        # instead of twice.  `traffic_streaming_rule` = every read priced as a stream (rounds 1 - 4 reported that one).
        traffic_stream_rule = ctr[1].get("dominant_kernel_hbm_bytes") if ctr else None
        traffic_counter = ctr[1].get("dominant_kernel_hbm_bytes_gather_rule") if ctr else None      # FETCH_SIZE x 1 + WRITE_SIZE: the counters alone
        traffic_correction = None
        if traffic_counter is not None:
            streamed_reads = rec_bytes - counted["n_closest_rays"] * 16 if engine == "wavefront" else 0      # (the 16-B hit records are stores)
            traffic_correction = int(max(0, streamed_reads) // 2)      # MODELLED: the path state the kernel streams, which FETCH_SIZE counted at half
            traffic = int(traffic_counter + traffic_correction)
        else:
            traffic = traffic_stream_rule
        fr_hbm_measured = traffic / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
        valu_model = cd.get("valu_busy_model", cd.get("valu_busy_frac"))
        # which roof binds.  Measured evidence of THIS build first (profiles/*_bound_evidence.json: what one more load / 16 - 32 more
        # v_mov / 64 idle cycles per node step cost, normalised by what they add): the memory side if a load costs more than the
        # VALU instructions do, else the vector ALU.  Without it: the modelled VALU busy against the measured HBM share, and
        # without counters the structural rule (a cache-resident tree cannot be HBM bound).
        ev = bound_evidence(wl.name) if engine == "wavefront" else None
        near_both = False
        if ev:
            v = ev[1]["variants"]
            s_valu = sum(v["valu"]["sensitivity_lo_hi"]) / 2.0
            s_mem = v["load"].get("sensitivity") or 0.0
            bound = "hbm" if s_mem > s_valu else "valu"
            near_both = abs(s_mem - s_valu) < 0.1
        elif valu_model is not None and fr_hbm_measured is not None:
            bound = "valu" if valu_model >= fr_hbm_measured else "hbm"
            near_both = abs(valu_model - fr_hbm_measured) < 0.2      # the kernel sits against both roofs: say so (roof["bound_detail"])
        else:
            bound = "valu" if (cache_resident or fr_hbm > 1.0) else "hbm"
        roof = {"kernel": dom_name, "kernel_ms": round(t_ms, 3), "launches": int(trace_launches), "bound": bound,
                "node_tests": int(counted["n_node_tests"]), "node_children": 4 if wide else 2, "tri_tests": int(counted["n_tri_tests"]),
                "rays": int(rays_c), "bvh_bytes": int(info["total_bytes"])}
        if bound == "valu":
            roof.update({"achieved": round(valu_ach, 3), "peak": VALU_PEAK_TOPS, "unit": "Tops/s", "frac": round(fr_valu, 5),
                         "algorithmic_ops": int(ops),
                         "note": "VALU-issue bound (%s); ops = %d per node test + %d per triangle test + %d per ray (f32 vector operations as "
                                 "written in rt_trace.h), peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"
                                 % ("measured: " + ev[1]["verdict_from"] if ev
                                    else "BVH of %.1f MB is L2 / Infinity-Cache resident" % (info["total_bytes"] / 1e6) if cache_resident
                                    else "counters: VALU busy %.2f of its cycles (model), HBM at %.2f of its peak" % (valu_model, fr_hbm_measured) if valu_model is not None and fr_hbm_measured is not None
                                    else "algorithmic bytes exceed what HBM can deliver: part of them is cache-served" + (" (measured HBM traffic: %.2f of the peak -- about %.2f of what a copy reaches; this walk sits close to both roofs)" % (fr_hbm_measured, fr_hbm_measured / 0.79) if fr_hbm_measured is not None else ""),
                                    ops_node, OPS_TRI, OPS_RAY)})
        else:
            roof.update({"achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fr_hbm, 5),
                         "algorithmic_bytes": int(trav_bytes + rec_bytes),
                         "note": "algorithmic bytes per SURVEY 8(d): N_node x %d + N_tri x %d + ray / hit records; the BVH (%.0f MB) "
                                 "exceeds L2 + Infinity Cache%s" % (node_record_bytes, info["tri_bytes"], info["total_bytes"] / 1e6,
                                                                     "; measured: " + ev[1]["verdict_from"] if ev else "")})
        roof["valu_frac"], roof["hbm_algorithmic_frac"] = round(fr_valu, 5), round(fr_hbm, 5)
        # the three ways to price the same traversal work (all <= 1): `valu_frac` = textbook operation count (52 / 54 / 9) against
        # one lane-operation per lane per clock; `flop_frac` = the same count against the f32 FLOP peak (packed FMA: 4 flops
        # per lane per clock -- reachable only by v_pk_fma); `issued_frac` = the VALU instructions the shipped code issues
        # per unit (30 / 37.5 / 9; 40 per node test in the loop over 32-B records), as lane-instructions, against the issue rate
        # of the fastest instructions
        issued_node = ISSUED_NODE_WIDE if wide else ISSUED_NODE_32B if node_loop_32b else ISSUED_NODE
        issued = counted["n_node_tests"] * issued_node + counted["n_tri_tests"] * ISSUED_TRI + rays_c * ISSUED_RAY
        roof["node_record_bytes"] = node_record_bytes
        roof["flop_frac"] = round(ops / (t_ms * 1e-3) / 1e12 / FLOP_PEAK_TFLOPS, 5)
        roof["issued_valu_instr"] = int(issued)
        roof["issued_frac"] = round(issued / (t_ms * 1e-3) / VALU_ISSUE_SLOTS, 5)
        # `traffic` is an ESTIMATE: the counter figure (record gathers priced 1 : 1) + a modelled correction for the path state the
        # kernel streams, which FETCH_SIZE counts at half; both parts beside it
        roof["traffic"] = traffic
        if traffic_counter is not None:
            roof["traffic_counter"], roof["traffic_stream_correction_modelled"] = traffic_counter, traffic_correction
        if traffic_stream_rule is not None and traffic_stream_rule != traffic:
            roof["traffic_streaming_rule"] = traffic_stream_rule
        if ev or (valu_model is not None and fr_hbm_measured is not None):
            roof["bound_detail"] = "valu+hbm" if near_both else bound
        if ev:
            c = ev[1]
            roof["bound_evidence"] = {"source": os.path.relpath(ev[0], ROOT), "verdict": c["verdict"],
                                      "valu_busy_measured_lo_hi": c["valu_busy_measured_lo_hi"], "valu_lanes_per_instr": c["valu_lanes_per_instr"],
                                      "valu_sensitivity_lo_hi": c["variants"]["valu"]["sensitivity_lo_hi"], "vmem_read_sensitivity": c["variants"]["load"].get("sensitivity"),
                                      "cost_of_64_idle_cycles_per_node_step": c["variants"]["idle"]["dt_over_t"],
                                      "cost_of_more_valu_per_node_step": c["variants"]["valu"]["dt_over_t"], "cost_of_one_more_load_per_node_step": c["variants"]["load"]["dt_over_t"]}
        if ctr:
            roof["traffic_source"] = os.path.relpath(ctr[0], ROOT)
            if fr_hbm_measured is not None:
                roof["hbm_measured_frac"] = round(fr_hbm_measured, 5)
            if valu_model is not None:
                roof["valu_busy_model"] = valu_model      # static instruction mix x cycles per class; the measured interval: bound_evidence
            for k in ("valu_cycles_per_instr", "valu_lanes_per_instr", "valu_useful_frac", "l2_hit_rate"):
                if k in cd:
                    roof[k] = cd[k]
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1024x1024 256spp" if wl.name == "pa4-cbox-path_mis" else "Mrays/sec (primary+secondary)",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            # the triangles: a shipped scene of the reference flattened into tests/golden/ (tools/make_goldens.py), or a seeded generator
            "data": ("reference scene (" + wl.generator + ")") if "tests/golden/" in wl.generator else "synthetic (" + wl.generator + ")",
            "config": {"workload": wl.name, "baseline_config": wl.config, "generator": wl.generator,
                       "integrator": sc.integrator.type, "width": width, "height": height, "spp": spp,
                       "triangles": info["n_triangles"],
                       "parallelism": f"{split}-split x{world} + RCCL {args.merge}" if world > 1 else "single GPU",
                       "rays_per_step": int(rays_total), "seed_mode": "per_sample", "engine": engine},
            "roofline": roof,
            "pass": {"kernel_ms": round(k_ms, 3), "trace_ms": round(float(np.mean(trace_ms)), 3), "shade_ms": round(float(np.mean(shade_ms)), 3),
                     "film_ms": round(float(np.mean(film_ms)), 3),
                     # wavefront engine, calls of two or more batches: wf_finish launches that ran beside the next batch on their own CUs
                     "tail_beside_ms": round(float(last.get("tail_ms", 0.0)), 3), "tail_cus": int(last.get("tail_cus", 0)),
                     "hbm_measured_bytes": (ctr[1].get("pass_hbm_bytes") if ctr else None)},
            "accel": dict({k: (round(v, 3) if isinstance(v, float) else v) for k, v in info.items()},
                          # auto (nori_hip_build_accel): the device's builder (PLOC), the host's SAH builder as its fall-back
                          builder=args.builder if args.builder != "auto" else ("ploc" if info["built_on_device"] else "host"), builder_asked=args.builder),
        }
        if world > 1 or merge_ms:
            out["ranks"] = [{"rank": k, "ms_per_step": round(float(v[0]), 3), "kernel_ms": round(float(v[1]), 3), "merge_ms": round(float(v[2]), 3),
                             "rays_per_step": int(v[3])} for k, v in enumerate(per_rank)]
    # the CPU leg runs on rank 0 (the other ranks wait); the parity render that follows is the distributed one -- every rank its
    # share of the sample the CPU rendered, merged on rank 0 -- so the verdict is about the frame a multi-GPU run produces
    if not args.no_cpu_baseline:
        sample = torch.zeros(3, dtype=torch.int64, device=dev)
        if rank == 0:
            cpu, A, mod, s_cpu = cpu_baseline(wl, args, whole_tiles=world > 1, min_spp=world if (world > 1 and split == "sample") else 1)
            sample = torch.tensor([mod, s_cpu, 1], dtype=torch.int64, device=dev)
        if world > 1:
            dist.broadcast(sample, src=0)
        mod, s_cpu = int(sample[0]), int(sample[1])
        if world > 1:
            gst = step(want_stats=True, spp_count=s_cpu)
            g = torch.tensor([float(gst["n_closest_rays"] + gst["n_shadow_rays"])], dtype=torch.float64, device=dev)
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            rays_gpu = int(g[0])
        else:
            frame.zero_()
            gst = r.render_into(frame, spp_count=s_cpu, spp_begin=0, tile_mod=mod, tile_rem=0)
            rays_gpu = int(gst["n_closest_rays"] + gst["n_shadow_rays"])
        if rank == 0:
            if frame.is_cuda:
                torch.cuda.synchronize()
            out["cpu_baseline"] = cpu
            out["parity"] = parity_block(wl, A, frame.cpu().numpy(), r.border, cpu["rays"], rays_gpu)
            if world > 1:
                out["parity"]["frame"] = f"merged on rank 0 from {world} ranks ({split} split, {args.merge} merge)"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


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


def cpu_baseline(wl, args, whole_tiles=False, min_spp=1):
    """The oracle (kind 'port': the CPU restatement of the path, compiled for THIS host with -O3 -march=native
    -ffp-contract=off, SAH BVH, std::thread workers pulling 32x32 blocks as src/main.cpp:85-113 does) on all
    host cores, on a bounded sample: every k-th tile of the frame at s samples per pixel, sized for
    ~cpu_seconds (whole_tiles: every tile -- the sample a multi-GPU run can share out like the frame itself -- at s >= min_spp).
    Returns (block, RGBW frame of the sample, k, s): the GPU then renders exactly that sample (same tiles, same sample
    indices, same per-sample seeds) and the two images are compared: the parity verdict of this timed run."""
    from tests.backends import Oracle, use_native_oracle
    native = use_native_oracle()          # builds oracle/_native/liboracle_native_<cpu>.so on this host if it can
    sc = wl.scene
    spp = sc.sample_count
    cores = usable_cores()
    t0 = time.perf_counter()
    o = Oracle(sc, use_bvh=wl.name != "c1-bunny-normals")
    build_s = time.perf_counter() - t0
    # probe: every 16th tile at 1 spp -> seconds per (tile x spp)
    tiles = ((sc.camera.width + 15) // 16) * ((sc.camera.height + 15) // 16)
    probe_mod = 16 if tiles >= 256 else 1
    _, st = o.render_host(spp_count=1, tile_mod=probe_mod, tile_rem=0, threads=cores)
    per_tile_spp = max(st["kernel_ms"] * 1e-3, 1e-4) / max(1, tiles // probe_mod)
    budget = args.cpu_seconds / per_tile_spp                  # (tile x spp) units we can afford
    if budget >= tiles * spp:
        mod, s = 1, spp
    elif budget >= tiles or whole_tiles:
        mod, s = 1, max(1, int(budget // tiles))
    else:
        mod, s = int(min(tiles, -(-tiles // max(1.0, budget)))), 1
    s = min(spp, max(s, min_spp))
    A, st = o.render_host(spp_count=s, tile_mod=mod, tile_rem=0, threads=cores)
    rays = st["n_closest_rays"] + st["n_shadow_rays"]
    sec = st["kernel_ms"] * 1e-3
    cpu = {"value": round(rays / sec / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "flags": "-O3 -march=native -ffp-contract=off" if native else "-O2 -ffp-contract=off (native build unavailable)",
           "accel": "binned-SAH BVH" if wl.name != "c1-bunny-normals" else "brute force over all triangles (the reference's Accel, src/accel.cpp:30-40)",
           "accel_build_s": round(build_s, 2), "rays": int(rays), "seconds": round(sec, 2),
           "sample": f"{sc.camera.width}x{sc.camera.height}, every {mod}. tile, {s} spp of {spp} ({rays} rays in {sec:.1f} s, "
                     f"std::thread x{cores}, per-sample seeding)"}
    return cpu, A, mod, s


def parity_block(wl, A, B, border, rays_cpu, rays_gpu):
    """CPU frame A against device frame B of the same sample."""
    import numpy as np
    from nori_amd.render import develop_host
    a, b = develop_host(A, border), develop_host(B, border)
    covered = A[border:A.shape[0] - border, border:A.shape[1] - border, 3] > 0
    rel = (np.abs(a - b) / np.maximum(np.abs(a), 1e-2)).max(axis=-1)[covered]
    parity = {"frac_within_1e-3": round(float((rel <= 1e-3).mean()), 6), "mean_rel": float(f"{rel.mean():.3e}"),
              "tolerance": ">= 0.999 of pixels within 1e-3 relative, mean relative error <= 1e-4 (SURVEY 8(d), per-sample seeding)",
              "pixels": int(covered.sum()), "rays_cpu": int(rays_cpu), "rays_gpu": int(rays_gpu),
              "w_channel_max_abs_diff": float(np.abs(A[..., 3] - B[..., 3]).max())}
    parity["ok"] = bool(parity["frac_within_1e-3"] >= 0.999 and parity["mean_rel"] <= 1e-4 and parity["rays_cpu"] == parity["rays_gpu"])
    bsdfs = {m.bsdf.type for m in wl.scene.meshes}
    if "dielectric" in bsdfs:      # what the comparison cannot vouch for: nothing in the reference pins this plugin
        parity["unpinned_by_reference"] = "Dielectric::sample (src/dielectric.cpp:31-33 is a stub; no reference test touches it): oracle and device share the documented choice -- Fresnel-weighted reflect / refract, weight 1"
    return parity


if __name__ == "__main__":
    main()
