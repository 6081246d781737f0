"""A PROJECTION of the 1 / 2 / 4 / 8-GPU bench lines from ONE GPU (review of round 5, item 6c) -- not a measurement of scaling: no
multi-GPU node has been available to any round, and the driver computes scaling efficiency itself from real N-GPU runs when it has one.

Per workload and N: the time of ONE device's share of the frame exactly as bench.py --gpus N shards it (tile split: every N-th 16x16
tile, tile_mod = N, the slowest of the shares rem = 0 and rem = N - 1; sample split: spp / N samples per pixel), HIP events around the
whole render call, median of REPS; plus the merge: a one-rank RCCL sum-reduce of the frame's bytes through torch.distributed (the
launch floor of the call bench.py makes; with more ranks the bytes cross xGMI: modelled beside it as bytes x (N - 1) / N over one
153 GB/s link).  projected ms(N) = share + modelled merge; speedup = ms(1) / ms(N).
    python tools/project_scaling.py [out.json]      (GPU box; default profiles/r6_projected_scaling.json)"""
import json, os, statistics, sys
sys.path.insert(0, ".")
import torch
import torch.distributed as dist
from nori_amd import workloads
from nori_amd.render import Renderer

REPS = int(os.environ.get("REPS", 5))
XGMI_LINK_GBS = 153.0
out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r6_projected_scaling.json"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)


def median_ms(fn):
    ts = []
    for _ in range(REPS + 1):
        ts.append(fn())
    return statistics.median(ts[1:])      # (the first call allocates)


result = {"what": "PROJECTION from one GPU -- not a measured scaling curve (no multi-GPU node was available); see tools/project_scaling.py",
          "xgmi_link_GBs_assumed": XGMI_LINK_GBS, "reps": REPS, "workloads": {}}
for name, spp in (("pa4-cbox-path_mis", None), ("c4-table-mis", 128), ("c5-terrain-10m", 128)):
    wl = workloads.load(name, spp=spp)
    sc = wl.scene
    r = Renderer(0).upload(sc, builder=2)
    frame = torch.zeros(r.frame_shape(), device="cuda")
    n_spp = sc.sample_count
    rows = []
    for n in (1, 2, 4, 8):
        def share(rem):
            def run():
                frame.zero_()
                kw = dict(tile_mod=n, tile_rem=rem) if wl.split == "tile" else dict(spp_begin=0, spp_count=max(1, n_spp // n))
                return r.render_into(frame, **kw)["kernel_ms"]
            return median_ms(run)
        share_ms = max(share(0), share(n - 1)) if (wl.split == "tile" and n > 1) else share(0)

        def merge():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM); e1.record(); e1.synchronize()
            return e0.elapsed_time(e1)
        merge_floor = median_ms(merge)
        nbytes = frame.numel() * 4
        merge_model = merge_floor + (nbytes * (n - 1) / n / (XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0)
        rows.append({"n_gpus": n, "share_ms": round(share_ms, 3), "merge_one_rank_measured_ms": round(merge_floor, 3), "merge_modelled_ms": round(merge_model, 3),
                     "projected_ms": round(share_ms + merge_model, 3)})
    t1 = rows[0]["projected_ms"]
    for row in rows:
        row["projected_speedup"] = round(t1 / row["projected_ms"], 2)
        row["projected_efficiency"] = round(t1 / row["projected_ms"] / row["n_gpus"], 3)
        row["share_of_ideal"] = round(rows[0]["share_ms"] / row["n_gpus"] / row["share_ms"], 3)
    result["workloads"][name] = {"split": wl.split, "spp": n_spp, "frame_bytes": frame.numel() * 4, "rows": rows}
    print(name, wl.split, "spp", n_spp)
    for row in rows:
        print("   N=%d share %.2f ms (%.0f %% of ideal) + merge %.2f ms (one-rank call %.2f) -> %.2f ms, projected speed-up %.2f" %
              (row["n_gpus"], row["share_ms"], 100 * row["share_of_ideal"], row["merge_modelled_ms"], row["merge_one_rank_measured_ms"], row["projected_ms"], row["projected_speedup"]))
    r.close()
json.dump(result, open(out_path, "w"), indent=1)
dist.destroy_process_group()
print(out_path)
