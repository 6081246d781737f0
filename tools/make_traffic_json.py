"""profiles/<tag>_<engine>_pmc_{FETCH_SIZE,WRITE_SIZE}.csv  ->  profiles/r1_traffic.json entries.

The CSVs come from tools/collect_profiles.sh: `bench.py --steps 3 --warmup 1 --no-cpu-baseline
--engine E` under `rocprofv3 --pmc X` = 1 counting pass + 4 plain passes.  HBM bytes of a kernel =
2 x FETCH_SIZE + WRITE_SIZE (KB; FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section; the read side
is calibrated by film_gather, whose doubled FETCH_SIZE equals its 6.4 GB sample store).
usage: python tools/make_traffic_json.py <tag> [workload key prefix]"""
import collections, csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "pa4-cbox-path_mis:1024x1024:256"
path = os.path.join(ROOT, "profiles", "r1_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}


def short(n):
    return re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", n).split("(")[0]


for eng in ("wavefront", "megakernel"):
    per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "calls": 0})
    ok = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(ROOT, "profiles", f"{tag}_{eng}_pmc_{c}.csv")
        if not os.path.exists(f):
            ok = False
            break
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                per[k]["calls"] += 1
    if not ok:
        continue
    ent = {"source": f"profiles/{tag}_{eng}_pmc_FETCH_SIZE.csv, profiles/{tag}_{eng}_pmc_WRITE_SIZE.csv", "kernels": {}}
    total = dom = 0.0
    for k, v in sorted(per.items()):
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        is_trace = k.startswith("wf_extend") or k.startswith("render_kernel")
        counting = False
        if is_trace:
            args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            counting = (args[2] if k.startswith("wf_extend") else args[2]) == "true"      # <STACK, SPILL, COUNT, FIRST> / <INTEG, STACK, COUNT>
        if counting:
            continue                    # the instrumented pass is not a timed one
        passes = 4 if is_trace else 5   # kernels without a counting variant ran in all 5 passes
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / passes
        ent["kernels"][k] = {"launches_per_pass": v["calls"] / passes, "fetch_kb_per_pass": round(v["FETCH_SIZE"] / passes, 1),
                             "write_kb_per_pass": round(v["WRITE_SIZE"] / passes, 1), "hbm_bytes_per_pass": int(b)}
        total += b
        if is_trace:
            dom += b
    ent["hbm_bytes_per_launch"] = int(total)        # whole render pass
    ent["dominant_kernel_bytes"] = int(dom)         # wf_extend / render_kernel launches of one pass
    out[f"{prefix}:{eng}"] = ent
    print(eng, "pass %.1f GB, dominant kernel %.1f GB" % (total / 1e9, dom / 1e9))
out["_note"] = ("hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 summed over the kernels of one render pass; "
                "separate rocprofv3 --pmc passes (tools/collect_profiles.sh); WRITE_SIZE uncalibrated")
json.dump(out, open(path, "w"), indent=1)
