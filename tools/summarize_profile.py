"""rocprofv3 counter CSVs of ONE render pass (tools/profile_round.sh) -> <dir>/<tag>_counters.json, the file bench.py
reads for `roofline.traffic` and the VALU-issue evidence.   usage: python tools/summarize_profile.py <dir> <tag> <workload> [engine]

Derived per kernel (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles; FETCH_SIZE reads half of a wide
coalesced stream on gfx950 -> doubled; counters are summed over the chip):
  elapsed_cycles       GRBM_GUI_ACTIVE / 8 XCDs                         (cross-check: SQ_BUSY_CYCLES / 32 shader engines)
  valu_busy_model      SQ_INSTS_VALU x cycles_per_instr / (1024 SIMDs x elapsed_cycles)   share of SIMD time the VALU is occupied -- a MODEL:
                       SQ_ACTIVE_INST_VALU counts one unit per instruction, not its cycles (profiles/r3_valu_counter_calibration.txt);
                       cycles_per_instr = the kernel's instruction mix x the measured cycles per class (tools/valu_mix.py ->
                       profiles/*_valu_mix.json; 3.3 when no mix is on file), every instruction of the kernel text weighted equally; not
                       clamped.  The measured counterpart (dynamic mix by class, sensitivities) is tools/bound_evidence.py.
  valu_lanes_per_instr SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU                  active lanes per VALU instruction (of 64)
  valu_useful_frac     valu_busy_frac x lanes / 64                                  share of VALU lane-cycles doing work
  hbm_bytes            (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- the guide's rule, calibrated on wide coalesced streams (film_gather).
                       It does NOT hold for gathers: tools/ubench_fetch.hip (profiles/r5_03_fetch_write_size_calibration.txt) reads known
                       byte counts through each access pattern -- a stream reads 2.00 bytes per counted byte, random 64-B and 96-B records
                       (the wide node, the leaf pair) 0.99 and 1.00, random 32-B records 0.30 (a 128-B line per record); WRITE_SIZE counts
                       streamed stores 1 : 1 and scattered 16-B nontemporal stores at 2 : 1 (a 32-B sector each -- bus traffic all the same).
  hbm_bytes_gather_rule  (1 x FETCH_SIZE + WRITE_SIZE) x 1024 -- what the kernel moved if ALL its reads were record gathers.  The traversal
                       kernels (wf_extend, wf_finish, render_kernel) lie between the two; bench.py places them with the bytes of path
                       state they are known to stream (`traffic` = gather rule + half the streamed reads, which FETCH_SIZE counted once).
  l2_hit_rate          TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import collections, csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
engine = sys.argv[4] if len(sys.argv) > 4 else "wavefront"


def short(n):
    return re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", n).split("(")[0]


def source_sha():
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "nori_amd", "csrc", "device", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    m = re.search(r"^HIP_FLAGS = (.*)$", open(os.path.join(ROOT, "__graft_entry__.py")).read(), re.M)      # as bench.py device_source_sha
    h.update((m.group(1) if m else "").encode())
    return h.hexdigest()[:16]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(d, f"{tag}_*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith(("__amd", "at::", "Cijk", "hip")):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1


def valu_mix():
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_mix.json"))):
        try:
            best = json.load(open(f))["kernels"]
        except (OSError, ValueError, KeyError):
            pass
    return best or {}


MIX = valu_mix()


def derive(c, kernel=None):
    out = {}
    el = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if el > 0:
        out["elapsed_cycles"] = el
    insts = c.get("SQ_INSTS_VALU", c.get("SQ_ACTIVE_INST_VALU"))
    if el > 0 and insts is not None:
        cyc = MIX.get(kernel, {}).get("cycles_per_instr", 3.3)
        out["valu_cycles_per_instr"] = cyc
        # a MODEL (static instruction mix x cycles per class), not clamped: above 1 it says the model overprices the mix
        # (the measured counterpart: tools/bound_evidence.py, valu_busy_measured_lo_hi)
        out["valu_busy_model"] = round(insts * cyc / (1024.0 * el), 4)
    if c.get("SQ_ACTIVE_INST_VALU"):
        out["valu_lanes_per_instr"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"], 2)
    if "valu_busy_model" in out and "valu_lanes_per_instr" in out:
        out["valu_useful_frac"] = round(min(1.0, out["valu_busy_model"]) * out["valu_lanes_per_instr"] / 64.0, 4)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        out["hbm_bytes"] = int((2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024)
        out["hbm_bytes_gather_rule"] = int((c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024)
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("SQ_WAVE_CYCLES"):
        for k2, n in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
            if k2 in c:
                out[n] = round(c[k2] / c["SQ_WAVE_CYCLES"], 4)
    return out


kernels = {}
for k, c in sorted(acc.items()):
    kernels[k] = {"launches": max(launches[k].values()), "counters": {n: v for n, v in sorted(c.items())}, "derived": derive(c, k)}
dom_prefix = "wf_extend" if engine == "wavefront" else "render_kernel"
dom = collections.defaultdict(float)
for k, c in acc.items():
    if k.startswith(dom_prefix):
        for n, v in c.items():
            dom[n] += v
dd = derive(dom, next((k for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)) if k.startswith(dom_prefix)), None))
out = {"tag": tag, "workload": workload, "engine": engine, "device_source_sha": source_sha(),
       "collected_with": "tools/profile_round.sh: one rocprofv3 --kernel-trace --pmc run per counter group, one render pass each (tools/wf_probe.py, REPS=1)",
       "kernels": kernels, "dominant_kernel": dict(dd, name=dom_prefix + " (all launches of the pass)"),
       "dominant_kernel_hbm_bytes": dd.get("hbm_bytes"), "dominant_kernel_hbm_bytes_gather_rule": dd.get("hbm_bytes_gather_rule"),
       "pass_hbm_bytes": int(sum(v["derived"].get("hbm_bytes", 0) for v in kernels.values())) or None}
path = os.path.join(d, f"{tag}_counters.json")
json.dump(out, open(path, "w"), indent=1)
print(path)
for k, v in kernels.items():
    print(f"{k}: launches {v['launches']} {json.dumps(v['derived'])}")
print("dominant:", json.dumps(out["dominant_kernel"]))
